#!/bin/bash
# which layer sizes profit from the BN-backward reduction inside the dgrad epilogue?  (ResNet-50 bs1024, interleaved)
out=gpurun_out/bench_bnfuse_window.txt
: > $out
for cfg in "0 1e9" "0 0" "0 450" "0 250" "0 120" "120 1e9" "250 1e9" "0 1e9"; do
  set -- $cfg
  echo "== min $1 MB max $2 MB" >> $out
  DPA_FUSE_BN_BWD_MIN_MB=$1 DPA_FUSE_BN_BWD_MAX_MB=$2 timeout -k 10 200 python bench.py --steps 15 --warmup 5 >> $out 2>&1 || { echo FAILED >> $out; exit 1; }
done
