#!/bin/bash
# direct depthwise kernels vs the block-diagonal implicit-GEMM path, interleaved on one box
out=gpurun_out/bench_dw_ab.txt
: > $out
for cfg in "mobilenet_v2 256" "mnasnet1_0 256" "efficientnet_b0 256" "mobilenet_v3_large 256" "convnext_tiny 128" "efficientnet_v2_s 128"; do
  set -- $cfg
  for d in 0 1; do
    echo "== $1 bs$2 DPA_DW_DIRECT=$d" >> $out
    DPA_DW_DIRECT=$d timeout -k 10 150 python bench.py --arch $1 --batch-size $2 --steps 10 --warmup 4 >> $out 2>&1 || { echo "FAILED rc=$?" >> $out; exit 1; }
  done
done
