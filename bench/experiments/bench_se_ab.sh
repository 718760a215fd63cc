#!/bin/bash
out=gpurun_out/bench_se_ab.txt
: > $out
for cfg in "efficientnet_b0 256" "regnet_y_1_6gf 256" "mobilenet_v3_large 256" "regnet_y_8gf 128"; do
  set -- $cfg
  for f in 0 1; do
    echo "== $1 bs$2 DPA_SE_FUSED=$f" >> $out
    DPA_SE_FUSED=$f timeout -k 10 150 python bench.py --arch $1 --batch-size $2 --steps 12 --warmup 4 >> $out 2>&1 || { echo "FAILED" >> $out; exit 1; }
  done
done
echo "== efficientnet_b0 graph + ddp rehearsal" >> $out
timeout -k 10 150 python bench.py --arch efficientnet_b0 --batch-size 128 --steps 8 --warmup 6 --graph 1 >> $out 2>&1 || { echo "FAILED" >> $out; exit 1; }
timeout -k 10 150 python bench.py --arch efficientnet_b0 --batch-size 128 --steps 8 --warmup 4 --force-ddp >> $out 2>&1 || { echo "FAILED" >> $out; exit 1; }
