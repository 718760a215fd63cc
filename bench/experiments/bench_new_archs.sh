#!/bin/bash
# throughput of the zoo families added late in round 1 (bf16, full training step, 1 GPU)
out=gpurun_out/bench_new_archs.txt
: > $out
for cfg in "regnet_y_400mf 256" "regnet_y_1_6gf 256" "regnet_y_8gf 128" "mobilenet_v3_large 256" "mobilenet_v3_small 256" "efficientnet_b0 256" "efficientnet_b3 128" "efficientnet_v2_s 128"; do
  set -- $cfg
  echo "== $1 bs$2" >> $out
  timeout -k 10 150 python bench.py --arch $1 --batch-size $2 --steps 10 --warmup 4 >> $out 2>&1 || { echo "FAILED rc=$?" >> $out; exit 1; }
done
# DDP rehearsal on one GPU for an SE model (autograd-delivered gradients next to kernel-delivered ones)
echo "== efficientnet_b0 bs256 --force-ddp" >> $out
timeout -k 10 150 python bench.py --arch efficientnet_b0 --batch-size 256 --steps 10 --warmup 4 --force-ddp >> $out 2>&1 || { echo "FAILED rc=$?" >> $out; exit 1; }
