#!/bin/bash
# after the BN + smooth-activation fusion: the families it touches, and ResNet-50 (whose BN kernels must not move)
out=gpurun_out/bench_bnact.txt
: > $out
for cfg in "efficientnet_b0 256" "mobilenet_v3_large 256" "efficientnet_v2_s 128" "resnet50 1024" "resnet50 256"; do
  set -- $cfg
  echo "== $1 bs$2" >> $out
  timeout -k 10 200 python bench.py --arch $1 --batch-size $2 --steps 15 --warmup 5 >> $out 2>&1 || { echo "FAILED rc=$?" >> $out; exit 1; }
done
