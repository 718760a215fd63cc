#!/bin/bash
# Stock PyTorch-ROCm vs this framework on ONE box, the configs BASELINE.md names besides the headline
# (fp32 b256 ResNet-50, VGG-16 b128 bf16, AlexNet b512 bf16).  Each step under its own time limit.
set -o pipefail
O=gpurun_out/r3_baselines
mkdir -p $O
run() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed rc=$?"; tail -3 $O/$name.err; return 1; }
  cat $O/$name.json
}
run stock_resnet50_fp32_b256 500 python bench/torch_baseline.py --arch resnet50 --fp32 --batch-size 256 &&
run ours_resnet50_fp32_b256 200 python bench.py --dtype fp32 --batch-size 256 &&
run stock_vgg16_bf16_b128 500 python bench/torch_baseline.py --arch vgg16 --batch-size 128 &&
run ours_vgg16_bf16_b128 200 python bench.py --arch vgg16 --batch-size 128 &&
run stock_alexnet_bf16_b512 400 python bench/torch_baseline.py --arch alexnet --batch-size 512 &&
run ours_alexnet_bf16_b512 200 python bench.py --arch alexnet --batch-size 512
