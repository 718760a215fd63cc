#!/bin/bash
out=gpurun_out/ab_bn_grid_models.txt; : > $out
run() { timeout -k 10 240 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SG', '$*', '->', d['value'], d['ms_per_step'])" | tee -a $out || exit 1; }
for args in "--arch mobilenet_v2 --batch-size 256" "--arch efficientnet_b0 --batch-size 256" "--arch densenet121 --batch-size 128" "--arch resnet18 --batch-size 256" "--arch regnet_x_8gf --batch-size 256" "--batch-size 256"; do
  for SG in 2048 1280 1536; do
    export DPA_BN_GRID=$SG
    run --steps 20 --warmup 8 $args
  done
done
