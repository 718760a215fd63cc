#!/bin/bash
# A/B of the streaming-kernel workgroup cap (DPA_STREAM_GRID) and the BN grid on several families, one box, interleaved
out=gpurun_out/ab_stream_grid.txt; : > $out
run() { timeout -k 10 240 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SG', '$*', '->', d['value'], d['ms_per_step'])" | tee -a $out || exit 1; }
for args in "--arch mobilenet_v2 --batch-size 256" "--arch efficientnet_b0 --batch-size 256" "--arch convnext_tiny --batch-size 128" "--arch densenet121 --batch-size 128" "--arch vit_b_16 --batch-size 128" "--arch swin_t --batch-size 128" ""; do
  for SG in 4096 2560 1280 5120; do
    export DPA_STREAM_GRID=$SG
    run --steps 20 --warmup 8 $args
  done
done
