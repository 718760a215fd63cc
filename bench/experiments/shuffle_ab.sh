#!/bin/bash
set -o pipefail
out=gpurun_out/shuffle_ab.txt
: > $out
run() { timeout -k 10 240 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $out || exit 1; }
run --steps 40 --warmup 10 --arch shufflenet_v2_x1_0 --batch-size 256
run --steps 40 --warmup 10 --arch shufflenet_v2_x2_0 --batch-size 256
run --steps 40 --warmup 10 --arch shufflenet_v2_x1_0 --batch-size 256 --force-ddp
run --steps 40 --warmup 10 --arch shufflenet_v2_x1_0 --batch-size 256 --graph 1
run --steps 40 --warmup 10 --arch shufflenet_v2_x0_5 --batch-size 256
