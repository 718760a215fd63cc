#!/bin/bash
set -o pipefail
out=gpurun_out/premask_ab.txt
: > $out
run() {
  echo "== premask=$1 $2 b$3" >> $out
  env DPA_POOL_PREMASK=$1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
run 0 vgg16 128 && run 1 vgg16 128 && run 0 vgg16 128 && run 1 vgg16 128 && run 0 alexnet 512 && run 1 alexnet 512 && run 0 vgg11 128 && run 1 vgg11 128
cat $out
