#!/bin/bash
set -o pipefail
out=gpurun_out/dense_ab.txt
: > $out
run() {
  echo "== $1 $2 b$3" >> $out
  env $1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
run DPA_IGEMM_TAIL_MIN=0 densenet121 128 && run DPA_IGEMM_TAIL_MIN=16 densenet121 128 && run DPA_IGEMM_TAIL_MIN=0 densenet121 128 && run DPA_IGEMM_TAIL_MIN=16 densenet121 128 && run DPA_X=1 swin_b 64 && run DPA_X=1 swin_b 64
cat $out
