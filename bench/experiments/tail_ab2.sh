#!/bin/bash
set -o pipefail
out=gpurun_out/tail_ab2.txt
: > $out
run() {
  echo "== tail_min=$1 $2 b$3" >> $out
  env DPA_IGEMM_TAIL_MIN=$1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
run 0 mobilenet_v3_large 256 && run 16 mobilenet_v3_large 256 && run 0 mobilenet_v3_large 256 && run 16 mobilenet_v3_large 256 && run 48 mobilenet_v3_large 256 &&
run 0 shufflenet_v2_x0_5 256 && run 16 shufflenet_v2_x0_5 256 && run 0 shufflenet_v2_x0_5 256 && run 16 shufflenet_v2_x0_5 256
cat $out
