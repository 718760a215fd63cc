#!/bin/bash
set -o pipefail
out=gpurun_out/tail_ab.txt
: > $out
run() {
  echo "== tail_min=$1 $2 b$3" >> $out
  env DPA_IGEMM_TAIL_MIN=$1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
run 0 regnet_y_8gf 128 && run 40 regnet_y_8gf 128 &&
run 0 regnet_x_8gf 256 && run 40 regnet_x_8gf 256 &&
run 0 regnet_x_1_6gf 256 && run 40 regnet_x_1_6gf 256 && run 24 regnet_x_1_6gf 256 &&
run 0 regnet_y_1_6gf 256 && run 24 regnet_y_1_6gf 256 &&
run 0 regnet_y_400mf 256 && run 8 regnet_y_400mf 256 && run 16 regnet_y_400mf 256
cat $out
