#!/bin/bash
set -o pipefail
out=gpurun_out/grow3_ab.txt
: > $out
run() {
  echo "== row3=$1 $2 b$3" >> $out
  env DPA_IGEMM_ROW3=$1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
run -1 resnext50_32x4d 256 && run -1 resnext101_32x8d 128 && run -1 regnet_x_8gf 256 && run -1 regnet_y_8gf 128 && run -1 regnet_x_1_6gf 256
cat $out
