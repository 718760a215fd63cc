#!/bin/bash
set -o pipefail
out=gpurun_out/final_ab.txt
: > $out
run() {  # VAR=val arch batch
  echo "== $1 $2 b$3" >> $out
  env $1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 40 --warmup 12 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" >> $out || return 1
}
for i in 1 2 3; do run DPA_DGRAD_XMASK=0 vgg16 128 && run DPA_DGRAD_XMASK=1 vgg16 128 || exit 1; done
run DPA_IGEMM_TAIL_MIN=-16 regnet_y_8gf 128 && run DPA_IGEMM_TAIL_MIN=16 regnet_y_8gf 128 &&
run DPA_IGEMM_TAIL_MIN=-16 regnet_x_8gf 256 && run DPA_IGEMM_TAIL_MIN=16 regnet_x_8gf 256 &&
run DPA_IGEMM_TAIL_MIN=-16 regnet_x_1_6gf 256 && run DPA_IGEMM_TAIL_MIN=16 regnet_x_1_6gf 256 &&
run DPA_IGEMM_TAIL_MIN=-16 regnet_y_1_6gf 256 && run DPA_IGEMM_TAIL_MIN=16 regnet_y_1_6gf 256 &&
run DPA_IGEMM_TAIL_MIN=-16 inception_v3 128 && run DPA_IGEMM_TAIL_MIN=16 inception_v3 128
cat $out
