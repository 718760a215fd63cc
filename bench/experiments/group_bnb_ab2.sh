#!/bin/bash
set -o pipefail
out=gpurun_out/group_bnb_ab2.txt
: > $out
run() {
  echo "== $1 $2 b$3" >> $out
  env DPA_FUSE_BN_BWD_GROUPED=$1 timeout -k 10 240 python bench.py --arch $2 --batch-size $3 --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out || return 1
}
run 1 resnext50_32x4d 256 && run 0 resnext50_32x4d 256 && run 1 resnext50_32x4d 256 && run 0 resnext50_32x4d 256 &&
run 1 resnext101_32x8d 128 && run 0 resnext101_32x8d 128 && run 1 resnext101_32x8d 128 && run 0 resnext101_32x8d 128
cat $out
