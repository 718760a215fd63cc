#!/bin/bash
# A/B: grouped dgrad fused BN-backward reduction (DPA_FUSE_BN_BWD) and aligned group merging (DPA_GROUP_ALIGN)
set -o pipefail
out=gpurun_out/group_bnb_ab.txt
: > $out
run() {  # label, env..., -- arch batch
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "== $label $1 b$2" >> $out
  env "${envs[@]}" timeout -k 10 240 python bench.py --arch $1 --batch-size $2 --steps 20 --warmup 8 2>/dev/null | tail -1 >> $out || return 1
}
run base DPA_X=1 -- resnext50_32x4d 256 &&
run base DPA_X=1 -- regnet_y_8gf 128 &&
run align256 DPA_GROUP_ALIGN=256 -- regnet_y_8gf 128 &&
run base DPA_X=1 -- regnet_x_1_6gf 256 &&
run align128 DPA_GROUP_ALIGN=128 -- regnet_x_1_6gf 256 &&
run align256 DPA_GROUP_ALIGN=256 -- regnet_x_1_6gf 256 &&
run base DPA_X=1 -- regnet_y_1_6gf 256 &&
run align256 DPA_GROUP_ALIGN=256 -- regnet_y_1_6gf 256
cat $out
