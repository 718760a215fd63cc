#!/bin/bash
# kernel-time tables of zoo families (looking for kernels that are far off their bandwidth floor)
# usage: prof_families.sh ["arch batch image" ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fam
if [ $# -eq 0 ]; then
  set -- "resnext50_32x4d 256 224" "densenet121 128 224" "googlenet 256 224" "vgg16 128 224" "inception_v3 128 299" "regnet_x_1_6gf 256 224" "convnext_tiny 128 224" "squeezenet1_1 256 224"
fi
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/fam/p_$1 -- python3 bench.py --arch $1 --batch-size $2 --image-size $3 --steps 6 --warmup 2 > gpurun_out/fam/$1.log 2>&1 || { echo "FAILED $1"; exit 1; }
  python scripts/summarize_rocprof_db.py gpurun_out/fam/p_$1 8 "$1 bs$2 bf16" > gpurun_out/fam/$1.md
  rm -rf gpurun_out/fam/p_$1
  echo "done $1"
done
