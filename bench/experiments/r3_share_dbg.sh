#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python bench.py --gpus 2 --share-gpu --arch resnet50 --batch-size 256 --steps 6 --warmup 4 > $O/dbg_loop.json 2> $O/dbg_loop.err
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v amdgpu.ids $O/dbg_loop.err | tail -60; exit 1; fi
done
