#!/bin/bash
set -o pipefail
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_distributed.py -x -v --timeout 180 --timeout-method thread -k "hip_graph or world1 or bootstraps" > $O/graph_tests.log 2>&1
rc=$?; grep -v "^E  " $O/graph_tests.log | tail -40; [ $rc -eq 0 ] || exit $rc
for extra in "" "--graph 1"; do
  timeout -k 10 300 python bench.py --force-ddp --batch-size 256 --steps 20 --warmup 6 $extra > $O/tmp.json 2>$O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  cat $O/tmp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-ddp b256 $extra', d['value'], d['ms_per_step'], d['config']['hip_graph'], d['final_loss'])"
done
for m in "alexnet 512" "shufflenet_v2_x2_0 256"; do set -- $m
 for extra in "" "--force-ddp" "--force-ddp --graph 1" "--graph 1"; do
  timeout -k 10 300 python bench.py --arch $1 --batch-size $2 --steps 30 --warmup 8 $extra > $O/tmp.json 2>$O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
  cat $O/tmp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $extra', d['value'], d['ms_per_step'], d['config']['hip_graph'], d['final_loss'])"
 done
done
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --batch-size 256 --steps 10 --warmup 6 --graph 1 > $O/bench_share2_b256_graph.json 2>$O/tmp.err || { tail -20 $O/tmp.err; exit 1; }
cat $O/bench_share2_b256_graph.json
