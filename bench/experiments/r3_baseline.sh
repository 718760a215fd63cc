#!/bin/bash
# start-of-round-3 baseline: headline bench, forced-DDP overhead on launch-bound models, kernel profile
set -o pipefail
O=gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_start.json 2> $O/bench_start.err || exit 1
cat $O/bench_start.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-size 256 > $O/bench_b256_start.json 2>> $O/bench_start.err || exit 1
cat $O/bench_b256_start.json
for cfg in "alexnet 512" "shufflenet_v2_x2_0 256" "vgg16 128"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --arch $1 --batch-size $2 --steps 30 --warmup 8 > $O/${1}_plain_start.json 2>> $O/bench_start.err || exit 1
  timeout -k 10 200 python bench.py --arch $1 --batch-size $2 --steps 30 --warmup 8 --force-ddp > $O/${1}_ddp_start.json 2>> $O/bench_start.err || exit 1
  cat $O/${1}_plain_start.json $O/${1}_ddp_start.json
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_start -- python bench.py --steps 6 --warmup 3 > $O/prof_start.log 2>&1 || exit 1
python scripts/summarize_rocprof.py $O/prof_start 9 "ResNet-50 bf16 b1024, start of round 3" > $O/resnet50_bs1024_start_r3.md
