#!/bin/bash
set -o pipefail
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --batch-size 256 --steps 10 --warmup 4 > $O/bench_share2_b256.json 2> $O/bench_share2.err || { tail -30 $O/bench_share2.err; exit 1; }
cat $O/bench_share2_b256.json
timeout -k 10 300 python bench.py --gpus 4 --share-gpu --batch-size 128 --steps 10 --warmup 4 --grad-wire fp32 > $O/bench_share4_b128_fp32wire.json 2>> $O/bench_share2.err || { tail -30 $O/bench_share2.err; exit 1; }
cat $O/bench_share4_b128_fp32wire.json
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --batch-size 64 --arch vgg16 --steps 10 --warmup 4 > $O/bench_share2_vgg16.json 2>> $O/bench_share2.err || { tail -30 $O/bench_share2.err; exit 1; }
cat $O/bench_share2_vgg16.json
timeout -k 10 300 python bench.py --force-ddp --steps 20 --warmup 5 > $O/bench_forceddp_bf16wire.json 2>> $O/bench_share2.err || { tail -30 $O/bench_share2.err; exit 1; }
cat $O/bench_forceddp_bf16wire.json
