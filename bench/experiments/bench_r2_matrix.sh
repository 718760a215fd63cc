#!/bin/bash
# round-2 end-of-round matrix (one box): prints "args -> img/s ms"
mkdir -p gpurun_out/r2
out=gpurun_out/r2/bench_matrix_end.txt; : > $out
run() { timeout -k 10 240 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], d['ms_per_step'], d['config'].get('peak_mem_gb'))" | tee -a $out || exit 1; }
run --steps 20 --warmup 8
run --steps 20 --warmup 8 --batch-size 256
run --steps 10 --warmup 4 --batch-size 2048
run --steps 20 --warmup 8 --force-ddp
run --steps 20 --warmup 8 --force-ddp --sync-bn
run --steps 20 --warmup 8 --batch-size 256 --dtype fp16
run --steps 10 --warmup 4 --batch-size 256 --dtype fp32
run --steps 20 --warmup 8 --arch resnet101 --batch-size 128
run --steps 20 --warmup 8 --arch wide_resnet50_2 --batch-size 128
run --steps 20 --warmup 8 --arch resnext50_32x4d --batch-size 256
run --steps 20 --warmup 8 --arch densenet121 --batch-size 128
run --steps 20 --warmup 8 --arch vgg16 --batch-size 128
run --steps 20 --warmup 8 --arch vit_b_16 --batch-size 128
run --steps 20 --warmup 8 --arch swin_t --batch-size 128
run --steps 20 --warmup 8 --arch convnext_tiny --batch-size 128
run --steps 20 --warmup 8 --arch mobilenet_v2 --batch-size 256
run --steps 20 --warmup 8 --arch efficientnet_b0 --batch-size 256
run --steps 20 --warmup 8 --arch regnet_x_8gf --batch-size 256
