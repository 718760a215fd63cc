#!/bin/bash
# end-of-round-2 matrix (one box): "args -> img/s ms peak_GB"; usage: bench_final_matrix.sh <part 1|2>
mkdir -p gpurun_out/${ROUND:-r2}
out=gpurun_out/${ROUND:-r2}/bench_matrix_final_$1.txt; : > $out
run() { timeout -k 10 240 python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], d['ms_per_step'], d['config'].get('peak_mem_gb'))" | tee -a $out || exit 1; }
if [ "$1" = "1" ]; then
run --steps 20 --warmup 8 --batch-size 1024
run --steps 20 --warmup 8 --batch-size 256
run --steps 10 --warmup 4 --batch-size 2048
run --steps 20 --warmup 8 --batch-size 1024 --force-ddp
run --steps 20 --warmup 8 --batch-size 1024 --force-ddp --sync-bn
run --steps 20 --warmup 8 --batch-size 256 --dtype fp16
run --steps 10 --warmup 4 --batch-size 256 --dtype fp32
run --steps 20 --warmup 8 --arch resnet18 --batch-size 256
run --steps 20 --warmup 8 --arch resnet101 --batch-size 128
run --steps 20 --warmup 8 --arch wide_resnet50_2 --batch-size 128
run --steps 20 --warmup 8 --arch resnext50_32x4d --batch-size 256
run --steps 20 --warmup 8 --arch vgg16 --batch-size 128
run --steps 20 --warmup 8 --arch alexnet --batch-size 512
run --steps 20 --warmup 8 --arch densenet121 --batch-size 128
run --steps 20 --warmup 8 --arch squeezenet1_1 --batch-size 256
run --steps 20 --warmup 8 --arch googlenet --batch-size 256
run --steps 20 --warmup 8 --arch inception_v3 --batch-size 128 --image-size 299
run --steps 20 --warmup 8 --arch regnet_x_1_6gf --batch-size 256
run --steps 20 --warmup 8 --arch regnet_x_8gf --batch-size 256
run --steps 20 --warmup 8 --arch regnet_y_400mf --batch-size 256
run --steps 20 --warmup 8 --arch regnet_y_1_6gf --batch-size 256
run --steps 20 --warmup 8 --arch regnet_y_8gf --batch-size 128
else
run --steps 20 --warmup 8 --arch mobilenet_v2 --batch-size 256
run --steps 20 --warmup 8 --arch mnasnet1_0 --batch-size 256
run --steps 20 --warmup 8 --arch mobilenet_v3_large --batch-size 256
run --steps 20 --warmup 8 --arch mobilenet_v3_small --batch-size 256
run --steps 20 --warmup 8 --arch efficientnet_b0 --batch-size 256
run --steps 20 --warmup 8 --arch efficientnet_b3 --batch-size 128
run --steps 20 --warmup 8 --arch efficientnet_v2_s --batch-size 128
run --steps 20 --warmup 8 --arch convnext_tiny --batch-size 128
run --steps 20 --warmup 8 --arch convnext_base --batch-size 64
run --steps 20 --warmup 8 --arch shufflenet_v2_x0_5 --batch-size 256
run --steps 20 --warmup 8 --arch shufflenet_v2_x1_0 --batch-size 256
run --steps 20 --warmup 8 --arch shufflenet_v2_x2_0 --batch-size 256
run --steps 20 --warmup 8 --arch vit_b_16 --batch-size 128
run --steps 20 --warmup 8 --arch vit_b_32 --batch-size 128
run --steps 20 --warmup 8 --arch vit_l_16 --batch-size 64
run --steps 20 --warmup 8 --arch swin_t --batch-size 128
run --steps 20 --warmup 8 --arch swin_b --batch-size 64
run --steps 20 --warmup 8 --arch swin_v2_t --batch-size 128
run --steps 20 --warmup 8 --arch maxvit_t --batch-size 128
fi
