#!/bin/bash
# one box, one run: the README / profiles/r1_bench_configs.md table
run() { timeout -k 10 300 python bench.py "$@" --steps 15 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('| %s %s, %d | %s | %.1f | %.1f GB |' % (d['config']['model'], d['dtype'], d['config']['per_gpu_batch'], format(round(d['value']), ','), d['ms_per_step'], d['config']['peak_mem_gb']))" || exit 1; }
run
run --batch-size 256
run --batch-size 2048
run --batch-size 256 --dtype fp16
run --batch-size 256 --dtype fp32
run --arch resnet18 --batch-size 256
run --arch resnet101 --batch-size 128
run --arch wide_resnet50_2 --batch-size 128
run --arch resnext50_32x4d --batch-size 256
run --arch vgg16 --batch-size 128
run --arch alexnet --batch-size 512
run --arch densenet121 --batch-size 128
run --arch squeezenet1_1 --batch-size 256
run --arch googlenet --batch-size 256
run --arch mobilenet_v2 --batch-size 256
run --arch mnasnet1_0 --batch-size 256
run --arch regnet_x_1_6gf --batch-size 256
run --arch regnet_x_8gf --batch-size 256
run --arch regnet_y_400mf --batch-size 256
run --arch regnet_y_1_6gf --batch-size 256
run --arch regnet_y_8gf --batch-size 128
run --arch mobilenet_v3_large --batch-size 256
run --arch mobilenet_v3_small --batch-size 256
run --arch efficientnet_b0 --batch-size 256
run --arch efficientnet_b3 --batch-size 128
run --arch efficientnet_v2_s --batch-size 128
run --arch inception_v3 --batch-size 128 --image-size 299
run --arch convnext_tiny --batch-size 128
run --arch convnext_base --batch-size 64
run --force-ddp
DPA_TORCH_BASELINE=1 run --batch-size 256
