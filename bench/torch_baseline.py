#!/usr/bin/env python
"""Stock PyTorch-ROCm comparison point for bench.py (BASELINE.md "baselines to measure ourselves").

NOT part of the framework: plain ``torch.nn`` modules (MIOpen convolutions / batch norm, rocBLAS linear, ATen
elementwise), ``channels_last``, ``torch.autocast(bfloat16)``, ``torch.optim.SGD`` -- the way a user of stock
PyTorch would run the reference's ``imagenet_ddp_apex.py`` recipe on this GPU with its documented fast path.
Same workload as bench.py: ResNet-50 (torchvision v1.5 definition, re-stated here because torchvision is not
installed), synthetic uint8 224x224 batches from pinned host memory normalised on the GPU, forward + loss +
zero_grad + backward + SGD step in the timed region, images/sec = batch * steps / time.

    python bench/torch_baseline.py [--arch resnet50|vgg16|alexnet] --batch-size 256 --steps 20 --warmup 5 [--fp32] [--foreach 0]
"""
import argparse
import json
import sys
import threading
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(width * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = F.relu(self.bn2(self.conv2(y)), inplace=True)
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt, inplace=True)


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        layers, cin = [], 64
        for width, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            for b in range(blocks):
                layers.append(Bottleneck(cin, width, stride if b == 0 else 1, b == 0))
                cin = width * 4
        self.layers = nn.Sequential(*layers)
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, 2, 1)
        x = self.layers(x)
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class VGG16(nn.Module):
    """torchvision vgg16 (configuration D, no BatchNorm): 138,357,544 parameters."""

    def __init__(self, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for v in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"):
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(7)
        self.classifier = nn.Sequential(nn.Linear(512 * 49, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class AlexNet(nn.Module):
    """torchvision alexnet: 61,100,840 parameters."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2))
        self.avgpool = nn.AdaptiveAvgPool2d(6)
        self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(256 * 36, 4096), nn.ReLU(True),
                                        nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


ARCHS = {"resnet50": (ResNet50, 25557032), "vgg16": (VGG16, 138357544), "alexnet": (AlexNet, 61100840)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50", choices=sorted(ARCHS))
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--foreach", type=int, default=1)
    ap.add_argument("--miopen-benchmark", type=int, default=1)
    args = ap.parse_args()
    progress = {"phase": "init", "step": 0}

    def heartbeat():          # MIOpen's first-call kernel search is silent for minutes; the GPU runner kills silent jobs
        t0 = time.time()
        while not progress.get("done"):
            time.sleep(30)
            print("[torch_baseline] %4.0fs %s step %d" % (time.time() - t0, progress["phase"], progress["step"]),
                  file=sys.stderr, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)       # the reference sets this (imagenet_ddp.py:158)
    torch.manual_seed(1234)
    ctor, nparams = ARCHS[args.arch]
    model = ctor().to(dev).to(memory_format=torch.channels_last).train()
    assert sum(p.numel() for p in model.parameters()) == nparams
    lr = 0.1 * args.batch_size / 256.0 * 0.01
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4, foreach=bool(args.foreach))
    crit = nn.CrossEntropyLoss()
    ring = [(torch.randint(0, 256, (args.batch_size, 3, 224, 224), dtype=torch.uint8).contiguous(
        memory_format=torch.channels_last).pin_memory(), torch.randint(0, 1000, (args.batch_size,)).pin_memory())
        for _ in range(4)]
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1) * 255
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1) * 255
    copy = torch.cuda.Stream(dev)

    def fetch(i):
        with torch.cuda.stream(copy):
            x, t = ring[i % 4]
            x = x.to(dev, non_blocking=True).float().sub_(mean).div_(std)
            return x, t.to(dev, non_blocking=True)

    nxt = fetch(0)

    def step(i):
        nonlocal nxt
        torch.cuda.current_stream().wait_stream(copy)
        x, t = nxt
        x.record_stream(torch.cuda.current_stream())
        t.record_stream(torch.cuda.current_stream())
        nxt = fetch(i + 1)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
            loss = crit(model(x), t)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    progress["phase"] = "warmup"
    for i in range(args.warmup):
        loss = step(i)
        torch.cuda.synchronize()
        progress["step"] = i + 1
    progress["phase"] = "timed"
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    progress["done"] = True
    print(json.dumps({
        "metric": "images/sec " + args.arch + " training, STOCK PyTorch-ROCm (torch.nn + autocast + channels_last + torch.optim.SGD)",
        "value": round(args.batch_size * args.steps / dt, 2), "unit": "images/sec", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
        "dtype": "fp32" if args.fp32 else "bf16 autocast", "final_loss": float(loss.item()),
        "config": {"model": args.arch, "per_gpu_batch": args.batch_size, "torch": torch.__version__,
                   "foreach_sgd": bool(args.foreach), "miopen_benchmark": bool(args.miopen_benchmark),
                   "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}}), flush=True)


if __name__ == "__main__":
    main()
