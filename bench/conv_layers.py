#!/usr/bin/env python
"""Per-layer microbenchmark of the implicit-GEMM kernels over the ResNet-50 convolution shapes.

Times fprop / dgrad / wgrad of every distinct (H, Cin, Cout, k, stride) at per-GPU batch N and prints
achieved TFLOP/s and the "ideal" HBM time (each operand read / written once at 5.5 TB/s), so the tile
dispatcher can be tuned against measurements.  Run on the GPU box:
    python bench/conv_layers.py --batch 256 [--dtype bf16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (H_in, Cin, Cout, k, stride, pad, count in ResNet-50)
R50 = [
    (224, 3, 64, 7, 2, 3, 1),
    (56, 64, 64, 1, 1, 0, 1), (56, 64, 64, 3, 1, 1, 3), (56, 64, 256, 1, 1, 0, 4), (56, 256, 64, 1, 1, 0, 2),
    (56, 256, 128, 1, 1, 0, 1), (56, 128, 128, 3, 2, 1, 1), (28, 128, 512, 1, 1, 0, 4), (56, 256, 512, 1, 2, 0, 1),
    (28, 512, 128, 1, 1, 0, 3), (28, 128, 128, 3, 1, 1, 3),
    (28, 512, 256, 1, 1, 0, 1), (28, 256, 256, 3, 2, 1, 1), (14, 256, 1024, 1, 1, 0, 6), (28, 512, 1024, 1, 2, 0, 1),
    (14, 1024, 256, 1, 1, 0, 5), (14, 256, 256, 3, 1, 1, 5),
    (14, 1024, 512, 1, 1, 0, 1), (14, 512, 512, 3, 2, 1, 1), (7, 512, 2048, 1, 1, 0, 3), (14, 1024, 2048, 1, 2, 0, 1),
    (7, 2048, 512, 1, 1, 0, 2), (7, 512, 512, 3, 1, 1, 2),
]


def timeit(fn, iters=10):
    """GPU time per call.  The calls are captured into a hipGraph and replayed: a Python-launched sequence is
    CPU bound below ~50 us per call (autograd Function + allocator + pybind), which hid every kernel-side
    change on the small layers."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    except Exception:  # noqa: BLE001
        g = None
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if g is not None:
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (2 * iters) * 1e-3
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def sweep(args, hip_ops):
    """fprop / dgrad time of every tile configuration (igemm_force_config) per layer shape."""
    C = hip_ops.C()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    N = args.batch
    ncfg = C.igemm_num_configs()
    print("ms per call; columns = config ids 0..%d, 'a' = automatic choice" % (ncfg - 1))
    totals = {}
    for (H, Cin, Cout, k, s, p, cnt) in R50:
        if Cin % 8:
            continue
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(N, H, H, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, OH, OH, Cout, device="cuda").to(dt)
        for kind, fn in (("fprop", lambda: hip_ops.conv2d(x, w, None, (s, s), (p, p), False, None, True, False)),
                         ("dgrad", lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (s, s), (p, p)))):
            ts = []
            for cfg in list(range(ncfg)) + [-1]:
                C.igemm_force_config(cfg)
                ts.append(timeit(fn, 6) * 1e3)
            C.igemm_force_config(-1)
            best = min(range(ncfg), key=lambda i: ts[i])
            totals.setdefault(kind, [0.0, 0.0])
            totals[kind][0] += ts[-1] * cnt
            totals[kind][1] += ts[best] * cnt
            print("%3dx%-3d %4d->%-4d k%d s%d x%d %s  %s  a=%.3f best=%d (%.3f)" % (
                H, H, Cin, Cout, k, s, cnt, kind, " ".join("%.3f" % t for t in ts[:-1]), ts[-1], best, ts[best]),
                flush=True)
    for kind, (a, b) in totals.items():
        print("weighted %s: auto %.2f ms, best-per-layer %.2f ms" % (kind, a, b))


def sweep_wgrad(args, hip_ops):
    C = hip_ops.C()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    N = args.batch
    ncfg = C.wgrad_num_configs()
    C.wgrad_set_target_blocks(args.target_blocks)
    print("wgrad ms per call (kernel + slab fold); columns = config ids 0..%d, 'a' = automatic; target blocks %d" % (
        ncfg - 1, args.target_blocks))
    tot = [0.0, 0.0]
    for (H, Cin, Cout, k, s, p, cnt) in R50:
        Cx = Cin if Cin % 8 == 0 else 8
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(N, H, H, Cx, device="cuda").to(dt)
        w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, OH, OH, Cout, device="cuda").to(dt)
        fn = lambda: hip_ops.conv_wgrad(dy, x, w, (s, s), (p, p))
        ts = []
        for cfg in list(range(ncfg)) + [-1]:
            C.wgrad_force_config(cfg)
            ts.append(timeit(fn, 6) * 1e3)
        C.wgrad_force_config(-1)
        best = min(range(ncfg), key=lambda i: ts[i])
        tot[0] += ts[-1] * cnt
        tot[1] += ts[best] * cnt
        print("%3dx%-3d %4d->%-4d k%d s%d x%d  %s  a=%.3f best=%d (%.3f)" % (
            H, H, Cin, Cout, k, s, cnt, " ".join("%.3f" % t for t in ts[:-1]), ts[-1], best, ts[best]), flush=True)
    print("weighted wgrad: auto %.2f ms, best-per-layer %.2f ms" % tuple(tot))


def sweep_wgrad(args, hip_ops):
    C = hip_ops.C()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    N = args.batch
    ncfg = C.wgrad_num_configs()
    C.wgrad_set_target_blocks(args.target_blocks)
    print("wgrad ms per call (kernel + slab fold); columns = config ids 0..%d, 'a' = automatic; target blocks %d" % (
        ncfg - 1, args.target_blocks))
    tot = [0.0, 0.0]
    for (H, Cin, Cout, k, s, p, cnt) in R50:
        Cx = Cin if Cin % 8 == 0 else 8
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(N, H, H, Cx, device="cuda").to(dt)
        w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, OH, OH, Cout, device="cuda").to(dt)
        fn = lambda: hip_ops.conv_wgrad(dy, x, w, (s, s), (p, p))
        ts = []
        for cfg in list(range(ncfg)) + [-1]:
            C.wgrad_force_config(cfg)
            ts.append(timeit(fn, 6) * 1e3)
        C.wgrad_force_config(-1)
        best = min(range(ncfg), key=lambda i: ts[i])
        tot[0] += ts[-1] * cnt
        tot[1] += ts[best] * cnt
        print("%3dx%-3d %4d->%-4d k%d s%d x%d  %s  a=%.3f best=%d (%.3f)" % (
            H, H, Cin, Cout, k, s, cnt, " ".join("%.3f" % t for t in ts[:-1]), ts[-1], best, ts[best]), flush=True)
    print("weighted wgrad: auto %.2f ms, best-per-layer %.2f ms" % tuple(tot))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--sweep", action="store_true", help="time every igemm tile configuration per layer")
    ap.add_argument("--target-blocks", type=int, default=768)
    ap.add_argument("--sweep-wgrad", action="store_true", help="time every wgrad tile configuration per layer")
    args = ap.parse_args()
    from distributed_pytorch_amd.ops import hip_ops

    if args.sweep:
        return sweep(args, hip_ops)
    if args.sweep_wgrad:
        return sweep_wgrad(args, hip_ops)

    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    es = 2 if dt != torch.float32 else 4
    N = args.batch
    tot = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    ideal = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-34s %10s %10s %10s   (ms | TFLOP/s | ideal-HBM ms)" % ("layer", "fprop", "dgrad", "wgrad"))
    for (H, Cin, Cout, k, s, p, cnt) in R50:
        Cx = Cin if Cin % 8 == 0 else 8
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(N, H, H, Cx, device="cuda").to(dt)
        w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, OH, OH, Cout, device="cuda").to(dt)
        flops = 2.0 * N * OH * OH * Cout * Cin * k * k
        bx, by = x.numel() * es, dy.numel() * es
        res = {}
        res["fprop"] = timeit(lambda: hip_ops.conv2d(x, w, None, (s, s), (p, p), False, None, True, False))
        if Cx == Cin:
            res["dgrad"] = timeit(lambda: hip_ops.conv_dgrad(dy, w, tuple(x.shape), (s, s), (p, p)))
        res["wgrad"] = timeit(lambda: hip_ops.conv_wgrad(dy, x, w, (s, s), (p, p)))
        cells = []
        for kind in ("fprop", "dgrad", "wgrad"):
            if kind not in res:
                cells.append("%22s" % "-")
                continue
            t = res[kind]
            idl = (bx + by) / 5.5e12
            tot[kind] += t * cnt
            ideal[kind] += idl * cnt
            cells.append("%6.3f|%5.0f|%6.3f" % (t * 1e3, flops / t / 1e12, idl * 1e3))
        print("%3dx%-3d %4d->%-4d k%d s%d  x%d   %s" % (H, H, Cin, Cout, k, s, cnt, "  ".join(cells)))
    print("weighted totals (ms/step): " + "  ".join("%s %.2f (ideal %.2f)" % (k, v * 1e3, ideal[k] * 1e3)
                                                     for k, v in tot.items()))


if __name__ == "__main__":
    main()
