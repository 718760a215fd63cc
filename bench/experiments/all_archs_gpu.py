"""Every registered architecture: one bf16 (AMP O2) training step + one eval forward on the GPU at a small batch."""
import sys
import time

import torch

from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.ops import functional as F
from distributed_pytorch_amd.optim import FusedSGD

only = set(a[1:] for a in sys.argv[1:] if a.startswith("+"))
skip = set(a for a in sys.argv[1:] if not a.startswith("+"))
bad = []
for arch in models.list_models():
    if arch in skip or (only and arch not in only):
        print("%-22s skipped" % arch, flush=True)
        continue
    size = 299 if arch == "inception_v3" else 224
    t0 = time.time()
    try:
        amp.state().reset()
        torch.manual_seed(0)
        m = models.create_model(arch).cuda().train()
        opt = FusedSGD(m.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4)
        m, opt = amp.initialize(m, opt, opt_level="O2", dtype="bf16")
        x = torch.randint(0, 256, (8, size, size, 3), dtype=torch.uint8, device="cuda")
        t = torch.randint(0, 1000, (8,), device="cuda")
        losses = []
        for _ in range(2):
            loss, _ = F.loss_and_metrics(m(x), t)
            opt.zero_grad()
            with amp.scale_loss(loss, opt) as s:
                s.backward()
            opt.step()
            losses.append(float(loss))
        m.eval()
        with torch.no_grad():
            out = m(x)
        ok = all(l == l and abs(l) < 50 for l in losses) and bool(torch.isfinite(out.float()).all()) \
            and out.shape == (8, 1000)
        print("%-22s %s loss %.3f -> %.3f  %.1fs  %.1f GB" % (arch, "ok " if ok else "BAD", losses[0], losses[1],
                                                             time.time() - t0, torch.cuda.max_memory_allocated() / 2**30),
              flush=True)
        if not ok:
            bad.append(arch)
    except Exception as e:  # noqa: BLE001
        print("%-22s EXC %s: %s" % (arch, type(e).__name__, str(e)[:200]), flush=True)
        bad.append(arch)
    del m, opt
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
print("FAILED:", bad)
sys.exit(1 if bad else 0)
