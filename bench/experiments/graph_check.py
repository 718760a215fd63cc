import copy, sys, time, torch
sys.path.insert(0, '.')
from distributed_pytorch_amd import amp, models
from distributed_pytorch_amd.optim import FusedSGD
from distributed_pytorch_amd.utils.graph import GraphedTrainStep
arch, bs = sys.argv[1], int(sys.argv[2])
torch.manual_seed(0)
base = models.create_model(arch).cuda().train()
n = 10
xs = [torch.randint(0, 256, (bs, 224, 224, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
ts = [torch.randint(0, 1000, (bs,), device="cuda") for _ in range(n)]
for use_graph in (False, True):
    amp.state().reset()
    model = copy.deepcopy(base)
    opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    model, opt = amp.initialize(model, opt, opt_level="O2", dtype="bf16")
    st = GraphedTrainStep(model, opt, warmup=2, enabled=use_graph)
    losses = []
    for i in range(n):
        if i == 4:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _ = st(xs[i], ts[i])
        losses.append(loss.detach().clone())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (n - 4)
    print(arch, bs, "graph" if use_graph else "eager", "replays", st.replays, "ms/step %.2f" % (dt * 1e3),
          " ".join("%.4f" % float(l) for l in losses), flush=True)
