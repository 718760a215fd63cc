import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_bnfold as T
for planes, shape in [(64, (4, 14, 14)), (128, (3, 9, 11)), (64, (32, 28, 28))]:
    net, x, gy, y1, dx1, g1, folded = T._run(True, planes, shape)
    _, _, _, y0, dx0, g0, none = T._run(False, planes, shape)
    print("planes", planes, shape, "folded", folded, none, "fwd equal", torch.equal(y1, y0))
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    print("  dx fold vs nofold: %.4g" % rel(dx1, dx0))
    for n in g1:
        print("  %-28s fold vs nofold %.4g   |g| %.4g" % (n, rel(g1[n], g0[n]), g0[n].norm().item()))
