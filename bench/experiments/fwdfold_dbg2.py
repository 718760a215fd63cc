import sys, torch
import torch.nn.functional as TF
sys.path.insert(0, '.')
from distributed_pytorch_amd.ops import hip_ops
torch.manual_seed(0)
dev = "cuda"
NB, H, W, C3, Cd, N = 4, 14, 14, 64, 128, 256
a = torch.randn(NB, H, W, C3, device=dev).relu().bfloat16()
x = torch.randn(NB, H, W, Cd, device=dev).relu().bfloat16()
w3 = torch.randn(N, C3, 1, 1, device=dev) * 0.1
wd = torch.randn(N, Cd, 1, 1, device=dev) * 0.1
M = NB * H * W
one = hip_ops._tap_table(a.device, ((0, 0, 0),))
C = hip_ops.C()
def stats(inp, Cc, w):
    gram = hip_ops._wgrad_raw(inp, inp, M, Cc, Cc, H, W)
    cs = hip_ops._colsum(inp.view(M, Cc))
    wg = torch.empty((N, Cc), device=dev)
    hip_ops._igemm(w, gram, wg, None, None, one, N, Cc, Cc, 1, Cc, Cc, Cc, 1, N, 1, 1, 1, N, dtype=torch.float32, out_dtype=torch.float32)
    coef = torch.empty((4, N), device=dev); wsc = torch.empty((N, Cc), dtype=torch.bfloat16, device=dev)
    g = torch.ones(N, device=dev); b = torch.zeros(N, device=dev)
    C.bnfold_fwd(wg.data_ptr(), w.data_ptr(), cs.data_ptr(), float(M), g.data_ptr(), b.data_ptr(), 1e-5, 0.1, 0, 0, 0, coef.data_ptr(), wsc.data_ptr(), N, Cc, hip_ops._dt(inp), hip_ops._stream())
    z = inp.float().view(M, Cc) @ w.view(N, Cc).t()
    print("mean err", (coef[0] - z.mean(0)).abs().max().item(), "invstd rel err", ((coef[1] - (z.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / coef[1]).max().item(),
          "gram err", (gram - inp.float().view(M, Cc).t() @ inp.float().view(M, Cc)).abs().max().item() / gram.abs().max().item())
    return coef, wsc, z
c3, s3, z3 = stats(a, C3, w3)
cd, sd, zd = stats(x, Cd, wd)
ref = torch.relu((z3 - z3.mean(0)) * (z3.var(0, unbiased=False) + 1e-5).rsqrt() + (zd - zd.mean(0)) * (zd.var(0, unbiased=False) + 1e-5).rsqrt())
shift = c3[3] + cd[3]
out = torch.empty((NB, H, W, N), dtype=torch.bfloat16, device=dev)
mask = torch.empty(M * N // 8, dtype=torch.uint8, device=dev)
hip_ops._igemm(a, s3, out, shift, None, one, M, N, C3, 1, C3, C3, N, H, W, 1, 1, H, W, dtype=torch.bfloat16, out_dtype=torch.bfloat16, cat=(x, Cd, Cd, sd, Cd), relu_mask=mask)
torch.cuda.synchronize()
o = out.float().view(M, N)
print("fused rel err", ((o - ref).norm() / ref.norm()).item())
# pieces
p3 = torch.relu(a.float().view(M, C3) @ s3.float().t() + x.float().view(M, Cd) @ sd.float().t() + shift)
print("vs emulation", ((o - p3).norm() / p3.norm()).item())
p_only3 = a.float().view(M, C3) @ s3.float().t()
print("first operand only? ", ((o - torch.relu(p_only3 + shift)).norm() / p3.norm()).item())
