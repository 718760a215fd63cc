"""One depthwise shape, a few launches of each kernel (for rocprofv3 counter passes)."""
import sys
import torch
sys.path.insert(0, ".")
from distributed_pytorch_amd import backend

ext = backend.require()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
N, H, W, C, K, st = [int(a) for a in sys.argv[1:7]] if len(sys.argv) > 6 else (256, 56, 56, 144, 3, 1)
p = K // 2
OH, OW = (H + 2 * p - K) // st + 1, (W + 2 * p - K) // st + 1
x = torch.randn(N, H, W, C, device=dev, dtype=torch.bfloat16)
w = torch.randn(C, K, K, device=dev)
y = torch.empty(N, OH, OW, C, device=dev, dtype=torch.bfloat16)
dy = torch.randn(N, OH, OW, C, device=dev, dtype=torch.bfloat16)
dx = torch.empty_like(x)
rows = ext.dw_conv_rows(N * OH * OW, C, 1)
stats = torch.empty(rows * 2 * C + rows, device=dev)
wrows = ext.dw_wgrad_rows(N * OH * OW, C, K, 1)
partial = torch.empty(wrows * C * K * K, device=dev)
for _ in range(3):
    ext.dw_conv(x.data_ptr(), w.data_ptr(), 0, 0, y.data_ptr(), stats.data_ptr(), 0, N, H, W, C, OH, OW, K, K, st, st, p, p, 0, 1, s)
    ext.dw_conv(dy.data_ptr(), w.data_ptr(), 0, 0, dx.data_ptr(), 0, 1, N, H, W, C, OH, OW, K, K, st, st, p, p, 0, 1, s)
    ext.dw_wgrad(dy.data_ptr(), x.data_ptr(), partial.data_ptr(), N, H, W, C, OH, OW, K, K, st, st, p, p, 1, s)
torch.cuda.synchronize()
