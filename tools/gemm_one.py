import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss_asr_amd import ops
M, N, K = 4096, 4096, 4096
a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.zeros(M, N, device='cuda')
for _ in range(3): ops.gemm(a, b, out=out)
torch.cuda.synchronize()
