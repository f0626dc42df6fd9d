"""One GEMM shape, three launches: the subject of tools/gemm_pmc.sh's counter passes.
python tools/gemm_one.py [M N K [ta tb [batch]]]   (SSASR_GEMM_TILE in the environment forces a tile shape)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss_asr_amd import ops
a = [int(v) for v in sys.argv[1:]]
M, N, K = a[:3] if len(a) >= 3 else (4096, 4096, 4096)
ta, tb = (a[3], a[4]) if len(a) >= 5 else (0, 0)
nb = a[5] if len(a) >= 6 else 1
x = torch.randn((nb, K, M) if ta else (nb, M, K), device='cuda')
y = torch.randn((nb, K, N) if tb else (nb, N, K), device='cuda')
out = torch.zeros(nb, M, N, device='cuda')
if nb == 1:
    x, y, out = x[0], y[0], out[0]
for _ in range(3): ops.gemm(x, y, ta=bool(ta), tb=bool(tb), out=out)
torch.cuda.synchronize()
