import sys, time; sys.path.insert(0, '.')
import torch
from ss_asr_amd import ops
dev = 'cuda'
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [  # name, ta, tb, M, N, K, splitk
 ('i2h L2 NT', 0, 0, 7648, 1024, 1024, 1), ('i2h L1 NT', 0, 0, 15296, 1024, 80, 1),
 ('dX L2 NN', 0, 1, 7648, 1024, 1024, 1), ('dW_ih L2 TN sk2', 1, 1, 1024, 1024, 7648, 2),
 ('dW_ih L2 TN sk4', 1, 1, 1024, 1024, 7648, 4), ('dW_ih L2 TN sk8', 1, 1, 1024, 1024, 7648, 8),
 ('dW_hh L2 TN sk8', 1, 1, 1024, 256, 7616, 8), ('dW_hh L1 TN sk8', 1, 1, 1024, 256, 15264, 8),
 ('dW_ih L1 TN sk16', 1, 1, 1024, 80, 15296, 16), ('dW_ih L1 TN sk32', 1, 1, 1024, 80, 15296, 32),
 ('dW_hh L2 TN sk16', 1, 1, 1024, 256, 7616, 16), ('dW_hh L1 TN sk16', 1, 1, 1024, 256, 15264, 16), ('dW_hh L1 TN sk32', 1, 1, 1024, 256, 15264, 32),
 ('dW_ih L2 TN sk16', 1, 1, 1024, 1024, 7648, 16), ('psi NT', 0, 0, 1920, 128, 512, 1),
 ('big NT 4096^3', 0, 0, 4096, 4096, 4096, 1)]
for name, ta, tb, M, N, K, sk in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((K, N) if tb else (N, K), device=dev)
    out = torch.zeros(M, N, device=dev)
    us = t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk))
    ref = t(lambda: torch.matmul(a.t() if ta else a, b if tb else b.t()))
    print('%-20s %6dx%5dx%6d  ours %8.1f us %6.1f TF | torch(rocBLAS) %8.1f us %6.1f TF' % (
        name, M, N, K, us, 2.0 * M * N * K / us / 1e6, ref, 2.0 * M * N * K / ref / 1e6))
