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
shapes = [  # name, ta, tb, M, N, K, splitk   (full-length batch: 800 frames x 32 utterances)
 ('i2h L1 NT', 0, 0, 25600, 1024, 80, 1), ('i2h L2 NT', 0, 0, 12800, 1024, 1024, 1), ('i2h L3 NT', 0, 0, 6400, 1024, 1024, 1),
 ('i2h L4 NT', 0, 0, 3200, 1024, 1024, 1),
 ('dX L2 NN', 0, 1, 12800, 1024, 1024, 1), ('dX L3 NN', 0, 1, 6400, 1024, 1024, 1),
 ('dW_ih L2 seg TN sk2', 1, 1, 1024, 1024, 3200, 2), ('dW_ih L2 seg TN sk4', 1, 1, 1024, 1024, 3200, 4),
 ('dW_hh L2 seg TN sk8', 1, 1, 1024, 256, 3200, 8), ('dW_hh L1 seg TN sk8', 1, 1, 1024, 256, 6400, 8),
 ('dW_ih L1 seg TN sk16', 1, 1, 1024, 80, 6400, 16),
 ('psi NT', 0, 0, 3200, 128, 512, 1), ('big NT 4096^3', 0, 0, 4096, 4096, 4096, 1)]
for name, ta, tb, M, N, K, sk in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((K, N) if tb else (N, K), device=dev)
    out = torch.zeros(M, N, device=dev)
    us = t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk))
    ref = t(lambda: torch.matmul(a.t() if ta else a, b if tb else b.t()))
    print('%-20s %6dx%5dx%6d  ours %8.1f us %6.1f TF | torch(rocBLAS) %8.1f us %6.1f TF' % (
        name, M, N, K, us, 2.0 * M * N * K / us / 1e6, ref, 2.0 * M * N * K / ref / 1e6))
