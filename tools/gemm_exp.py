import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss_asr_amd import ops, _lib
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode in (0, 1):
    _lib.set_option('SSASR_GEMM_X6', mode)
    res = []
    for name, ta, tb, M, N, K, sk in [('NT 4096^3', 0, 0, 4096, 4096, 4096, 1), ('i2h L2 NT', 0, 0, 12800, 1024, 1024, 1),
                                      ('dX L2 NN', 0, 1, 12800, 1024, 1024, 1), ('dX L3 NN', 0, 1, 6400, 1024, 1024, 1),
                                      ('dW TT sk2', 1, 1, 1024, 1024, 3200, 2), ('dWhh TT sk8', 1, 1, 1024, 256, 3200, 8),
                                      ('dWih1 TT sk16', 1, 1, 1024, 80, 6400, 16), ('TN', 1, 0, 2048, 2048, 2048, 1)]:
        a = torch.randn((K, M) if ta else (M, K), device='cuda'); b = torch.randn((K, N) if tb else (N, K), device='cuda')
        out = torch.zeros(M, N, device='cuda')
        us = t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk))
        res.append('%s %.0f us %.0f TF' % (name, us, 2.0 * M * N * K / us / 1e6))
    print('x6=%d' % mode, ' | '.join(res))
