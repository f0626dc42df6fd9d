import sys; sys.path.insert(0, '.'); sys.path.insert(0, '/root/repo')
import torch
from ss_asr_amd import ops
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(7648, 1024, 1024), (8192, 1024, 1024), (4096, 1024, 1024), (2048, 1024, 1024), (1024, 1024, 1024),
                (512, 1024, 1024), (8192, 1024, 4096), (8192, 1024, 256), (8192, 1024, 64), (8192, 1024, 32),
                (4096, 4096, 1024), (4096, 4096, 4096), (16384, 1024, 1024), (32768, 1024, 1024)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
    out = torch.zeros(M, N, device='cuda')
    us = t(lambda: ops.gemm(a, b, out=out))
    ref = t(lambda: torch.matmul(a, b.t()))
    blocks = ((M + 127) // 128) * ((N + 127) // 128)
    print('%6dx%5dx%5d blocks %5d ours %8.1f us %6.1f TF  us/kstep/round %.2f | rocBLAS %8.1f us %6.1f TF' % (
        M, N, K, blocks, us, 2.0 * M * N * K / us / 1e6, us / (K / 32) / max(1, (blocks + 511) // 512), ref, 2.0 * M * N * K / ref / 1e6))
