"""Tile-shape sweep of the fp32 GEMM over the row counts of the bench's layers (diagnostic):
time with 128 x 128 and with 64 x 64 tiles (SSASR_GEMM_TILE), NN (input projection) and NT (input gradient)."""
import os, sys; sys.path.insert(0, '.')
import torch
from ss_asr_amd import _lib, ops
dev = 'cuda'
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = K = 1024
for tb, name in ((0, 'NN'), (1, 'NT')):
    for M in (1280, 1920, 2560, 3200, 3744, 4000, 5120, 6400, 7488, 8000, 9440, 10880, 12800):
        a = torch.randn(M, K, device=dev)
        b = torch.randn((K, N) if tb else (N, K), device=dev)
        out = torch.zeros(M, N, device=dev)
        r = []
        for tile in ('128', '64'):
            _lib.set_option('SSASR_GEMM_TILE', int(tile))
            r.append(t(lambda: ops.gemm(a, b, ta=False, tb=bool(tb), out=out)))
        _lib.set_option('SSASR_GEMM_TILE', 0)
        auto = t(lambda: ops.gemm(a, b, ta=False, tb=bool(tb), out=out))
        print('%s M=%6d  128: %7.1f us %6.1f TF | 64: %7.1f us %6.1f TF | auto %7.1f' % (
            name, M, r[0], 2.0 * M * N * K / r[0] / 1e6, r[1], 2.0 * M * N * K / r[1] / 1e6, auto), flush=True)
