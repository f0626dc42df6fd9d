"""The one-plane (bf16-operand) variant of the GEMM launcher's products beside the default (exact split, fp32 products):
time and error against float64, on the products of a 32 x 800-frame train step.  One box, one process, variants in turn.
python tools/gemm_bf16.py [reps]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss_asr_amd import _lib, ops
dev = 'cuda'
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def t(fn, n=reps):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [  # name, ta, tb, batch, M, N, K, splitk
    ('i2h L2 NT x2', 0, 0, 2, 12800, 1024, 1024, 1), ('i2h L3 NT x2', 0, 0, 2, 6400, 1024, 1024, 1),
    ('i2h L2 470fr x2', 0, 0, 2, 7520, 1024, 1024, 1),
    ('dX L2 NN K2048', 0, 1, 1, 12800, 1024, 2048, 1),
    ('dW_ih TT x2 sk3', 1, 1, 2, 1024, 1024, 3200, 3), ('dW_hh TT x2 sk8', 1, 1, 2, 1024, 256, 3200, 8),
    ('big NT 4096^3', 0, 0, 1, 4096, 4096, 4096, 1), ('big TT 4096^3', 1, 1, 1, 4096, 4096, 4096, 1)]
VARIANTS = [('fp32 auto', 0, 0), ('bf16 auto', 1, 0), ('bf16 64', 1, 64), ('bf16 128', 1, 128)]
for name, ta, tb, nb, M, N, K, sk in shapes:
    a = torch.randn((nb, K, M) if ta else (nb, M, K), device=dev)
    b = torch.randn((nb, K, N) if tb else (nb, N, K), device=dev)
    out = torch.zeros(nb, M, N, device=dev)
    a64 = (a.transpose(1, 2) if ta else a)[:, :256].double()
    b64 = (b if tb else b.transpose(1, 2)).double()
    want = a64 @ b64                               # the first 256 rows are enough for the error
    if nb == 1:
        a, b, out = a[0], b[0], out[0]
    best, err = {}, {}
    for rnd in range(3):
        for v in (VARIANTS if rnd % 2 == 0 else VARIANTS[::-1]):
            assert lib.ssasr_set_option(b'SSASR_GEMM_BF16', v[1]) == 0
            assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', v[2]) == 0
            out.zero_()
            ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk)
            got = out.reshape(nb, M, N)[:, :256].double()
            err[v[0]] = ((got - want).norm() / want.norm()).item()
            us = t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk))
            best[v[0]] = min(best.get(v[0], 1e30), us)
    lib.ssasr_set_option(b'SSASR_GEMM_TILE', 0)
    lib.ssasr_set_option(b'SSASR_GEMM_BF16', 0)
    print('%-18s %2d x %5dx%5dx%5d | %s' % (name, nb, M, N, K, ' | '.join(
        '%s %7.1f us %6.1f TF err %.1e' % (v[0], best[v[0]], 2.0 * nb * M * N * K / best[v[0]] / 1e6, err[v[0]]) for v in VARIANTS)), flush=True)
