"""A/B of the GEMM tile shapes on one box, one process: the launcher's own choice (SSASR_GEMM_TILE 0), forced
128 x 128, forced 256 x 128 (the wide form), on the products of a 32 x 800-frame train step.
python tools/gemm_wide.py [reps]"""
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
    ('i2h L4 NT x2', 0, 0, 2, 3200, 1024, 1024, 1), ('i2h L2 470fr x2', 0, 0, 2, 7520, 1024, 1024, 1),
    ('dX L2 NN K2048', 0, 1, 1, 12800, 1024, 2048, 1), ('dX L3 NN K2048', 0, 1, 1, 6400, 1024, 2048, 1),
    ('dW_ih TT x2 sk2', 1, 1, 2, 1024, 1024, 3200, 2), ('dW_ih TT x2 sk3', 1, 1, 2, 1024, 1024, 3200, 3),
    ('dW_hh TT x2 sk8', 1, 1, 2, 1024, 256, 3200, 8),
    ('big NT 4096^3', 0, 0, 1, 4096, 4096, 4096, 1), ('big NN 4096^3', 0, 1, 1, 4096, 4096, 4096, 1),
    ('big TT 4096^3', 1, 1, 1, 4096, 4096, 4096, 1)]
want = sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] != '-' else None
TILES = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 128, 255, 256]
for name, ta, tb, nb, M, N, K, sk in shapes:
    if want and not any(w in name for w in want):
        continue
    a = torch.randn((nb, K, M) if ta else (nb, M, K), device=dev)
    b = torch.randn((nb, K, N) if tb else (nb, N, K), device=dev)
    out = torch.zeros(nb, M, N, device=dev)
    if nb == 1:
        a, b, out = a[0], b[0], out[0]
    row = []
    ref = None
    best = {}
    for rnd in range(3):                 # the variants in turn, three rounds: the chip's clock drifts with its load history
        for tile in (TILES if rnd % 2 == 0 else TILES[::-1]):
            assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', tile) == 0
            if sk > 1:
                out.zero_()
            us = t(lambda: ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, splitk=sk))
            best[tile] = min(best.get(tile, 1e30), us)
            if sk == 1 and rnd == 0:
                cur = out.clone()
                if ref is None:
                    ref = cur
                else:
                    assert (cur - ref).abs().max().item() <= 1e-3 * ref.abs().max().item(), name
    for tile in TILES:
        us = best[tile]
        row.append('%s %7.1f us %6.1f TF' % ({0: 'auto', 64: '64', 128: '128', 256: 'wideSK', 255: 'wide'}[tile], us, 2.0 * nb * M * N * K / us / 1e6))
    lib.ssasr_set_option(b'SSASR_GEMM_TILE', 0)
    print('%-18s %2d x %5dx%5dx%5d | %s' % (name, nb, M, N, K, ' | '.join(row)), flush=True)
