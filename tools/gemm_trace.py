"""Phase stamps of the wide (stream-K) GEMM kernel's workgroups: s_memtime at part start / prologue done / steady
state done / K loop done / next part opened / epilogue done, per run and part.
python tools/gemm_trace.py [M N K ta tb batch]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ss_asr_amd import _lib, ops
lib = _lib.load()
a = [int(v) for v in sys.argv[1:]]
M, N, K = a[:3] if len(a) >= 3 else (12800, 1024, 1024)
ta, tb = (a[3], a[4]) if len(a) >= 5 else (0, 0)
nb = a[5] if len(a) >= 6 else 2
x = torch.randn((nb, K, M) if ta else (nb, M, K), device='cuda')
y = torch.randn((nb, K, N) if tb else (nb, N, K), device='cuda')
out = torch.zeros(nb, M, N, device='cuda')
lib.ssasr_set_option(b'SSASR_GEMM_TILE', 256)
for _ in range(3): ops.gemm(x, y, ta=bool(ta), tb=bool(tb), out=out)
torch.cuda.synchronize()
G = 512
tr = torch.zeros(G * 8 * 8, dtype=torch.int64, device='cuda')
p = tr.data_ptr()
lib.ssasr_set_option(b'SSASR_GEMM_TRACE_LO', (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else (p & 0xffffffff))
lib.ssasr_set_option(b'SSASR_GEMM_TRACE_HI', p >> 32)
ops.gemm(x, y, ta=bool(ta), tb=bool(tb), out=out)
torch.cuda.synchronize()
lib.ssasr_set_option(b'SSASR_GEMM_TRACE_LO', 0); lib.ssasr_set_option(b'SSASR_GEMM_TRACE_HI', 0)
t = tr.cpu().numpy().reshape(G, 8, 8).astype(np.float64)
runs = [r for r in range(G) if t[r, 0, 0] > 0]
t0 = min(t[r, 0, 0] for r in runs)
names = ['prologue', 'steady', 'tail', 'open next', 'epilogue']
print('%d runs; shader cycles (s_memtime) per phase, median over runs [min .. max]' % len(runs))
for part in range(8):
    rows = [r for r in runs if t[r, part, 5] > 0]
    if not rows:
        break
    d = np.array([[t[r, part, k + 1] - t[r, part, k] for k in range(5)] for r in rows])
    print('part %d (%d runs): ' % (part, len(rows)) + '  '.join('%s %.0f [%.0f..%.0f]' % (n, np.median(d[:, k]), d[:, k].min(), d[:, k].max()) for k, n in enumerate(names)))
print('first part: start -> loaders open %.0f, -> first tile split + stored %.0f (median cycles)' % (
    np.median([t[r, 0, 6] - t[r, 0, 0] for r in runs]), np.median([t[r, 0, 7] - t[r, 0, 0] for r in runs])))
end = max(t[r, p_, 5] for r in runs for p_ in range(8))
print('first start -> last end: %.0f cycles; run spans: median %.0f, max %.0f' % (
    end - t0, np.median([max(t[r, :, 5]) - t[r, 0, 0] for r in runs]), max(max(t[r, :, 5]) - t[r, 0, 0] for r in runs)))
