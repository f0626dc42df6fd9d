"""What an input-projection GEMM costs the FORWARD recurrence it would run beside (overlap layer
k + 1's projection with layer k's recurrence, middle-out).  The forward recurrence keeps all 256 CUs (256 workgroups,
one per CU), so a projection on a second stream has to share CUs with it.  Measured here, before building the
event plumbing: the first layer's recurrence (S = 470, N = 32, I = 80: fused input projection, one kernel) alone and
beside layer 2's projection (7,520 rows x 1,024 -> 2 x 1,024: one launch, both directions) issued on a second
stream `lag` us after the recurrence started, whole and as the middle-out half that would be legal (3,760 rows).
Prints the recurrence's duration, the GEMM's duration beside it, and what the step would gain:
    gain = (GEMM time taken off the critical path) - (recurrence slowdown)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from ss_asr_amd import _lib, ops
dev = torch.device('cuda', 0)
lib = _lib.load()
S, N, H, I = 470, 32, 256, 80
g = torch.Generator(device='cpu').manual_seed(6)
x = (torch.randn(S, N, I, generator=g) / 4).to(dev)
w = [(torch.randn(4 * H, I, generator=g) / 8).to(dev), (torch.randn(4 * H, H, generator=g) / 16).to(dev),
     torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)] * 2
y = torch.empty(S, N, 2 * H, device=dev)
gates = torch.empty(2, S * N, 4 * H, device=dev); hs = torch.empty(2, S * N, H, device=dev)
hx = torch.empty(int(lib.ssasr_bilstm_fwd_hx_floats(S, N, H)), device=dev)
tsave = torch.empty(int(lib.ssasr_bilstm_tsave_floats(S, N, H)), device=dev)
sync = torch.zeros(8, device=dev, dtype=torch.int32)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def rec():
    ops.check(lib.ssasr_bilstm_fwd(p(x), N * I, I, S, N, I, H, None, *[p(t) for t in w], p(y), N * 2 * H, 2 * H,
                                   p(gates), None, p(hs), p(hx), p(sync), 0, p(tsave), st()), 'fwd')
# layer 2's projection: rows = (S / 2) * N, K = 4 H (the pyramid's pair of frames), both directions as two batches
K2 = 4 * H
def make_gemm(rows):
    a = (torch.randn(rows, K2, generator=g) / 4).to(dev)
    wi = (torch.randn(2, 4 * H, K2, generator=g) / 32).to(dev)
    b = torch.zeros(4 * H, device=dev)
    out = torch.empty(2, rows, 4 * H, device=dev)
    def run():
        ops.check(lib.ssasr_gemm_f32(0, 0, rows, 4 * H, K2, C.c_float(1.0), p(a), K2, p(wi), K2, C.c_float(0.0), p(out), 4 * H,
                                     p(b), 0, 2, 0, 4 * H * K2, rows * 4 * H, 1, st()), 'gemm')
    return run
rows_all = (S // 2) * N
gemm_all, gemm_half = make_gemm(rows_all), make_gemm(rows_all // 2)
side = torch.cuda.Stream()
spin = torch.empty(1, device=dev)

def timed(fn, reps=7):
    v = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1) * 1e3)
    v.sort()
    return v[len(v) // 2]

rec(); gemm_all(); gemm_half(); torch.cuda.synchronize()
t_rec = timed(rec)
t_all, t_half = timed(gemm_all), timed(gemm_half)
print('alone: recurrence %.1f us (%.3f us / step), projection of all rows %.1f us, of the middle half %.1f us'
      % (t_rec, t_rec / S, t_all, t_half))

def beside(gemm, lag_us):
    """recurrence on the current stream; `gemm` on the side stream, started ~lag_us later (host sleep)"""
    out = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rec(); e1.record()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e6 < lag_us:
            pass
        with torch.cuda.stream(side):
            g0.record(); gemm(); g1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 1e3, g0.elapsed_time(g1) * 1e3, e0.elapsed_time(g1) * 1e3))
    out.sort()
    return out[len(out) // 2]

for name, gemm, t_g in (('all rows', gemm_all, t_all), ('middle half', gemm_half, t_half)):
    for lag in (0, 300, 550):
        r, gdur, gend = beside(gemm, lag)
        hidden = max(0.0, min(r, gend) - (gend - gdur))            # GEMM time that ran inside the recurrence's window
        print('%-12s lag %3d us: recurrence %7.1f us (+%5.1f), GEMM %6.1f us beside (alone %.1f), ends at %6.1f; '
              'hidden %.0f us of GEMM for %.0f us of slowdown -> gain %+.0f us'
              % (name, lag, r, r - t_rec, gdur, t_g, gend, hidden, r - t_rec, min(hidden, t_g) * (t_g / max(gdur, 1e-9)) - (r - t_rec)))
ops.check_persistent_status()
