"""What slows a recurrence that shares the GPU: the 400-step BPTT launch of bench.recurrence_roofline timed alone
and beside four kinds of neighbour on a second stream -- a device-to-device copy (HBM streaming), a fill (write
traffic), this library's fp32 (split-bf16) GEMM on operands that stay in L2 / MALL (matrix pipe, little HBM traffic)
and the same GEMM on operands far larger than the caches.  Diagnostic behind DESIGN.md section 9 (2)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from ss_asr_amd import _lib, ops
dev = torch.device('cuda', 0)
lib = _lib.load()
S, N, H = 400, 32, 256
I = 4 * H
g = torch.Generator(device='cpu').manual_seed(6)
x = (torch.randn(S, N, I, generator=g) / 4).to(dev)
w = [(torch.randn(4 * H, I, generator=g) / 32).to(dev), (torch.randn(4 * H, H, generator=g) / 16).to(dev),
     torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)] * 2
y = torch.empty(S, N, 2 * H, device=dev)
dy = (torch.randn(S, N, 2 * H, generator=g) / 8).to(dev)
gates = torch.empty(2, S * N, 4 * H, device=dev); cs = torch.empty(2, S * N, H, device=dev); hs = torch.empty(2, S * N, H, device=dev)
hx = torch.empty(int(lib.ssasr_bilstm_fwd_hx_floats(S, N, H)), device=dev)
gx = torch.empty(int(lib.ssasr_bilstm_bwd_gx_floats(S, N, H)), device=dev)
ws_t = torch.empty(2, H, 4 * H, device=dev); ws_dc = torch.empty(2, 2, N, H, device=dev)
tsave = torch.empty(int(lib.ssasr_bilstm_tsave_floats(S, N, H)), device=dev)
sync = torch.zeros(8, device=dev, dtype=torch.int32)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def fwd():
    ops.check(lib.ssasr_bilstm_fwd(p(x), N * I, I, S, N, I, H, None, *[p(t) for t in w], p(y), N * 2 * H, 2 * H,
                                   p(gates), p(cs), p(hs), p(hx), p(sync), 0, p(tsave), st()), 'fwd')
def bwd():
    ops.check(lib.ssasr_bilstm_bwd(p(dy), N * 2 * H, 2 * H, p(x), N * I, I, S, N, I, H, None, p(w[0]), p(w[1]), p(w[4]), p(w[5]),
                                   p(gates), p(cs), p(hs), None, N * I, I, None, None, None, None, None, None,
                                   p(ws_t), p(ws_dc), p(gx), p(sync), 0, p(tsave), st()), 'bwd')
side = torch.cuda.Stream()
big_a = torch.empty(64 << 20, device=dev); big_b = torch.empty(64 << 20, device=dev)          # 256 MB each
sa = torch.randn(1024, 1024, device=dev); sb = torch.randn(1024, 1024, device=dev); sc = torch.empty(1024, 1024, device=dev)
la = torch.randn(16384, 4096, device=dev); lb = torch.randn(4096, 4096, device=dev); lc = torch.empty(16384, 4096, device=dev)
def n_copy(): big_b.copy_(big_a)
def n_fill(): big_b.fill_(1.0)
def n_gemm_small(): ops.gemm(sa, sb, out=sc)          # 2 GFLOP on 12 MB: 64 tiles of 128 x 128 or 256 of 64 x 64
def n_gemm_large(): ops.gemm(la, lb, out=lc)          # 550 GFLOP-scale streaming product
def measure(neigh, reps=6):
    out = []
    for _ in range(reps):
        fwd(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if neigh is not None:
            with torch.cuda.stream(side):
                t0 = torch.cuda.Event(enable_timing=True); t0.record()
                for _ in range(neigh[1]): neigh[0]()
                t1 = torch.cuda.Event(enable_timing=True); t1.record()
        e0.record(); bwd(); e1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 1e3, (t0.elapsed_time(t1) * 1e3) if neigh is not None else 0.0))
    out.sort()
    return out[len(out) // 2]
for name, neigh in (('alone', None), ('copy 256 MB x 12', (n_copy, 12)), ('fill 256 MB x 20', (n_fill, 20)),
                    ('gemm 1024^3 (cache resident) x 120', (n_gemm_small, 120)), ('gemm 16384 x 4096 x 4096 x 3', (n_gemm_large, 3))):
    us, tn = measure(neigh)
    print('%-40s BPTT launch %7.1f us = %.2f us / step   (neighbour busy %.0f us)' % (name, us, us / S, tn))
ops.check_persistent_status()
