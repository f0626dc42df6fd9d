"""Is the BPTT slower when its saved activations come from HBM instead of the Infinity Cache?
Times ssasr_bilstm_bwd (dx = dw = NULL: ring fill + the persistent BPTT kernel) at the layer-2
shape right after the forward call (saves still cache-resident) and after a 1 GB write that
evicts them."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss_asr_amd import _lib, ops
lib = _lib.load()
dev = torch.device('cuda:0')
S, N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 400, 32, 256
I = 4 * H
g = torch.Generator().manual_seed(6)
x = (torch.randn(S, N, I, generator=g) / 4).to(dev)
w = [(torch.randn(4 * H, I, generator=g) / 32).to(dev), (torch.randn(4 * H, H, generator=g) / 16).to(dev),
     torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)] * 2
y = torch.empty(S, N, 2 * H, device=dev)
dy = (torch.randn(S, N, 2 * H, generator=g) / 8).to(dev)
gates = torch.empty(2, S * N, 4 * H, device=dev)
hs = torch.empty(2, S * N, H, device=dev)
hx = torch.empty(int(lib.ssasr_bilstm_fwd_hx_floats(S, N, H)), device=dev)
gx = torch.empty(int(lib.ssasr_bilstm_bwd_gx_floats(S, N, H)), device=dev)
ws_t = torch.empty(2, H, 4 * H, device=dev); ws_dc = torch.empty(2, 2, N, H, device=dev)
tsave = torch.empty(int(lib.ssasr_bilstm_tsave_floats(S, N, H)), device=dev)
sync = torch.zeros(8, device=dev, dtype=torch.int32)
junk = torch.empty(256 * 1024 * 1024, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def fwd():
    ops.check(lib.ssasr_bilstm_fwd(p(x), N * I, I, S, N, I, H, None, *[p(t) for t in w], p(y), N * 2 * H, 2 * H,
                                   p(gates), None, p(hs), p(hx), p(sync), 0, p(tsave), st), 'fwd')
def bwd():
    ops.check(lib.ssasr_bilstm_bwd(p(dy), N * 2 * H, 2 * H, p(x), N * I, I, S, N, I, H, None, p(w[0]), p(w[1]), p(w[4]), p(w[5]),
                                   p(gates), None, p(hs), None, N * I, I, None, None, None, None, None, None,
                                   p(ws_t), p(ws_dc), p(gx), p(sync), 0, p(tsave), st), 'bwd')
for cold in (False, True, False, True):
    ts = []
    for _ in range(4):
        fwd()
        if cold:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / S)
    print('S=%d  saves %s: %.3f us / step (runs %s)' % (S, 'evicted' if cold else 'resident', sorted(ts)[1], ' '.join('%.2f' % t for t in ts)), flush=True)
assert int(sync[4]) == 0
