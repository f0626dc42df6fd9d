"""Phase stamps of the split-T attention kernel (diagnostic build -DSSASR_ATTN_VARIANT=7):
SSASR_LIB=tools/ab/attn_v7.so python tools/attn_trace.py [T]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ss_asr_amd import _lib, ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 375
B, A, E, D = 32, 128, 512, 256
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
feat = torch.randn(B, T, E, generator=g).to(dev)
comp = torch.tanh(torch.randn(B, T, A, generator=g)).to(dev)
state = torch.randn(B, D, generator=g).to(dev)
w_phi = (torch.randn(A, D, generator=g) / 16).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(20):
    ops.attn_step(state, w_phi, comp, feat, lens)
torch.cuda.synchronize()
lib = _lib.load()
NS = int(lib.ssasr_attn_step_ws_floats(B, T, A, E)) // (2 * B * 544)
n = B * NS
buf = np.zeros(n * 8, dtype=np.uint64)
lib.ssasr_debug_attn_trace.argtypes = [C.c_void_p, C.c_int64]
assert lib.ssasr_debug_attn_trace(buf.ctypes.data, n * 8) == 0
t = buf.reshape(n, 8).astype(np.float64) / 100.0          # us
t0 = t[:, 0].min()
names = ['loads issued', 'partial in LDS', 'published', 'gather verified', 'after barrier', 'end']
print('workgroups %d (NS %d); times in us since the first workgroup issued its loads' % (n, NS))
for k, nm in enumerate(names):
    col = t[:, k] - t0
    print('%-16s min %6.2f  median %6.2f  max %6.2f' % (nm, col.min(), np.median(col), col.max()))
for k in range(1, 6):
    d = t[:, k] - t[:, k - 1]
    print('phase %d -> %d: median %6.2f  max %6.2f' % (k - 1, k, np.median(d), d.max()))
