"""Per-phase timing of the persistent decoder backward chain (diagnostic).
    ./tools/build_trace_lib.sh && SSASR_LIB=tools/ab/trace.so python tools/chaintrace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from ss_asr_amd import _lib, ops
from ss_asr_amd.asr import ASR

lib = _lib.load()
torch.manual_seed(5)
model = ASR(50, 256, 256, 128, 80, 0.9).to('cuda:0')
B = 32
Tp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
U = int(sys.argv[2]) if len(sys.argv) > 2 else 52
feat = torch.randn(B, Tp, 512, device='cuda', requires_grad=True)
enc_len = torch.full((B,), Tp, dtype=torch.int32, device='cuda')
teacher = torch.randint(3, 50, (B, U + 2), device='cuda').to(torch.int32)
modes = [0] * U
uniforms = torch.rand(U, B, device='cuda')
for it in range(3):
    comp = ops.attn_precompute(feat, model.attention.psi.weight, model.attention.psi.bias)
    logits, att, chars = ops.decoder_loop(feat, comp, enc_len, teacher, modes, uniforms, model._decoder_params())
    model.zero_grad()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    logits.sum().backward()
    e1.record()
    torch.cuda.synchronize()
ops.check_persistent_status()
print('decoder backward (incl. GEMMs): %.1f us' % (e0.elapsed_time(e1) * 1e3))
STEPS, SLOTS, WG = 64, 8, 256
NSL = 2 if Tp <= 128 else 4 if Tp <= 256 else 6
NATT = NSL * B
buf = np.zeros(WG * STEPS * SLOTS, dtype=np.uint64)
lib.ssasr_debug_dtrace.restype = C.c_int
lib.ssasr_debug_dtrace.argtypes = [C.c_void_p, C.c_size_t]
assert lib.ssasr_debug_dtrace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(WG, STEPS, SLOTS).astype(np.int64)
for name, wgs, nslot in (('attention (%d wg)' % NATT, range(0, NATT), 4), ('cell (64 wg)', range(NATT, NATT + 64), 6)):
    t = tr[list(wgs), 5:min(U, STEPS) - 3]
    rel = (t - t[:, :, :1]) * 0.01
    per = (t[:, 1:, 0] - t[:, :-1, 0]) * 0.01
    print('%-20s period %.2f us;' % (name, np.median(per)),
          ' '.join('s%d=%.2f' % (k, np.median(rel[:, :, k])) for k in range(1, nslot)))
