"""Per-phase timing of the persistent decode loop (diagnostic).
    ./tools/build_trace_lib.sh && SSASR_LIB=tools/ab/trace.so python tools/dectrace.py
Prints, per role, the median time of each stamp relative to the step's first stamp and the
step period (s_memrealtime, 10 ns ticks)."""
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
B, Tp, U = 32, 100, 52
feat = torch.randn(B, Tp, 512, device='cuda')
enc_len = torch.full((B,), Tp, dtype=torch.int32, device='cuda')
teacher = torch.randint(3, 50, (B, U + 2), device='cuda').to(torch.int32)
modes = [0] * U
uniforms = torch.rand(U, B, device='cuda')
comp = ops.attn_precompute(feat, model.attention.psi.weight, model.attention.psi.bias)
for _ in range(3):
    logits, att, chars = ops.decoder_loop(feat, comp, enc_len, teacher, modes, uniforms, model._decoder_params())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
logits, att, chars = ops.decoder_loop(feat, comp, enc_len, teacher, modes, uniforms, model._decoder_params())
e1.record()
torch.cuda.synchronize()
ops.check_persistent_status()
print('decoder_loop forward: %.1f us total, %.2f us / decode step' % (e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / U))
STEPS, SLOTS, WG = 64, 8, 256
buf = np.zeros(WG * STEPS * SLOTS, dtype=np.uint64)
lib.ssasr_debug_dtrace.restype = C.c_int
lib.ssasr_debug_dtrace.argtypes = [C.c_void_p, C.c_size_t]
assert lib.ssasr_debug_dtrace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(WG, STEPS, SLOTS).astype(np.int64)
for name, wgs, nslot in (('attention (64 wg)', range(0, 64), 5), ('phi+compute (16 wg)', range(64, 80), 7),
                         ('compute (112 wg)', range(80, 192), 7)):
    t = tr[list(wgs), 5:U - 2]                     # steady-state steps
    rel = (t - t[:, :, :1]) * 0.01
    per = (t[:, 1:, 0] - t[:, :-1, 0]) * 0.01
    print('%-22s period %.2f us;' % (name, np.median(per)),
          ' '.join('s%d=%.2f' % (k, np.median(rel[:, :, k])) for k in range(1, nslot)))
# cross-role: when does the first / last compute workgroup publish h1_t relative to attention's step start
a0 = tr[0:64, 5:U - 2, 0]; a1 = tr[0:64, 5:U - 2, 1]; a4 = tr[0:64, 5:U - 2, 4]
c0 = tr[64:192, 5:U - 2, 0]; c1 = tr[64:192, 5:U - 2, 1]; c3 = tr[64:192, 5:U - 2, 3]; c4 = tr[64:192, 5:U - 2, 4]; c6 = tr[64:192, 5:U - 2, 6]
print('per step, medians over steps (us): last h1 publish -> last phi q seen by attention %.2f; q seen -> last ctx published %.2f; '
      'last ctx published -> last compute saw ctx %.2f; saw ctx -> last h1 published %.2f' % (
          np.median((a1.max(0)[1:] - c6.max(0)[:-1])) * 0.01, np.median(a4.max(0) - a1.max(0)) * 0.01,
          np.median(c4.max(0) - a4.max(0)) * 0.01, np.median(c6.max(0) - c4.max(0)) * 0.01))

# start-up and wind-down: first stamp of every role's step 0 against the earliest stamp in the launch, the
# first steps' periods, and the last stamp against the last step's start
first = tr[:, 0, 0]
t_begin = first[first > 0].min()
print('loop entry after the earliest stamp (us): attention median %.1f max %.1f; compute median %.1f max %.1f' % (
    np.median(tr[0:64, 0, 0] - t_begin) * 0.01, (tr[0:64, 0, 0] - t_begin).max() * 0.01,
    np.median(tr[64:192, 0, 0] - t_begin) * 0.01, (tr[64:192, 0, 0] - t_begin).max() * 0.01))
per0 = (tr[64:192, 1:9, 0] - tr[64:192, 0:8, 0]) * 0.01
print('compute workgroups, period of steps 0..7 (us): ' + ' '.join('%.2f' % np.median(per0[:, k]) for k in range(8)))
print('whole loop by the stamps (us): %.1f for %d steps' % ((tr[64:192, U - 1, 6].max() - t_begin) * 0.01, U))
