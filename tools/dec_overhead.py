"""Fixed cost of the persistent decode launches: ssasr_decoder_fwd (+ backward) timed for several U."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ss_asr_amd import ops
from ss_asr_amd.asr import ASR
dev = torch.device('cuda:0')
torch.manual_seed(5)
model = ASR(**bench.DIMS).to(dev)
B, T = 32, 100
feat = torch.randn(B, T, 512, device=dev)
enc_len = torch.full((B,), T, dtype=torch.int32, device=dev)
comp = ops.attn_precompute(feat, model.attention.psi.weight, model.attention.psi.bias).detach()
for U in (2, 7, 12, 27, 52, 102):
    teacher = torch.randint(3, 50, (B, U + 2), device=dev).to(torch.int32)
    modes = [0] * U
    def run():
        with torch.no_grad():
            return ops.decoder_loop(feat, comp, enc_len, teacher, modes, None, model._decoder_params())
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print('U=%3d  forward call %.1f us  (%.2f us / step)' % (U, us, us / U), flush=True)
ops.check_persistent_status()
