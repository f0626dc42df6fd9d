"""Decode loop forward / backward: time of ops.decoder_loop and of its backward for several U at one
encoder length, fitted as c + a * U (what a launch costs besides its steps), teacher-forced and with
10 % sampled steps.  usage: dec_startup.py [T'] [B]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, bench
from ss_asr_amd import ops
from ss_asr_amd.asr import ASR
T = int(sys.argv[1]) if len(sys.argv) > 1 else 58
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda', 0)
torch.manual_seed(5)
model = ASR(**bench.DIMS).to(dev)
feat = torch.randn(B, T, 512, device=dev)
enc_len = torch.full((B,), T, dtype=torch.int32, device=dev)
psi = (model.attention.psi.weight, model.attention.psi.bias)
for sampled in (0.0, 0.1):
    Us, fw, bw = (12, 24, 47, 94), [], []
    for U in Us:
        teacher = torch.randint(3, 50, (B, U + 2), device=dev).to(torch.int32)
        random.seed(3)
        modes = [1 if random.random() < sampled else 0 for _ in range(U)]
        uniforms = torch.rand(U, B, device=dev)
        f = feat.clone().requires_grad_(True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf, tb = [], []
        for it in range(6):
            torch.cuda.synchronize()
            ev[0].record()
            logits, _, _ = ops.decoder_loop(f, None, enc_len, teacher, modes, uniforms, model._decoder_params(), psi=psi)
            ev[1].record()
            logits.backward(torch.ones_like(logits) * 1e-3)
            ev[2].record()
            ops.join_side_stream(); torch.cuda.synchronize()
            if it > 1:
                tf.append(ev[0].elapsed_time(ev[1]) * 1e3); tb.append(ev[1].elapsed_time(ev[2]) * 1e3)
        fw.append(min(tf)); bw.append(min(tb))
        print('sampled %.1f U=%3d  forward %.1f us (%.2f / step)  backward %.1f us (%.2f / step)' % (sampled, U, fw[-1], fw[-1] / U, bw[-1], bw[-1] / U), flush=True)
    for name, v in (('forward', fw), ('backward', bw)):
        a, c = np.polyfit(np.array(Us, dtype=float), np.array(v), 1)
        print('  %s: %.2f us per step + %.1f us per call' % (name, a, c))
ops.check_persistent_status()
