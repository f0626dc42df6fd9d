"""Runs forward + backward of a golden fixture repeatedly and reports which gradients vary
from run to run (diagnostic for races in the persistent kernels)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from ss_asr_amd import ops
from ss_asr_amd.asr import ASR
from ss_asr_amd.optim import FlatParameters

def seeded_weights(model, seed):          # any reproducible non-degenerate weights will do here
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5 if p.dim() > 1 else 0.0))

fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'full_b16_t400.npz'))
dims = [int(v) for v in fx['dims']]
torch.manual_seed(0)
model = ASR(*dims, float(fx['tf_rate']))
seeded_weights(model, 3)
model = model.to('cuda:0')
flat = FlatParameters(model)
x = torch.from_numpy(fx['x']).cuda(); y = torch.from_numpy(fx['y']).cuda()
lens = [int(v) for v in fx['lens']]; ans_len = int(fx['ans_len'])
names = [n for n, _ in model.named_parameters()]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run_once():
    flat.zero_grad()
    random.seed(int(fx['rng_seed']))
    _, logits, _ = model(x, int(fx['decode_steps']), teacher=y, state_len=lens)
    loss = ops.masked_ce_loss(logits, y, ans_len)
    loss.backward()
    ops.join_side_stream(); torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in model.parameters()], float(loss)


# reference gradients: per-step kernels, no segments
from ss_asr_amd import _lib
old_np, old_seg = _lib.set_option('SSASR_NO_PERSISTENT', 1), ops.bptt_segments
ops.bptt_segments = 1
ref, ref_loss = run_once()
_lib.set_option('SSASR_NO_PERSISTENT', old_np); ops.bptt_segments = old_seg
scale = [float(r.abs().max()) + 1e-30 for r in ref]
dev = np.zeros((N, len(ref)))
for it in range(N):
    g, loss = run_once()
    dev[it] = [float((a - b).abs().max()) / s for a, b, s in zip(g, ref, scale)]
ops.check_persistent_status()
base = np.median(dev, axis=0)
print('median elementwise deviation from the per-step reference (max |dg| / max |g|): worst', '%.2e' % base.max(), names[int(base.argmax())])
for it in range(N):
    odd = [(names[i], '%.1e' % dev[it, i]) for i in range(len(ref)) if dev[it, i] > max(4 * base[i], 2e-5)]
    if odd:
        print(' run', it, odd[:16])
print('done')
