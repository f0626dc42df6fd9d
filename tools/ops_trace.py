"""Which Python lines launch the small torch kernels of a train step: wraps the tensor methods that
copy / fill / cast and prints, for one step, each call that touches a CUDA tensor with its innermost
frames inside this repository (diagnostic for DESIGN.md 4.5 / 9 (5))."""
import os, sys, random, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config2_batches
T = int(sys.argv[1]) if len(sys.argv) > 1 else 470
dev = torch.device('cuda', 0)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
model = ASR(**bench.DIMS).to(dev); model.train()
st = ASRTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
x, y, lens = next(iter(config2_batches(1, batch_size=32, feat_dim=80, seed=1, rank=0, hi=T)))
_, al = label_geometry(y)
x, y = x.to(dev), y.to(dev)
print('x', tuple(x.shape), 'max len', max(lens), 'y', y.dtype)
for _ in range(3): st(x, y, lens, al)
torch.cuda.synchronize()
on = [False]
def where():
    fr = [f for f in traceback.extract_stack()[:-2] if '/ss_asr_amd/' in f.filename or f.filename.endswith('bench.py')]
    return ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if on[0]:
            ts = [t for t in list(a) + [r] if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in ts):
                big = max(ts, key=lambda t: t.numel())
                same = isinstance(r, torch.Tensor) and len(a) and isinstance(a[0], torch.Tensor) and r.data_ptr() == a[0].data_ptr() and name in ('contiguous', 'to')
                if not same:
                    print('%-12s %-22s %s' % (name, tuple(big.shape), where()))
        return r
    setattr(owner, name, f)
for n in ('contiguous', 'to', 'copy_', 'fill_', 'zero_', 'clone', 'float', 'int', 'long'):
    wrap(torch.Tensor, n)
for n in ('zeros', 'rand', 'cat', 'zeros_like', 'ones', 'full', 'tensor'):
    wrap(torch, n)
on[0] = True
st(x, y, lens, al)
torch.cuda.synchronize()
on[0] = False
