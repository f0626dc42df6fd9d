"""In-process alternating A/B of one library option on the fixed train step: blocks of `n` steps with the option at
A, then at B, in turn, `rounds` times (one process, one box, one thermal state: what ab_env.sh cannot give).
python tools/ab_option.py NAME|env:VAR|ops:ATTR A B [frames] [n] [rounds]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ss_asr_amd import _lib
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config2_batches
name, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 470
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 5
lib = _lib.load()


def set_option(v):
    """NAME is a library option, or env:VAR for a switch the host code reads per call"""
    if name.startswith('env:'):
        os.environ[name[4:]] = str(v)
    elif name.startswith('ops:'):           # a module-level setting of ss_asr_amd.ops (e.g. ops:bptt_segments)
        from ss_asr_amd import ops
        assert hasattr(ops, name[4:])
        setattr(ops, name[4:], v)
    else:
        assert lib.ssasr_set_option(name.encode(), v) == 0
dev = torch.device('cuda', 0)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
model = ASR(**bench.DIMS).to(dev); model.train()
stepper = ASRTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
best = None
for x, y, lens in config2_batches(40, batch_size=32, feat_dim=80, seed=1, rank=0, hi=800):
    if best is None or abs(max(lens) - frames) < abs(max(best[2]) - frames):
        best = (x, y, lens)
x, y, lens = best
_, ans_len = label_geometry(y)
x, y = x.to(dev), y.to(dev)
for v in (A, B):
    set_option(v)
    for _ in range(4): stepper(x, y, lens, ans_len)
torch.cuda.synchronize()
res = {A: [], B: []}
for r in range(rounds):
    for v in ((A, B) if r % 2 == 0 else (B, A)):
        set_option(v)
        stepper(x, y, lens, ans_len)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): stepper(x, y, lens, ans_len)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / n * 1e3)
stepper.finish()
for v in (A, B):
    print('%s=%d  frames %d: %s  median %.3f ms/step' % (name, v, max(lens), ' '.join('%.3f' % t for t in res[v]), float(np.median(res[v]))))
