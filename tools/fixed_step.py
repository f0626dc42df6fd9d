"""Runs a few train steps over ONE fixed batch (32 utterances, all `frames` long unless ragged)
so that a rocprofv3 kernel trace of it can be read as a timeline (tools/timeline.py)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config2_batches
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 800
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
print('batch: max frames %d, mean %.0f, label steps %d' % (max(lens), sum(lens) / len(lens), ans_len))
import time
for i in range(steps):
    if i == steps - 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    stepper(x, y, lens, ans_len)
torch.cuda.synchronize()
print('last 3 steps: %.3f ms/step' % ((time.perf_counter() - t0) / 3 * 1e3))
