"""Runs a few train steps of BASELINE.json configs[3]'s batch (synthetic.config4_batch: 32 utterances of
1500-3000 frames, ~300 label steps) so that a rocprofv3 kernel trace of it can be summarised
(tools/summarize_profile.py) or read as a timeline (tools/timeline.py).
usage: config4_step.py [steps] [joint]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config4_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
joint = len(sys.argv) > 2 and sys.argv[2] == 'joint'
dev = torch.device('cuda', 0)
random.seed(4); np.random.seed(4); torch.manual_seed(4)
if joint:
    from ss_asr_amd.ctc import JointCTCASR, JointCTCTrainStep
    model = JointCTCASR(ctc_weight=0.3, **bench.DIMS).to(dev); model.train()
    stepper = JointCTCTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
else:
    model = ASR(**bench.DIMS).to(dev); model.train()
    stepper = ASRTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
x, y, lens = config4_batch()
_, ans_len = label_geometry(y)
x, y = x.to(dev), y.to(dev)
print('batch: max frames %d, mean %.0f, label steps %d' % (max(lens), sum(lens) / len(lens), ans_len))
for i in range(steps):
    if i == steps - 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    stepper(x, y, lens, ans_len)
torch.cuda.synchronize()
stepper.finish()
print('last 3 steps: %.3f ms/step' % ((time.perf_counter() - t0) / 3 * 1e3))
