"""240 (SOAK_STEPS) train steps on eight bench batches; prints the loss every 40 steps, the last step's (grad norm,
skipped) and the number of NaN-skipped steps.  Run twice (SSASR_GEMM_X6=1 / 0) to compare the
trajectories of the two matrix-product forms: equal to 4 decimals for ~160 steps, then the usual
divergence of a sampled (tf_rate 0.9) training run from rounding-level differences."""
import os, sys, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config2_batches
dev = torch.device('cuda', 0)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
model = ASR(**bench.DIMS).to(dev); model.train()
st = ASRTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
bs = []
for x, y, lens in config2_batches(8, batch_size=32, feat_dim=80, seed=1, rank=0, hi=800):
    _, al = label_geometry(y); bs.append((x.to(dev), y.to(dev), lens, al))
out = []
for i in range(int(os.environ.get("SOAK_STEPS", "240"))):
    l = st(*bs[i % 8])
    if i % 40 == 39: out.append(round(float(l), 4))
print(os.environ.get('SSASR_GEMM_X6', '1'), out, st.finish(), st.skipped_steps)
# SOAK_LONG=n: n more steps on configs[3]'s batch (1500-3000 frames, ~300 label steps: the long decode loop and
# the six-slice backward chain).  A persistent kernel that gave up waiting raises from the step / finish().
n_long = int(os.environ.get("SOAK_LONG", "0"))
if n_long:
    from ss_asr_amd.synthetic import config4_batch
    x, y, lens = config4_batch()
    _, al = label_geometry(y)
    x, y = x.to(dev), y.to(dev)
    ls = []
    for i in range(n_long):
        l = st(x, y, lens, al)
        if i % 50 == 49: ls.append(round(float(l), 4))
    print('long', ls, st.finish(), st.skipped_steps)
