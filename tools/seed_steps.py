"""A few steps of ONE leg of the Seed loop at the bench shape (bench.config5_bench's models and batches), for a
profiler: `rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sae -- python3 tools/seed_steps.py sae 10`.
Legs: tae, adv, sae, or `round`: all three in the reference's order on the one ASR object, cycling over four of the
corpus' batches (a soak: every 50 rounds the step objects' status words are collected -- a hand-off time-out raises).
Prints the wall time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.discriminator import Discriminator
from ss_asr_amd.engine import ADVTrainStep, SAETrainStep, TAETrainStep
from ss_asr_amd.speech_autoencoder import SpeechAutoEncoder
from ss_asr_amd.synthetic import config2_batches
from ss_asr_amd.text_autoencoder import TextAutoEncoder

leg = sys.argv[1] if len(sys.argv) > 1 else 'sae'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
torch.manual_seed(5)
D = bench.DIMS
asr = ASR(**D).to(dev)
tae = TextAutoEncoder(D['output_dim'], emb_dim=128, state_size=D['encoder_state_size'], num_layers=2).to(dev)
x, y, lens = config2_batches(4, batch_size=32, feat_dim=D['feature_dim'], seed=1)[0]
x800 = torch.zeros(x.shape[0], 800, x.shape[2])
x800[:, :x.shape[1]] = x
x, y, x800 = x.to(dev), y.to(dev), x800.to(dev)
if leg == 'sae':
    sae = SpeechAutoEncoder(asr.encoder.out_dim, D['feature_dim'], [[1, 36], [5, 1], [3, 1]], [32, 64, 256],
                            [[3, 1], [5, 1], [50, 40]]).to(dev)
    step = SAETrainStep(asr, sae)
    run = lambda: step(x800, lens)
elif leg == 'adv':
    step = ADVTrainStep(asr, tae, Discriminator(asr.encoder.get_outdim(), 256).to(dev))
    run = lambda: step(x, lens, y)
elif leg == 'round':
    sae = SpeechAutoEncoder(asr.encoder.out_dim, D['feature_dim'], [[1, 36], [5, 1], [3, 1]], [32, 64, 256],
                            [[3, 1], [5, 1], [50, 40]]).to(dev)
    steps3 = (TAETrainStep(asr, tae), ADVTrainStep(asr, tae, Discriminator(asr.encoder.get_outdim(), 256).to(dev)),
              SAETrainStep(asr, sae))
    data = []
    for bx, by, bl in config2_batches(4, batch_size=32, feat_dim=D['feature_dim'], seed=1):
        b800 = torch.zeros(bx.shape[0], 800, bx.shape[2])
        b800[:, :bx.shape[1]] = bx
        data.append((bx.to(dev), by.to(dev), bl, [int(v) + 1 for v in (by != 0).sum(-1)], b800.to(dev)))
    count = [0]

    class step:                                    # (finish() of all three)
        @staticmethod
        def finish():
            for s3 in steps3:
                s3.finish()

    def run():
        bx, by, bl, yl, b800 = data[count[0] % len(data)]
        steps3[0](by, by, yl, yl)
        steps3[1](bx, bl, by)
        loss = steps3[2](b800, bl)
        count[0] += 1
        if count[0] % 50 == 0:
            step.finish()
            print('round %d: sae loss %.4f' % (count[0], float(loss)), flush=True)
else:
    step = TAETrainStep(asr, tae)
    y_lens = [int(v) + 1 for v in (y != 0).sum(-1)]
    run = lambda: step(y, y, y_lens, y_lens)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run()
torch.cuda.synchronize()
step.finish()
print('%s: %.3f ms per step over %d steps' % (leg, (time.perf_counter() - t0) / steps * 1e3, steps))
