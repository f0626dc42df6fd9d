"""Where a slow block of train steps comes from: blocks of 20 steps of the fixed 470-frame step, wall time per block,
beside the interpreter's garbage collections (generation, duration) that fell inside each block.
python tools/hiccup.py [blocks] [unfrozen]"""
import gc, os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import config2_batches
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
freeze = not (len(sys.argv) > 2 and sys.argv[2] == 'unfrozen')
dev = torch.device('cuda', 0)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
model = ASR(**bench.DIMS).to(dev); model.train()
stepper = ASRTrainStep(model, lr=1.0, eps=1e-8, grad_clip=5.0)
best = None
for x, y, lens in config2_batches(40, batch_size=32, feat_dim=80, seed=1, rank=0, hi=800):
    if best is None or abs(max(lens) - 470) < abs(max(best[2]) - 470):
        best = (x, y, lens)
x, y, lens = best
_, ans_len = label_geometry(y)
x, y = x.to(dev), y.to(dev)
for _ in range(8): stepper(x, y, lens, ans_len)
torch.cuda.synchronize()
events, t_start = [], [0.0]


def on_gc(phase, info):
    if phase == 'start':
        t_start[0] = time.perf_counter()
    else:
        events.append((info['generation'], (time.perf_counter() - t_start[0]) * 1e3, info['collected']))


gc.callbacks.append(on_gc)
if not freeze:
    gc.unfreeze()                     # (the step object froze the collector after its first step: undo, to show the hiccup)
for b in range(blocks):
    events.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): stepper(x, y, lens, ans_len)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    big = [e for e in events if e[0] == 2 or e[1] > 1.0]
    print('block %2d  %.3f ms/step  gc: %d collections, %.2f ms in all%s' % (
        b, ms, len(events), sum(e[1] for e in events), ''.join('  [gen %d %.1f ms, %d objects]' % e for e in big)), flush=True)
stepper.finish()
print('objects tracked by the collector: %d' % len(gc.get_objects()))
