"""Per-launch start-up cost of the persistent recurrences: bench.recurrence_roofline at several layer
lengths, least-squares fit us(S) = c + a * S (the BPTT of a layer runs as several launches over step
ranges, so c is paid per range)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, bench
dev = torch.device('cuda', 0)
Ss = (30, 60, 120, 240, 400)
fw, bw = [], []
for S in Ss:
    bptt, fwd, _ = bench.recurrence_roofline(dev, S=S, N=32, H=256, reps=8)
    fw.append(fwd['us_per_launch']); bw.append(bptt['us_per_launch'])
    print('S=%4d  forward %.1f us  BPTT %.1f us' % (S, fw[-1], bw[-1]), flush=True)
for name, v in (('forward', fw), ('BPTT', bw)):
    a, c = np.polyfit(np.array(Ss, dtype=float), np.array(v), 1)
    print('%s: %.3f us per step + %.1f us per launch' % (name, a, c))
