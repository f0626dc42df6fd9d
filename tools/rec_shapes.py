"""Per-step time of the two persistent recurrences alone for a few column counts / layer
lengths (bench.recurrence_roofline at other shapes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
dev = torch.device('cuda', 0)
for (S, N) in ((400, 16), (400, 32), (400, 48), (400, 64), (100, 32), (800, 32)):
    bptt, fwd, _gemm = bench.recurrence_roofline(dev, S=S, N=N, H=256, reps=5)
    print('S=%4d N=%3d  forward %.3f us/step   BPTT %.3f us/step' % (S, N, fwd['us_per_step'], bptt['us_per_step']), flush=True)
