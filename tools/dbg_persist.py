import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch, las_oracle as lo
from ss_asr_amd import ops
torch.manual_seed(0)
def rnd(*s, scale=1.0): return torch.randn(*s, dtype=torch.float64) * scale
for (N, T, I, H, lens) in [(3, 10, 80, 256, [10, 8, 5]), (32, 40, 80, 256, list(range(40, 8, -1))), (16, 12, 64, 64, [12]*16)]:
    x = rnd(N, T, I)
    for i, l in enumerate(lens): x[i, l:] = 0
    w = []
    for d in range(2): w += [rnd(4*H, I, scale=I**-0.5), rnd(4*H, H, scale=H**-0.5), rnd(4*H, scale=0.1), rnd(4*H, scale=0.1)]
    S = max(lens)
    yr = lo.bilstm_explicit(x[:, :S].transpose(0, 1), lens, w).transpose(0, 1)
    ld = torch.tensor(lens, dtype=torch.int32, device='cuda')
    yd = ops.bilstm(x.float().cuda(), ld, S, True, [t.float().cuda() for t in w])
    torch.cuda.synchronize()
    ops.check_persistent_status()
    err = (yd.cpu().double() - yr).abs()
    print('N=%d S=%d H=%d max err %.3e' % (N, S, H, err.max().item()), 'per-step max err fwd dir:', [round(err[:, s, :H].max().item(), 6) for s in range(min(S, 6))], 'bwd dir:', [round(err[:, s, H:].max().item(), 6) for s in range(min(S, 6))])
