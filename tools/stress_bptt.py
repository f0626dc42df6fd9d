"""Race detector for the persistent LSTM recurrences: the forward and BPTT kernels have no
atomics, so every run over the same inputs must be bit-identical.  Reports, per shape, how many
of `runs` repetitions differ from the first and which time steps / directions differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ss_asr_amd import ops

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = 'cuda:0'
shapes = [(400, 16, 80, 256), (200, 16, 1024, 256), (100, 16, 1024, 256), (16, 100, 1024, 256), (800, 32, 80, 256)]
for (S, N, I, H) in shapes:
    g = torch.Generator().manual_seed(S + N)
    x = (torch.randn(S, N, I, generator=g) * 0.5).to(dev).requires_grad_(True)
    w = []
    for d in range(2):
        w += [(torch.randn(4 * H, I, generator=g) * I ** -0.5).to(dev).requires_grad_(True),
              (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(dev).requires_grad_(True),
              torch.zeros(4 * H, device=dev, requires_grad=True), torch.zeros(4 * H, device=dev, requires_grad=True)]
    lens = torch.randint(S // 2, S + 1, (N,), generator=g).to(torch.int32)
    lens[0] = S
    lens = lens.to(dev)
    dy = (torch.randn(S, N, 2 * H, generator=g) * 0.1).to(dev)
    first_y = first_dx = None
    bad_f = bad_b = 0
    where = []
    for r in range(runs):
        x.grad = None
        y = ops.bilstm(x, lens, S, False, tuple(w))
        y.backward(dy)
        torch.cuda.synchronize()
        if first_y is None:
            first_y, first_dx = y.detach().clone(), x.grad.clone()
            continue
        if not torch.equal(y.detach(), first_y):
            bad_f += 1
        if not torch.equal(x.grad, first_dx):
            bad_b += 1
            diff = (x.grad != first_dx).any(dim=2)        # [S, N]
            steps = diff.any(dim=1).nonzero().flatten().tolist()
            cols = diff.any(dim=0).nonzero().flatten().tolist()
            mag = float((x.grad - first_dx).abs().max() / first_dx.abs().max())
            if len(where) < 6:
                where.append((r, 'steps %d..%d (%d)' % (steps[0], steps[-1], len(steps)), 'cols', cols[:8], 'mag %.1e' % mag))
    ops.check_persistent_status()
    print('S=%d N=%d I=%d H=%d: forward differs in %d / %d runs, backward (dx) in %d' % (S, N, I, H, bad_f, runs - 1, bad_b), flush=True)
    for wline in where:
        print('   ', wline)
