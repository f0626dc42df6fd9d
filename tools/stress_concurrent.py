"""Train-step reproducibility with another tenant on the GPU: large rocBLAS products run on a
third stream while the forward and backward passes execute, which delays and scatters the
workgroup starts of the persistent kernels (as RCCL kernels or a second model would).
Forward results must stay bit-identical to a quiet run; gradients within summation-order noise."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ss_asr_amd import ops
from ss_asr_amd.asr import ASR
from ss_asr_amd.optim import FlatParameters

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = 'cuda:0'
for (B, T, U) in [(16, 400, 24), (32, 800, 40)]:
    torch.manual_seed(1)
    model = ASR(50, 256, 256, 128, 80, 0.9).to(dev)     # the bench workload's sizes
    flat = FlatParameters(model)
    g = torch.Generator().manual_seed(B)
    lens = sorted([int(v) for v in torch.randint(T // 2, T + 1, (B,), generator=g)], reverse=True)
    lens[0] = T
    x = torch.randn(B, T, 80, generator=g) * 0.5
    for b, l in enumerate(lens):
        x[b, l:] = 0
    y = torch.randint(1, 30, (B, U + 1), generator=g)
    y[:, 0] = 0
    x, y = x.to(dev), y.to(dev)
    a = torch.randn(4096, 4096, device=dev); bmat = torch.randn(4096, 4096, device=dev); c = torch.empty_like(a)
    bg = torch.cuda.Stream()

    def once(load):
        flat.zero_grad()
        random.seed(7); torch.manual_seed(7)
        if load:
            with torch.cuda.stream(bg):
                for _ in range(load):
                    torch.mm(a, bmat, out=c)
        _, logits, _ = model(x, U, teacher=y, state_len=lens)
        loss = ops.masked_ce_loss(logits, y, U)
        loss.backward()
        ops.join_side_stream(); torch.cuda.synchronize()
        return logits.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]

    ref_logits, ref_g = once(0)
    top = max(float(t.abs().max()) for t in ref_g)
    names = [n for n, _ in model.named_parameters()]
    bad_f = 0; worst = (0.0, '')
    for r in range(runs):
        logits, gr = once(4 + 4 * (r % 3))
        if not torch.equal(logits, ref_logits):
            bad_f += 1
        for n, u, v in zip(names, gr, ref_g):
            s = float(v.abs().max())
            if s < 1e-4 * top: continue
            d = float((u - v).abs().max()) / s
            if d > worst[0]: worst = (d, n)
    ops.check_persistent_status()
    print('B=%d T=%d: forward differs in %d / %d loaded runs; worst gradient deviation %.2e (%s)' %
          (B, T, bad_f, runs, worst[0], worst[1]), flush=True)
