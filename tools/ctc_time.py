"""Times the CTC branch's two kernels (ssasr_ctc_loss_fwd / _bwd) at the encoder geometries of
BASELINE.json configs[1] (T' = 100, <= 60 characters) and configs[3] (T' = 375, <= 300)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ss_asr_amd import ops  # noqa: E402

d = torch.device('cuda:0')
for B, T, V, lmax in ((32, 100, 50, 60), (32, 375, 50, 300), (16, 375, 50, 300)):
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(B, T, V, generator=g)).to(d).requires_grad_(True)
    frame_lens = torch.randint(T // 2, T + 1, (B,), generator=g).to(d, torch.int32)
    label_lens = torch.randint(lmax // 2, lmax + 1, (B,), generator=g).to(d, torch.int32)
    y = torch.randint(2, V, (B, lmax), generator=g).to(d, torch.int32)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(12):
        logits.grad = None
        ev[0].record()
        loss = ops.ctc_loss(logits, frame_lens, y, label_lens, lmax)
        ev[1].record()
        loss.backward()
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print('B=%d T=%d Lmax=%d: forward %.1f us, backward %.1f us, loss %.3f' % (B, T, lmax, tf * 100, tb * 100, float(loss)))
