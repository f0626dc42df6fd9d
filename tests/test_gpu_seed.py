"""GPU parity of the Seed loop's extra kernels (csrc/seed.hip; BASELINE.json configs[4]'s ADV and SAE legs)
against plain torch on the CPU, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TORCH_ACT = {None: lambda v: v, 'tanh': torch.tanh, 'relu': torch.relu, 'leaky_relu': F.leaky_relu,
             'sigmoid': torch.sigmoid}


@pytest.mark.parametrize('rows,K,N,act', [(37, 50, 1, 'sigmoid'), (96, 24, 24, 'relu'), (800, 512, 256, 'relu'),
                                           (3200, 768, 768, 'leaky_relu'), (3200, 768, 640, None), (130, 70, 33, 'tanh')])
def test_linear_matches_torch(rows, K, N, act):
    """ssasr_linear_fwd / _bwd (nn.Linear + activation, src/discriminator.py:38-52, src/speech_autoencoder.py:183-188)
    against torch in float64: output, input gradient, weight and bias gradients (accumulated into given buffers)."""
    from ss_asr_amd import seed_ops
    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(4, rows // 4, K, generator=g) if rows % 4 == 0 else torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(*x.shape[:-1], N, generator=g)
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = TORCH_ACT[act](F.linear(xr, wr, br))
    yr.backward(dy.double())
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, w, b))
    y = seed_ops.linear(xg, wg, bg, act)
    y.backward(dy.cuda())
    scale = lambda t: max(1.0, float(t.detach().abs().max()))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-6 * scale(yr)
    for got, want, what in ((xg.grad, xr.grad, 'dx'), (wg.grad, wr.grad, 'dw'), (bg.grad, br.grad, 'db')):
        err = float((got.cpu().double() - want).abs().max())
        assert err < 3e-6 * scale(want), (what, err, scale(want))
    # a frozen layer (the generator pass of ADVTrainer) gives the input gradient alone
    x2 = x.cuda().requires_grad_()
    seed_ops.linear(x2, wg.detach(), bg.detach(), act).backward(dy.cuda())
    assert float((x2.grad.cpu().double() - xr.grad).abs().max()) < 3e-6 * scale(xr.grad)


@pytest.mark.parametrize('n,target', [(8 * 25, 0.9), (32 * 100, 0.0), (32 * 100, 1.0), (7, 0.9)])
def test_bce_matches_torch(n, target):
    """ssasr_bce_fwd / _bwd against nn.BCELoss: ordinary probabilities, and saturated ones (the -100 clamp of the
    log terms and the 1e-12 floor of the derivative's denominator)."""
    from ss_asr_amd import seed_ops
    g = torch.Generator().manual_seed(n)
    p = torch.sigmoid(3.0 * torch.randn(n, generator=g))
    p[0], p[n - 1] = 0.0, 1.0                      # saturated scores
    pr = p.clone().requires_grad_()
    want = F.binary_cross_entropy(pr, torch.full((n,), target))
    want.backward()
    pg = p.cuda().requires_grad_()
    got = seed_ops.bce_loss(pg, target)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    np.testing.assert_allclose(pg.grad.cpu().numpy(), pr.grad.numpy(), rtol=2e-5, atol=1e-9)
