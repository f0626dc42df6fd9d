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


# ------------------------------------------------------------------ SAE: the speech autoencoder's kernels ----
def _lib_call():
    import ctypes as C
    from ss_asr_amd import _lib
    lib = _lib.load()
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return lib, p, st


@pytest.mark.parametrize('B,T,W,C,F,kh,kw', [
    (3, 40, 20, 1, 8, 1, 7),        # one kernel row over mel, C = 1: in place, unaligned windows (conv_1 of the yaml)
    (2, 30, 9, 32, 16, 5, 1),       # kernel rows as K segments, kw * C = 32 (conv_2)
    (2, 12, 9, 64, 24, 3, 1),       # kw * C = 64 (conv_3)
    (2, 25, 11, 1, 6, 7, 1),        # kernel over time with C = 1: im2col path (the class docstring's conv_1)
    (2, 14, 12, 6, 10, 1, 3),       # one kernel row, K = 18
    (2, 10, 8, 12, 20, 2, 2),       # both axes, kw * C = 24: im2col path
    (2, 16, 10, 16, 8, 3, 2),       # both axes, kw * C = 32: in place
])
def test_conv2d_matches_torch(B, T, W, C, F, kh, kw):
    """ssasr_conv2d_fwd / _bwd (channels-last; the overlapping windows read in place or through im2col) against
    torch.nn.functional.conv2d in float64: output, weight gradient (accumulated into a non-zero buffer), input
    gradient from the zero-bordered output gradient."""
    lib, p, st = _lib_call()
    g = torch.Generator().manual_seed(B * T + C * kh)
    x = torch.randn(B, T, W, C, generator=g)
    w = torch.randn(F, C, kh, kw, generator=g) / (C * kh * kw) ** 0.5
    To, Wo = T - kh + 1, W - kw + 1
    dy = torch.randn(B, To, Wo, F, generator=g)
    xr, wr = x.permute(0, 3, 1, 2).double().requires_grad_(), w.double().requires_grad_()
    yr = F_conv(xr, wr)
    yr.backward(dy.permute(0, 3, 1, 2).double())
    xg, wg = x.cuda(), w.cuda()
    ws = torch.empty(max(int(lib.ssasr_conv2d_ws_floats(B, T, W, C, F, kh, kw)), 4), device='cuda')
    y = torch.empty(B, To, Wo, F, device='cuda')
    assert lib.ssasr_conv2d_fwd(p(xg), p(wg), p(y), B, T, W, C, F, kh, kw, p(ws), st()) == 0
    want = yr.detach().permute(0, 2, 3, 1)
    assert float((y.cpu().double() - want).abs().max()) < 3e-6 * max(1.0, float(want.abs().max()))
    # backward: dense dy for the weight gradient alone, bordered dy for both
    dw0 = torch.randn(F, C, kh, kw, generator=g)
    dw = dw0.cuda()
    assert lib.ssasr_conv2d_bwd(p(dy.cuda()), 0, p(xg), p(wg), None, p(dw), B, T, W, C, F, kh, kw, p(ws), st()) == 0
    tol_w = 3e-6 * max(1.0, float(wr.grad.abs().max())) * max(1.0, (B * To * Wo) ** 0.5 / 8)
    assert float((dw.cpu().double() - dw0.double() - wr.grad).abs().max()) < tol_w
    dyb = torch.zeros(B, To + 2 * (kh - 1), Wo + 2 * (kw - 1), F)
    dyb[:, kh - 1:kh - 1 + To, kw - 1:kw - 1 + Wo] = dy
    dx = torch.empty(B, T, W, C, device='cuda')
    dw2 = torch.zeros(F, C, kh, kw, device='cuda')
    rc = lib.ssasr_conv2d_bwd(p(dyb.cuda()), 1, p(xg), p(wg), p(dx), p(dw2), B, T, W, C, F, kh, kw, p(ws), st())
    assert rc == 0
    assert float((dw2.cpu().double() - wr.grad).abs().max()) < tol_w
    want_dx = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu().double() - want_dx).abs().max()) < 3e-6 * max(1.0, float(want_dx.abs().max()))


def F_conv(x, w):
    return F.conv2d(x, w)


@pytest.mark.parametrize('B,T,W,C,ph,pw,border', [(3, 21, 9, 32, 3, 1, (4, 0)), (2, 13, 8, 64, 5, 1, (2, 0)),
                                                  (4, 50, 45, 16, 50, 40, (0, 0)), (2, 9, 11, 20, 2, 3, (1, 2)),
                                                  (2, 70, 40, 256, 64, 33, (2, 0)), (3, 7, 5, 6, 1, 1, (0, 0))])
def test_batch_norm_relu_pool_matches_torch(B, T, W, C, ph, pw, border):
    """ssasr_bn_stats + ssasr_bn_relu_pool_fwd / _bwd against BatchNorm2d (training mode) -> ReLU -> MaxPool2d in
    float64: pooled output, running statistics, gradient of the convolution output (inside a zero border),
    dgamma / dbeta (accumulated); then eval mode against the running statistics."""
    lib, p, st = _lib_call()
    g = torch.Generator().manual_seed(T * W + C)
    y = torch.randn(B, T, W, C, generator=g) * 1.5 + 0.3
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    rm0, rv0 = 0.1 * torch.randn(C, generator=g), 1.0 + 0.1 * torch.rand(C, generator=g)
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    yr = y.permute(0, 3, 1, 2).double().requires_grad_()
    pr = F.max_pool2d(torch.relu(bn(yr)), (ph, pw))
    dp = torch.randn(B, T // ph, W // pw, C, generator=g)
    pr.backward(dp.permute(0, 3, 1, 2).double())
    yg, gam, bet, rm, rv = (t.cuda() for t in (y, gamma, beta, rm0.clone(), rv0.clone()))
    ws = torch.empty(max(int(lib.ssasr_bn_ws_floats(C)), int(lib.ssasr_pool_ws_floats(B, T, W, C, ph, pw))), device='cuda')
    save = torch.empty(4 * C, device='cuda')
    assert lib.ssasr_bn_stats(p(yg), B * T * W, C, p(gam), p(bet), p(rm), p(rv), 0.1, 1e-5, 1, p(ws), p(save), st()) == 0
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), atol=2e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=3e-6)
    To, Wo = T // ph, W // pw
    pg = torch.empty(B, To, Wo, C, device='cuda')
    idx = torch.empty(B, To, Wo, C, device='cuda', dtype=torch.int32)
    assert lib.ssasr_bn_relu_pool_fwd(p(yg), p(save), B, T, W, C, ph, pw, p(pg), p(idx), p(ws), st()) == 0
    want = pr.detach().permute(0, 2, 3, 1)
    assert float((pg.cpu().double() - want).abs().max()) < 3e-6 * max(1.0, float(want.abs().max()))
    bt, bw = border
    dy = torch.full((B, T + 2 * bt, W + 2 * bw, C), float('nan'), device='cuda')
    dg0, db0 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dg, db = dg0.cuda(), db0.cuda()
    assert lib.ssasr_bn_relu_pool_bwd(p(dp.cuda()), p(pg), p(idx), p(yg), p(save), p(gam), B, T, W, C, ph, pw, bt, bw, p(dy),
                                      p(dg), p(db), p(ws), st()) == 0
    dy = dy.cpu()
    want_dy = yr.grad.permute(0, 2, 3, 1)
    scale = max(1.0, float(want_dy.abs().max()))
    assert float((dy[:, bt:bt + T, bw:bw + W].double() - want_dy).abs().max()) < 5e-6 * scale
    inner = torch.zeros_like(dy, dtype=torch.bool)
    inner[:, bt:bt + T, bw:bw + W] = True
    assert float(dy[~inner].abs().sum()) == 0.0 if (bt or bw) else True
    n_terms = max(1.0, (B * To * Wo) ** 0.5 / 8)
    assert float((dg.cpu().double() - dg0.double() - bn.weight.grad).abs().max()) < 5e-6 * n_terms * max(1.0, float(bn.weight.grad.abs().max()))
    assert float((db.cpu().double() - db0.double() - bn.bias.grad).abs().max()) < 5e-6 * n_terms * max(1.0, float(bn.bias.grad.abs().max()))
    # eval mode: the running statistics (left untouched)
    bn.eval()
    pe = F.max_pool2d(torch.relu(bn(y.permute(0, 3, 1, 2).double())), (ph, pw)).permute(0, 2, 3, 1)
    rm1, rv1 = rm.clone(), rv.clone()
    assert lib.ssasr_bn_stats(p(yg), B * T * W, C, p(gam), p(bet), p(rm), p(rv), 0.1, 1e-5, 0, p(ws), p(save), st()) == 0
    assert lib.ssasr_bn_relu_pool_fwd(p(yg), p(save), B, T, W, C, ph, pw, p(pg), p(idx), p(ws), st()) == 0
    assert torch.equal(rm, rm1) and torch.equal(rv, rv1)
    assert float((pg.cpu().double() - pe.detach()).abs().max()) < 3e-6 * max(1.0, float(pe.detach().abs().max()))


@pytest.mark.parametrize('ph,pw', [(2, 2), (50, 8)])
def test_batch_norm_relu_pool_lets_a_nan_through(ph, pw):
    """nn.ReLU and nn.MaxPool2d propagate NaN (the reference's loss is NaN when an activation is, and Solver.step then
    skips the update, src/trainer.py:131-148); the fused kernels -- the small-window form and the chunked large-window
    form -- must not turn it into 0.  Eval mode: with running statistics the NaN stays in its own value."""
    lib, p, st = _lib_call()
    g = torch.Generator().manual_seed(3)
    B, T, W, C = 2, 100, 16, 8
    y = torch.randn(B, T, W, C, generator=g)
    y[1, 57, 9, 3] = float('nan')
    gamma, beta = 0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
        want = F.max_pool2d(torch.relu(bn(y.permute(0, 3, 1, 2))), (ph, pw)).permute(0, 2, 3, 1)
    yg, gam, bet, rm, rv = (t.cuda() for t in (y, gamma, beta, rm0.clone(), rv0.clone()))
    ws = torch.empty(max(int(lib.ssasr_bn_ws_floats(C)), int(lib.ssasr_pool_ws_floats(B, T, W, C, ph, pw))), device='cuda')
    save = torch.empty(4 * C, device='cuda')
    assert lib.ssasr_bn_stats(p(yg), B * T * W, C, p(gam), p(bet), p(rm), p(rv), 0.1, 1e-5, 0, p(ws), p(save), st()) == 0
    pg = torch.empty(B, T // ph, W // pw, C, device='cuda')
    idx = torch.empty(B, T // ph, W // pw, C, device='cuda', dtype=torch.int32)
    assert lib.ssasr_bn_relu_pool_fwd(p(yg), p(save), B, T, W, C, ph, pw, p(pg), p(idx), p(ws), st()) == 0
    got = pg.cpu()
    assert int(torch.isnan(want).sum()) == 1 and torch.equal(torch.isnan(got), torch.isnan(want))
    ok = ~torch.isnan(want)
    assert float((got[ok] - want[ok]).abs().max()) < 1e-5


def test_frame_decoder_input_and_smooth_l1_match_torch():
    """ssasr_sae_concat_* and ssasr_smooth_l1_* against torch: [listener frame | global encoding] rows with the
    broadcast's sum in backward; the loss of src/trainer.py:811-818 with its zero padding up to batch_t (R < bt:
    the padded rows count in the mean and pass no gradient), both branches of the Huber law."""
    from ss_asr_amd import seed_ops
    g = torch.Generator().manual_seed(5)
    B, Tq, L, G, Fd = 3, 7, 24, 10, 6
    lis, enc = torch.randn(B, Tq, L, generator=g), torch.randn(B, G, generator=g)
    lr, er = lis.double().requires_grad_(), enc.double().requires_grad_()
    dr = torch.cat((lr, er.unsqueeze(1).expand(-1, Tq, -1)), dim=2)
    up = torch.randn(B, Tq, L + G, generator=g)
    dr.backward(up.double())
    lg, eg = lis.cuda().requires_grad_(), enc.cuda().requires_grad_()
    dg = seed_ops.sae_concat(lg, eg)
    dg.backward(up.cuda())
    assert torch.equal(dg.detach().cpu(), dr.detach().float())
    assert torch.equal(lg.grad.cpu(), lr.grad.float())
    np.testing.assert_allclose(eg.grad.cpu().numpy(), er.grad.numpy(), atol=2e-6)
    R, bt, Tx = 8 * Tq, 8 * Tq + 5, 8 * Tq + 9
    pred = 2.0 * torch.randn(B, R, Fd, generator=g)
    x = torch.randn(B, Tx, Fd, generator=g)
    prr = pred.double().requires_grad_()
    full = torch.zeros(B, bt, Fd, dtype=torch.float64)
    full[:, :R] = prr
    want = F.smooth_l1_loss(full, x[:, :bt].double())
    (3.0 * want).backward()
    pg = pred.cuda().requires_grad_()
    got = seed_ops.sae_loss(pg, x.cuda(), bt)
    (3.0 * got).backward()
    assert abs(float(got) - float(want)) < 2e-6 * max(1.0, float(want))
    np.testing.assert_allclose(pg.grad.cpu().numpy(), prr.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_speech_autoencoder_module_matches_the_oracle_module():
    """speech_autoencoder.SpeechAutoEncoder against the oracle's module from the same state_dict: the training-mode
    forward (and the running statistics it leaves), `just_first` (src/speech_autoencoder.py:66-67: the first Listener
    frame alone), the eval-mode forward, and a kernel that does not fit its input (torch raises there too)."""
    import las_oracle as lo
    from ss_asr_amd.speech_autoencoder import SpeechAutoEncoder
    cfg = ([[1, 5], [3, 1], [2, 2]], [8, 12, 16], [[2, 1], [3, 1], [4, 3]])
    ref = lo.OracleSpeechAutoEncoder(24, 12, *cfg)
    lo.seeded_generic_weights(ref, 77)
    mine = SpeechAutoEncoder(24, 12, *cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    g = torch.Generator().manual_seed(3)
    # 36 frames x 12 mels: conv [1, 5] -> 36 x 8, pool [2, 1] -> 18 x 8, conv [3, 1] -> 16 x 8, pool [3, 1] -> 5 x 8,
    # conv [2, 2] -> 4 x 7; a last window of [4, 3] leaves 1 x 2 values per filter (refused), [4, 7] leaves one
    x = torch.randn(3, 36, 12, generator=g)
    lis = torch.randn(3, 4, 24, generator=g)
    with pytest.raises(RuntimeError):
        mine(x.cuda(), lis.cuda())                       # last pooling leaves 1 x 2 values per filter
    cfg = ([[1, 5], [3, 1], [2, 2]], [8, 12, 16], [[2, 1], [3, 1], [4, 7]])
    ref = lo.OracleSpeechAutoEncoder(24, 12, *cfg)
    lo.seeded_generic_weights(ref, 78)
    mine = SpeechAutoEncoder(24, 12, *cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    for kw in ({}, {'just_first': True}):
        want = ref(x, lis, **kw)
        got = mine(x.cuda(), lis.cuda(), **kw)
        assert got.shape == want.shape
        assert float((got.detach().cpu() - want.detach()).abs().max()) < 2e-5
    for k, v in ref.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            np.testing.assert_allclose(mine.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    ref.eval(); mine.eval()
    with torch.no_grad():
        assert float((mine(x.cuda(), lis.cuda()).cpu() - ref(x, lis)).abs().max()) < 2e-5
    with pytest.raises(RuntimeError):
        mine(x[:, :5].cuda(), lis.cuda())                # 5 frames: nothing left for the second block
