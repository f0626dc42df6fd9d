"""Kernel-level parity: each C-ABI entry point against the CPU oracle
(oracle/las_oracle.py, float64 restatement) on seeded inputs.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import las_oracle as lo

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def close(got, want, atol, what=''):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    tol = atol * max(1.0, want.abs().max().item())     # absolute for O(1) data, relative above
    assert err <= tol, '%s: max abs err %.3e > %.1e' % (what, err, tol)


# ----------------------------------------------------------------- GEMM ----
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(64, 64, 16), (200, 136, 80), (129, 50, 37), (1024, 256, 300),
                                   (33, 1024, 1024)])
def test_gemm_orientations(ta, tb, M, N, K):
    from ss_asr_amd import ops
    a = rnd(K, M, seed=1) if ta else rnd(M, K, seed=1)
    b = rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)
    want = (a.t() if ta else a) @ (b if tb else b.t())
    got = ops.gemm(a.float().to(dev()), b.float().to(dev()), ta=bool(ta), tb=bool(tb))
    close(got, want, 2e-4 * max(1.0, K ** 0.5 / 4), 'gemm %d%d %dx%dx%d' % (ta, tb, M, N, K))


def test_gemm_is_exact_fp32_fma_chain():
    """A = I with an asymmetric B catches a transposed C write; small integers
    make every product exact."""
    from ss_asr_amd import ops
    n = 96
    a = torch.eye(n)
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 17) - 3.0
    got = ops.gemm(a.to(dev()), b.to(dev()), tb=True)
    assert torch.equal(got.cpu(), b)
    got = ops.gemm(a.to(dev()), b.to(dev()), tb=False)
    assert torch.equal(got.cpu(), b.t())


@pytest.mark.parametrize('ta,tb,M,N,K,nb', [(0, 0, 9000, 512, 1024, 2),      # 288 wide tiles on 256 runs: 32 tiles cut in two
                                            (0, 1, 8192, 1280, 512, 1),     # NN, 320 tiles
                                            (1, 1, 4608, 1024, 288, 2),     # TT, 288 tiles of 9 K steps
                                            (0, 0, 5000, 1000, 992, 3)])    # edge tiles in M and N, 480 tiles
def test_gemm_wide_stream_k_grid_equals_the_tile_kernels(ta, tb, M, N, K, nb):
    """The wide kernel's stream-K grid (csrc/gemm.hip, gemm_x6w_kernel: equal runs of K steps per CU; a tile cut between two
    runs goes through a slab and is finished by the part that arrives last) against float64 and against the tile kernels,
    at shapes whose tile count is not a multiple of the CU count -- with a bias, which the last arriver's epilogue adds once.
    Then the same product with its parts ADDED atomically (split-K callers: every part an atomic add)."""
    from ss_asr_amd import _lib, ops
    lib = _lib.load()
    a = rnd(nb, K, M, seed=11) if ta else rnd(nb, M, K, seed=11)
    b = rnd(nb, K, N, seed=12) if tb else rnd(nb, N, K, seed=12)
    bias = rnd(N, seed=13)
    want = (a.transpose(1, 2) if ta else a) @ (b if tb else b.transpose(1, 2)) + bias
    ad, bd, biasd = a.float().to(dev()), b.float().to(dev()), bias.float().to(dev())
    try:
        assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', 256) == 0
        wide = ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=biasd)
        again = ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=biasd)
        acc = torch.zeros(nb, M, N, device=dev())
        ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), out=acc, splitk=4)
        assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', 128) == 0
        tiles = ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=biasd)
    finally:
        lib.ssasr_set_option(b'SSASR_GEMM_TILE', 0)
    tol = 2e-4 * max(1.0, K ** 0.5 / 4)
    close(wide, want, tol, 'wide stream-K %d%d %dx%dx%d' % (ta, tb, M, N, K))
    assert torch.equal(wide, again)                       # two parts per cut tile: the sum does not depend on who arrives last
    close(acc, want - bias, tol, 'wide stream-K, parts added')
    assert float((wide - tiles).abs().max()) < 1e-4 * float(tiles.abs().max())


def test_gemm_stream_k_launches_on_several_streams_do_not_share_a_workspace():
    """The stream-K grid's tickets and slabs (csrc/gemm.hip, launch_wide) belong to the stream that first asked for
    them: products in flight on six streams at once -- two more than there are regions, so the last two take the
    tile kernels -- each give their own result, twice over."""
    from ss_asr_amd import _lib, ops
    lib = _lib.load()
    M, N, K, nb = 9000, 512, 1024, 2                      # 288 wide tiles on 256 runs: 32 tiles cut in two
    streams = [torch.cuda.Stream() for _ in range(6)]
    ops_in = [(rnd(nb, M, K, seed=40 + i).float().to(dev()), rnd(nb, N, K, seed=60 + i).float().to(dev())) for i in range(6)]
    torch.cuda.synchronize()
    try:
        assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', 256) == 0
        outs = []
        for rep in range(2):
            for s, (a, b) in zip(streams, ops_in):
                with torch.cuda.stream(s):
                    outs.append(ops.gemm(a, b, ta=False, tb=False))
        torch.cuda.synchronize()
        assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', 128) == 0
        for i, (a, b) in enumerate(ops_in):
            want = ops.gemm(a, b, ta=False, tb=False)
            for rep in range(2):
                got = outs[rep * 6 + i]
                assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max()), (i, rep)
    finally:
        lib.ssasr_set_option(b'SSASR_GEMM_TILE', 0)


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('tile', [64, 128])
def test_gemm_bf16_variant_is_the_product_of_the_rounded_operands(ta, tb, tile):
    """SSASR_GEMM_BF16=1 (csrc/gemm.hip, the tile kernels with ONE plane per operand): the result is the fp32-accumulated
    product of the operands rounded to bf16 (round to nearest even) -- checked against float64 on the rounded operands at
    the fp32 kernels' tolerance, with edge tiles in M and N, a K that is not a multiple of the K step, a bias, and the
    split-K form (parts added atomically).  The default stays the exact three-plane split (the test below)."""
    from ss_asr_amd import _lib, ops
    lib = _lib.load()
    M, N, K, nb = 333, 200, 1000, 2
    a = rnd(nb, K, M, seed=21) if ta else rnd(nb, M, K, seed=21)
    b = rnd(nb, K, N, seed=22) if tb else rnd(nb, N, K, seed=22)
    bias = rnd(N, seed=23)
    ar, br = a.float().bfloat16().double(), b.float().bfloat16().double()
    want = (ar.transpose(1, 2) if ta else ar) @ (br if tb else br.transpose(1, 2))
    ad, bd, biasd = a.float().to(dev()), b.float().to(dev()), bias.float().to(dev())
    try:
        assert lib.ssasr_set_option(b'SSASR_GEMM_BF16', 1) == 0
        assert lib.ssasr_set_option(b'SSASR_GEMM_TILE', tile) == 0
        got = ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=biasd)
        acc = torch.zeros(nb, M, N, device=dev())
        ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), out=acc, splitk=3)
        assert lib.ssasr_set_option(b'SSASR_GEMM_BF16', 0) == 0
        exact = ops.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=biasd)
    finally:
        lib.ssasr_set_option(b'SSASR_GEMM_BF16', 0)
        lib.ssasr_set_option(b'SSASR_GEMM_TILE', 0)
    def worst(x, y):
        return float((x.double().cpu() - y).abs().max())
    # |products| ~ 1, sums ~ 30: fp32 accumulation lands within ~1e-4 of float64; the operands' rounding moves the
    # result by ~0.1 -- a variant that kept more (or less) than the rounded operands would not pass both bounds
    assert worst(got, want + bias) < 2e-3, 'bf16 variant %d%d tile %d' % (ta, tb, tile)
    assert worst(acc, want) < 2e-3, 'bf16 variant, split-K'
    full = (a.transpose(1, 2) if ta else a) @ (b if tb else b.transpose(1, 2)) + bias
    assert worst(exact, full) < 2e-3, 'default after the switch is cleared'
    assert worst(got, full) > 2e-2


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_on_bf16_pieces_is_as_exact_as_the_fp32_instruction(ta, tb):
    """The fp32 product as six bf16 MFMAs on the exact three-way split of both operands
    (csrc/gemm.hip, SSASR_GEMM_X6) against the same product on v_mfma_f32_16x16x4_f32, both
    measured against float64: wide-range operands (magnitudes over 12 decades, so that an
    operand's low pieces matter), K = 1000, edge tiles in M and N.  The split form must not be
    less exact than the fp32 instruction's own rounding."""
    from ss_asr_amd import _lib, ops
    M, N, K = 200, 136, 1000
    g = torch.Generator().manual_seed(77)
    def wide(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float64) * torch.exp(
            6.0 * torch.randn(*shape, generator=g, dtype=torch.float64))
    a = wide(K, M) if ta else wide(M, K)
    b = wide(K, N) if tb else wide(N, K)
    af, bf = a.float(), b.float()
    want = (af.double().t() if ta else af.double()) @ (bf.double() if tb else bf.double().t())
    scale = ((af.double().abs().t() if ta else af.double().abs()) @
             (bf.double().abs() if tb else bf.double().abs().t()))           # sum |a_k b_k| per element
    errs = {}
    for mode in (0, 1):
        old = _lib.set_option('SSASR_GEMM_X6', mode)
        try:
            got = ops.gemm(af.to(dev()), bf.to(dev()), ta=bool(ta), tb=bool(tb))
        finally:
            _lib.set_option('SSASR_GEMM_X6', old)
        errs[mode] = float(((got.double().cpu() - want).abs() / scale).max())
    assert errs[0] < 2e-6, errs               # the fp32 instruction: ~sqrt(K) roundings of 6e-8
    assert errs[1] < max(2 * errs[0], 5e-7), errs


def test_gemm_bias_tanh_beta_batched_splitk():
    from ss_asr_amd import ops
    a, b, bias = rnd(300, 72, seed=3), rnd(40, 72, seed=4), rnd(40, seed=5)
    got = ops.gemm(a.float().to(dev()), b.float().to(dev()), bias=bias.float().to(dev()), act=1)
    close(got, torch.tanh(a @ b.t() + bias), 1e-5, 'bias+tanh')
    c0 = rnd(300, 40, seed=6)
    out = c0.float().to(dev())
    ops.gemm(a.float().to(dev()), b.float().to(dev()), out=out, alpha=0.5, beta=2.0)
    close(out, 0.5 * (a @ b.t()) + 2.0 * c0, 1e-4, 'alpha/beta')
    ab, bb = rnd(5, 70, 33, seed=7), rnd(5, 20, 33, seed=8)
    got = ops.gemm(ab.float().to(dev()), bb.float().to(dev()))
    close(got, ab @ bb.transpose(1, 2), 1e-4, 'batched')
    a, b = rnd(5000, 48, seed=9), rnd(5000, 24, seed=10)
    got = ops.gemm(a.float().to(dev()), b.float().to(dev()), ta=True, tb=True, splitk=8)
    close(got, a.t() @ b, 2e-3, 'split-K')


# --------------------------------------------------------------- BiLSTM ----
def lstm_weights(I, H, seed):
    w = []
    for d in range(2):
        w += [rnd(4 * H, I, seed=seed + 10 * d, scale=I ** -0.5),
              rnd(4 * H, H, seed=seed + 10 * d + 1, scale=H ** -0.5),
              rnd(4 * H, seed=seed + 10 * d + 2, scale=0.1),
              rnd(4 * H, seed=seed + 10 * d + 3, scale=0.1)]
    return w


@pytest.mark.parametrize('N,T,I,H,lens', [
    (4, 12, 12, 32, [12, 9, 9, 4]),
    (5, 9, 20, 16, [7, 6, 3, 2, 1]),          # input longer than max(lens)
    (33, 6, 8, 16, [6] * 20 + [3] * 13),      # more than one 32-column chunk
    (3, 10, 80, 256, [10, 8, 5]),             # production widths (vector path)
    # 48 / 64 columns at H = 256: the forward takes two 32-column chunks of 256 workgroups in all,
    # the second chunk of N = 48 half empty (its padded lanes once aliased chunk 0's column, which is
    # what timed out under an exchange ring: DESIGN.md 4.2); the BPTT runs 3 / 4 chunks without halves
    (48, 40, 64, 256, [40] * 10 + [33] * 20 + [9] * 18),
    (64, 24, 64, 256, [24] * 30 + [11] * 34),
    # H = 256, N <= 32: full chunks with ragged lengths; a last chunk of 13 live columns; the fused 80-bin input
    # projection (I = 80) with 13 columns
    (32, 37, 64, 256, [37 - k for k in range(32)]),
    (29, 21, 48, 256, [21] * 9 + [14] * 12 + [2] * 8),
    (13, 26, 80, 256, [26, 26, 25, 20, 19, 19, 12, 9, 9, 5, 3, 2, 1]),
    # more utterances than a persistent launch takes: two column windows (128 + 22) WITH lengths, fused 80-bin input;
    # and a two-column last window at H = 64
    (150, 40, 80, 256, sorted([40 - (k * 7) % 33 for k in range(150)], reverse=True)),
    (130, 36, 48, 64, sorted([36 - (k * 5) % 30 for k in range(130)], reverse=True)),
])
def test_bilstm_packed_forward_backward(N, T, I, H, lens):
    from ss_asr_amd import ops
    x = rnd(N, T, I, seed=11)
    for i, l in enumerate(lens):
        x[i, l:] = 0
    w = lstm_weights(I, H, 20)
    S = max(lens)
    # oracle (float64, autograd)
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in w]
    yr = lo.bilstm_explicit(xr[:, :S].transpose(0, 1), lens, wr).transpose(0, 1)
    gy = rnd(N, S, 2 * H, seed=12)
    (yr * gy).sum().backward()
    # HIP
    xd = x.float().to(dev()).requires_grad_(True)
    wd = [t.float().to(dev()).requires_grad_(True) for t in w]
    ld = torch.tensor(lens, dtype=torch.int32, device=dev())
    yd = ops.bilstm(xd, ld, S, True, wd)
    close(yd, yr, 2e-5, 'y')
    (yd * gy.float().to(dev())).sum().backward()
    close(xd.grad, xr.grad, 5e-5, 'dx')
    for name, a, b in zip(['w_ih', 'w_hh', 'b_ih', 'b_hh'] * 2, wd, wr):
        close(a.grad, b.grad, 2e-4, 'd' + name)


@pytest.mark.parametrize('fused', [1, 0])
@pytest.mark.parametrize('N,S,I,H,segments', [(20, 150, 24, 64, 4), (9, 130, 16, 128, 3), (20, 150, 24, 64, 1),
                                              (32, 96, 80, 256, 4),        # the first layer's shape: I = 80 (a partly filled column tile)
                                              (16, 64, 1024, 256, 2),      # layers 2-3: I = 1024
                                              (140, 130, 24, 64, 3)])      # two column windows x three step ranges (ring regions and dc_state per window)
def test_bilstm_segmented_bptt_with_overlapped_weight_gradients(N, S, I, H, segments, fused, monkeypatch):
    """The path the train step takes: gradients live in an optimizer-owned flat buffer, so the
    BPTT runs in step-range segments (ring and dc state carried across launches) and the weight
    gradients are accumulated from a second stream, range by range -- as ONE launch per range
    (SSASR_WGRAD_FUSED, default: dG^T . [X | H_prev] as the column segments of one GEMM launch, the
    bias gradients as column sums of the dG rows its first column tile streams) or as the separate
    products and column sums."""
    from ss_asr_amd import _lib, ops
    from ss_asr_amd.optim import FlatParameters
    monkeypatch.setattr(ops, 'bptt_segments', segments)
    old_fused = _lib.set_option('SSASR_WGRAD_FUSED', fused)
    try:
        _segmented_bptt_case(N, S, I, H)
    finally:
        _lib.set_option('SSASR_WGRAD_FUSED', old_fused)


def _segmented_bptt_case(N, S, I, H):
    from ss_asr_amd import ops
    from ss_asr_amd.optim import FlatParameters
    lens = sorted(np.random.default_rng(3).integers(S // 3, S + 1, size=N).tolist(), reverse=True)
    lens[0] = S
    x = rnd(N, S, I, seed=51)
    for i, l in enumerate(lens):
        x[i, l:] = 0
    w = lstm_weights(I, H, 60)
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in w]
    yr = lo.bilstm_explicit(xr.transpose(0, 1), lens, wr).transpose(0, 1)
    gy = rnd(N, S, 2 * H, seed=52)
    (yr * gy).sum().backward()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(t.float().to(dev())) for t in w])
    holder = Holder()
    flat = FlatParameters(holder)
    flat.grad.fill_(0.25)                                   # gradients are ACCUMULATED into the buffer
    xd = x.float().to(dev()).requires_grad_(True)
    ld = torch.tensor(lens, dtype=torch.int32, device=dev())
    yd = ops.bilstm(xd, ld, S, True, list(holder.ps))
    close(yd, yr, 2e-5, 'y')
    (yd * gy.float().to(dev())).sum().backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    ops.check_persistent_status()
    close(xd.grad, xr.grad, 5e-5, 'dx')
    for name, a, b in zip(['w_ih', 'w_hh', 'b_ih', 'b_hh'] * 2, holder.ps, wr):
        close(a.grad - 0.25, b.grad, 3e-4, 'd' + name)


@pytest.mark.parametrize('pct,inline,want_dx', [(10, 1, False), (100, 1, False), (60, 0, False), (35, 1, True)])
def test_bilstm_segmented_bptt_range_options(pct, inline, want_dx, monkeypatch):
    """The shape of the step ranges is a tuning choice, not part of the result: a short or an equal last range
    (SSASR_LAST_SEG_PCT), the last range's weight gradients on the main stream (SSASR_TAIL_INLINE; taken by a
    layer that is asked for no input gradient, as the first layer of the Listener) or on the second one."""
    from ss_asr_amd import ops, _lib
    from ss_asr_amd.optim import FlatParameters
    monkeypatch.setattr(ops, 'bptt_segments', 4)
    N, S, I, H = 20, 150, 24, 64
    lens = sorted(np.random.default_rng(5).integers(S // 3, S + 1, size=N).tolist(), reverse=True)
    lens[0] = S
    x = rnd(N, S, I, seed=71)
    for i, l in enumerate(lens):
        x[i, l:] = 0
    w = lstm_weights(I, H, 72)
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in w]
    yr = lo.bilstm_explicit(xr.transpose(0, 1), lens, wr).transpose(0, 1)
    gy = rnd(N, S, 2 * H, seed=73)
    (yr * gy).sum().backward()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(t.float().to(dev())) for t in w])
    holder = Holder()
    flat = FlatParameters(holder)
    flat.grad.zero_()
    old = (_lib.set_option('SSASR_LAST_SEG_PCT', pct), _lib.set_option('SSASR_TAIL_INLINE', inline))
    try:
        xd = x.float().to(dev()).requires_grad_(want_dx)
        ld = torch.tensor(lens, dtype=torch.int32, device=dev())
        yd = ops.bilstm(xd, ld, S, True, list(holder.ps))
        (yd * gy.float().to(dev())).sum().backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        ops.check_persistent_status()
    finally:
        _lib.set_option('SSASR_LAST_SEG_PCT', old[0])
        _lib.set_option('SSASR_TAIL_INLINE', old[1])
    if want_dx:
        close(xd.grad, xr.grad, 5e-5, 'dx')
    for name, a, b in zip(['w_ih', 'w_hh', 'b_ih', 'b_hh'] * 2, holder.ps, wr):
        close(a.grad, b.grad, 3e-4, 'd' + name)


@pytest.mark.parametrize('S,N,I,H', [(7, 40, 64, 16),
                                     (12, 150, 64, 64),       # two column windows of the persistent recurrences (128 + 22)
                                     (32, 375, 96, 256),      # blstm_4 at BASELINE.json configs[3]: 375 encoder frames = 128 + 128 + 119
                                     (9, 129, 32, 128)])      # a one-column last window
def test_bilstm_sequence_major_no_lengths(S, N, I, H):
    """blstm_4 form: recurrence over dim 0, every column full length.  More than 128 columns run as consecutive
    persistent launches over column windows of the same buffers (csrc/rnn.hip, PERSIST_WINDOW), forward and BPTT,
    with ONE input projection, input gradient and weight-gradient pass over all columns."""
    from ss_asr_amd import _lib, ops
    if N > 128 and H % 64 == 0:
        assert int(_lib.load().ssasr_bilstm_fwd_hx_floats(S, N, H)) > 0 and int(_lib.load().ssasr_bilstm_tsave_floats(S, N, H)) > 0
    x = rnd(S, N, I, seed=13)
    w = lstm_weights(I, H, 40)
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in w]
    yr = lo.bilstm_explicit(xr, None, wr)
    gy = rnd(S, N, 2 * H, seed=14)
    (yr * gy).sum().backward()
    xd = x.float().to(dev()).requires_grad_(True)
    wd = [t.float().to(dev()).requires_grad_(True) for t in w]
    yd = ops.bilstm(xd, None, S, False, wd)
    close(yd, yr, 2e-5, 'y')
    (yd * gy.float().to(dev())).sum().backward()
    torch.cuda.synchronize()
    ops.check_persistent_status()
    close(xd.grad, xr.grad, 5e-5, 'dx')
    for a, b in zip(wd, wr):
        close(a.grad, b.grad, 2e-4, 'dw')


def test_a_missing_producer_times_out_drains_and_is_reported():
    """Fault injection (SSASR_TEST_DROP_TILE): one unit tile of the persistent forward recurrence
    never publishes its h.  Its consumers must give up after their bounded spins, the rest of the
    launch must drain through the latch instead of spinning out every later step (32 steps here:
    seconds, not minutes), and the status word must say which kernel and step gave up first."""
    import time
    from ss_asr_amd import _lib, ops
    N, S, I, H = 8, 32, 64, 256
    x = rnd(N, S, I, seed=81).float().to(dev())
    w = [t.float().to(dev()) for t in lstm_weights(I, H, 82)]
    ops.check_persistent_status()
    old = _lib.set_option('SSASR_TEST_DROP_TILE', 5)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.bilstm(x, None, S, True, w)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        _lib.set_option('SSASR_TEST_DROP_TILE', old)
    with pytest.raises(RuntimeError, match='encoder forward recurrence.*step 1'):
        ops.check_persistent_status()
    assert elapsed < 20.0, elapsed
    # and the next launch is healthy again
    y = ops.bilstm(x, None, S, True, w)
    ops.check_persistent_status()
    assert bool(torch.isfinite(y).all())


# ------------------------------------------------------------ attention ----
@pytest.mark.parametrize('B,T,A,E,D,lens', [
    (4, 8, 16, 64, 32, [8, 7, 6, 5]),
    (3, 100, 128, 512, 256, [100, 57, 1]),
    (2, 375, 128, 512, 256, [375, 200]),
    # split-T form (T > 128): slices past an utterance's length, a one-frame utterance, a length on a
    # slice edge, and the shapes of BASELINE.json configs[3] (T' = 187 .. 375) at 32 utterances
    (5, 129, 128, 512, 256, [129, 128, 33, 32, 1]),
    (32, 188, 128, 512, 256, [188 - 5 * k for k in range(32)]),
    (32, 375, 128, 512, 256, [375 - 11 * k for k in range(32)]),
    (3, 1000, 128, 512, 256, [1000, 999, 17]),
])
def test_attention_step_forward_backward(B, T, A, E, D, lens):
    from ss_asr_amd import ops
    feat, state = rnd(B, T, E, seed=21), rnd(B, D, seed=22)
    w_phi = rnd(A, D, seed=23, scale=D ** -0.5)
    w_psi, b_psi = rnd(A, E, seed=24, scale=E ** -0.5), rnd(A, seed=25, scale=0.1)
    ga, gc = rnd(B, T, seed=26), rnd(B, E, seed=27)
    leaves = [t.clone().requires_grad_(True) for t in (feat, state, w_phi, w_psi, b_psi)]
    f, s, wp, ws, bs = leaves
    comp = torch.tanh(f @ ws.t() + bs)
    alpha, ctx = lo.attention_step_explicit(s, f, comp, lens, wp)
    ((alpha * ga).sum() + (ctx * gc).sum()).backward()

    dl = [t.float().to(dev()).requires_grad_(True) for t in (feat, state, w_phi, w_psi, b_psi)]
    fd, sd, wpd, wsd, bsd = dl
    compd = ops.attn_precompute(fd, wsd, bsd)
    close(compd, comp, 1e-5, 'comp')
    ld = torch.tensor(lens, dtype=torch.int32, device=dev())
    ad, cd = ops.attn_step(sd, wpd, compd, fd, ld)
    close(ad, alpha, 2e-6, 'alpha')      # fp32 energies (|e| ~ 10, ulp 1e-6) under a peaked softmax
    close(cd, ctx, 1e-5, 'ctx')
    ((ad * ga.float().to(dev())).sum() + (cd * gc.float().to(dev())).sum()).backward()
    for name, a, b in zip(['feat', 'state', 'w_phi', 'w_psi', 'b_psi'], dl, leaves):
        close(a.grad, b.grad, 1e-4, 'd' + name)


def test_split_attention_repeated_calls_share_one_workspace():
    """Back-to-back calls on one workspace (the decode loop's per-step launches) give the same
    result every time: a call exchanges through one of the workspace's two buffers and re-arms the
    other, so the buffer the NEXT call will use always holds the fill pattern again."""
    from ss_asr_amd import ops
    B, T, A, E, D = 32, 300, 128, 512, 256
    feat = rnd(B, T, E, seed=61).float().to(dev())
    comp = torch.tanh(rnd(B, T, A, seed=62)).float().to(dev())
    w_phi = rnd(A, D, seed=63, scale=D ** -0.5).float().to(dev())
    ld = torch.tensor([300 - 7 * k for k in range(B)], dtype=torch.int32, device=dev())
    outs = []
    for k in range(6):
        state = rnd(B, D, seed=70 + k % 2).float().to(dev())
        a, c = ops.attn_step(state, w_phi, comp, feat, ld)
        outs.append((a.clone(), c.clone()))
    torch.cuda.synchronize()
    for k in range(2, 6):
        assert torch.equal(outs[k][0], outs[k % 2][0]) and torch.equal(outs[k][1], outs[k % 2][1])
    ws, nxt = ops.attn_workspace(B, T, A, E, dev())
    assert ws is not None and bool((ws.view(torch.int32).view(2, -1)[nxt] == 0x7FC0DEAD).all())


def test_split_attention_missing_slice_times_out_and_is_reported():
    """Fault injection (SSASR_TEST_DROP_ATTN_SLICE): slice 2 of utterance 0 never publishes its partial
    softmax.  The utterance's other workgroups give up after their bounded wait, say so in the
    status word (this kernel used to `break` and consume the fill pattern silently), and
    the next call on the same workspace is healthy."""
    import time
    from ss_asr_amd import _lib, ops
    B, T, A, E, D = 32, 300, 128, 512, 256
    feat = rnd(B, T, E, seed=61).float().to(dev())
    comp = torch.tanh(rnd(B, T, A, seed=62)).float().to(dev())
    w_phi = rnd(A, D, seed=63, scale=D ** -0.5).float().to(dev())
    state = rnd(B, D, seed=64).float().to(dev())
    ld = torch.tensor([300 - 7 * k for k in range(B)], dtype=torch.int32, device=dev())
    good_a, good_c = ops.attn_step(state, w_phi, comp, feat, ld)
    ops.check_persistent_status()
    old = _lib.set_option('SSASR_TEST_DROP_ATTN_SLICE', 2)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.attn_step(state, w_phi, comp, feat, ld)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        _lib.set_option('SSASR_TEST_DROP_ATTN_SLICE', old)
    with pytest.raises(RuntimeError, match='split-T attention'):
        ops.check_persistent_status()
    assert elapsed < 20.0, elapsed
    for _ in range(2):                         # both exchange buffers
        a, c = ops.attn_step(state, w_phi, comp, feat, ld)
        ops.check_persistent_status()
        assert torch.equal(a, good_a) and torch.equal(c, good_c)


def test_attention_first_step_is_uniform_over_valid_frames():
    """s = 0 and phi has no bias => alpha = 1/len on valid frames."""
    from ss_asr_amd import ops
    B, T, A, E, D = 3, 10, 16, 32, 16
    lens = [10, 6, 3]
    feat = rnd(B, T, E, seed=31).float().to(dev())
    comp = torch.tanh(rnd(B, T, A, seed=32)).float().to(dev())
    w_phi = rnd(A, D, seed=33).float().to(dev())
    ld = torch.tensor(lens, dtype=torch.int32, device=dev())
    att, _ = ops.attn_step(torch.zeros(B, D, device=dev()), w_phi, comp, feat, ld)
    for b, l in enumerate(lens):
        np.testing.assert_allclose(att[b, :l].cpu().numpy(), 1.0 / l, rtol=1e-6)
        assert float(att[b, l:].abs().sum()) == 0.0


def test_integration_stub_computes_an_attention_step():
    """INTEGRATION.md's documented ctypes stub, executed as written, against the oracle."""
    from test_host_cpu import integration_stub
    ns = integration_stub()
    B, T, A, E, D = 3, 100, 128, 512, 256
    lens = [100, 57, 1]
    feat, state = rnd(B, T, E, seed=21), rnd(B, D, seed=22)
    w_phi = rnd(A, D, seed=23, scale=D ** -0.5)
    comp = torch.tanh(rnd(B, T, A, seed=24))
    alpha, ctx = lo.attention_step_explicit(state, feat, comp, lens, w_phi)
    ld = torch.tensor(lens, dtype=torch.int32, device=dev())
    ad, cd = ns['attention_step'](state.float().to(dev()), w_phi.float().to(dev()), comp.float().to(dev()),
                                  feat.float().to(dev()), ld)
    close(ad, alpha, 2e-6, 'alpha')
    close(cd, ctx, 1e-5, 'ctx')


# ------------------------------------------------------------- LSTM cell ----
@pytest.mark.parametrize('N,I,H', [(4, 96, 32), (32, 768, 256), (37, 50, 16)])
def test_lstm_cell_forward_backward(N, I, H):
    from ss_asr_amd import ops
    t = [rnd(N, I, seed=41), rnd(N, H, seed=42), rnd(N, H, seed=43),
         rnd(4 * H, I, seed=44, scale=I ** -0.5), rnd(4 * H, H, seed=45, scale=H ** -0.5),
         rnd(4 * H, seed=46, scale=0.1), rnd(4 * H, seed=47, scale=0.1)]
    gh, gc = rnd(N, H, seed=48), rnd(N, H, seed=49)
    r = [v.clone().requires_grad_(True) for v in t]
    h1, c1 = lo.lstm_cell_explicit(*r)
    ((h1 * gh).sum() + (c1 * gc).sum()).backward()
    d = [v.float().to(dev()).requires_grad_(True) for v in t]
    h1d, c1d = ops.lstm_cell(*d)
    close(h1d, h1, 1e-5, 'h')
    close(c1d, c1, 1e-5, 'c')
    ((h1d * gh.float().to(dev())).sum() + (c1d * gc.float().to(dev())).sum()).backward()
    for name, a, b in zip(['x', 'h', 'c', 'w_ih', 'w_hh', 'b_ih', 'b_hh'], d, r):
        close(a.grad, b.grad, 1e-4, 'd' + name)


# ----------------------------------------------------------- loss / step ----
def test_masked_ce_loss_forward_backward():
    from ss_asr_amd import ops
    B, U, V = 5, 9, 50
    logits = rnd(B, U, V, seed=51)
    y = torch.zeros(B, U + 3, dtype=torch.long)
    g = torch.Generator().manual_seed(52)
    for b, l in enumerate([9, 7, 4, 2, 1]):
        y[b, 1:1 + l] = torch.randint(1, V, (l,), generator=g)
    lr = logits.clone().requires_grad_(True)
    want = lo.masked_ce_loss(lr, y, U)
    want.backward()
    ld = logits.float().to(dev()).requires_grad_(True)
    got = ops.masked_ce_loss(ld, y.to(dev()), U)
    close(got, want, 1e-6, 'loss')
    got.backward()
    close(ld.grad, lr.grad, 1e-7, 'dlogits')


@pytest.mark.parametrize('n', [1000, 4096 * 3 + 5, 1 << 20])
def test_clip_adadelta_matches_oracle(n):
    from ss_asr_amd import ops
    for scale, steps in ((0.001, 2), (3.0, 2)):       # below and above the clip norm
        p, g = rnd(n, seed=61).float(), (rnd(n, seed=62) * scale).float()
        pr, sq, ad = [p.clone()], [torch.zeros(n)], [torch.zeros(n)]
        pd, sqd, add = p.to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
        ws, stats = ops.clip_adadelta_ws(n, dev()), torch.zeros(2, device=dev())
        for _ in range(steps):
            norm, ok = lo.clip_adadelta_explicit(pr, [g.clone()], sq, ad)
            ops.clip_adadelta_(pd, g.to(dev()), sqd, add, ws, stats)
            s = stats.cpu()
            assert ok and s[1] == 0
            assert abs(float(s[0]) - norm) <= 1e-5 * max(1.0, norm)
        close(pd, pr[0], 2e-6, 'param')
        close(sqd, sq[0], 1e-7, 'square_avg')


def test_clip_adadelta_skips_on_nan_and_honours_grad_scale():
    from ss_asr_amd import ops
    n = 5000
    p, g = rnd(n, seed=63).float().to(dev()), rnd(n, seed=64).float().to(dev())
    sq, ad = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    ws, stats = ops.clip_adadelta_ws(n, dev()), torch.zeros(2, device=dev())
    bad = g.clone()
    bad[17] = float('nan')
    before = p.clone()
    ops.clip_adadelta_(p, bad, sq, ad, ws, stats)
    assert stats[1].item() == 1.0 and torch.equal(p, before) and float(sq.abs().sum()) == 0
    # grad_scale = 1/4 must equal pre-dividing the gradient (DDP mean)
    p2, sq2, ad2 = before.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    ops.clip_adadelta_(p, g, sq, ad, ws, stats, grad_scale=0.25)
    ops.clip_adadelta_(p2, g * 0.25, sq2, ad2, ws, stats)
    close(p, p2, 1e-7, 'grad_scale')


def test_frame_lengths():
    from ss_asr_amd import ops
    x = rnd(6, 50, 80, seed=71).float()
    lens = [50, 44, 30, 17, 2, 1]
    for i, l in enumerate(lens):
        x[i, l:] = 0
    assert ops.frame_lengths(x.to(dev())).cpu().tolist() == lens


def test_first_layer_fused_input_projection_equals_gemm(monkeypatch):
    """I = 80, H = 256: the forward recurrence's helper wave forms W_ih x + b itself
    (rnn_kernels.h, KI); SSASR_NO_FUSED_INPUT=1 takes the GEMM.  Same MFMA products in a
    different summation order: outputs and gradients agree to rounding."""
    from ss_asr_amd import _lib, ops
    torch.manual_seed(5)
    N, S, I, H = 20, 70, 80, 256
    x = (torch.randn(N, S + 3, I) * 0.5).to(dev())
    lens = torch.tensor(sorted(torch.randint(S // 2, S + 1, (N,)).tolist(), reverse=True), dtype=torch.int32)
    lens[0] = S
    w = []
    for _ in range(2):
        w += [(torch.randn(4 * H, I) * I ** -0.5).to(dev()), (torch.randn(4 * H, H) * H ** -0.5).to(dev()),
              (torch.randn(4 * H) * 0.1).to(dev()), (torch.randn(4 * H) * 0.1).to(dev())]
    dy = (torch.randn(N, S, 2 * H) * 0.1).to(dev())
    out = []
    for fused in (True, False):
        old = _lib.set_option('SSASR_NO_FUSED_INPUT', 0 if fused else 1)     # switches are read once: set, not setenv
        try:
            xs = x.clone().requires_grad_(True)
            ws = [t.clone().requires_grad_(True) for t in w]
            y = ops.bilstm(xs, lens.to(dev()), S, True, tuple(ws))
            y.backward(dy)
            torch.cuda.synchronize()
        finally:
            _lib.set_option('SSASR_NO_FUSED_INPUT', old)
        out.append((y.detach(), xs.grad, [t.grad for t in ws]))
    ops.check_persistent_status()
    (ya, dxa, dwa), (yb, dxb, dwb) = out
    assert float((ya - yb).abs().max()) < 2e-6
    assert float((dxa - dxb).abs().max()) < 2e-6 * max(1.0, float(dxb.abs().max()))
    for a, b in zip(dwa, dwb):
        assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(b.abs().max()))


def test_label_copy_cache_follows_the_tensor_not_its_address():
    """ops.as_i32 remembers the int32 copy of the step's label matrix (teacher forcing and loss
    share it).  The next batch's labels are usually allocated where the last batch's were: the
    cache must miss then."""
    from ss_asr_amd import ops
    y1 = torch.randint(1, 30, (8, 12), device=dev())
    c1 = ops.as_i32(y1)
    assert ops.as_i32(y1) is c1                      # same object, same version: remembered
    addr = y1.data_ptr()
    del y1
    y2 = torch.randint(1, 30, (8, 12), device=dev())  # the caching allocator hands the block out again
    c2 = ops.as_i32(y2)
    assert torch.equal(c2.long(), y2)
    if y2.data_ptr() == addr:
        assert c2 is not c1
    y2[0, 0] = 31                                     # in-place edit bumps the version
    assert int(ops.as_i32(y2)[0, 0]) == 31


def test_second_backward_over_the_same_graph_is_refused():
    """The layers overwrite their saved gates with the gate derivatives (no second copy of the
    largest activation): a second backward pass must fail loudly, not return garbage."""
    from ss_asr_amd import ops
    torch.manual_seed(2)
    N, S, I, H = 4, 6, 16, 64
    x = torch.randn(N, S, I, device=dev(), requires_grad=True)
    w = []
    for _ in range(2):
        w += [torch.randn(4 * H, I, device=dev(), requires_grad=True) * 0.1,
              torch.randn(4 * H, H, device=dev(), requires_grad=True) * 0.1,
              torch.zeros(4 * H, device=dev(), requires_grad=True), torch.zeros(4 * H, device=dev(), requires_grad=True)]
    w = [t.detach().requires_grad_(True) for t in w]
    lens = torch.full((N,), S, dtype=torch.int32, device=dev())
    y = ops.bilstm(x, lens, S, True, tuple(w)).sum()
    y.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='second pass'):
        y.backward()
