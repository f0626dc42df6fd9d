"""Model-level parity on the GPU against the golden vectors captured from the
real reference (tests/golden, made by oracle/make_golden.py), and against the
CPU oracle at sizes the goldens do not cover.  Tolerances: activations 2e-5
abs, loss 1e-4 abs (north star: 1e-3), per-parameter gradient norms 2e-5 rel.
"""
import random

import numpy as np
import pytest
import torch

import las_oracle as lo
from conftest import fixture_att, fixture_xy

pytestmark = pytest.mark.gpu

# BASELINE.json configs[1] at its own size: bench.py's longest (32 x 800 frames, 81 label steps:
# four BPTT segments, two-chunk recurrences, the 32-utterance persistent decode loop) and median bucket
BENCH_SHAPES = ['bench_b32_t800', 'bench_b32_median']
# BASELINE.json configs[3]'s shape at full size (attention loss only: the reference has no CTC): 32
# utterances of 1500-3000 frames, T' = 375 (split-T attention inside the per-step decode loop, ~300
# decode steps), 3000 / 1500 / 750 persistent recurrence steps with the exchange ring wrapping ~370 times
LONG_SHAPES = ['long_b32_t3000']
GRAD_NORM_RTOL = 2e-5      # measured on MI355X (round 3): <= 3.3e-6 on every fixture
W1_ATOL = 3e-6             # post-step weights, absolute (an Adadelta first update is <= 3.2e-4 per element)
UPDATE_NORM_RTOL = 2e-4    # per-tensor norm of the first update


def build(fx):
    from ss_asr_amd.asr import ASR
    dims = [int(v) for v in fx['dims']]
    torch.manual_seed(0)
    model = ASR(*dims, float(fx['tf_rate']))
    ws = int(fx['weights_seed'])
    if ws >= 0:
        lo.seeded_weights(model, ws)
    else:
        model.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files
                               if k.startswith('w0/')})
    return model.to('cuda:0')


def forward(fx, model):
    from ss_asr_amd import ops
    x, y = (t.cuda() for t in fixture_xy(fx))
    lens = [int(v) for v in fx['lens']]
    assert ops.frame_lengths(x).cpu().tolist() == lens
    ans_len = int(fx['ans_len'])
    random.seed(int(fx['rng_seed']))
    enc_len, logits, att = model(x, int(fx['decode_steps']),
                                 teacher=y if int(fx['teacher']) else None, state_len=lens)
    loss = ops.masked_ce_loss(logits, y, ans_len)
    torch.cuda.synchronize()
    return enc_len, logits, att, loss


@pytest.mark.parametrize('name', ['small_tf1', 'small_odd', 'small_padded', 'small_greedy',
                                  'full_b4', 'edge_b1', 'edge_short', 'full_b40', 'full_b16_t400'] + BENCH_SHAPES + LONG_SHAPES)
def test_forward_matches_reference(golden, name):
    fx = golden(name)
    model = build(fx)
    taps = {}
    hooks = [getattr(model.encoder, n).register_forward_hook(
        lambda m, i, o, n=n: taps.__setitem__(n, o[0].detach())) for n in
        ('blstm_1', 'blstm_2', 'blstm_3')]
    enc_len, logits, att, loss = forward(fx, model)
    for h in hooks:
        h.remove()
    assert enc_len == [int(v) for v in fx['enc_len']]
    for n, v in taps.items():
        if 'act_' + n in fx.files:
            np.testing.assert_allclose(v.cpu().numpy(), fx['act_' + n], atol=2e-5, rtol=0,
                                       err_msg=n)
        elif 'act_' + n + '_sample' in fx.files:      # compact fixtures: a strided sample and a checksum
            flat = v.reshape(-1)
            got = flat[::max(1, flat.numel() // 512)][:512].cpu().numpy()
            np.testing.assert_allclose(got, fx['act_' + n + '_sample'], atol=2e-5, rtol=0, err_msg=n)
            ref_sum = float(fx['act_' + n + '_abs_sum'])
            assert abs(float(flat.double().abs().sum()) - ref_sum) < 2e-5 * ref_sum, n
    np.testing.assert_allclose(fixture_att(fx, att.numpy()), fx['att'], atol=2e-6, rtol=0)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), fx['logits'], atol=5e-5, rtol=0)
    assert abs(float(loss) - float(fx['loss'])) < 1e-4
    if 'acc' in fx.files:
        # the validation metric itself (north star: "loss/CER within 1e-3"): the reference's calc_acc on the
        # reference's logits (captured through oracle/ref_harness.py) against the product's on the GPU's logits
        from ss_asr_amd.postprocess import calc_acc
        ans_len = int(fx['ans_len'])
        label = torch.from_numpy(fx['y'])[:, 1:ans_len + 1]
        acc = calc_acc(logits[:, :ans_len], label)
        print('acc %s: %.6f (reference %.6f)' % (name, acc, float(fx['acc'])))
        assert abs(acc - float(fx['acc'])) < 1e-3


@pytest.mark.parametrize('name', ['full_b16_t400'] + BENCH_SHAPES)
def test_bf16_operand_variant_stays_within_the_stated_loss_tolerance(golden, name):
    """SSASR_GEMM_BF16=1 (the bf16-storage VARIANT of BASELINE.json configs[1]: the launcher's GEMM operands
    rounded to bf16, fp32 accumulation; recurrences, attention, decoder and loss unchanged) against the
    REFERENCE's fp32 loss and accuracy on the same inputs: north_star's "loss/CER within 1e-3".  Not the
    default, never the headline; the default path's tolerance (1e-4) is test_forward_matches_reference's."""
    from ss_asr_amd import _lib
    from ss_asr_amd.postprocess import calc_acc
    fx = golden(name)
    model = build(fx)
    lib = _lib.load()
    assert lib.ssasr_set_option(b'SSASR_GEMM_BF16', 1) == 0
    try:
        enc_len, logits, att, loss = forward(fx, model)
    finally:
        assert lib.ssasr_set_option(b'SSASR_GEMM_BF16', 0) == 0
    d_loss = abs(float(loss) - float(fx['loss']))
    d_logit = float(np.abs(logits.detach().cpu().numpy() - fx['logits']).max())
    print('bf16 variant %s: |d loss| %.2e, max |d logit| %.2e' % (name, d_loss, d_logit))
    assert enc_len == [int(v) for v in fx['enc_len']]
    assert d_loss < 1e-3
    assert d_logit > 1e-6            # (the variant really ran: fp32 products land within 5e-5)
    if 'acc' in fx.files:
        ans_len = int(fx['ans_len'])
        label = torch.from_numpy(fx['y'])[:, 1:ans_len + 1]
        assert abs(calc_acc(logits[:, :ans_len], label) - float(fx['acc'])) < 1e-3


@pytest.mark.parametrize('name', ['small_tf1', 'small_odd', 'small_padded', 'full_b4', 'edge_b1', 'edge_short', 'full_b40',
                                  'full_b16_t400'] + BENCH_SHAPES + LONG_SHAPES)
def test_backward_and_solver_step_match_reference(golden, name):
    from ss_asr_amd.optim import FlatParameters, FusedAdadelta
    fx = golden(name)
    model = build(fx)
    flat = FlatParameters(model)
    optim = FusedAdadelta(flat, lr=1.0, eps=1e-8)
    optim.zero_grad()
    names = [str(n) for n in fx['param_names']]
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    _, _, _, loss = forward(fx, model)
    loss.backward()
    torch.cuda.synchronize()          # weight-gradient GEMMs run on a side stream
    params = dict(model.named_parameters())
    got = np.array([params[n].grad.double().norm().item() for n in names])
    # per-parameter gradient norms: GRAD_NORM_RTOL relative (VERDICT r2 found 1e-3 loose beside a loss that
    # agrees to 2e-6; the measured maxima are printed below)
    np.testing.assert_allclose(got, fx['grad_norms'], rtol=GRAD_NORM_RTOL, atol=1e-6)
    print('max rel grad-norm error %s: %.3g' % (name, np.max(np.abs(got - fx['grad_norms']) / np.maximum(fx['grad_norms'], 1e-6))))
    for k in fx.files:
        if k.startswith('g/'):
            np.testing.assert_allclose(params[k[2:]].grad.cpu().numpy(), fx[k], atol=2e-5,
                                       rtol=0, err_msg=k)
        if k.startswith('g_head/'):
            np.testing.assert_allclose(params[k[7:]].grad.reshape(-1)[:256].cpu().numpy(), fx[k],
                                       atol=2e-5, rtol=0, err_msg=k)
    optim.clip_and_step(max_norm=5.0)
    norm, skipped = optim.poll(wait=True)
    assert not skipped and abs(norm - float(fx['grad_norm'])) < 1e-4
    upd = np.array([(params[n].detach() - before[n]).double().norm().item() for n in names])
    # The first Adadelta update is at most sqrt(eps / 0.1) = 3.2e-4 per element, so the post-step weights are
    # held to W1_ATOL = 3e-6 absolute (1 % of an update; VERDICT r3: 1e-4 passed an update that is 30 % wrong)
    # and the per-tensor update norms to UPDATE_NORM_RTOL; measured maxima are printed.
    np.testing.assert_allclose(upd, fx['update_norms'], rtol=UPDATE_NORM_RTOL, atol=1e-7)
    print('max rel update-norm error %s: %.3g' % (name, np.max(np.abs(upd - fx['update_norms']) / np.maximum(fx['update_norms'], 1e-7))))
    checked, worst = 0, 0.0
    for k in fx.files:
        if k.startswith('w1/'):
            got_w = params[k[3:]].detach().cpu().numpy()
            worst = max(worst, float(np.abs(got_w - fx[k]).max()))
            np.testing.assert_allclose(got_w, fx[k], atol=W1_ATOL, rtol=0, err_msg=k)
            checked += 1
        if k.startswith('w1_head/'):       # compact fixtures: the first 256 post-step weights of selected tensors
            got_w = params[k[8:]].detach().reshape(-1)[:256].cpu().numpy()
            worst = max(worst, float(np.abs(got_w - fx[k]).max()))
            np.testing.assert_allclose(got_w, fx[k], atol=W1_ATOL, rtol=0, err_msg=k)
            checked += 1
    assert checked > 0, 'fixture %s holds no post-step weights' % name
    print('max abs post-step weight error %s: %.3g' % (name, worst))


def test_backward_matches_reference_on_the_fp32_mfma_instruction(golden):
    """The GEMMs' other arithmetic form (SSASR_GEMM_X6 = 0: v_mfma_f32_16x16x4_f32 instead of six
    bf16 MFMAs on the split operands) against the same reference fixture, same tolerances: both
    forms are the reference's fp32 arithmetic."""
    from ss_asr_amd import _lib
    old = _lib.set_option('SSASR_GEMM_X6', 0)
    try:
        fx = golden('full_b4')
        model = build(fx)
        _, _, _, loss = forward(fx, model)
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss) - float(fx['loss'])) < 1e-4
        params = dict(model.named_parameters())
        names = [str(n) for n in fx['param_names']]
        got = np.array([params[n].grad.double().norm().item() for n in names])
        np.testing.assert_allclose(got, fx['grad_norms'], rtol=GRAD_NORM_RTOL, atol=1e-6)
    finally:
        _lib.set_option('SSASR_GEMM_X6', old)


def test_module_level_loop_equals_fused_loop(golden):
    """Driving Attention / Speller step by step (the way TextAutoEncoder does,
    src/text_autoencoder.py:55-88) gives the fused decode loop's result."""
    fx = golden('small_tf1')
    model = build(fx)
    x = torch.from_numpy(fx['x']).cuda()
    y = torch.from_numpy(fx['y']).cuda()
    lens = [int(v) for v in fx['lens']]
    steps = int(fx['decode_steps'])
    feat, enc_len = model.encoder(x, lens)
    model.decoder.init_rnn(x.shape[0], x.device)
    model.attention.reset_enc_mem()
    teacher = model.embed(y)
    last = model.embed(torch.zeros(x.shape[0], dtype=torch.long, device=x.device))
    outs = []
    for t in range(steps):
        att, ctx = model.attention(model.decoder.state_list[0], feat, enc_len)
        out = model.decoder(torch.cat([last, ctx], dim=-1))
        outs.append(torch.nn.functional.linear(out, model.char_trans.weight,
                                               model.char_trans.bias))
        last = teacher[:, t + 1, :]
    logits = torch.stack(outs, dim=1)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), fx['logits'], atol=5e-5, rtol=0)
    logits.sum().backward()
    assert model.encoder.blstm_1.layer.weight_hh_l0.grad is not None


def test_sampled_steps_follow_the_categorical_law():
    """tf_rate = 0: every next character is drawn in-kernel.  The empirical
    law of the first drawn character must match softmax(logits[:, 0])."""
    from ss_asr_amd.asr import ASR
    torch.manual_seed(3)
    model = ASR(50, 32, 32, 16, 12, 0.0).to('cuda:0')
    B, T = 64, 16
    x = torch.randn(1, T, 12).repeat(B, 1, 1).cuda()
    y = torch.randint(3, 50, (B, 6)).cuda()
    y[:, 0] = 0
    counts = torch.zeros(50)
    probs = None
    for it in range(40):
        _, logits, _ = model(x, 3, teacher=y, state_len=[T] * B)
        # all utterances are identical up to blstm_4's utterance-axis recurrence,
        # so compare per utterance position: use utterance 0 only
        p0 = torch.softmax(logits[0, 0].detach().cpu().double(), -1)
        probs = p0 if probs is None else probs
        counts[int(model.last_chars[1, 0])] += 1
    # chi-square-ish sanity: the most likely character is drawn at a rate near its probability
    top = int(torch.argmax(probs))
    rate = counts[top].item() / counts.sum().item()
    assert abs(rate - probs[top].item()) < 0.25


@pytest.mark.parametrize('dims,frames,chars,tf', [
    ((50, 256, 256, 128, 80), [168, 160, 152, 120, 96, 90, 64, 40], [14, 12, 12, 9, 8, 7, 5, 3], 0.5),   # persistent loop
    ((50, 32, 32, 16, 12), [64, 56, 48, 40], [10, 7, 5, 3], 0.5),                                      # per-step kernels
    # BASELINE.json configs[1] at its own size and teacher-forcing rate: bench.py's longest bucket (32 x 800
    # frames, 81 label steps) at tf_rate 0.9 -- the goldens of that shape are teacher forced
    ((50, 256, 256, 128, 80), 'bench_b32_t800', None, 0.9)])
def test_sampled_steps_match_the_oracle_draw(dims, frames, chars, tf):
    """The sampled branch of the decode loop (src/asr.py:94-98) against the oracle.  The kernel
    draws by inverse CDF from caller-visible uniforms, so every draw can be checked exactly: the
    oracle replays the loop on the characters the kernel fed (forced_chars), and for every
    sampled (step, utterance) the kernel's character c must satisfy run[c-1] <= u * total <
    run[c] on the ORACLE's logits, up to a CDF-boundary tolerance of 1e-5 * total (fp32 rounding
    of the cumulative sums); teacher-forced steps must feed the label.  Logits, loss and the
    gradient norm of that tf_rate 0.5 pass then have to match the oracle's."""
    from ss_asr_amd import ops
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.optim import FlatParameters
    from ss_asr_amd.synthetic import make_batch
    if isinstance(frames, str):
        from ss_asr_amd.synthetic import config2_batches
        x, y, lens = config2_batches(8, batch_size=32, feat_dim=dims[4], seed=1)[0]
        frames = lens
        torch.set_num_threads(min(16, torch.get_num_threads()))
    else:
        x, y, lens = make_batch(np.array(frames), np.array(chars), dims[4], seed=17)
    ans_len = max(lo.label_lengths(y)) - 1
    torch.manual_seed(0)
    ref = lo.OracleASR(*dims, tf)
    lo.seeded_weights(ref, 31)
    model = ASR(*dims, tf)
    lo.seeded_weights(model, 31)
    model = model.to('cuda:0')
    flat = FlatParameters(model)
    flat.zero_grad()
    random.seed(123)
    torch.manual_seed(123)
    _, logits, _ = model(x.cuda(), ans_len, teacher=y.cuda(), state_len=lens)
    loss = ops.masked_ce_loss(logits, y.cuda(), ans_len)
    loss.backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    ops.check_persistent_status()
    modes = list(model.last_modes)
    random.seed(123)
    assert modes == [0 if random.random() <= tf else 1 for _ in range(ans_len)]   # the reference's coin flips
    assert 1 in modes and 0 in modes
    fed = model.last_chars.cpu().long()                     # [U + 1, B]
    uni = model.last_uniforms.cpu().numpy()                 # [U, B]

    _, ref_logits, _ = ref(x, ans_len, teacher=y, state_len=lens, forced_chars=fed)
    ref_loss = lo.masked_ce_loss(ref_logits, y, ans_len)
    ref_loss.backward()
    rl = ref_logits.detach().numpy()
    assert (fed[0] == 0).all()
    exact = total = 0
    for t, mode in enumerate(modes):
        for b in range(len(frames)):
            c = int(fed[t + 1, b])
            if mode == 0:
                assert c == int(y[b, t + 1]), (t, b)
                continue
            run, tot = lo.inverse_cdf_bounds(rl[b, t])
            target = float(uni[t, b]) * tot
            lo_edge = float(run[c - 1]) if c > 0 else 0.0
            assert lo_edge - 1e-5 * tot <= target < float(run[c]) + 1e-5 * tot, (t, b, c, target, lo_edge, float(run[c]))
            total += 1
            exact += int(np.argmax(run > np.float32(target))) == c
    assert total > 0 and exact >= total - 1, (exact, total)        # at most one draw on a CDF boundary
    np.testing.assert_allclose(logits.detach().cpu().numpy(), rl, atol=5e-5, rtol=0)
    assert abs(float(loss) - float(ref_loss)) < 1e-4
    got = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())).item()
    want = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ref.parameters())).item()
    print('sampled pass: global grad norm %.6f (oracle %.6f)' % (got, want))
    assert abs(got - want) < GRAD_NORM_RTOL * max(1.0, want), (got, want)


def test_persistent_decode_loop_equals_multi_launch_loop():
    """Production sizes take the single-launch persistent decode loop; with the
    same coin flips and uniforms it must pick the same characters and give the
    same logits / attention / gradients as the one-launch-per-stage loop
    (also exercises in-kernel sampling and greedy steps)."""
    import ctypes as C
    from ss_asr_amd import _lib, ops
    from ss_asr_amd.asr import ASR
    torch.manual_seed(5)
    model = ASR(50, 256, 256, 128, 80, 0.5).to('cuda:0')
    B, T, U = 9, 96, 14
    feat = torch.randn(B, T // 8, 512, device='cuda')
    enc_len = torch.tensor([12, 12, 11, 9, 8, 8, 5, 3, 1], dtype=torch.int32, device='cuda')
    teacher = torch.randint(3, 50, (B, U + 2), device='cuda').to(torch.int32)
    modes = [0, 1, 0, 0, 2, 1, 1, 0, 0, 0, 1, 0, 2, 0]
    uniforms = torch.rand(U, B, device='cuda')
    results = []
    for per_step in (0, 1):
        _lib.set_option('SSASR_NO_PERSISTENT_DECODER', per_step)
        f = feat.clone().requires_grad_(True)
        comp = ops.attn_precompute(f, model.attention.psi.weight, model.attention.psi.bias)
        logits, att, chars = ops.decoder_loop(f, comp, enc_len, teacher, modes, uniforms,
                                              model._decoder_params())
        model.zero_grad()
        (logits * torch.linspace(0.5, 1.5, 50, device='cuda')).sum().backward()
        torch.cuda.synchronize()
        ops.check_persistent_status()
        results.append((logits.detach().cpu(), att.cpu(), chars.cpu(), f.grad.cpu(),
                        model.decoder.layer_1.weight_ih.grad.cpu().clone(),
                        model.attention.phi.weight.grad.cpu().clone()))
    _lib.set_option('SSASR_NO_PERSISTENT_DECODER', 0)
    a, b = results
    assert torch.equal(a[2], b[2])                                  # same characters fed
    np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), atol=2e-6, rtol=0)
    for x, y in zip(a[3:], b[3:]):
        np.testing.assert_allclose(x.numpy(), y.numpy(), atol=5e-5 * max(1.0, float(y.abs().max())), rtol=0)


@pytest.mark.parametrize('B,Tp,U,lens', [
    (32, 375, 24, [375 - 7 * k for k in range(32)]),          # BASELINE.json configs[3]: six slices of 64 frames
    (5, 129, 9, [129, 128, 65, 64, 1]),                       # three slices, lengths on slice edges, a one-frame utterance
    (16, 250, 12, [250 - 13 * k for k in range(16)]),         # four slices; utterances that end before the last slice starts
    (7, 384, 8, [384, 383, 321, 320, 192, 64, 3]),            # the longest encoder output the backward chain takes
])
def test_long_encoder_persistent_decode_loop_equals_multi_launch_loop(B, Tp, U, lens):
    """Encoder outputs longer than 128 frames take the register-resident persistent decode loop
    (csrc/decoder_long.h: frame slices of 64 per workgroup, partial softmaxes combined in-launch)
    and the frame-sliced persistent backward chain.  With the same coin flips and uniforms they
    must feed the same characters and give the same logits / attention / gradients as the
    one-launch-per-stage loops (split-T attention kernel per step), teacher-forced, sampled and
    greedy steps alike."""
    from ss_asr_amd import _lib, ops
    from ss_asr_amd.asr import ASR
    assert int(_lib.load().ssasr_decoder_fwd_part_floats(B, Tp, 128, 512, 256, 50)) > 0, 'long form not taken'
    torch.manual_seed(5)
    model = ASR(50, 256, 256, 128, 80, 0.5).to('cuda:0')
    feat = torch.randn(B, Tp, 512, device='cuda')
    enc_len = torch.tensor(lens, dtype=torch.int32, device='cuda')
    for b, l in enumerate(lens):
        feat[b, l:] = 0
    teacher = torch.randint(3, 50, (B, U + 2), device='cuda').to(torch.int32)
    modes = ([0, 1, 0, 0, 2, 1, 1, 0, 0, 0, 1, 0, 2, 0] * 2)[:U]
    uniforms = torch.rand(U, B, device='cuda')
    results = []
    try:
        for per_step in (0, 1):
            _lib.set_option('SSASR_NO_PERSISTENT_DECODER', per_step)
            _lib.set_option('SSASR_NO_PERSISTENT_DECODER_BWD', per_step)
            f = feat.clone().requires_grad_(True)
            comp = ops.attn_precompute(f, model.attention.psi.weight, model.attention.psi.bias)
            logits, att, chars = ops.decoder_loop(f, comp, enc_len, teacher, modes, uniforms,
                                                  model._decoder_params())
            model.zero_grad()
            (logits * torch.linspace(0.5, 1.5, 50, device='cuda')).sum().backward()
            torch.cuda.synchronize()
            ops.check_persistent_status()
            results.append((logits.detach().cpu(), att.cpu(), chars.cpu(), f.grad.cpu(),
                            model.decoder.layer_1.weight_ih.grad.cpu().clone(),
                            model.decoder.layer_2.weight_hh.grad.cpu().clone(),
                            model.attention.phi.weight.grad.cpu().clone(),
                            model.attention.psi.weight.grad.cpu().clone(),
                            model.embed.weight.grad.cpu().clone()))
    finally:
        _lib.set_option('SSASR_NO_PERSISTENT_DECODER', 0)
        _lib.set_option('SSASR_NO_PERSISTENT_DECODER_BWD', 0)
    a, b = results
    assert torch.equal(a[2], b[2])                                  # same characters fed
    np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), atol=2e-6, rtol=0)
    for b_idx, l in enumerate(lens):                                # nothing attends past an utterance's end
        assert float(a[1][b_idx, :, l:].abs().sum()) == 0.0
    for x, y in zip(a[3:], b[3:]):
        np.testing.assert_allclose(x.numpy(), y.numpy(), atol=5e-5 * max(1.0, float(y.abs().max())), rtol=0)


def test_long_encoder_decode_loop_with_a_missing_record_times_out_and_is_reported():
    """Fault injection (SSASR_TEST_DROP_DEC_SLICE): slice 1 of utterance 0 stops publishing its partial-softmax
    record after the first decode step of the long-encoder persistent loop.  Its peers' bounded waits give
    up, every other wait of the launch drains through the latch (40 steps x 192 + 64 workgroups: seconds,
    not minutes), the status word names the decode loop, and the next call is healthy."""
    import time
    from ss_asr_amd import _lib, ops
    from ss_asr_amd.asr import ASR
    torch.manual_seed(5)
    model = ASR(50, 256, 256, 128, 80, 1.0).to('cuda:0')
    B, Tp, U = 32, 375, 40
    feat = torch.randn(B, Tp, 512, device='cuda')
    enc_len = torch.full((B,), Tp, dtype=torch.int32, device='cuda')
    teacher = torch.randint(3, 50, (B, U + 2), device='cuda').to(torch.int32)
    modes = [0] * U
    with torch.no_grad():
        comp = ops.attn_precompute(feat, model.attention.psi.weight, model.attention.psi.bias)
        good, _, _ = ops.decoder_loop(feat, comp, enc_len, teacher, modes, None, model._decoder_params())
        torch.cuda.synchronize()
        ops.check_persistent_status()
        old = _lib.set_option('SSASR_TEST_DROP_DEC_SLICE', 1)
        try:
            t0 = time.perf_counter()
            ops.decoder_loop(feat, comp, enc_len, teacher, modes, None, model._decoder_params())
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        finally:
            _lib.set_option('SSASR_TEST_DROP_DEC_SLICE', old)
        with pytest.raises(RuntimeError, match='decode loop forward'):
            ops.check_persistent_status()
        assert elapsed < 30.0, elapsed
        again, _, _ = ops.decoder_loop(feat, comp, enc_len, teacher, modes, None, model._decoder_params())
        torch.cuda.synchronize()
        ops.check_persistent_status()
        assert torch.equal(again, good)


@pytest.mark.slow
def test_long_utterances_config4_shape_against_the_oracle():
    """BASELINE.json configs[3] in miniature: 1,500+ frame utterances (T' = 192 > 128, so the
    decode loop takes the long-encoder persistent form of csrc/decoder_long.h and the 4-slice
    backward chain -- the per-step split-T loop is what
    test_long_encoder_persistent_decode_loop_equals_multi_launch_loop forces and compares it with --
    while the encoder runs 1,536 / 768 / 384 persistent steps with the exchange ring wrapping
    around hundreds of times and the BPTT cut into segments).  One train step against the CPU
    oracle on the same seeded weights."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.engine import ASRTrainStep, label_geometry
    from ss_asr_amd.synthetic import make_batch
    from ss_asr_amd import ops
    dims = (50, 256, 256, 128, 80)
    frames = np.array([1536, 1400, 1111])
    chars = np.array([60, 52, 41])
    x, y, lens = make_batch(frames, chars, 80, seed=11)
    _, ans_len = label_geometry(y)
    torch.manual_seed(0)
    ref = lo.OracleASR(*dims, 1.0)
    lo.seeded_weights(ref, 9)
    ropt = lo.make_optimizer(ref)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref_loss, ref_norm = lo.train_step(ref, ropt, x, y)
    model = ASR(*dims, 1.0)
    lo.seeded_weights(model, 9)
    model = model.to('cuda:0')
    step = ASRTrainStep(model)
    random.seed(0)
    loss = float(step(x.cuda(), y.cuda(), lens, ans_len))
    norm, skipped = step.optim.poll(wait=True)
    ops.check_persistent_status()
    assert not skipped
    assert abs(loss - ref_loss) < 1e-4, (loss, ref_loss)
    assert abs(norm - ref_norm) < 2e-3 * max(1.0, ref_norm), (norm, ref_norm)
    got = dict(model.named_parameters())
    for k, v in ref.named_parameters():
        if k in ('encoder.blstm_1.layer.weight_hh_l0', 'decoder.layer_1.weight_ih', 'attention.psi.weight'):
            np.testing.assert_allclose(got[k].detach().cpu().numpy(), v.detach().numpy(), atol=2e-4, rtol=0, err_msg=k)


def test_four_train_steps_follow_the_oracle_trajectory():
    """Several consecutive steps (optimizer state carried over, different batches, ragged
    lengths): loss, clipped gradient norm and the weights after the last step against the CPU
    oracle started from the same seeded weights."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.engine import ASRTrainStep, label_geometry
    from ss_asr_amd.synthetic import make_batch
    from ss_asr_amd import ops
    dims = (50, 64, 64, 32, 80)
    torch.manual_seed(0)
    ref = lo.OracleASR(*dims, 1.0)
    lo.seeded_weights(ref, 21)
    ropt = lo.make_optimizer(ref)
    model = ASR(*dims, 1.0)
    lo.seeded_weights(model, 21)
    model = model.to('cuda:0')
    step = ASRTrainStep(model)
    shapes = [([72, 64, 40, 17], [9, 7, 6, 3]), ([41, 33], [5, 4]), ([96, 95, 94, 50, 49, 8], [11, 10, 9, 6, 5, 2]),
              ([56, 24, 16], [7, 3, 2])]
    for k, (fr, ch) in enumerate(shapes):
        x, y, lens = make_batch(np.array(fr), np.array(ch), 80, seed=30 + k)
        _, ans_len = label_geometry(y)
        ref_loss, ref_norm = lo.train_step(ref, ropt, x, y)
        loss = float(step(x.cuda(), y.cuda(), lens, ans_len).detach())
        norm, skipped = step.optim.poll(wait=True)
        assert not skipped
        assert abs(loss - ref_loss) < 2e-4, (k, loss, ref_loss)
        assert abs(norm - ref_norm) < 2e-3 * max(1.0, ref_norm), (k, norm, ref_norm)
    ops.check_persistent_status()
    got = dict(model.named_parameters())
    worst = max(float((got[k].detach().cpu() - v.detach()).abs().max()) for k, v in ref.named_parameters())
    assert worst < 5e-4, worst


def test_two_step_objects_alternating_in_one_process_do_not_disturb_each_other():
    """BASELINE.json configs[4]'s Seed loop alternates trainers in one process.  The step objects share
    process-wide plumbing (the second stream, the weight-gradient listener, the status rows): an ASR step
    object and a joint CTC + attention step object on a second model take turns, and the ASR model's
    losses, norms and final weights must equal a run in which it trained alone ("allows ONE
    train-step object per process")."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.ctc import JointCTCASR, JointCTCTrainStep
    from ss_asr_amd.engine import ASRTrainStep, label_geometry
    from ss_asr_amd.synthetic import make_batch
    dims = (50, 64, 64, 32, 80)
    shapes = [([72, 64, 40, 17], [9, 7, 6, 3]), ([96, 95, 94, 50, 49, 8], [11, 10, 9, 6, 5, 2]), ([56, 24, 16], [7, 3, 2])]
    batches = []
    for k, (fr, ch) in enumerate(shapes):
        x, y, lens = make_batch(np.array(fr), np.array(ch), 80, seed=60 + k)
        batches.append((x.cuda(), y.cuda(), lens, label_geometry(y)[1]))

    def run(interleave):
        torch.manual_seed(0)
        model = ASR(*dims, 1.0)
        lo.seeded_weights(model, 31)
        step = ASRTrainStep(model.to('cuda:0'))
        other = None
        if interleave:
            joint = JointCTCASR(*dims, 1.0, ctc_weight=0.3)
            lo.seeded_weights(joint, 32)
            other = JointCTCTrainStep(joint.to('cuda:0'))
        out = []
        for k, (x, y, lens, ans_len) in enumerate(batches):
            loss = float(step(x, y, lens, ans_len).detach())
            if other is not None:
                xo, yo, lo_, ao = batches[(k + 1) % len(batches)]
                assert np.isfinite(float(other(xo, yo, lo_, ao).detach()))
            out.append((loss,) + tuple(step.optim.poll(wait=True)))
        if other is not None:
            other.finish()
        step.finish()
        return out, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()

    alone, w_alone = run(False)
    mixed, w_mixed = run(True)
    for (l0, n0, s0), (l1, n1, s1) in zip(alone, mixed):
        assert not s0 and not s1
        assert abs(l0 - l1) < 1e-6 and abs(n0 - n1) < 1e-6 * max(1.0, n0)
    assert float((w_alone - w_mixed).abs().max()) < 1e-6


@pytest.mark.parametrize('name', ['full_b16_t400', 'bench_b32_t800'])
def test_backward_is_reproducible_with_overlapped_streams(golden, name):
    """The recurrences have no atomics, so repeated backward passes over the same batch may
    differ only by the summation order of the weight-gradient accumulations (~1e-6).  Larger
    run-to-run differences mean a race between the persistent kernels and the work that
    shares the GPU with them (weight-gradient GEMMs on the second stream delay workgroup
    starts by microseconds, which once let one workgroup of a tile overwrite saved gates the
    other still had to read)."""
    from ss_asr_amd import ops
    from ss_asr_amd.optim import FlatParameters
    fx = golden(name)
    model = build(fx)
    flat = FlatParameters(model)
    names = [n for n, _ in model.named_parameters()]
    first = None
    worst = (0.0, '')
    for _ in range(25):
        flat.zero_grad()
        _, _, _, loss = forward(fx, model)
        loss.backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        grads = [p.grad.detach().clone() for p in model.parameters()]
        if first is None:
            first = grads
            top = max(float(g.abs().max()) for g in grads)
            continue
        for n, a, b in zip(names, grads, first):
            scale = float(b.abs().max())
            if scale < 1e-4 * top:
                continue            # gradients that are zero up to rounding (the psi bias)
            dev = float((a - b).abs().max()) / scale
            if dev > worst[0]:
                worst = (dev, n)
    ops.check_persistent_status()
    assert worst[0] < 5e-5, worst


@pytest.mark.parametrize('frames,chars', [([16], [2]),                       # one utterance, two encoder frames
                                          ([9], [1]),                        # T' = 1: a single attention frame
                                          (list(range(120, 54, -2)), [3 + k % 9 for k in range(33)]),   # 33 > 32 utterances
                                          ([200, 8], [12, 1])])              # very ragged pair
def test_train_step_edge_shapes_match_the_oracle(frames, chars):
    """Batch sizes and lengths at the edges of the persistent forms (one utterance; an encoder
    output of one or two frames; more utterances than the 32 the persistent decode loop and the
    two-chunk recurrences take; a pair whose second utterance ends after one encoder frame): one
    train step at the full layer sizes against the CPU oracle on the same seeded weights."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.engine import ASRTrainStep, label_geometry
    from ss_asr_amd.synthetic import make_batch
    from ss_asr_amd import ops
    dims = (50, 256, 256, 128, 80)
    x, y, lens = make_batch(np.array(frames), np.array(chars), 80, seed=5)
    _, ans_len = label_geometry(y)
    torch.manual_seed(0)
    ref = lo.OracleASR(*dims, 1.0)
    lo.seeded_weights(ref, 13)
    ropt = lo.make_optimizer(ref)
    ref_loss, ref_norm = lo.train_step(ref, ropt, x, y)
    model = ASR(*dims, 1.0)
    lo.seeded_weights(model, 13)
    model = model.to('cuda:0')
    step = ASRTrainStep(model)
    random.seed(0)
    loss = float(step(x.cuda(), y.cuda(), lens, ans_len))
    norm, skipped = step.optim.poll(wait=True)
    ops.check_persistent_status()
    assert not skipped
    assert abs(loss - ref_loss) < 1e-4, (loss, ref_loss)
    assert abs(norm - ref_norm) < 2e-3 * max(1.0, ref_norm), (norm, ref_norm)


def test_encoder_alone_forward_backward_as_the_other_trainers_call_it():
    """`asr.encoder(x, x_lens)` on its own with an arbitrary gradient fed into its output, at 32
    utterances: how SAETrainer and ADVTrainer use the shared Listener (src/trainer.py:810, :988 --
    no attention, no speller, no ASR.forward around it).  Against the oracle's Listener: features,
    lengths, every encoder parameter's gradient, and the input gradient."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.synthetic import config2_batches
    from ss_asr_amd import ops
    dims = (50, 256, 256, 128, 80)
    x, _, lens = config2_batches(8, batch_size=32, seed=11)[3]
    ref = lo.OracleASR(*dims, 1.0)
    lo.seeded_weights(ref, 21)
    xr = x.clone().requires_grad_(True)
    feat_r, len_r = ref.encoder(xr, lens)
    g = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(feat_r.shape)).astype(np.float32))
    (feat_r * g).sum().backward()

    model = ASR(*dims, 1.0)
    lo.seeded_weights(model, 21)
    model = model.to('cuda:0')
    xd = x.cuda().requires_grad_(True)
    feat_d, len_d = model.encoder(xd, lens)
    assert [int(v) for v in len_d] == [int(v) for v in len_r]
    np.testing.assert_allclose(feat_d.detach().cpu().numpy(), feat_r.detach().numpy(), atol=2e-5, rtol=0)
    (feat_d * g.cuda()).sum().backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    ops.check_persistent_status()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), atol=5e-5, rtol=0)
    ref_p = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        if n.startswith('encoder.'):
            want = ref_p[n].grad
            scale = max(1.0, float(want.abs().max()))
            np.testing.assert_allclose(p.grad.cpu().numpy(), want.numpy(), atol=2e-4 * scale, rtol=0, err_msg=n)
        else:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n


@pytest.mark.parametrize('name', ['tae_full_b12', 'tae_full_b40'])
def test_text_autoencoder_on_the_kernels_matches_reference(golden, name):
    """SURVEY.md 8 f4 (config 5's shared decoder): ss_asr_amd.text_autoencoder.TextAutoEncoder runs
    the reference's attend-and-spell loop (src/text_autoencoder.py:55-94) through ssasr_decoder_fwd /
    _bwd with the text encoder's output as the listener features.  Logits, TAETrainer's loss and the
    gradients (of the text encoder and of the shared asr.attention / decoder / embed / char_trans)
    against the fixture captured from the reference; 12 rows take the persistent decode loop, 40 rows
    (more than its 32) the per-step kernels."""
    from ss_asr_amd import ops
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.text_autoencoder import TextAutoEncoder, tae_loss
    fx = golden(name)
    dims = [int(v) for v in fx['dims']]
    torch.manual_seed(0)
    asr = ASR(*dims, float(fx['tf_rate']))
    lo.seeded_weights(asr, int(fx['asr_weights_seed']))
    tae = TextAutoEncoder(dims[0], *[int(v) for v in fx['tae_dims']])
    lo.seeded_tae_weights(tae, int(fx['tae_weights_seed']))
    ref_keys = lo.OracleTextAutoEncoder(dims[0], *[int(v) for v in fx['tae_dims']]).state_dict().keys()
    assert list(tae.state_dict().keys()) == list(ref_keys)          # checkpoint compatibility
    asr, tae = asr.to('cuda:0'), tae.to('cuda:0')
    y, y_noise = torch.from_numpy(fx['y']).cuda(), torch.from_numpy(fx['y_noise']).cuda()
    random.seed(int(fx['rng_seed']))
    lens, logits = tae(asr, y, y_noise, int(fx['decode_step']), noise_lens=[int(v) for v in fx['noise_lens']])
    assert lens == [int(v) for v in fx['noise_lens']]
    np.testing.assert_allclose(logits.detach().cpu().numpy(), fx['logits'], atol=5e-5, rtol=0)
    loss = tae_loss(logits, y)
    assert abs(float(loss) - float(fx['loss'])) < 1e-4
    loss.backward()
    torch.cuda.synchronize()
    ops.check_persistent_status()
    grads = {('tae.' + k): p.grad for k, p in tae.named_parameters()}
    grads.update({('asr.' + k): p.grad for k, p in asr.named_parameters() if p.grad is not None})
    names = [str(n) for n in fx['grad_names']]
    assert sorted(grads) == names                                    # the Listener gets no gradient
    got = np.array([grads[k].double().norm().item() for k in names])
    np.testing.assert_allclose(got, fx['grad_norms'], rtol=GRAD_NORM_RTOL, atol=1e-6)
    print('max rel grad-norm error %s: %.3g' % (name, np.max(np.abs(got - fx['grad_norms']) / np.maximum(fx['grad_norms'], 1e-6))))
    for k in fx.files:
        if k.startswith('g_head/'):
            np.testing.assert_allclose(grads[k[7:]].reshape(-1)[:256].cpu().numpy(), fx[k], atol=2e-5, rtol=0,
                                       err_msg=k)
