"""CTC branch of BASELINE.json configs[3] ("Joint CTC+attention loss").  Build-defined: the
reference has no CTC, so the checker is the one SURVEY.md 8(c) names,
torch.nn.functional.ctc_loss (CPU, float64), and parity is unpinned by the reference."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import las_oracle as lo

pytestmark = pytest.mark.gpu


def make_case(B, T, V, lmax, seed, frame_lens=None, label_lens=None, repeats=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g, dtype=torch.float64) * 2.0
    if frame_lens is None:
        frame_lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
        frame_lens[0] = T
    if label_lens is None:
        label_lens = torch.randint(0, lmax + 1, (B,), generator=g)
        label_lens[0] = lmax
    y = torch.zeros(B, lmax + 2, dtype=torch.long)          # extra columns: labels are read in place
    for b in range(B):
        hi = 3 if repeats else V                             # few classes => many repeated neighbours
        y[b, :int(label_lens[b])] = torch.randint(1, hi, (int(label_lens[b]),), generator=g)
    return logits, torch.as_tensor(frame_lens), y, torch.as_tensor(label_lens)


def torch_ctc(logits, frame_lens, y, label_lens, lmax):
    lr = logits.clone().requires_grad_(True)
    lp = F.log_softmax(lr, dim=-1).transpose(0, 1)
    loss = F.ctc_loss(lp, y[:, :lmax], frame_lens, label_lens, blank=0, reduction='mean', zero_infinity=True)
    loss.backward()
    return float(loss.detach()), lr.grad


CASES = [
    dict(B=3, T=8, V=5, lmax=3, seed=1),
    dict(B=4, T=12, V=4, lmax=5, seed=2, repeats=True),                     # repeated labels need blanks
    dict(B=32, T=100, V=50, lmax=40, seed=3),                                # configs[1] geometry
    dict(B=8, T=375, V=50, lmax=300, seed=4),                                # configs[3]: T' = 3000 / 8, 300 characters
    dict(B=2, T=20, V=6, lmax=8, seed=5, frame_lens=[20, 5], label_lens=[8, 7]),   # row 1 has no alignment
    dict(B=3, T=9, V=7, lmax=4, seed=6, label_lens=[4, 0, 1]),               # an empty label row
    dict(B=2, T=700, V=50, lmax=60, seed=7),                                 # table > 128 KB: workspace path
    dict(B=1, T=1, V=3, lmax=1, seed=8, frame_lens=[1], label_lens=[1]),     # one frame, one label
    dict(B=2, T=600, V=50, lmax=511, seed=9, frame_lens=[600, 590], label_lens=[511, 280], repeats=False),  # 1023 states
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B%d_T%d_V%d_L%d' % (c['B'], c['T'], c['V'], c['lmax']))
def test_ctc_loss_and_gradient_match_torch(case):
    from ss_asr_amd import ops
    lmax = case['lmax']
    logits, frame_lens, y, label_lens = make_case(**case)
    want, want_grad = torch_ctc(logits, frame_lens, y, label_lens, lmax)
    d = torch.device('cuda:0')
    ld = logits.float().to(d).requires_grad_(True)
    got = ops.ctc_loss(ld, frame_lens.to(d, torch.int32), y.to(d, torch.int32), label_lens.to(d, torch.int32), lmax)
    assert abs(float(got) - want) <= 2e-6 * max(1.0, abs(want)), (float(got), want)
    got.backward()
    err = (ld.grad.double().cpu() - want_grad).abs().max().item()
    assert err <= 5e-7, err                                 # gradients are O(1 / B) and below
    # frames past an utterance's length, and rows with no alignment, get exactly zero
    for b in range(case['B']):
        assert float(ld.grad[b, int(frame_lens[b]):].abs().sum()) == 0.0


def test_ctc_head_is_linear_plus_ctc():
    """The fused head node (GEMM + CTC, bias gradient out of the CTC backward kernel) against
    torch's Linear + ctc_loss."""
    from ss_asr_amd import ops
    B, T, E, V, lmax = 6, 50, 64, 50, 20
    _, frame_lens, y, label_lens = make_case(B, T, V, lmax, seed=11)
    g = torch.Generator().manual_seed(12)
    feat = torch.randn(B, T, E, generator=g, dtype=torch.float64)
    w = torch.randn(V, E, generator=g, dtype=torch.float64) / 8
    bias = torch.randn(V, generator=g, dtype=torch.float64) / 4
    ref = [t.clone().requires_grad_(True) for t in (feat, w, bias)]
    lp = F.log_softmax(F.linear(*ref), dim=-1).transpose(0, 1)
    want = F.ctc_loss(lp, y[:, :lmax], frame_lens, label_lens, blank=0, reduction='mean', zero_infinity=True)
    (3.0 * want).backward()
    d = torch.device('cuda:0')
    dev = [t.float().to(d).requires_grad_(True) for t in (feat, w, bias)]
    got = ops.ctc_head_loss(*dev, frame_lens.to(d, torch.int32), y.to(d, torch.int32),
                            label_lens.to(d, torch.int32), lmax)
    assert abs(float(got) - float(want)) <= 2e-5 * max(1.0, float(want))
    (3.0 * got).backward()
    for name, a, r in zip(('dfeat', 'dw', 'db'), dev, ref):
        err = (a.grad.double().cpu() - r.grad).abs().max().item()
        assert err <= 2e-5 * max(1.0, r.grad.abs().max().item()), (name, err)


def test_ctc_argument_errors():
    from ss_asr_amd import ops
    d = torch.device('cuda:0')
    logits = torch.zeros(2, 4, 5, device=d)
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32, device=d)
    y = torch.zeros(2, 3, dtype=torch.int32, device=d)
    with pytest.raises(ValueError):
        ops.ctc_loss(logits, i32(4, 4), y, i32(1, 1), 7)                  # Lmax beyond the label matrix
    with pytest.raises(ValueError):
        ops.ctc_loss(logits, i32(4, 4), torch.zeros(2, 600, dtype=torch.int32, device=d), i32(1, 1), 600)
    with pytest.raises(TypeError):
        ops.ctc_loss(logits, i32(4, 4), y.long(), i32(1, 1), 2)
    with pytest.raises(RuntimeError):
        ops.ctc_loss(logits.cpu(), i32(4, 4), y, i32(1, 1), 2)             # no CPU path


@pytest.mark.parametrize('frames,chars,weight', [(list(range(240, 80, -10)), [3 + 2 * k for k in range(16)], 0.3),
                                                 ([64, 48, 16], [6, 1, 2], 0.5)])
def test_joint_ctc_attention_step_matches_the_oracle(frames, chars, weight):
    """One joint train step (loss = w * ctc + (1 - w) * masked CE; forward, backward, clip,
    Adadelta) of JointCTCTrainStep against the CPU oracle with torch's ctc_loss on the oracle's
    Listener, from the same seeded weights: the two losses and every updated parameter."""
    from ss_asr_amd import ops
    from ss_asr_amd.ctc import JointCTCASR, JointCTCTrainStep
    from ss_asr_amd.engine import label_geometry
    from ss_asr_amd.synthetic import make_batch
    dims = (50, 64, 64, 32, 40)
    x, y, lens = make_batch(np.array(frames), np.array(chars), 40, seed=21)
    _, ans_len = label_geometry(y)
    torch.manual_seed(0)
    ref = lo.OracleASR(*dims, 1.0)
    lo.seeded_weights(ref, 17)
    head = torch.nn.Linear(128, 50)
    g = torch.Generator().manual_seed(18)
    head.weight.data = torch.randn(50, 128, generator=g) / 12
    head.bias.data = torch.randn(50, generator=g) / 10
    ropt = torch.optim.Adadelta(list(ref.parameters()) + list(head.parameters()), lr=1.0, eps=1e-8)
    want = lo.joint_train_step(ref, head, ropt, x, y, weight)

    model = JointCTCASR(*dims, 1.0, ctc_weight=weight)
    lo.seeded_weights(model, 17)            # same names, same draws; the head is set below
    g = torch.Generator().manual_seed(18)
    model.ctc_head.weight.data = torch.randn(50, 128, generator=g) / 12
    model.ctc_head.bias.data = torch.randn(50, generator=g) / 10
    model = model.to('cuda:0')
    step = JointCTCTrainStep(model)
    random.seed(0)
    loss = float(step(x.cuda(), y.cuda(), lens, ans_len))
    norm, skipped = step.finish()
    ops.check_persistent_status()
    assert not skipped
    assert abs(float(step.last_att_loss) - want[1]) < 1e-4
    assert abs(float(step.last_ctc_loss) - want[2]) < 2e-5 * max(1.0, want[2])
    assert abs(loss - want[0]) < 1e-4 * max(1.0, want[0])
    sd = model.state_dict()
    worst = max(float((sd[k].cpu() - v).abs().max()) for k, v in ref.state_dict().items())
    worst = max(worst, float((sd['ctc_head.weight'].cpu() - head.weight).abs().max()),
                float((sd['ctc_head.bias'].cpu() - head.bias).abs().max()))
    assert worst < 5e-4, worst


def test_joint_step_at_config4_full_size_matches_the_oracle_fixture(golden):
    """BASELINE.json configs[3] at its own size -- 32 utterances of 1500-3000 frames, 150-300
    characters, ctc_weight 0.3: ONE joint train step (Listener at 3000 / 1500 / 750 persistent steps,
    the long-encoder persistent decode loop, the CTC lattice at T' = 375 with 601 states, clip,
    Adadelta) against tests/golden/joint_long_b32_t3000.npz, which oracle/make_joint_golden.py made
    with the CPU oracle's joint_train_step (build-defined branch: the reference has no CTC; the
    attention half of the oracle is pinned to the reference at this shape by long_b32_t3000)."""
    from conftest import fixture_xy
    from ss_asr_amd import ops
    from ss_asr_amd.ctc import JointCTCASR, JointCTCTrainStep
    fx = golden('joint_long_b32_t3000')
    x, y = fixture_xy(fx)
    lens = [int(v) for v in fx['lens']]
    dims = [int(v) for v in fx['dims']]
    model = JointCTCASR(*dims, 1.0, ctc_weight=float(fx['ctc_weight']))
    lo.seeded_weights(model, int(fx['weights_seed']))          # same names, same draws; the head is set below
    g = torch.Generator().manual_seed(int(fx['head_seed']))
    model.ctc_head.weight.data = torch.randn(dims[0], 512, generator=g) / 512 ** 0.5
    model.ctc_head.bias.data = torch.randn(dims[0], generator=g) / 10
    model = model.to('cuda:0')
    names = [str(n) for n in fx['param_names']]
    step = JointCTCTrainStep(model)
    # gradients are zeroed by the fused update: capture them through the optimizer's view before it runs
    grads = {}
    real = step.optim.clip_and_step

    def capture(*a, **k):
        ops.join_side_stream()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            grads[n] = p.grad.detach().clone()
        return real(*a, **k)
    step.optim.clip_and_step = capture
    random.seed(0)
    loss = float(step(x.cuda(), y.cuda(), lens, int(fx['ans_len'])))
    norm, skipped = step.finish()
    ops.check_persistent_status()
    assert not skipped
    assert abs(float(step.last_att_loss) - float(fx['att_loss'])) < 1e-4
    assert abs(float(step.last_ctc_loss) - float(fx['ctc_loss'])) < 2e-5 * max(1.0, float(fx['ctc_loss']))
    assert abs(loss - float(fx['loss'])) < 1e-4 * max(1.0, float(fx['loss']))
    assert abs(norm - float(fx['grad_norm'])) < 2e-4 * max(1.0, float(fx['grad_norm']))
    got = np.array([grads[n].double().norm().item() for n in names])
    np.testing.assert_allclose(got, fx['grad_norms'], rtol=2e-4, atol=1e-6)
    logits = step.last_logits.detach().reshape(-1)[::997][:2048].cpu().numpy()
    np.testing.assert_allclose(logits, fx['logits_sample'], atol=5e-5, rtol=0)
    params = dict(model.named_parameters())
    for k in fx.files:
        if k.startswith('g_head/'):
            np.testing.assert_allclose(grads[k[7:]].reshape(-1)[:256].cpu().numpy(), fx[k], atol=2e-5, rtol=0, err_msg=k)
        if k.startswith('w1_head/'):       # an Adadelta first update is <= 3.2e-4 per element: 3e-6 = 1 % of it
            got_w = params[k[8:]].detach().reshape(-1)[:256].cpu().numpy()
            print('max abs post-step weight error %s: %.3g' % (k, float(np.abs(got_w - fx[k]).max())))
            np.testing.assert_allclose(got_w, fx[k], atol=3e-6, rtol=0, err_msg=k)
