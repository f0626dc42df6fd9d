"""Generates tests/golden/joint_long_b32_t3000.npz: one JOINT CTC + attention train step of the CPU
oracle (las_oracle.joint_train_step) at BASELINE.json configs[3]'s full size -- 32 utterances of
1500-3000 frames, 150-300 characters, ctc_weight 0.3.

BUILD-DEFINED, PARITY UNPINNED BY THE REFERENCE: the reference has no CTC anywhere (SURVEY.md
section 1), so this fixture is made by the ORACLE -- the attention half of it is the model that
tests/golden/long_b32_t3000.npz pins to the real reference at this very shape, the CTC half is
torch.nn.functional.ctc_loss, the checker SURVEY.md 8(c) names.  It exists so that the GPU box can
check the joint step at full size without minutes of CPU work per test run.

Run (build container or anywhere with torch; ~10 minutes on 8 cores):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_joint_golden.py
The fixture holds the recipe of the batch (synthetic.config4_batch), not its 31 MB of frames.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import las_oracle as lo  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
DIMS = (50, 256, 256, 128, 80)
HEAD_KEYS = ('attention.phi.weight', 'encoder.blstm_1.layer.weight_hh_l0', 'decoder.layer_1.weight_ih',
             'char_trans.bias', 'encoder.blstm_4.weight_ih_l0_reverse', 'ctc_head.weight', 'ctc_head.bias')


def seeded_head(seed, out_dim=50, in_dim=512):
    """The ctc_head of both sides of the test: a torch.Generator stream, scaled like init_parameters."""
    g = torch.Generator().manual_seed(seed)
    head = torch.nn.Linear(in_dim, out_dim)
    head.weight.data = torch.randn(out_dim, in_dim, generator=g) / in_dim ** 0.5
    head.bias.data = torch.randn(out_dim, generator=g) / 10
    return head


def main(weights_seed=16, head_seed=18, ctc_weight=0.3):
    from ss_asr_amd.synthetic import config4_batch
    x, y, lens = config4_batch()
    torch.manual_seed(0)
    model = lo.OracleASR(*DIMS, 1.0)
    lo.seeded_weights(model, weights_seed)
    head = seeded_head(head_seed)
    params = dict(model.named_parameters())
    params.update({'ctc_head.' + k: v for k, v in head.named_parameters()})
    optim = torch.optim.Adadelta(list(model.parameters()) + list(head.parameters()), lr=1.0, eps=1e-8)
    # joint_train_step, opened up to keep the gradients and the clipped norm
    t0 = time.time()
    ans_len = max(lo.label_lengths(y)) - 1
    optim.zero_grad()
    _, logits, _ = model(x, ans_len, teacher=y, state_len=lens)
    att = lo.masked_ce_loss(logits, y, ans_len)
    ctc = lo.ctc_branch_loss(model, head, x, y, lens)
    loss = ctc_weight * ctc + (1.0 - ctc_weight) * att
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in params.items()}
    norm, _ = lo.solver_step(list(model.parameters()) + list(head.parameters()), optim)
    print('joint step: %.0f s  loss %.6f  att %.6f  ctc %.6f  |g| %.6f' % (time.time() - t0, float(loss), float(att),
                                                                           float(ctc), float(norm)))
    names = list(params)
    out = dict(dims=np.array(DIMS), weights_seed=np.int64(weights_seed), head_seed=np.int64(head_seed),
               ctc_weight=np.float64(ctc_weight), y=y.numpy(), lens=np.array(lens), ans_len=np.int64(ans_len),
               recipe_config4=np.int64(1), recipe_batch_size=np.int64(32), recipe_seed=np.int64(4),
               x_abs_sum=np.float64(x.double().abs().sum().item()),
               loss=np.float64(float(loss)), att_loss=np.float64(float(att)), ctc_loss=np.float64(float(ctc)),
               grad_norm=np.float64(float(norm)), param_names=np.array(names),
               grad_norms=np.array([grads[k].double().norm().item() for k in names]),
               logits_sample=logits.detach().reshape(-1)[::997][:2048].numpy())
    for k in HEAD_KEYS:
        out['g_head/' + k] = grads[k].reshape(-1)[:256].numpy()
        out['w1_head/' + k] = params[k].detach().reshape(-1)[:256].numpy()
    path = os.path.join(OUT, 'joint_long_b32_t3000.npz')
    np.savez_compressed(path, **out)
    print(path, '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
