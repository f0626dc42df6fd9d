"""Joint CTC + attention training (BASELINE.json configs[3]).

BUILD-DEFINED: the reference has no CTC anywhere -- its only loss is the masked cross
entropy of src/trainer.py:426-434 -- so nothing here follows a
reference line.  The branch is the usual one (Watanabe et al. 2017): a Linear(2 * H -> V)
on the Listener's output, CTC over the utterance's T' = T // 8 encoder frames against the
characters between <sos> and <eos>, blank = class 0 (the <sos> / padding index, which no
label uses), and

    loss = ctc_weight * ctc + (1 - ctc_weight) * attention_loss.

The checker in tests/ is torch.nn.functional.ctc_loss on the CPU oracle's encoder.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .asr import ASR
from .engine import ASRTrainStep

BLANK = 0


class JointCTCASR(ASR):
    """ASR plus ``ctc_head``.  The base model's parameters are created (and seeded) first, so
    a JointCTCASR and an ASR built from one seed share every common weight, and a reference
    checkpoint loads with ``strict=False`` (missing: ctc_head.weight, ctc_head.bias)."""

    def __init__(self, output_dim, encoder_state_size, decoder_state_size, mlp_out_size,
                 feature_dim, tf_rate, ctc_weight=0.3):
        super().__init__(output_dim, encoder_state_size, decoder_state_size, mlp_out_size,
                         feature_dim, tf_rate)
        if not 0.0 <= ctc_weight <= 1.0:
            raise ValueError('ctc_weight must lie in [0, 1]')
        self.ctc_weight = ctc_weight
        self.ctc_head = nn.Linear(self.encoder.out_dim, output_dim)
        self.ctc_head.weight.data.normal_(0, 1. / math.sqrt(self.encoder.out_dim))   # as init_parameters
        self.ctc_head.bias.data.zero_()

    def load_state_dict(self, state_dict, *args, **kwargs):
        """A checkpoint of the plain ASR (the reference's, or this build's without the branch)
        loads too: the head then keeps its fresh initialisation."""
        if not any(k.startswith('ctc_head.') for k in state_dict):
            state_dict = dict(state_dict)
            state_dict.update({'ctc_head.' + k: v for k, v in self.ctc_head.state_dict().items()})
        return super().load_state_dict(state_dict, *args, **kwargs)

    def ctc_loss(self, y):
        """CTC of the most recent forward's encoder output against y [B, L] (prepare_y's matrix:
        <sos>, characters, <eos>, zero padding)."""
        feat, enc_len_dev = self.last_encoded
        y32 = ops.as_i32(y)
        label_lens = ((y32 != 0).sum(-1) - 1).clamp_(min=0).to(torch.int32)    # characters, without <eos>
        lmax = y32.shape[1] - 1
        return ops.ctc_head_loss(feat, self.ctc_head.weight, self.ctc_head.bias, enc_len_dev,
                                 y32[:, 1:], label_lens, lmax, BLANK)


class JointCTCTrainStep(ASRTrainStep):
    """ASRTrainStep whose loss is the joint one; everything else (all-reduce, clip, Adadelta,
    status polling) is inherited."""

    def forward_loss(self, x, y, x_lens, ans_len):
        att_loss, logits, att = super().forward_loss(x, y, x_lens, ans_len)
        lam = self.model.ctc_weight
        self.last_att_loss = att_loss.detach()
        self.last_ctc_loss = ctc = self.model.ctc_loss(y)
        return lam * ctc + (1.0 - lam) * att_loss, logits, att
