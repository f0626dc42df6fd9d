"""TextAutoEncoder with the surface of the reference's src/text_autoencoder.py (config 5's
text branch): a text encoder (embedding + 2-layer BiLSTM over the padded
character rows, :96-107) in place of the Listener, followed by the ASR model's OWN attention,
speller, embedding and char_trans (:55-94) -- so training it trains the LAS decoder.

Here the attend-and-spell loop is the same fused decode loop ASR.forward uses
(ssasr_decoder_fwd / _bwd through ops.decoder_loop, with the text encoder's output as the
listener features and the noised row lengths as the attention mask), and the encoder's BiLSTM
layers are ssasr_bilstm_fwd / _bwd without lengths (the reference does not pack: padded
positions are encoded too).  The nn.Embedding / nn.LSTM below are parameter containers with the
reference's state_dict keys; the character embedding lookup itself is torch's gather (index
plumbing with its scatter-add backward, no arithmetic of the path).
"""
import random

import torch
import torch.nn as nn

from . import ops


class TextEncoder(nn.Module):
    """src/text_autoencoder.py:96-107."""

    def __init__(self, char_dim, emb_dim, state_size, num_layers):
        super().__init__()
        self.emb = nn.Embedding(char_dim, emb_dim)
        self.blstm = nn.LSTM(input_size=emb_dim, hidden_size=state_size, num_layers=num_layers,
                             bidirectional=True, batch_first=True)
        self.num_layers = num_layers

    def _layer_weights(self, layer):
        return [getattr(self.blstm, '%s_l%d%s' % (n, layer, sfx)) for sfx in ('', '_reverse')
                for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]

    def forward(self, y):
        x = self.emb(y)                                   # [B, seq, emb]
        for layer in range(self.num_layers):
            x = ops.bilstm(x, None, x.shape[1], True, self._layer_weights(layer))
        return x                                          # [B, seq, 2 * state_size]


class TextAutoEncoder(nn.Module):
    """src/text_autoencoder.py:8-94; same constructor arguments and state_dict."""

    def __init__(self, char_dim, emb_dim=128, state_size=256, num_layers=2):
        super().__init__()
        self.encoder = TextEncoder(char_dim, emb_dim, state_size, num_layers)

    def forward(self, asr, y, y_noised, decode_step, noise_lens=None):
        """asr: an ss_asr_amd.asr.ASR; y [B, L] clean label rows, y_noised [B, <= L] rows with
        characters dropped, decode_step = longest clean row, noise_lens = lengths of the noised
        rows (the attention mask).  Returns (noise_lens, logits [B, decode_step, V])."""
        if noise_lens is None:
            raise AssertionError('noise_lens (the lengths of the noised rows) is required')
        feat = self.encoder(y_noised)
        dev = feat.device
        B = y_noised.shape[0]
        asr.decoder.init_rnn(B, dev)
        asr.attention.reset_enc_mem()
        # Teacher forcing as src/text_autoencoder.py:81-88: one host coin flip per step but the
        # last, whose successor (never consumed) is the arg-max branch.
        modes = [0 if random.random() <= asr.tf_rate else 1 for _ in range(decode_step - 1)] + [2]
        # the loop reads the character fed to step t from column t; one spare column for step U
        teacher = torch.cat([y.to(torch.int32), torch.zeros(B, 1, dtype=torch.int32, device=dev)], dim=1).contiguous()
        enc_len_dev, modes_dev = ops.upload_i32(dev, [int(v) for v in noise_lens], modes)
        uniforms = torch.rand(decode_step, B, device=dev) if 1 in modes else None
        logits, _att, chars = ops.decoder_loop(feat, None, enc_len_dev, teacher, modes, uniforms,
                                               asr._decoder_params(), modes_dev=modes_dev,
                                               psi=(asr.attention.psi.weight, asr.attention.psi.bias))
        self.last_chars, self.last_modes, self.last_uniforms = chars, modes, uniforms
        return noise_lens, logits


def tae_loss(logits, y):
    """TAETrainer's loss (src/trainer.py:662-672): cross entropy of output t against y[:, t]
    (index 0 ignored), summed per row, divided by the row's count of non-zero labels, mean
    over the batch -- on the masked-CE kernel, whose label of step t is column t + 1."""
    shifted = torch.cat([torch.zeros_like(y[:, :1]), y], dim=1)
    return ops.masked_ce_loss(logits, shifted, logits.shape[1])
