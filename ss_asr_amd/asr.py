"""Listen-Attend-Spell model with the nn.Module surface of the reference's
``src/asr.py`` (ASR :15-212, Listener :214-264, Speller :267-326, Attention
:328-392, pBLSTM :394-450), computing on MI355X through libssasr_hip.so.

The torch.nn modules held below (nn.LSTM, nn.LSTMCell, nn.Linear,
nn.Embedding) are parameter containers only: they give the same
``state_dict`` keys, parameter order and constructor-time RNG consumption as
the reference, so checkpoints and seeded initialisations interchange.  Their
``forward`` is never called; all arithmetic goes through ``ss_asr_amd.ops``.
"""
import math
import os
import random

import torch
import torch.nn as nn

from . import ops


def _dev_i32(values, device):
    """Host list of lengths -> int32 device tensor.  Staged through pinned memory and
    copied asynchronously: a copy from pageable memory (torch.tensor(..., device=cuda))
    waits for the stream to drain, which stalled the host 3.5 ms twice per train step."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int32)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)

EOS_INDEX = 1   # '>' in TOKENS + ALL_CHARS (src/preprocess.py:17-27)


def _check_lengths(state_len, tmax):
    """pack_padded_sequence's input contract (src/asr.py:413)."""
    prev = None
    for l in state_len:
        if l <= 0:
            raise RuntimeError('Length of all samples has to be greater than 0, '
                               'but found an element in \'lengths\' that is <= 0')
        if prev is not None and l > prev:
            raise RuntimeError('`lengths` array must be sorted in decreasing order when '
                               '`enforce_sorted` is True.')
        prev = l
    if state_len[0] > tmax:
        raise RuntimeError('Expected sequence length to be larger than or equal to the '
                           'maximum length in `lengths`')


def _lstm_weights(lstm):
    return [getattr(lstm, n + sfx) for sfx in ('', '_reverse')
            for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]


class pBLSTM(nn.Module):
    """src/asr.py:394-450: packed BiLSTM, then concatenation of frame pairs."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.layer = nn.LSTM(in_dim, out_dim, bidirectional=True, batch_first=True)

    def forward(self, input_x, state=None, state_len=None, pack_input=False, len_dev=None, slots=None):
        if state is not None:
            raise NotImplementedError('initial states are always zero on the hot path')
        if pack_input:
            assert state_len is not None, \
                "Please specify seq len for pack_padded_sequence."
            state_len = [int(s) for s in state_len]
            _check_lengths(state_len, input_x.shape[1])
            steps = state_len[0]
            if len_dev is None:
                len_dev = _dev_i32(state_len, input_x.device)
        else:
            steps, len_dev = input_x.shape[1], None
        output = ops.bilstm(input_x, len_dev, steps, True, _lstm_weights(self.layer), slots=slots)
        output = self.downsample(output)
        if state_len is not None:
            return output, None, [int(s / 2) for s in state_len]
        return output, None

    def downsample(self, x):
        """[B, T, F] -> [B, T//2, 2F]; an odd last frame is dropped (src/asr.py:429-450)."""
        t_dim, f_dim = x.shape[1], x.shape[2]
        if t_dim % 2 != 0:
            # a view, not a copy: pairs of frames merge into rows of 2F floats whatever the utterance stride is, and
            # the next layer's kernels take strides (ops._BiLSTM.forward)
            x = x[:, :t_dim - 1, :]
            t_dim -= 1
            if x.stride(2) == 1 and x.stride(1) == f_dim and os.environ.get('SSASR_DOWNSAMPLE_COPY') != '1':
                return x.view([x.shape[0], int(t_dim / 2), f_dim * 2])
        return x.contiguous().view([-1, int(t_dim / 2), f_dim * 2])


class Listener(nn.Module):
    """src/asr.py:214-264."""

    def __init__(self, state_size, feature_dim):
        super().__init__()
        self.state_size = state_size
        self.out_dim = 2 * self.state_size
        self.blstm_1 = pBLSTM(feature_dim, self.state_size)
        self.blstm_2 = pBLSTM(self.state_size * 2 * 2, self.state_size)
        self.blstm_3 = pBLSTM(self.state_size * 2 * 2, self.state_size)
        # Not batch_first in the reference (src/asr.py:237-238): the recurrence
        # runs over the utterance axis, independently per encoder time index.
        self.blstm_4 = nn.LSTM(self.state_size * 2 * 2, self.state_size, bidirectional=True)

    def get_outdim(self):
        return self.out_dim

    @staticmethod
    def layer_lengths(state_len):
        """Frame counts seen by blstm_1..3 and by the decoder (each pBLSTM halves them,
        src/asr.py:425)."""
        out = [[int(s) for s in state_len]]
        for _ in range(3):
            out.append([int(s / 2) for s in out[-1]])
        return out

    def forward(self, x, state_len, pack_input=True, len_devs=None, arenas=None):
        """len_devs: the first three lists of layer_lengths(state_len) as int32 device
        tensors when the caller has uploaded them already.  arenas: (forward, backward)
        ops.ExchangeArena of this pass; the four layers' exchange workspaces are reserved in
        them so that one fill arms all of them."""
        if pack_input and len_devs is None:
            len_devs = ops.upload_i32(x.device, *self.layer_lengths(state_len)[:3])
        slots = [None] * 4
        if arenas is not None and pack_input:
            lens = self.layer_lengths(state_len)
            B, H = x.shape[0], self.state_size
            shapes = [(lens[0][0], B), (lens[1][0], B), (lens[2][0], B), (B, lens[3][0])]   # (steps, columns)
            for k, (S, N) in enumerate(shapes):
                hx, ring = ops.bilstm_exchange_floats(S, N, H)
                slots[k] = (arenas[0], arenas[0].reserve(hx) if hx else None,
                            arenas[1], arenas[1].reserve(ring) if ring else None)
        for k, layer in enumerate((self.blstm_1, self.blstm_2, self.blstm_3)):
            x, _, state_len = layer(x, state_len=state_len, pack_input=pack_input,
                                    len_dev=len_devs[k] if pack_input else None, slots=slots[k])
        x = ops.bilstm(x, None, x.shape[0], False, _lstm_weights(self.blstm_4), slots=slots[3])      # src/asr.py:262
        return x, state_len


class Speller(nn.Module):
    """src/asr.py:267-326."""

    def __init__(self, state_size, encoder_out_size):
        super().__init__()
        self.layer_1 = nn.LSTMCell(input_size=encoder_out_size + state_size, hidden_size=state_size)
        self.layer_2 = nn.LSTMCell(input_size=state_size, hidden_size=state_size)
        self._zero_state = None
        self._state_list = []
        self._cell_list = []
        self.state_size = state_size
        self.num_layers = 2

    def init_rnn(self, batch_size, device):
        """Zero states (src/asr.py:284-290).  Allocated on first use: the fused decode loop
        never reads them, only step-by-step callers do."""
        self._zero_state = (batch_size, device)
        self._state_list = self._cell_list = None

    def _materialise(self):
        if self._state_list is None and self._zero_state is not None:
            b, dev = self._zero_state
            self._state_list = [torch.zeros(b, self.state_size, device=dev)] * self.num_layers
            self._cell_list = [torch.zeros(b, self.state_size, device=dev)] * self.num_layers

    @property
    def state_list(self):
        self._materialise()
        return self._state_list

    @state_list.setter
    def state_list(self, v):
        self._materialise()
        self._state_list = v

    @property
    def cell_list(self):
        self._materialise()
        return self._cell_list

    @cell_list.setter
    def cell_list(self, v):
        self._materialise()
        self._cell_list = v

    @property
    def hidden_state(self):
        return [s.clone().detach().cpu() for s in self.state_list], \
            [c.clone().detach().cpu() for c in self.cell_list]

    @hidden_state.setter
    def hidden_state(self, state):
        device = self.state_list[0].device
        self.state_list = [s.to(device) for s in state[0]]
        self.cell_list = [c.to(device) for c in state[1]]

    def forward(self, input_context):
        l1, l2 = self.layer_1, self.layer_2
        self.state_list[0], self.cell_list[0] = ops.lstm_cell(
            input_context, self.state_list[0], self.cell_list[0],
            l1.weight_ih, l1.weight_hh, l1.bias_ih, l1.bias_hh)
        self.state_list[1], self.cell_list[1] = ops.lstm_cell(
            self.state_list[0], self.state_list[1], self.cell_list[1],
            l2.weight_ih, l2.weight_hh, l2.bias_ih, l2.bias_hh)
        return self.state_list[-1]


class Attention(nn.Module):
    """src/asr.py:328-392 (content based; psi projection cached per utterance batch)."""

    def __init__(self, mlp_out_size, encoder_out_size, decoder_state_size):
        super().__init__()
        self.phi = nn.Linear(decoder_state_size, mlp_out_size, bias=False)
        self.psi = nn.Linear(encoder_out_size, mlp_out_size)
        self.comp_listener_feature = None
        self.state_mask = None      # here: int32 lengths on the device

    def reset_enc_mem(self):
        self.comp_listener_feature = None
        self.state_mask = None

    def forward(self, decoder_state, listener_feature, state_len):
        if self.comp_listener_feature is None:
            self.state_mask = _dev_i32(state_len, listener_feature.device)
            self.comp_listener_feature = ops.attn_precompute(
                listener_feature, self.psi.weight, self.psi.bias)
        return ops.attn_step(decoder_state, self.phi.weight, self.comp_listener_feature,
                             listener_feature, self.state_mask)


class ASR(nn.Module):
    """src/asr.py:15-212; same constructor arguments and state_dict."""

    def __init__(self, output_dim, encoder_state_size, decoder_state_size, mlp_out_size,
                 feature_dim, tf_rate):
        super().__init__()
        enc_out_dim = encoder_state_size * 2
        self.encoder = Listener(encoder_state_size, feature_dim)
        self.attention = Attention(mlp_out_size, enc_out_dim, decoder_state_size)
        self.decoder = Speller(decoder_state_size, enc_out_dim)
        self.embed = nn.Embedding(output_dim, decoder_state_size)
        self.char_trans = nn.Linear(decoder_state_size, output_dim)
        self.tf_rate = tf_rate
        # When True (default) the training-mode attention map is copied to the
        # host asynchronously (pinned memory); call torch.cuda.synchronize()
        # or self.att_event.synchronize() before reading it.  eval() mode
        # always returns a finished copy.
        self.async_att = True
        # False: the attention map stays on the device (ASRTrainStep sets this around a train step)
        self.att_on_host = True
        self.att_event = None
        self.last_chars = self.last_modes = self.last_uniforms = None
        self.last_encoded = None
        self.init_parameters()

    def init_parameters(self):
        """src/asr.py:175-212."""
        for p in self.parameters():
            data = p.data
            if data.dim() == 1:
                data.zero_()
            elif data.dim() == 2:
                data.normal_(0, 1. / math.sqrt(data.size(1)))
            else:
                raise NotImplementedError
        self.embed.weight.data.normal_(0, 1)
        for bias in (self.decoder.layer_1.bias_ih, self.decoder.layer_2.bias_ih):
            n = bias.size(0)
            bias.data[n // 4:n // 2].fill_(1.)

    def _decoder_params(self):
        l1, l2 = self.decoder.layer_1, self.decoder.layer_2
        return dict(w_phi=self.attention.phi.weight,
                    w_ih1=l1.weight_ih, w_hh1=l1.weight_hh, b_ih1=l1.bias_ih, b_hh1=l1.bias_hh,
                    w_ih2=l2.weight_ih, w_hh2=l2.weight_hh, b_ih2=l2.bias_ih, b_hh2=l2.bias_hh,
                    embed=self.embed.weight, w_ct=self.char_trans.weight,
                    b_ct=self.char_trans.bias)

    def forward(self, audio_feature, decode_step, teacher=None, state_len=None):
        """Returns (encode_len: list[int], logits [B,U,V] on the device,
        attention [B,U,T'] on the host, detached) -- src/asr.py:52-110."""
        dev = audio_feature.device
        # One host coin flip per step, as the reference draws them (src/asr.py:94).
        if teacher is not None:
            modes = [0 if random.random() <= self.tf_rate else 1 for _ in range(decode_step)]
            teacher_i32 = ops.as_i32(teacher)
        else:
            modes = [2] * decode_step
            teacher_i32 = None
        # every per-step integer the device needs, in one upload
        lens = Listener.layer_lengths(state_len)
        l1, l2, l3, enc_len_dev, modes_dev = ops.upload_i32(dev, lens[0], lens[1], lens[2], lens[3], modes)
        # exchange workspaces of the whole pass: two allocations, each armed by one fill
        arenas = (ops.ExchangeArena(dev), ops.ExchangeArena(dev))
        dec_slots = ops.decoder_reserve(arenas[0], arenas[1], audio_feature.shape[0], lens[3][0], decode_step,
                                        A=self.attention.phi.weight.shape[0], E=self.encoder.out_dim,
                                        D=self.decoder.state_size, V=self.char_trans.weight.shape[0])
        encode_feature, encode_len = self.encoder(audio_feature, state_len, len_devs=(l1, l2, l3), arenas=arenas)
        # what a second head on the encoder (ss_asr_amd/ctc.py) reads: output and its int32 frame counts
        self.last_encoded = (encode_feature, enc_len_dev)
        self.decoder.init_rnn(encode_feature.shape[0], dev)
        self.attention.reset_enc_mem()
        uniforms = None
        if 1 in modes:
            # Categorical(...).sample() of the reference (src/asr.py:97): one
            # uniform per (step, utterance), inverse-CDF draw inside the kernel.
            uniforms = torch.rand(decode_step, encode_feature.shape[0], device=dev)
        # (the cached projection tanh(psi(h)) of src/asr.py:381 is computed inside the loop's node)
        logits, att, chars = ops.decoder_loop(encode_feature, None, enc_len_dev, teacher_i32,
                                              modes, uniforms, self._decoder_params(),
                                              modes_dev=modes_dev if decode_step else None,
                                              psi=(self.attention.psi.weight, self.attention.psi.bias),
                                              slots=dec_slots)
        # what the loop was driven by: characters fed to each step [U+1, B], the per-step modes
        # (0 teacher, 1 sample, 2 argmax) and the uniforms of the sampled steps [U, B] or None
        self.last_chars, self.last_modes, self.last_uniforms = chars, modes, uniforms
        if not self.att_on_host:
            host = att.detach()         # left on the device (train steps never look at it)
        elif self.training and self.async_att:
            host = torch.empty(att.shape, dtype=att.dtype, pin_memory=True)
            host.copy_(att.detach(), non_blocking=True)
            self.att_event = torch.cuda.Event()
            self.att_event.record()
        else:
            host = att.detach().cpu()
        return encode_len, logits, host
