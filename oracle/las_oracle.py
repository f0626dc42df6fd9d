"""CPU oracle for the ASR training hot path  --  TEST INFRASTRUCTURE ONLY.

A plain-PyTorch-CPU restatement of the Listen-Attend-Spell train step of
cadia-lvl/ss_asr.  It exists to *check* the HIP path; it is never the thing
shipped or measured as the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this file.  ``ss_asr_amd`` never imports it.

Parity status: PINNED.  ``oracle/make_golden.py`` ran the real reference
(imported from /root/reference in the build container) on seeded inputs and
committed its inputs/outputs under ``tests/golden/``; ``tests/test_oracle.py``
checks every function here against those vectors.  The reference holds no
tests, fixtures or golden vectors of its own.

Two layers live here:

* ``OracleASR`` and friends use ``torch.nn.LSTM`` / ``LSTMCell`` the way the
  reference does, so that timing it on the host cores is a fair "reference CPU
  path" (bench.py ``cpu_baseline.kind == "port"``).
* the ``*_explicit`` functions spell the same arithmetic out gate by gate
  (float64 capable).  Kernel-level GPU tests compare against these.

Reference citations are ``file:line`` under /root/reference.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

# vocabulary constants, src/preprocess.py:17-27
CHARS = 'abcdefghijklmnoprstuvxy0123456789'
ICE_CHARS = 'áéíóúýæöþð'
SPECIAL_CHARS = ' .,?'
ALL_CHARS = CHARS + ICE_CHARS + SPECIAL_CHARS
TOKENS = '<>$'
VOCAB = len(TOKENS + ALL_CHARS)  # 50


# --------------------------------------------------------------------------
# nn-based model (mirrors the reference's op structure)
# --------------------------------------------------------------------------
class OraclePyramid(nn.Module):
    """pBLSTM, src/asr.py:394-450: packed BiLSTM then pairwise frame concat."""

    def __init__(self, in_dim, hid):
        super().__init__()
        self.layer = nn.LSTM(in_dim, hid, bidirectional=True, batch_first=True)

    def forward(self, x, lens):
        packed = pack_padded_sequence(x, lens, batch_first=True)      # asr.py:413
        out, _ = self.layer(packed)                                   # asr.py:414
        out, out_lens = pad_packed_sequence(out, batch_first=True)    # asr.py:417
        t = out.shape[1] - (out.shape[1] % 2)                         # asr.py:443-446
        out = out[:, :t, :].contiguous().view(out.shape[0], t // 2, -1)
        return out, [int(l / 2) for l in out_lens.tolist()]           # asr.py:424


class OracleListener(nn.Module):
    """Listener, src/asr.py:214-264."""

    def __init__(self, hid, feat):
        super().__init__()
        self.state_size = hid
        self.out_dim = 2 * hid
        self.blstm_1 = OraclePyramid(feat, hid)
        self.blstm_2 = OraclePyramid(4 * hid, hid)
        self.blstm_3 = OraclePyramid(4 * hid, hid)
        # asr.py:237-238 -- NOT batch_first: recurs over the utterance axis.
        self.blstm_4 = nn.LSTM(4 * hid, hid, bidirectional=True)

    def forward(self, x, lens):
        x, lens = self.blstm_1(x, lens)
        x, lens = self.blstm_2(x, lens)
        x, lens = self.blstm_3(x, lens)
        x, _ = self.blstm_4(x)                                        # asr.py:262
        return x, lens


class OracleAttention(nn.Module):
    """Attention, src/asr.py:328-392 (content based, cached psi projection)."""

    def __init__(self, mlp, enc_dim, dec_dim):
        super().__init__()
        self.phi = nn.Linear(dec_dim, mlp, bias=False)
        self.psi = nn.Linear(enc_dim, mlp)
        self.comp = None
        self.mask = None

    def reset_enc_mem(self):
        self.comp = None
        self.mask = None

    def forward(self, state, feat, lens):
        if self.comp is None:
            idx = torch.arange(feat.shape[1]).unsqueeze(0)
            self.mask = idx >= torch.as_tensor(list(lens)).unsqueeze(1)   # asr.py:374-380
            self.comp = torch.tanh(self.psi(feat))                        # asr.py:381
        q = torch.tanh(self.phi(state))                                   # asr.py:383
        energy = torch.bmm(self.comp, q.unsqueeze(2)).squeeze(2)          # asr.py:385-386
        energy = energy.masked_fill(self.mask, float('-inf'))             # asr.py:387
        alpha = torch.softmax(energy, dim=-1)                             # asr.py:388
        ctx = torch.bmm(alpha.unsqueeze(1), feat).squeeze(1)              # asr.py:389-390
        return alpha, ctx


class OracleSpeller(nn.Module):
    """Speller, src/asr.py:267-326: two stacked LSTMCells."""

    def __init__(self, hid, enc_dim):
        super().__init__()
        self.layer_1 = nn.LSTMCell(enc_dim + hid, hid)
        self.layer_2 = nn.LSTMCell(hid, hid)
        self.state_size = hid
        self.state_list, self.cell_list = [], []

    def init_rnn(self, batch, device=None):
        z = torch.zeros(batch, self.state_size)
        self.state_list = [z, z]
        self.cell_list = [z, z]

    def forward(self, inp):
        h1, c1 = self.layer_1(inp, (self.state_list[0], self.cell_list[0]))
        h2, c2 = self.layer_2(h1, (self.state_list[1], self.cell_list[1]))
        self.state_list = [h1, h2]
        self.cell_list = [c1, c2]
        return h2


class OracleASR(nn.Module):
    """ASR, src/asr.py:15-212.  Same constructor kwargs, same state_dict keys."""

    def __init__(self, output_dim, encoder_state_size, decoder_state_size,
                 mlp_out_size, feature_dim, tf_rate):
        super().__init__()
        self.encoder = OracleListener(encoder_state_size, feature_dim)
        self.attention = OracleAttention(mlp_out_size, 2 * encoder_state_size,
                                         decoder_state_size)
        self.decoder = OracleSpeller(decoder_state_size, 2 * encoder_state_size)
        self.embed = nn.Embedding(output_dim, decoder_state_size)
        self.char_trans = nn.Linear(decoder_state_size, output_dim)
        self.tf_rate = tf_rate
        self.init_parameters()

    def init_parameters(self):
        """src/asr.py:175-212: biases 0, matrices N(0, 1/sqrt(fan_in)),
        embedding N(0,1), speller bias_ih forget slice = 1."""
        for p in self.parameters():
            if p.dim() == 1:
                p.data.zero_()
            elif p.dim() == 2:
                p.data.normal_(0, 1.0 / math.sqrt(p.size(1)))
            else:
                raise NotImplementedError
        self.embed.weight.data.normal_(0, 1)
        for cell in (self.decoder.layer_1, self.decoder.layer_2):
            n = cell.bias_ih.numel()
            cell.bias_ih.data[n // 4:n // 2].fill_(1.0)

    def forward(self, audio_feature, decode_step, teacher=None, state_len=None, forced_chars=None):
        """forced_chars ([U+1, B] int64, test hook, not in the reference): the character fed
        to step t + 1 is forced_chars[t + 1] whatever the step's mode, so that a loop whose
        sampled characters were drawn elsewhere (the HIP kernel's inverse-CDF draw) can be
        replayed and each draw checked against this model's logits (inverse_cdf_bounds)."""
        feat, enc_len = self.encoder(audio_feature, state_len)            # asr.py:63
        emb_teacher = self.embed(teacher) if teacher is not None else None
        batch = audio_feature.shape[0]
        self.decoder.init_rnn(batch)
        self.attention.reset_enc_mem()
        last = self.embed(torch.zeros(batch, dtype=torch.long))           # asr.py:73
        logits, atts = [], []
        for t in range(decode_step):                                      # asr.py:79
            alpha, ctx = self.attention(self.decoder.state_list[0], feat, enc_len)
            out = self.decoder(torch.cat([last, ctx], dim=-1))
            cur = self.char_trans(out)
            if forced_chars is not None:
                last = self.embed(forced_chars[t + 1])
            elif emb_teacher is not None:
                if random.random() <= self.tf_rate:                       # asr.py:94
                    last = emb_teacher[:, t + 1, :]
                else:
                    pick = torch.distributions.Categorical(
                        F.softmax(cur, dim=-1)).sample()
                    last = self.embed(pick)
            else:
                last = self.embed(torch.argmax(cur, dim=-1))              # asr.py:100
            logits.append(cur)
            atts.append(alpha.detach())
        return enc_len, torch.stack(logits, dim=1), torch.stack(atts, dim=1)


class OracleTextEncoder(nn.Module):
    """TextEncoder, src/text_autoencoder.py:96-107: embedding + 2-layer BiLSTM over the padded
    character rows (no packing: padded positions are encoded too)."""

    def __init__(self, char_dim, emb_dim, state_size, num_layers):
        super().__init__()
        self.emb = nn.Embedding(char_dim, emb_dim)
        self.blstm = nn.LSTM(input_size=emb_dim, hidden_size=state_size, num_layers=num_layers,
                             bidirectional=True, batch_first=True)

    def forward(self, y):
        out, _ = self.blstm(self.emb(y))
        return out


class OracleTextAutoEncoder(nn.Module):
    """TextAutoEncoder, src/text_autoencoder.py:8-94: a text encoder in place of the Listener,
    then the ASR model's own attention / speller / embed / char_trans (shared parameters)."""

    def __init__(self, char_dim, emb_dim=128, state_size=256, num_layers=2):
        super().__init__()
        self.encoder = OracleTextEncoder(char_dim, emb_dim, state_size, num_layers)

    def forward(self, asr, y, y_noised, decode_step, noise_lens=None):
        feat = self.encoder(y_noised)                                     # text_autoencoder.py:52
        batch = y_noised.shape[0]
        asr.decoder.init_rnn(batch)                                       # :55-56
        asr.attention.reset_enc_mem()
        emb = asr.embed(y)                                                # :59
        last = asr.embed(torch.zeros(batch, dtype=torch.long))            # :62-63
        outs = []
        for t in range(decode_step):                                      # :69
            _, ctx = asr.attention(asr.decoder.state_list[0], feat, noise_lens)
            cur = asr.char_trans(asr.decoder(torch.cat([last, ctx], dim=-1)))
            if t < decode_step - 1:                                       # :81
                if random.random() <= asr.tf_rate:
                    last = emb[:, t + 1, :]
                else:
                    pick = torch.distributions.Categorical(F.softmax(cur, dim=-1)).sample()
                    last = asr.embed(pick)
            else:
                last = asr.embed(torch.argmax(cur, dim=-1))               # :88
            outs.append(cur)
        return noise_lens, torch.stack(outs, dim=1)


def tae_loss(logits, y):
    """TAETrainer's loss, src/trainer.py:662-669: the label of output t is y[:, t]."""
    b, t, c = logits.shape
    per_tok = F.cross_entropy(logits.reshape(b * t, c), y.reshape(-1), ignore_index=0, reduction='none')
    per_utt = per_tok.view(b, t).sum(-1) / (y != 0).sum(-1).to(torch.float32)
    return per_utt.mean()


def inverse_cdf_bounds(logits_row):
    """Cumulative unnormalised probabilities of softmax(logits_row) in float32, summed in
    index order: (run [V], total).  A draw with uniform u picks the first v with
    run[v] > u * total -- the law of Categorical(softmax(logits)).sample() (src/asr.py:96-97)
    spelled out for a caller-supplied uniform."""
    l = np.asarray(logits_row, dtype=np.float32)
    p = np.exp(l - l.max()).astype(np.float32)
    run = np.cumsum(p, dtype=np.float32)
    return run, float(run[-1])


def masked_ce_loss(logits, y, ans_len):
    """src/trainer.py:426-434: CE(ignore 0) summed per utterance, divided by
    the count of non-zero labels in the whole row of y, mean over batch."""
    label = y[:, 1:ans_len + 1].contiguous()
    b, t, c = logits.shape
    per_tok = F.cross_entropy(logits.reshape(b * t, c), label.reshape(-1),
                              ignore_index=0, reduction='none')
    per_utt = per_tok.view(b, t).sum(-1) / (y != 0).sum(-1).to(torch.float32)
    return per_utt.mean()


def label_lengths(y):
    """src/ASRDataset.py:338: count(y != 0) + 1."""
    return [int(v) + 1 for v in (y != 0).sum(-1)]


def frame_lengths(x):
    """src/ASRDataset.py:314: frames whose feature sum is non-zero."""
    return [int(v) for v in (x.sum(-1) != 0).sum(-1)]


def solver_step(params, optim, grad_clip=5):
    """src/trainer.py:131-148.  Returns (grad_norm, stepped)."""
    norm = nn.utils.clip_grad_norm_(params, grad_clip)
    if math.isnan(float(norm)):
        return float(norm), False
    optim.step()
    return float(norm), True


def train_step(model, optim, x, y):
    """One iteration of ASRTrainer.exec, src/trainer.py:415-438.  Returns
    (loss, grad_norm)."""
    lens = frame_lengths(x)
    ans_len = max(label_lengths(y)) - 1
    optim.zero_grad()
    _, logits, _ = model(x, ans_len, teacher=y, state_len=lens)
    loss = masked_ce_loss(logits, y, ans_len)
    loss.backward()
    norm, _ = solver_step(list(model.parameters()), optim)
    return float(loss.detach()), norm


def ctc_branch_loss(model, head, x, y, lens, blank=0):
    """CTC branch of BASELINE.json configs[3].  BUILD-DEFINED (the reference has no CTC,
    SURVEY.md section 1; parity unpinned by the reference): the checker SURVEY.md 8(c) names,
    torch's ctc_loss, on this oracle's Listener: Linear head on the encoder output, the
    characters between <sos> and <eos> as labels, blank = class 0."""
    feat, enc_len = model.encoder(x, lens)
    lp = F.log_softmax(head(feat), dim=-1).transpose(0, 1)                 # [T', B, V]
    label_lens = ((y != 0).sum(-1) - 1).clamp(min=0)
    return F.ctc_loss(lp, y[:, 1:], torch.tensor([int(v) for v in enc_len]), label_lens, blank=blank,
                      reduction='mean', zero_infinity=True)


def joint_train_step(model, head, optim, x, y, ctc_weight):
    """train_step with loss = ctc_weight * ctc + (1 - ctc_weight) * masked CE; optim holds the
    model's parameters followed by the head's.  Returns (loss, attention loss, ctc loss)."""
    lens = frame_lengths(x)
    ans_len = max(label_lengths(y)) - 1
    optim.zero_grad()
    _, logits, _ = model(x, ans_len, teacher=y, state_len=lens)
    att = masked_ce_loss(logits, y, ans_len)
    ctc = ctc_branch_loss(model, head, x, y, lens)
    loss = ctc_weight * ctc + (1.0 - ctc_weight) * att
    loss.backward()
    solver_step(list(model.parameters()) + list(head.parameters()), optim)
    return float(loss.detach()), float(att.detach()), float(ctc.detach())


def make_optimizer(model):
    """src/trainer.py:401-403 with conf/default.yaml:2-4."""
    return torch.optim.Adadelta(model.parameters(), lr=1.0, eps=1e-8)


def make_tae_optimizer(tae, asr, lr=1e-4, kind='Adam'):
    """TAETrainer.set_model, src/trainer.py:633-641 with conf/default.yaml:43-45: ONE optimizer over the
    whole text autoencoder and the ASR model's embed / attention / decoder / char_trans (not its Listener)."""
    params = (list(tae.parameters()) + list(asr.embed.parameters()) + list(asr.attention.parameters()) +
              list(asr.decoder.parameters()) + list(asr.char_trans.parameters()))
    return getattr(torch.optim, kind)(params, lr=lr, eps=1e-8)


def tae_train_step(asr, tae, optim, y, y_noise):
    """One iteration of TAETrainer.exec, src/trainer.py:652-677: decode for max(y_lens) steps, the loss of
    :662-672, backward, then Solver.step on the TEXT AUTOENCODER's parameters -- so the norm that is
    clipped (and tested for NaN) is the text autoencoder's alone, while the optimizer steps the shared
    ASR parameters too.  Returns (loss, clipped-norm)."""
    y_lens, noise_lens = label_lengths(y), label_lengths(y_noise)
    optim.zero_grad()
    _, logits = tae(asr, y, y_noise, max(y_lens), noise_lens=noise_lens)
    loss = tae_loss(logits, y)
    loss.backward()
    norm, _ = solver_step(list(tae.parameters()), optim)
    return float(loss.detach()), norm


# --------------------------------------------------------------------------
# the Seed loop's other legs (BASELINE.json configs[4]): ADVTrainer, SAETrainer
# --------------------------------------------------------------------------
class OracleDiscriminator(nn.Module):
    """Discriminator, src/discriminator.py:4-54: Linear(in, hid) ReLU Linear(hid, hid) ReLU Linear(hid, 1),
    then a sigmoid; scores every frame of [batch, seq, in]."""

    def __init__(self, in_dim, hidden_dim=256):
        super().__init__()
        self.core = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim),
                                  nn.ReLU(), nn.Linear(hidden_dim, 1))

    def forward(self, x):
        return torch.sigmoid(self.core(x))


def make_adv_optimizers(asr, disc, g_opt=('Adadelta', 1.0), d_opt=('Adadelta', 1.0)):
    """ADVTrainer.set_model, src/trainer.py:938-948 with conf/default.yaml:63-68: the generator optimizer holds
    the Listener's parameters, the discriminator optimizer the discriminator's."""
    G = getattr(torch.optim, g_opt[0])(asr.encoder.parameters(), lr=g_opt[1], eps=1e-8)
    D = getattr(torch.optim, d_opt[0])(disc.parameters(), lr=d_opt[1], eps=1e-8)
    return G, D


def adv_train_step(asr, text_encoder, disc, G_optim, D_optim, x, y, label_smoothing=0.1):
    """One iteration of ADVTrainer.exec, src/trainer.py:968-1032, with the loss the reference names but never
    defines (`self.loss_metric`, :984) taken as nn.BCELoss (src/discriminator.py:17-19).  Returns
    (D_realloss, D_fakeloss, G_loss, D norm, G norm)."""
    bce = nn.BCELoss()
    lens = frame_lengths(x)
    batch = x.shape[0]
    disc.zero_grad()
    real = text_encoder(y)                                                  # :978
    d_out = disc(real)
    d_real = bce(d_out.squeeze(dim=2), torch.ones(batch, real.shape[1]) - label_smoothing)
    d_real.backward()
    fake, _ = asr.encoder(x, lens)                                          # :988
    d_out = disc(fake.detach())
    d_fake = bce(d_out.squeeze(dim=2), torch.zeros(batch, fake.shape[1]))
    d_fake.backward()
    d_norm, _ = solver_step(list(disc.parameters()), D_optim)               # :999
    asr.encoder.zero_grad()                                                 # :1005
    d_out = disc(fake)                                                      # the UPDATED discriminator
    g_loss = bce(d_out.squeeze(dim=2), torch.ones(batch, fake.shape[1]))
    g_loss.backward()
    g_norm, _ = solver_step(list(asr.encoder.parameters()), G_optim)        # :1029
    return float(d_real.detach()), float(d_fake.detach()), float(g_loss.detach()), d_norm, g_norm


class OracleSpeechAutoEncoder(nn.Module):
    """SpeechAutoEncoder, src/speech_autoencoder.py:5-203: a global encoder of three Conv2d(bias False) /
    BatchNorm2d / ReLU / MaxPool2d blocks over [batch, 1, seq, feat] (:118-160) and a decoder
    Linear LeakyReLU Linear LeakyReLU Linear (:183-188) applied to [listener frame | global encoding] once per
    Listener frame, each call predicting the 8 input frames under that Listener frame (:62-94)."""

    def __init__(self, listener_out_dim, feature_dim, kernel_sizes, num_filters, pool_kernel_sizes):
        super().__init__()
        self.feature_dim = feature_dim
        enc = nn.Module()
        chans = [1] + list(num_filters)
        for k in range(3):
            setattr(enc, 'conv_%d' % (k + 1), nn.Sequential(
                nn.Conv2d(chans[k], chans[k + 1], kernel_size=tuple(kernel_sizes[k]), padding=0, bias=False),
                nn.BatchNorm2d(chans[k + 1]), nn.ReLU(), nn.MaxPool2d(tuple(pool_kernel_sizes[k]))))
        self.encoder = enc
        dec = nn.Module()
        in_dim = num_filters[-1] + listener_out_dim
        dec.core = nn.Sequential(nn.Linear(in_dim, in_dim), nn.LeakyReLU(), nn.Linear(in_dim, in_dim), nn.LeakyReLU(),
                                 nn.Linear(in_dim, 8 * feature_dim))
        self.decoder = dec

    def encode(self, x):
        h = x.unsqueeze(1)
        for k in (1, 2, 3):
            h = getattr(self.encoder, 'conv_%d' % k)(h)
        return h.squeeze(2).squeeze(2)

    def forward(self, x, listener_out, just_first=False):
        g = self.encode(x)
        frames = []
        for i in range(1 if just_first else listener_out.shape[1]):
            out = self.decoder.core(torch.cat((listener_out[:, i, :], g), dim=1))
            frames.append(out.view(out.shape[0], 8, self.feature_dim))
        return torch.cat(frames, dim=1)


def make_sae_optimizer(sae, asr, lr=1e-4, kind='Adam'):
    """SAETrainer.set_model, src/trainer.py:789-794 with conf/default.yaml:24-26: ONE optimizer over the whole
    speech autoencoder and the ASR model's Listener."""
    return getattr(torch.optim, kind)(list(sae.parameters()) + list(asr.encoder.parameters()), lr=lr, eps=1e-8)


def sae_loss(pred, x, batch_t):
    """src/trainer.py:811-818: the prediction padded with zero frames up to batch_t against x[:, :batch_t]."""
    full = torch.zeros(pred.shape[0], batch_t, pred.shape[2])
    full[:, :pred.shape[1], :] = pred
    return F.smooth_l1_loss(full, x[:, :batch_t, :])


def sae_train_step(asr, sae, optim, x):
    """One iteration of SAETrainer.exec, src/trainer.py:803-820: Listener, speech autoencoder, smooth-L1 against
    the input frames, backward, then Solver.step on the SPEECH AUTOENCODER's parameters (the clipped norm is
    theirs alone; the optimizer steps the Listener too).  Returns (loss, clipped norm)."""
    lens = frame_lengths(x)
    optim.zero_grad()
    listener_out, _ = asr.encoder(x, lens)
    pred = sae(x, listener_out)
    loss = sae_loss(pred, x, max(lens))
    loss.backward()
    norm, _ = solver_step(list(sae.parameters()), optim)
    return float(loss.detach()), norm


def seeded_generic_weights(module, seed):
    """numpy-PCG64 parameters for the Seed loop's extra modules (Discriminator, SpeechAutoEncoder; reference
    or oracle: same names and order): matrices and convolution kernels N(0, 1/sqrt(fan_in)) with fan_in =
    everything but the leading axis; batch-norm scales 1 + N(0, 0.1); every other vector N(0, 0.1)."""
    rng = np.random.default_rng(seed)
    bn_scales = {id(m.weight) for m in module.modules() if isinstance(m, nn.BatchNorm2d)}
    with torch.no_grad():
        for _, p in module.named_parameters():
            draw = rng.standard_normal(tuple(p.shape))
            if p.dim() > 1:
                draw *= 1.0 / math.sqrt(int(np.prod(p.shape[1:])))
            elif id(p) in bn_scales:
                draw = 1.0 + 0.1 * draw
            else:
                draw *= 0.1
            p.copy_(torch.from_numpy(draw.astype(np.float32)))
    return module


# --------------------------------------------------------------------------
# explicit arithmetic (kernel-level oracle)
# --------------------------------------------------------------------------
def lstm_gates_explicit(pre, c_prev):
    """One LSTM cell update from gate pre-activations ``pre`` [.., 4H] in the
    PyTorch row order i, f, g, o."""
    h = pre.shape[-1] // 4
    i = torch.sigmoid(pre[..., 0:h])
    f = torch.sigmoid(pre[..., h:2 * h])
    g = torch.tanh(pre[..., 2 * h:3 * h])
    o = torch.sigmoid(pre[..., 3 * h:4 * h])
    c = f * c_prev + i * g
    return o * torch.tanh(c), c


def lstm_dir_explicit(x, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of a packed LSTM over time-major input.

    x: [S, N, I]; lens: N ints or None (full length).  Column n is updated at
    step s only while s < lens[n]; the reverse direction walks s downwards, so
    each column effectively starts from zero state at its own last frame.
    Outputs past a column's length are zero (pad_packed_sequence)."""
    steps, cols, _ = x.shape
    hid = w_hh.shape[1]
    h = x.new_zeros(cols, hid)
    c = x.new_zeros(cols, hid)
    out = x.new_zeros(steps, cols, hid)
    if lens is None:
        lens_t = torch.full((cols,), steps, dtype=torch.long)
    else:
        lens_t = torch.as_tensor(list(lens), dtype=torch.long)
    order = range(steps - 1, -1, -1) if reverse else range(steps)
    for s in order:
        pre = x[s] @ w_ih.t() + h @ w_hh.t() + b_ih + b_hh
        h_new, c_new = lstm_gates_explicit(pre, c)
        live = (s < lens_t).unsqueeze(1)
        h = torch.where(live, h_new, h)
        c = torch.where(live, c_new, c)
        out[s] = torch.where(live, h_new, torch.zeros_like(h_new))
    return out


def bilstm_explicit(x, lens, weights):
    """Bidirectional layer.  ``weights`` = (w_ih, w_hh, b_ih, b_hh) for the
    forward direction followed by the same four for the reverse direction.
    x is time-major [S, N, I]; returns [S, N, 2H]."""
    fwd = lstm_dir_explicit(x, lens, *weights[0:4], reverse=False)
    bwd = lstm_dir_explicit(x, lens, *weights[4:8], reverse=True)
    return torch.cat([fwd, bwd], dim=-1)


def pyramid_explicit(x_bt, lens, weights):
    """pBLSTM on batch-first input: packed BiLSTM up to max(lens) frames, drop
    an odd last frame, concatenate frame pairs (src/asr.py:406-450)."""
    tmax = max(lens)
    y = bilstm_explicit(x_bt[:, :tmax].transpose(0, 1), lens, weights)
    y = y.transpose(0, 1)
    t = tmax - (tmax % 2)
    y = y[:, :t].contiguous().view(y.shape[0], t // 2, -1)
    return y, [int(l / 2) for l in lens]


def attention_step_explicit(state, feat, comp, lens, w_phi):
    """One Attention.forward call after the cached psi projection
    (src/asr.py:383-390).  Returns (alpha [B,T'], ctx [B,E])."""
    q = torch.tanh(state @ w_phi.t())
    energy = torch.einsum('bta,ba->bt', comp, q)
    idx = torch.arange(feat.shape[1]).unsqueeze(0)
    energy = energy.masked_fill(idx >= torch.as_tensor(list(lens)).unsqueeze(1),
                                float('-inf'))
    alpha = torch.softmax(energy, dim=-1)
    return alpha, torch.einsum('bt,bte->be', alpha, feat)


def lstm_cell_explicit(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell arithmetic (src/asr.py:320-324)."""
    return lstm_gates_explicit(x @ w_ih.t() + h @ w_hh.t() + b_ih + b_hh, c)


def clip_adadelta_explicit(params, grads, sq_avg, acc_delta, max_norm=5.0,
                           lr=1.0, rho=0.9, eps=1e-8):
    """clip_grad_norm_ (src/trainer.py:144) followed by torch.optim.Adadelta
    (src/trainer.py:148, :401-403), on lists of tensors, in place.
    Returns (total_norm, stepped)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    if math.isnan(float(total)):
        return float(total), False
    coef = min(1.0, max_norm / (float(total) + 1e-6))
    for p, g, sq, ad in zip(params, grads, sq_avg, acc_delta):
        g = g * coef
        sq.mul_(rho).addcmul_(g, g, value=1 - rho)
        delta = (ad + eps).sqrt() / (sq + eps).sqrt() * g
        ad.mul_(rho).addcmul_(delta, delta, value=1 - rho)
        p.sub_(lr * delta)
    return float(total), True


def seeded_weights(model, seed):
    """Overwrites every parameter with numpy-PCG64 draws that follow the
    reference's init law (N(0, 1/sqrt(fan_in)), embedding N(0,1), biases 0,
    speller forget bias 1).  Used where a fixture must not carry 41 MB of
    weights: the same call reproduces them on any host."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.zero_()
            else:
                std = 1.0 if name == 'embed.weight' else 1.0 / math.sqrt(p.shape[1])
                p.copy_(torch.from_numpy(
                    (rng.standard_normal(tuple(p.shape)) * std).astype(np.float32)))
        for cell in (model.decoder.layer_1, model.decoder.layer_2):
            n = cell.bias_ih.numel()
            cell.bias_ih[n // 4:n // 2] = 1.0
    return model


def seeded_tae_weights(tae, seed):
    """numpy-PCG64 parameters for a TextAutoEncoder (reference or oracle: same parameter names and
    order): matrices N(0, 1/sqrt(fan_in)), the embedding N(0, 1), biases N(0, 0.1) -- non-zero on
    purpose, the ASR fixtures only ever see zero encoder biases."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in tae.named_parameters():
            std = 1.0 if name == 'encoder.emb.weight' else (1.0 / math.sqrt(p.shape[-1]) if p.dim() > 1 else 0.1)
            p.copy_(torch.from_numpy((rng.standard_normal(tuple(p.shape)) * std).astype(np.float32)))
    return tae
