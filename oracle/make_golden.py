"""Generates tests/golden/*.npz by RUNNING THE REAL REFERENCE on CPU.

Build-container only (needs /root/reference; see oracle/ref_harness.py).
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Every fixture is data: seeded inputs, the reference's outputs, and (for the
small model) its weights.  Full-size cases carry no weights; they are
re-created on any host by ``las_oracle.seeded_weights(model, seed)``, a numpy
PCG64 stream.  The loss / clip / Adadelta lines below restate
src/trainer.py:426-438 around the reference's own ``ASR`` class, because
``ASRTrainer`` itself cannot be constructed without tensorboardX and a dataset
on disk.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import import_reference  # noqa: E402
from las_oracle import seeded_tae_weights, seeded_weights  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def synth_batch(rng, lens, feat, y_lens, vocab=50):
    """Zero-padded N(0,1) fbanks with the given frame lengths and label rows
    '<' chars '>' padded with 0 (src/ASRDataset.py:131-151)."""
    b, t = len(lens), max(lens)
    x = rng.standard_normal((b, t, feat)).astype(np.float32)
    for i, l in enumerate(lens):
        x[i, l:] = 0
    width = max(y_lens) + 2
    y = np.zeros((b, width), dtype=np.int64)
    for i, l in enumerate(y_lens):
        y[i, 1:1 + l] = rng.integers(3, vocab, size=l)
        y[i, 1 + l] = 1
    return torch.from_numpy(x), torch.from_numpy(y)


def ref_loss(logits, y, ans_len):
    # src/trainer.py:426-434
    metric = nn.CrossEntropyLoss(ignore_index=0, reduction='none')
    label = y[:, 1:ans_len + 1].contiguous()
    b, t, c = logits.shape
    loss = metric(logits.view(b * t, c), label.view(-1))
    loss = torch.sum(loss.view(b, t), dim=-1) / torch.sum(y != 0, dim=-1).to(torch.float32)
    return torch.mean(loss)


def capture(asr_mod, name, dims, lens, y_lens, tf_rate, seed, weights_seed=None,
            teacher=True, pad_to=None, keep='all', extra_decode=0, xy=None, recipe=None):
    """xy: (x, y) made by the caller instead of synth_batch; with `recipe` (a dict of
    integers that lets a test rebuild the same x, y) the fixture carries the recipe and
    not the 8 MB input."""
    seed_all(seed)
    model = asr_mod.ASR(*dims, tf_rate)
    if weights_seed is not None:
        seeded_weights(model, weights_seed)
    rng = np.random.default_rng(seed + 1000)
    if xy is not None:
        x, y = xy
    else:
        x, y = synth_batch(rng, lens, dims[4], y_lens, dims[0])
    if pad_to is not None:       # dataset-wide zero padding beyond the batch max
        x = torch.cat([x, x.new_zeros(x.shape[0], pad_to - x.shape[1], x.shape[2])], 1)
    ans_len = int(max((y != 0).sum(-1) + 1)) - 1
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    taps = {}
    hooks = []
    for nm in ('blstm_1', 'blstm_2', 'blstm_3', 'blstm_4'):
        def hook(_m, _i, out, nm=nm):
            taps[nm] = out[0].detach().clone()
        hooks.append(getattr(model.encoder, nm).register_forward_hook(hook))

    optim = torch.optim.Adadelta(model.parameters(), lr=1.0, eps=1e-8)
    optim.zero_grad()
    seed_all(seed + 7)           # pins the coin flips / samples of the decode loop
    steps = ans_len + extra_decode
    enc_len, logits, att = model(x, steps, teacher=y if teacher else None,
                                 state_len=list(lens))
    for h in hooks:
        h.remove()
    loss = ref_loss(logits[:, :ans_len].contiguous(), y, ans_len)   # trainer.py:484-485
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    grad_norm = nn.utils.clip_grad_norm_(model.parameters(), 5)      # trainer.py:144
    optim.step()                                                     # trainer.py:148
    sd1 = model.state_dict()

    # the reference's own character accuracy (src/postprocess.py:7-29) on the first ans_len outputs, as
    # ASRTrainer logs it (src/trainer.py:443) and valid() averages it (:486); calc_err needs the absent
    # `editdistance` package and is not captured
    import postprocess as ref_post                              # reference module (path set by import_reference)
    acc = ref_post.calc_acc(logits[:, :ans_len], y[:, 1:ans_len + 1])
    out = dict(dims=np.array(dims), tf_rate=np.float64(tf_rate), seed=np.int64(seed), acc=np.float64(acc),
               weights_seed=np.int64(-1 if weights_seed is None else weights_seed),
               teacher=np.int64(teacher), rng_seed=np.int64(seed + 7),
               y=y.numpy(), lens=np.array(lens), ans_len=np.int64(ans_len),
               decode_steps=np.int64(steps),
               enc_len=np.array(enc_len), logits=logits.detach().numpy(),
               loss=np.float64(loss.item()),
               grad_norm=np.float64(float(grad_norm)))
    if recipe is None:
        out['x'] = x.numpy()
        out['att'] = att.numpy()
    else:
        for k, v in recipe.items():
            out['recipe_' + k] = np.int64(v)
        # input: a checksum instead of the tensor; attention map: every fourth utterance
        out['x_abs_sum'] = np.float64(x.double().abs().sum().item())
        out['att_rows'] = np.arange(0, att.shape[0], 4)
        out['att'] = att[::4].numpy()
    names = list(grads.keys())
    out['param_names'] = np.array(names)
    out['grad_norms'] = np.array([grads[k].double().norm().item() for k in names])
    out['update_norms'] = np.array([(sd1[k] - sd0[k]).double().norm().item() for k in names])
    for nm, v in taps.items():
        if keep in ('all', 'acts'):
            out['act_' + nm] = v.numpy()
        else:                     # compact: a strided sample plus a checksum
            flat = v.reshape(-1)
            out['act_' + nm + '_sample'] = flat[::max(1, flat.numel() // 512)][:512].numpy()
            out['act_' + nm + '_abs_sum'] = np.float64(flat.double().abs().sum().item())
    if keep == 'all':
        for k in names:
            out['w0/' + k] = sd0[k].numpy()
            out['g/' + k] = grads[k].numpy()
            out['w1/' + k] = sd1[k].numpy()
    else:
        for k in ('attention.phi.weight', 'encoder.blstm_1.layer.weight_hh_l0',
                  'decoder.layer_1.weight_ih', 'char_trans.bias',
                  'encoder.blstm_4.weight_ih_l0_reverse'):
            out['g_head/' + k] = grads[k].reshape(-1)[:256].numpy()
            out['w1_head/' + k] = sd1[k].reshape(-1)[:256].numpy()
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s loss %.6f  |g| %.6f  enc_len %s  %.1f KB' % (
        name, loss.item(), float(grad_norm), enc_len[:6], os.path.getsize(path) / 1024))


def noisy_text_batch(rng, y_lens, drop_rate, vocab=50):
    """Label rows '<' chars '>' padded with 0 and their noised copies (characters dropped with
    probability drop_rate, never the tokens: src/ASRDataset.py:111-127), as TAETrainer gets them."""
    rows, noisy = [], []
    for l in y_lens:
        chars = rng.integers(3, vocab, size=l).tolist()
        rows.append([0] + chars + [1])
        noisy.append([0] + [c for c in chars if rng.random() > drop_rate] + [1])
    def pad(rs):
        w = max(len(r) for r in rs)
        out = np.zeros((len(rs), w), dtype=np.int64)
        for i, r in enumerate(rs):
            out[i, :len(r)] = r
        return torch.from_numpy(out)
    return pad(rows), pad(noisy)


def capture_tae(asr_mod, tae_mod, name, dims, tae_dims, y_lens, tf_rate, seed, drop_rate=0.2):
    """TextAutoEncoder.forward (src/text_autoencoder.py:31-94) + TAETrainer's loss
    (src/trainer.py:662-672) + backward on the real reference, both models with seeded weights."""
    seed_all(seed)
    asr = asr_mod.ASR(*dims, tf_rate)
    seeded_weights(asr, seed + 100)
    tae = tae_mod.TextAutoEncoder(dims[0], *tae_dims)
    seeded_tae_weights(tae, seed + 200)
    rng = np.random.default_rng(seed + 1000)
    y, y_noise = noisy_text_batch(rng, y_lens, drop_rate, dims[0])
    y_l = [int(v) + 1 for v in (y != 0).sum(-1)]                 # prepare_y
    noise_l = [int(v) + 1 for v in (y_noise != 0).sum(-1)]
    decode_step = max(y_l)
    assert decode_step == y.shape[1]
    seed_all(seed + 7)
    _, logits = tae(asr, y, y_noise, decode_step, noise_lens=noise_l)
    metric = nn.CrossEntropyLoss(ignore_index=0, reduction='none')          # trainer.py:637-638
    b, t, c = logits.shape
    loss = metric(logits.view(b * t, c), y.view(-1))
    loss = torch.mean(torch.sum(loss.view(b, t), dim=-1) / torch.sum(y != 0, dim=-1).to(torch.float32))
    loss.backward()
    out = dict(dims=np.array(dims), tae_dims=np.array(tae_dims), tf_rate=np.float64(tf_rate),
               seed=np.int64(seed), rng_seed=np.int64(seed + 7), y=y.numpy(), y_noise=y_noise.numpy(),
               noise_lens=np.array(noise_l), decode_step=np.int64(decode_step),
               logits=logits.detach().numpy(), loss=np.float64(loss.item()))
    out['asr_weights_seed'] = np.int64(seed + 100)
    out['tae_weights_seed'] = np.int64(seed + 200)
    grads = {('tae.' + k): p.grad for k, p in tae.named_parameters()}
    grads.update({('asr.' + k): p.grad for k, p in asr.named_parameters() if p.grad is not None})
    names = sorted(grads)
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array([grads[k].double().norm().item() for k in names])
    for k in ('tae.encoder.emb.weight', 'tae.encoder.blstm.weight_hh_l0', 'tae.encoder.blstm.weight_ih_l1_reverse',
              'asr.decoder.layer_1.weight_ih', 'asr.attention.phi.weight', 'asr.attention.psi.weight',
              'asr.embed.weight', 'asr.char_trans.bias'):
        out['g_head/' + k] = grads[k].reshape(-1)[:256].numpy()
    out['asr_no_grad'] = np.array(sorted(k for k, p in asr.named_parameters() if p.grad is None))
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s loss %.6f  decode_step %d  %.1f KB' % (name, loss.item(), decode_step, os.path.getsize(path) / 1024))


def ref_tae_step(tae, asr, optim, y, y_noise):
    """The body of the reference's TAETrainer.exec (src/trainer.py:652-677) around the reference's own
    TextAutoEncoder / ASR classes (TAETrainer itself needs tensorboardX and an index on disk)."""
    y_l = [int(v) + 1 for v in (y != 0).sum(-1)]                 # prepare_y
    noise_l = [int(v) + 1 for v in (y_noise != 0).sum(-1)]
    optim.zero_grad()
    _, logits = tae(asr, y, y_noise, max(y_l), noise_lens=noise_l)
    metric = nn.CrossEntropyLoss(ignore_index=0, reduction='none')          # trainer.py:637-638
    b, t, c = logits.shape
    loss = metric(logits.view(b * t, c), y.view(-1))
    loss = torch.mean(torch.sum(loss.view(b, t), dim=-1) / torch.sum(y != 0, dim=-1).to(torch.float32))
    loss.backward()
    grad_norm = nn.utils.clip_grad_norm_(tae.parameters(), 5)   # Solver.step(self.text_autoenc.parameters(), ...)
    optim.step()
    return float(loss.item()), float(grad_norm)


def ref_tae_optimizer(tae, asr, lr):
    # trainer.py:633-641 (conf/default.yaml:43-45: Adam, 1e-4)
    return torch.optim.Adam(list(tae.parameters()) + list(asr.embed.parameters()) + list(asr.attention.parameters()) +
                            list(asr.decoder.parameters()) + list(asr.char_trans.parameters()), lr=lr, eps=1e-8)


def capture_tae_traj(asr_mod, tae_mod, name, dims, tae_dims, batches_y_lens, tf_rate, seed, lr=1e-4,
                     asr_batches=None, drop_rate=0.2):
    """A trajectory of TAETrainer steps on the real reference (config 5's first leg, src/trainer.py:594-758):
    for every entry of `batches_y_lens` one TAE step (Adam over the text autoencoder + the ASR model's
    embed / attention / decoder / char_trans; norm clipped over the text autoencoder only).  With
    `asr_batches` (frame lengths, label lengths per round) an ASRTrainer step (src/trainer.py:415-438,
    Adadelta over the whole ASR model) runs BEFORE each TAE step on the SAME ASR object: the two legs of
    the Seed loop sharing parameters."""
    seed_all(seed)
    asr = asr_mod.ASR(*dims, tf_rate)
    seeded_weights(asr, seed + 100)
    tae = tae_mod.TextAutoEncoder(dims[0], *tae_dims)
    seeded_tae_weights(tae, seed + 200)
    tae_opt = ref_tae_optimizer(tae, asr, lr)
    asr_opt = torch.optim.Adadelta(asr.parameters(), lr=1.0, eps=1e-8) if asr_batches else None
    rng = np.random.default_rng(seed + 1000)
    w0 = {('tae.' + k): v.clone() for k, v in tae.state_dict().items()}
    w0.update({('asr.' + k): v.clone() for k, v in asr.state_dict().items()})
    out = dict(dims=np.array(dims), tae_dims=np.array(tae_dims), tf_rate=np.float64(tf_rate), seed=np.int64(seed),
               lr=np.float64(lr), asr_weights_seed=np.int64(seed + 100), tae_weights_seed=np.int64(seed + 200),
               rounds=np.int64(len(batches_y_lens)), with_asr=np.int64(1 if asr_batches else 0))
    tae_loss_l, tae_norm_l, asr_loss_l, asr_norm_l = [], [], [], []
    for r, y_lens in enumerate(batches_y_lens):
        if asr_batches:
            lens, ylens = asr_batches[r]
            x, ya = synth_batch(rng, lens, dims[4], ylens, dims[0])
            ans_len = int(max((ya != 0).sum(-1) + 1)) - 1
            asr_opt.zero_grad()
            seed_all(seed + 7 + 2 * r)
            _, logits, _ = asr(x, ans_len, teacher=ya, state_len=list(lens))
            loss = ref_loss(logits, ya, ans_len)
            loss.backward()
            asr_norm_l.append(float(nn.utils.clip_grad_norm_(asr.parameters(), 5)))
            asr_opt.step()
            asr_loss_l.append(float(loss.item()))
            out['asr_x%d' % r], out['asr_y%d' % r], out['asr_lens%d' % r] = x.numpy(), ya.numpy(), np.array(lens)
        y, y_noise = noisy_text_batch(rng, y_lens, drop_rate, dims[0])
        seed_all(seed + 8 + 2 * r)
        l, n = ref_tae_step(tae, asr, tae_opt, y, y_noise)
        tae_loss_l.append(l)
        tae_norm_l.append(n)
        out['y%d' % r], out['y_noise%d' % r] = y.numpy(), y_noise.numpy()
        out['rng_seed%d' % r] = np.int64(seed + 8 + 2 * r)
        out['asr_rng_seed%d' % r] = np.int64(seed + 7 + 2 * r)
    out['tae_loss'], out['tae_norm'] = np.array(tae_loss_l), np.array(tae_norm_l)
    if asr_batches:
        out['asr_loss'], out['asr_norm'] = np.array(asr_loss_l), np.array(asr_norm_l)
    w1 = {('tae.' + k): v for k, v in tae.state_dict().items()}
    w1.update({('asr.' + k): v for k, v in asr.state_dict().items()})
    names = sorted(w1)
    out['param_names'] = np.array(names)
    out['update_norms'] = np.array([(w1[k] - w0[k]).double().norm().item() for k in names])
    small = sum(v.numel() for v in w1.values()) < 400000
    for k in names:
        if small:
            out['w1/' + k] = w1[k].numpy()
        else:
            out['w1_head/' + k] = w1[k].reshape(-1)[:256].numpy()
            out['dw_head/' + k] = (w1[k] - w0[k]).reshape(-1)[:256].numpy()
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s tae loss %s  norm %s  %.1f KB' % (name, np.round(tae_loss_l, 5), np.round(tae_norm_l, 5),
                                                  os.path.getsize(path) / 1024))


def ref_adv_step(disc, asr, text_encoder, D_optim, G_optim, x, x_lens, y, label_smoothing):
    """The body of the reference's ADVTrainer.exec (src/trainer.py:968-1032) around the reference's own
    Discriminator / ASR / TextAutoEncoder classes.  ADVTrainer itself cannot be constructed (tensorboardX, an
    index on disk, config keys the yaml lacks) and reads an attribute it never sets: `self.loss_metric`
    (:984) is nn.BCELoss here, the loss src/discriminator.py:17-19 names -- the ONE departure from the
    reference's text, recorded in the fixture."""
    loss_metric = nn.BCELoss()
    batch_size = x.shape[0]
    disc.zero_grad()
    real_data = text_encoder(y)
    D_out = disc(real_data)
    real_labels = torch.ones(batch_size, real_data.shape[1]) - label_smoothing
    D_realloss = loss_metric(D_out.squeeze(dim=2), real_labels)
    D_realloss.backward()
    fake_data, _ = asr.encoder(x, x_lens)
    D_out = disc(fake_data.detach())
    fake_labels = torch.zeros(batch_size, fake_data.shape[1])
    D_fakeloss = loss_metric(D_out.squeeze(dim=2), fake_labels)
    D_fakeloss.backward()
    d_norm = float(nn.utils.clip_grad_norm_(disc.parameters(), 5))          # Solver.step(discriminator.parameters(), D_optim)
    D_optim.step()
    asr.encoder.zero_grad()
    fake_labels = torch.ones(batch_size, fake_data.shape[1])
    D_out = disc(fake_data)
    G_loss = loss_metric(D_out.squeeze(dim=2), fake_labels)
    G_loss.backward()
    g_norm = float(nn.utils.clip_grad_norm_(asr.encoder.parameters(), 5))   # Solver.step(asr_model.encoder.parameters(), G_optim)
    G_optim.step()
    return float(D_realloss.item()), float(D_fakeloss.item()), float(G_loss.item()), d_norm, g_norm


def _weights_record(out, w0, w1, keep_all):
    names = sorted(w1)
    out['param_names'] = np.array(names)
    out['update_norms'] = np.array([(w1[k].double() - w0[k].double()).norm().item() for k in names])
    for k in names:
        if keep_all:
            out['w1/' + k] = w1[k].numpy()
        else:
            out['w1_head/' + k] = w1[k].reshape(-1)[:256].numpy()
            out['dw_head/' + k] = (w1[k] - w0[k]).reshape(-1)[:256].numpy()


def capture_adv_traj(asr_mod, tae_mod, disc_mod, name, dims, tae_dims, hidden, batches, seed, g_opt, d_opt,
                     label_smoothing=0.1):
    """A trajectory of ADVTrainer iterations on the real reference's classes (config 5's second leg,
    src/trainer.py:909-1124): per entry of `batches` (frame lengths, label lengths) one iteration -- the
    discriminator's two passes and step, then the generator's pass and step.  The batch itself is a recipe:
    ss_asr_amd.synthetic.make_batch(lens, ylens, feat, batch_seed)."""
    from las_oracle import seeded_generic_weights
    from ss_asr_amd.synthetic import make_batch
    seed_all(seed)
    asr = asr_mod.ASR(*dims, 1.0)
    seeded_weights(asr, seed + 100)
    tae = tae_mod.TextAutoEncoder(dims[0], *tae_dims)
    seeded_tae_weights(tae, seed + 200)
    disc = disc_mod.Discriminator(asr.encoder.get_outdim(), hidden_dim=hidden)
    seeded_generic_weights(disc, seed + 300)
    G_optim = getattr(torch.optim, g_opt[0])(asr.encoder.parameters(), lr=g_opt[1], eps=1e-8)     # trainer.py:938-943
    D_optim = getattr(torch.optim, d_opt[0])(disc.parameters(), lr=d_opt[1], eps=1e-8)            # :945-948
    w0 = {('disc.' + k): v.clone() for k, v in disc.state_dict().items()}
    w0.update({('asr.' + k): v.clone() for k, v in asr.state_dict().items()})
    out = dict(dims=np.array(dims), tae_dims=np.array(tae_dims), hidden=np.int64(hidden), seed=np.int64(seed),
               asr_weights_seed=np.int64(seed + 100), tae_weights_seed=np.int64(seed + 200),
               disc_weights_seed=np.int64(seed + 300), rounds=np.int64(len(batches)),
               g_opt=np.array([g_opt[0], repr(g_opt[1])]), d_opt=np.array([d_opt[0], repr(d_opt[1])]),
               label_smoothing=np.float64(label_smoothing), loss_metric=np.array('BCELoss'))
    rows = []
    for r, (lens, ylens) in enumerate(batches):
        x, y, _ = make_batch(np.array(lens), np.array(ylens), dims[4], seed + 1000 + r)
        rows.append(ref_adv_step(disc, asr, tae.encoder, D_optim, G_optim, x, list(lens), y, label_smoothing))
        out['lens%d' % r], out['ylens%d' % r], out['batch_seed%d' % r] = np.array(lens), np.array(ylens), np.int64(seed + 1000 + r)
    rows = np.array(rows)
    out['d_real'], out['d_fake'], out['g_loss'], out['d_norm'], out['g_norm'] = rows.T
    w1 = {('disc.' + k): v for k, v in disc.state_dict().items()}
    w1.update({('asr.' + k): v for k, v in asr.state_dict().items()})
    _weights_record(out, w0, w1, sum(v.numel() for v in w1.values()) < 400000)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s D real %s fake %s G %s  %.1f KB' % (name, np.round(out['d_real'], 5), np.round(out['d_fake'], 5),
                                                     np.round(out['g_loss'], 5), os.path.getsize(path) / 1024))


def ref_sae_step(sae, asr, optim, x, x_lens):
    """The body of the reference's SAETrainer.exec (src/trainer.py:803-820) around the reference's own
    SpeechAutoEncoder / ASR classes."""
    optim.zero_grad()
    listener_out, _ = asr.encoder(x, x_lens)
    autoenc_out = sae(x, listener_out)
    batch_t = max(x_lens)
    xt = x[:, :batch_t, :]
    enc_final = torch.zeros([autoenc_out.shape[0], batch_t, autoenc_out.shape[2]])
    enc_final[:, :autoenc_out.shape[1], :] = autoenc_out
    loss = nn.SmoothL1Loss()(enc_final, xt)                                 # trainer.py:796
    loss.backward()
    norm = float(nn.utils.clip_grad_norm_(sae.parameters(), 5))             # Solver.step(speech_autoenc.parameters(), optim)
    optim.step()
    return float(loss.item()), norm


def capture_sae_traj(asr_mod, sae_mod, name, dims, sae_cfg, batches, pad_to, seed, opt=('Adam', 1e-4)):
    """A trajectory of SAETrainer iterations on the real reference's classes (config 5's third leg,
    src/trainer.py:760-907): one optimizer over the speech autoencoder and the Listener (:789-794), the norm
    clipped over the speech autoencoder alone (:820).  Every batch is padded to `pad_to` frames, as the
    reference's dataset pads every utterance to the corpus maximum (src/ASRDataset.py:131-151): the global
    encoder convolves the padding too, the loss stops at the batch's longest utterance.  Afterwards the eval-
    mode loss of the first batch (running batch-norm statistics, SAETrainer.valid :853-868)."""
    from las_oracle import seeded_generic_weights
    from ss_asr_amd.synthetic import make_batch
    seed_all(seed)
    asr = asr_mod.ASR(*dims, 1.0)
    seeded_weights(asr, seed + 100)
    sae = sae_mod.SpeechAutoEncoder(asr.encoder.out_dim, dims[4], **sae_cfg)
    seeded_generic_weights(sae, seed + 300)
    optim = getattr(torch.optim, opt[0])(list(sae.parameters()) + list(asr.encoder.parameters()), lr=opt[1], eps=1e-8)
    w0 = {('sae.' + k): v.clone() for k, v in sae.state_dict().items()}
    w0.update({('asr.' + k): v.clone() for k, v in asr.state_dict().items()})
    out = dict(dims=np.array(dims), seed=np.int64(seed), asr_weights_seed=np.int64(seed + 100),
               sae_weights_seed=np.int64(seed + 300), rounds=np.int64(len(batches)), pad_to=np.int64(pad_to),
               opt=np.array([opt[0], repr(opt[1])]), kernel_sizes=np.array(sae_cfg['kernel_sizes']),
               num_filters=np.array(sae_cfg['num_filters']), pool_kernel_sizes=np.array(sae_cfg['pool_kernel_sizes']))
    rows = []
    for r, lens in enumerate(batches):
        x, _, _ = make_batch(np.array(lens), np.full(len(lens), 3), dims[4], seed + 1000 + r, pad_to=pad_to)
        rows.append(ref_sae_step(sae, asr, optim, x, list(lens)))
        out['lens%d' % r], out['batch_seed%d' % r] = np.array(lens), np.int64(seed + 1000 + r)
        if r == 0:
            with torch.no_grad():
                sae.eval()                                  # (a forward in eval mode moves no buffer)
                lis, _ = asr.encoder(x, list(lens))
                out['eval_pred_head'] = sae(x, lis).reshape(-1)[:512].numpy()
                sae.train()
    out['loss'], out['norm'] = np.array(rows).T
    w1 = {('sae.' + k): v for k, v in sae.state_dict().items()}
    w1.update({('asr.' + k): v for k, v in asr.state_dict().items()})
    w0f = {k: v.to(torch.float32) for k, v in w0.items()}
    w1f = {k: v.to(torch.float32) for k, v in w1.items()}
    _weights_record(out, w0f, w1f, sum(v.numel() for v in w1.values()) < 400000)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s loss %s  norm %s  %.1f KB' % (name, np.round(out['loss'], 6), np.round(out['norm'], 6),
                                               os.path.getsize(path) / 1024))


def main(only=None):
    os.makedirs(OUT, exist_ok=True)
    asr_mod = import_reference()
    if only:
        real = capture
        globals()['capture'] = lambda mod, name, *a, **k: real(mod, name, *a, **k) if name in only else None
    small = (50, 32, 32, 16, 12)         # output, enc H, dec H, mlp, feat
    full = (50, 256, 256, 128, 80)       # conf/default.yaml:6-9 with feature_dim 80
    capture(asr_mod, 'small_tf1', small, [64, 56, 48, 40], [10, 7, 5, 3], 1.0, 1)
    capture(asr_mod, 'small_odd', small, [63, 51, 33, 17], [9, 9, 4, 6], 1.0, 2,
            weights_seed=21, keep='acts')
    capture(asr_mod, 'small_padded', small, [50, 50, 41, 24, 9], [3, 8, 8, 2, 5], 1.0, 3,
            weights_seed=22, keep='acts', pad_to=72)
    capture(asr_mod, 'small_greedy', small, [64, 56, 48, 40], [10, 7, 5, 3], 1.0, 4,
            weights_seed=23, keep='acts', teacher=False, extra_decode=6)
    capture(asr_mod, 'small_sampled', small, [64, 56, 48, 40], [10, 7, 5, 3], 0.5, 5,
            weights_seed=24, keep='acts')
    capture(asr_mod, 'full_b4', full, [64, 56, 48, 40], [10, 7, 5, 3], 1.0, 6,
            weights_seed=11, keep='compact')
    # edges of the batch shape at the yaml's layer sizes: ONE utterance (every kernel's column count is 1; blstm_4 recurs over
    # a "sequence" of one step), and utterances so short that the encoder leaves one frame (8 frames -> T' = 1: attention
    # over a single frame) beside one of 17 (T' = 2), the longest of odd length at every layer (17 -> 8, 9 -> 4 ...)
    capture(asr_mod, 'edge_b1', full, [123], [7], 1.0, 8, weights_seed=13, keep='compact')
    capture(asr_mod, 'edge_short', full, [17, 9, 8], [3, 2, 2], 1.0, 9, weights_seed=14, keep='compact')
    # more utterances than the 32 a persistent launch takes as one column chunk / the persistent decode loop takes at all
    lens40 = sorted(np.random.default_rng(15).integers(24, 97, size=40).tolist(), reverse=True)
    capture(asr_mod, 'full_b40', full, lens40, np.random.default_rng(16).integers(2, 12, size=40).tolist(), 1.0, 10,
            weights_seed=15, keep='compact')
    lens16 = sorted(np.random.default_rng(9).integers(200, 401, size=16).tolist(),
                    reverse=True)
    lens16[0] = 400
    ylens16 = np.random.default_rng(10).integers(10, 41, size=16).tolist()
    capture(asr_mod, 'full_b16_t400', full, lens16, ylens16, 1.0, 7,
            weights_seed=12, keep='compact')
    # BASELINE.json configs[1] at its own size: the batches bench.py times (its longest bucket,
    # 32 utterances x 800 frames, and its median bucket), teacher forced so that the decode loop is
    # deterministic.  The fixture carries the recipe of the batch, not its 8 MB of frames.
    sys.path.insert(0, os.path.dirname(HERE))
    from ss_asr_amd.synthetic import config2_batches
    # config 5's shared decoder: the reference's TextAutoEncoder drives asr.attention / decoder /
    # embed / char_trans (src/text_autoencoder.py:55-94).  Full layer sizes: 12 rows (the persistent
    # decode loop's shape) and 40 rows (more than its 32: the per-step kernels); a small model too.
    import text_autoencoder as tae_mod                          # reference module (path set by import_reference)
    real_tae = capture_tae

    def cap_tae(name, *a, **k):
        if not only or name in only:
            real_tae(asr_mod, tae_mod, name, *a, **k)
    cap_tae('tae_full_b12', full, (128, 256, 2), [14, 12, 12, 11, 9, 9, 8, 6, 5, 5, 3, 2], 1.0, 31)
    cap_tae('tae_full_b40', full, (128, 256, 2), [3 + (7 * k) % 13 for k in range(40)], 1.0, 32)
    cap_tae('tae_small_tf05', small, (8, 32, 2), [10, 7, 5, 3, 3], 0.5, 33)
    # config 5's first leg as a trainer (TAETrainer, src/trainer.py:594-758): three Adam steps at full layer
    # sizes (12 rows: the persistent decode loop), and the two legs that this build has -- an ASRTrainer step
    # and a TAETrainer step -- alternating on ONE shared ASR object, small model, all weights kept
    if not only or 'tae_traj_full_b12' in only:
        capture_tae_traj(asr_mod, tae_mod, 'tae_traj_full_b12', full, (128, 256, 2),
                         [[14, 12, 12, 11, 9, 9, 8, 6, 5, 5, 3, 2], [13, 13, 10, 9, 9, 7, 7, 6, 4, 4, 3, 3],
                          [15, 11, 10, 10, 8, 8, 6, 6, 5, 3, 2, 2]], 1.0, 41)
    if not only or 'seed_alt_small' in only:
        capture_tae_traj(asr_mod, tae_mod, 'seed_alt_small', small, (8, 32, 2),
                         [[10, 7, 5, 3, 3], [9, 8, 4, 4, 2], [8, 8, 6, 3, 2]], 1.0, 42,
                         asr_batches=[([64, 56, 48, 40], [10, 7, 5, 3]), ([72, 50, 33, 17], [9, 6, 4, 2]),
                                      ([56, 56, 41, 24], [8, 8, 3, 5])])
    # config 5's second leg (ADVTrainer, src/trainer.py:909-1124): three iterations at full layer sizes with the
    # yaml's optimizers (Adadelta / Adadelta), and a small model with Adam on both sides
    import discriminator as disc_mod                            # reference module
    if not only or 'adv_traj_full_b8' in only:
        capture_adv_traj(asr_mod, tae_mod, disc_mod, 'adv_traj_full_b8', full, (128, 256, 2), 256,
                         [([200, 184, 160, 152, 120, 96, 80, 64], [20, 18, 15, 15, 12, 9, 8, 6]),
                          ([192, 176, 168, 144, 128, 104, 72, 40], [19, 17, 16, 14, 12, 10, 7, 4]),
                          ([208, 200, 136, 128, 112, 88, 56, 48], [21, 20, 13, 12, 11, 8, 5, 4])],
                         51, ('Adadelta', 1.0), ('Adadelta', 1.0))
    if not only or 'adv_traj_small_adam' in only:
        capture_adv_traj(asr_mod, tae_mod, disc_mod, 'adv_traj_small_adam', small, (8, 32, 2), 24,
                         [([64, 56, 48, 40], [10, 7, 5, 3]), ([72, 50, 33, 17], [9, 6, 4, 2]), ([56, 56, 41, 24], [8, 8, 3, 5])],
                         52, ('Adam', 1e-3), ('Adam', 1e-3))
    # config 5's third leg (SAETrainer, src/trainer.py:760-907): three iterations at full layer sizes with the
    # yaml's kernels and filters (conf/default.yaml:27-30; its last pooling window, [2000, 40], needs utterances
    # of 30,000 frames: the window here is what a 208-frame corpus leaves, [11, 40]), batches whose longest
    # utterance is shorter than the padding and not a multiple of 8; and a small model with the kernel
    # orientation of the class docstring (src/speech_autoencoder.py:108-110: time first), odd filter counts, Adam 1e-3
    import speech_autoencoder as sae_mod                        # reference module
    if not only or 'sae_traj_full_b8' in only:
        capture_sae_traj(asr_mod, sae_mod, 'sae_traj_full_b8', full,
                         dict(kernel_sizes=[[1, 36], [5, 1], [3, 1]], num_filters=[32, 64, 256],
                              pool_kernel_sizes=[[3, 1], [5, 1], [11, 40]]),
                         [[208, 184, 160, 152, 120, 96, 80, 64], [203, 176, 168, 144, 128, 104, 72, 40],
                          [192, 190, 136, 128, 112, 88, 56, 48]], 208, 61)
    if not only or 'sae_traj_small' in only:
        capture_sae_traj(asr_mod, sae_mod, 'sae_traj_small', small,
                         dict(kernel_sizes=[[7, 1], [1, 3], [2, 2]], num_filters=[6, 12, 20],
                              pool_kernel_sizes=[[2, 1], [1, 2], [28, 3]]),
                         [[64, 56, 48, 40], [61, 50, 33, 17], [56, 56, 41, 24]], 64, 62, opt=('Adam', 1e-3))
    # the yaml's speech autoencoder AS SHIPPED (conf/default.yaml:27-30): its last pooling window [2000, 40] only fits
    # utterances of 30,050 .. 60,000 frames (five to ten minutes of audio) -- two of them, one iteration; the
    # Listener's recurrences run 30,100 / 15,050 / 7,525 steps.  Listener of 64 units per direction (the smallest
    # width whose recurrences take the persistent kernels): at the full 256 the reference's CPU backward -- a 5 MB
    # weight-gradient accumulation per time step and direction -- did not finish within 50 minutes on 8 cores.
    # Minutes of reference CPU time all the same: on request
    if only and 'sae_yaml_b2_t30100' in only:
        capture_sae_traj(asr_mod, sae_mod, 'sae_yaml_b2_t30100', (50, 64, 64, 32, 80),
                         dict(kernel_sizes=[[1, 36], [5, 1], [3, 1]], num_filters=[32, 64, 256],
                              pool_kernel_sizes=[[3, 1], [5, 1], [2000, 40]]),
                         [[30100, 27013]], 30100, 63)
    for name, pick, seed, wseed in (('bench_b32_t800', 0, 8, 14), ('bench_b32_median', 4, 9, 15)):
        x, y, lens = config2_batches(8, batch_size=32, feat_dim=80, seed=1)[pick]
        ylens = [int(v) - 1 for v in (y != 0).sum(-1)]
        capture(asr_mod, name, full, lens, ylens, 1.0, seed, weights_seed=wseed, keep='compact',
                xy=(x, y), recipe=dict(n_batches=8, pick=pick, batch_size=32, corpus_seed=1))


def long_fixture(asr_mod):
    """BASELINE.json configs[3]'s shape at full size (32 utterances of 1500-3000 frames, 150-300
    characters; attention loss only -- the reference has no CTC): T' = 375, ~300 decode steps."""
    from ss_asr_amd.synthetic import config4_batch
    x, y, lens = config4_batch()
    ylens = [int(v) - 1 for v in (y != 0).sum(-1)]
    capture(asr_mod, 'long_b32_t3000', (50, 256, 256, 128, 80), lens, ylens, 1.0, 10, weights_seed=16,
            keep='compact', xy=(x, y), recipe=dict(config4=1, batch_size=32, seed=4))


if __name__ == '__main__':
    if sys.argv[1:] == ['long_b32_t3000']:       # minutes of reference CPU time: only on request
        sys.path.insert(0, os.path.dirname(HERE))
        long_fixture(import_reference())
    else:
        main(set(sys.argv[1:]))     # optional: names of the fixtures to (re)generate
