"""One ASR train step (the body of ASRTrainer.exec, src/trainer.py:415-438) as a
reusable object: forward, masked CE, backward, gradient all-reduce across
ranks, clip + NaN guard + Adadelta.  ASRTrainer.exec, bench.py, smoke() and the
tests all run THIS object, so that what is timed and checked is what trains."""
import os

import torch

from . import dist as sdist
from . import ops
from .optim import FlatParameters, FusedAdadelta, FusedAdam


_settled = False


def settle_collector(again=False):
    """Once per process, after its FIRST train step (any step object): collect, then move everything
    alive to the collector's permanent generation (gc.freeze).  A process holding the model, the
    optimizer and the library bindings tracks ~270 k long-lived objects; a full (generation-2)
    collection walks all of them -- 74 ms on the benchmark host, once per ~300 train steps, which a
    5.6 ms step cannot hide (tools/hiccup.py: one block of 20 steps at 7.47 instead of 5.60 ms/step;
    with the freeze none, and the young collections fall from 0.8 to 0.4 ms).  The collection this
    function itself runs costs the same ~80 ms, once, in the first step's shadow.  `again`: a caller
    that has built more long-lived state since (bench.py after its warm-up) repeats it.
    SSASR_GC_FREEZE=0 leaves the collector alone."""
    global _settled
    if (_settled and not again) or os.environ.get('SSASR_GC_FREEZE', '1') == '0':
        return
    _settled = True
    import gc
    gc.collect()
    gc.freeze()


class ASRTrainStep:
    def __init__(self, model, lr=1.0, eps=1e-8, rho=0.9, grad_clip=5.0):
        if not next(model.parameters()).is_cuda:
            raise RuntimeError('ASRTrainStep needs the model on the GPU (no CPU path)')
        self.model = model
        self.flat = FlatParameters.of(model)       # (a TAETrainStep over the same model may have made it already)
        sdist.broadcast_flat(self.flat.data)
        self.optim = FusedAdadelta(self.flat, lr=lr, rho=rho, eps=eps)
        self.grad_clip = grad_clip
        # gradient all-reduce in two buckets, the large one overlapped with the last layer's BPTT
        self.reducer = sdist.GradReducer(self.flat, list(model.encoder.blstm_1.parameters()))
        ops.set_wgrad_listener(self.reducer.wgrad_enqueued)
        self._one = torch.ones((), device=self.flat.data.device)
        self.last_logits = None        # [B, U, V] of the most recent step (for the trainer's logging)
        self.last_done = None          # (grad_norm, skipped) of the last step whose words have arrived
        self.skipped_steps = 0         # steps whose update was skipped because the gradient norm was NaN

    def forward_loss(self, x, y, x_lens, ans_len):
        _, logits, att = self.model(x, ans_len, teacher=y, state_len=x_lens)
        return ops.masked_ce_loss(logits, y, ans_len), logits, att

    def __call__(self, x, y, x_lens, ans_len):
        """x [B,T,F] float32, y [B,L] int64 (both on the GPU), x_lens host list
        (descending), ans_len = max label length - 1.  Returns the loss tensor
        (on the device; reading it synchronises).

        The verdict of the PREVIOUS step (gradient norm, NaN skip, and whether one of its
        persistent launches timed out) is picked up here when its words have reached the host
        -- no synchronisation; a timeout raises."""
        self._note(self.optim.poll())
        # the weight-gradient listener is process-wide: claim it for THIS step object (several step
        # objects may alternate in one process -- the ASR and joint steps, the trainers of the Seed loop)
        ops.set_wgrad_listener(self.reducer.wgrad_enqueued)
        if not self.flat.clean:           # (True right after a step whose update kernel zeroed the gradients)
            self.optim.zero_grad()
        self.flat.clean = False
        self.reducer.begin()
        # the attention map is only looked at by valid(): no device-to-host copy per train step
        keep, self.model.att_on_host = getattr(self.model, 'att_on_host', True), False
        try:
            # every persistent launch of the step reports into the optimizer's status row
            with ops.shared_status_row(self.optim.status_row):
                loss, self.last_logits, _ = self.forward_loss(x, y, x_lens, ans_len)
                loss.backward(self._one)
        finally:
            self.model.att_on_host = keep
        scale = self.reducer.finish()
        self.optim.clip_and_step(self.grad_clip, grad_scale=scale, zero_grad=True)
        self.flat.clean = True
        settle_collector()
        return loss

    def _note(self, done):
        if done is not None:
            self.last_done = done
            self.skipped_steps += int(done[1])

    def finish(self):
        """Waits for the last step's words: returns its (grad_norm, skipped); raises on a timeout."""
        self._note(self.optim.poll(wait=True))
        return self.last_done


class TAETrainStep:
    """One TAETrainer step (the body of TAETrainer.exec, src/trainer.py:652-677) as a reusable object: the
    text autoencoder's forward through the shared attend-and-spell loop (text_autoencoder.py), the loss of
    :662-672, backward, gradient all-reduce across ranks, then Solver.step as the reference calls it --
    norm, NaN guard and clip over the TEXT AUTOENCODER's parameters only (:676), Adam (:633-641,
    conf/default.yaml:43-45) over the text autoencoder AND the ASR model's embed / attention / decoder /
    char_trans.  Those shared parameters stay where the ASR model's flat buffer has them (everything behind
    the Listener is one run of it), so an ASRTrainStep and a TAETrainStep over the same ASR object train
    the same storage, in turn: the two legs of the Seed loop (src/trainer.py:1126-1177) this build has."""

    def __init__(self, asr_model, tae_model, lr=1e-4, eps=1e-8, grad_clip=5.0):
        if not next(tae_model.parameters()).is_cuda or not next(asr_model.parameters()).is_cuda:
            raise RuntimeError('TAETrainStep needs both models on the GPU (no CPU path)')
        self.asr, self.tae = asr_model, tae_model
        self.asr_flat = FlatParameters.of(asr_model)
        self.tae_flat = FlatParameters.of(tae_model)
        shared = (list(asr_model.attention.parameters()) + list(asr_model.decoder.parameters()) +
                  list(asr_model.embed.parameters()) + list(asr_model.char_trans.parameters()))
        self.lo, self.hi = self.asr_flat.range_of(shared)
        sdist.broadcast_flat(self.tae_flat.data)
        sdist.broadcast_flat(self.asr_flat.data)
        self.optim = FusedAdam([(self.tae_flat.data, self.tae_flat.grad, True),
                                (self.asr_flat.data[self.lo:self.hi], self.asr_flat.grad[self.lo:self.hi], False)],
                               lr=lr, eps=eps)
        self.grad_clip = grad_clip
        self._one = torch.ones((), device=self.tae_flat.data.device)
        self.last_logits = None
        self.last_done = None
        self.skipped_steps = 0

    def forward_loss(self, y, y_noise, decode_step, noise_lens):
        from .text_autoencoder import tae_loss
        _, logits = self.tae(self.asr, y, y_noise, decode_step, noise_lens=noise_lens)
        return tae_loss(logits, y), logits

    def __call__(self, y, y_noise, y_lens, noise_lens):
        """y [B, L] clean label rows, y_noise [B, <= L] noised rows (int64, on the GPU), their prepare_y
        lengths.  Returns the loss tensor.  The previous step's verdict (norm, NaN skip, time-outs) is
        picked up here without a synchronisation, as in ASRTrainStep."""
        self._note(self.optim.poll())
        ops.set_wgrad_listener(None)          # (an ASRTrainStep's reducer must not see this pass's gradients)
        for flat in (self.tae_flat, self.asr_flat):
            if not flat.clean:
                flat.zero_grad()
        self.tae_flat.clean = self.asr_flat.clean = False
        with ops.shared_status_row(self.optim.status_row):
            loss, self.last_logits = self.forward_loss(y, y_noise, max(y_lens), noise_lens)
            loss.backward(self._one)
        scale = sdist.allreduce_grad(self.tae_flat.grad)
        sdist.allreduce_grad(self.asr_flat.grad[self.lo:self.hi])
        self.optim.clip_and_step(self.grad_clip, grad_scale=scale, zero_grad=True)
        # the Listener's gradients were never touched by this pass: the whole ASR buffer is clean again
        self.tae_flat.clean = self.asr_flat.clean = True
        settle_collector()
        return loss

    _note = ASRTrainStep._note
    finish = ASRTrainStep.finish


def label_geometry(y_cpu):
    """prepare_y's lengths on the host (src/ASRDataset.py:338): returns
    (y_lens, ans_len)."""
    y_lens = [int(v) + 1 for v in (y_cpu != 0).sum(-1)]
    return y_lens, max(y_lens) - 1


def _fused_optimizer(kind, flat, span, lr, eps=1e-8):
    """The fused form of torch.optim.<kind>(params, lr=lr, eps=eps) over one run of a flat buffer, with
    Solver.step's clip over that same run; None for a kind this build has no kernel for."""
    lo, hi = span if span is not None else (0, flat.numel)
    if kind == 'Adadelta':
        return FusedAdadelta(flat, lr=lr, eps=eps, span=(lo, hi))
    if kind == 'Adam':
        return FusedAdam([(flat.data[lo:hi], flat.grad[lo:hi], True)], lr=lr, eps=eps)
    return None


class ADVTrainStep:
    """One ADVTrainer iteration (src/trainer.py:968-1032) as a reusable object: the discriminator is trained to
    score the text encoder's frames high (labels 1 - label_smoothing) and the Listener's low (labels 0), then
    the Listener -- the generator -- to be scored high (labels 1) by the UPDATED discriminator; Solver.step
    over the discriminator's parameters with D_opt, then over the Listener's with G_opt.  The reference leaves
    `self.loss_metric` undefined (:984): it is nn.BCELoss here, what
    src/discriminator.py:17-19 describes.

    The Listener runs ONCE per iteration (as in the reference, which reuses `fake_data`'s graph for the
    generator pass); gradients the reference computes and never uses -- the text encoder's from the real
    pass, the discriminator's from the generator pass (zeroed at :975 before anything reads them) -- are not
    computed.  The Listener's parameters stay where the ASR model's flat buffer has them: the other legs of
    the Seed loop train the same storage."""

    def __init__(self, asr_model, tae_model, discriminator, g_opt=('Adadelta', 1.0), d_opt=('Adadelta', 1.0),
                 label_smoothing=0.1, grad_clip=5.0):
        for m in (asr_model, tae_model, discriminator):
            if not next(m.parameters()).is_cuda:
                raise RuntimeError('ADVTrainStep needs every model on the GPU (no CPU path)')
        self.asr, self.tae, self.disc = asr_model, tae_model, discriminator
        self.asr_flat = FlatParameters.of(asr_model)
        self.d_flat = FlatParameters.of(discriminator)
        self.span = self.asr_flat.range_of(list(asr_model.encoder.parameters()))
        sdist.broadcast_flat(self.asr_flat.data)
        sdist.broadcast_flat(self.d_flat.data)
        # the text autoencoder is only READ here (its encoder's frames are the discriminator's real data): every
        # rank must read the same one; buffers (none today) ride along
        sdist.broadcast_module(tae_model)
        sdist.broadcast_buffers(asr_model, discriminator)
        self.G_optim = _fused_optimizer(g_opt[0], self.asr_flat, self.span, g_opt[1])
        self.D_optim = _fused_optimizer(d_opt[0], self.d_flat, None, d_opt[1])
        if self.G_optim is None or self.D_optim is None:
            raise NotImplementedError('ADVTrainStep: optimizer types %r / %r (Adadelta and Adam have kernels)'
                                      % (g_opt[0], d_opt[0]))
        self.label_smoothing = float(label_smoothing)
        self.grad_clip = grad_clip
        self._one = torch.ones((), device=self.d_flat.data.device)
        self.last_done = None          # (generator grad norm, skipped) of the last finished iteration
        self.last_done_d = None
        self.skipped_steps = 0

    def frames(self, x, x_lens, y):
        """The two encoders' frames: (real [B, seq, 512] from the text encoder, no graph; fake [B, T', 512] from
        the Listener, with its graph)."""
        with torch.no_grad():
            real = self.tae.encoder(y)                               # the data distribution, [B, seq, 512]
        fake, _ = self.asr.encoder(x, x_lens)
        return real, fake

    def __call__(self, x, x_lens, y):
        """x [B, T, F] padded fbanks, x_lens host list (descending), y [B, L] label rows (int64), all on
        the GPU.  Returns (D_realloss, D_fakeloss, G_loss) device tensors."""
        from .seed_ops import bce_loss
        done = self.D_optim.poll()
        if done is not None:
            self.last_done_d = done
            self.skipped_steps += int(done[1])
        self._note(self.G_optim.poll())
        ops.set_wgrad_listener(None)
        for flat in (self.d_flat, self.asr_flat):
            if not flat.clean:
                flat.zero_grad()
        self.d_flat.clean = self.asr_flat.clean = False
        lo, hi = self.span
        with ops.shared_status_row(self.G_optim.status_row):
            # --- discriminator: maximise log D(real) + log(1 - D(G(x)))
            real, fake = self.frames(x, x_lens, y)
            d_real = bce_loss(self.disc(real), 1.0 - self.label_smoothing)
            d_real.backward(self._one)
            d_fake = bce_loss(self.disc(fake.detach()), 0.0)
            d_fake.backward(self._one)
            scale = sdist.allreduce_grad(self.d_flat.grad)
            self.D_optim.clip_and_step(self.grad_clip, grad_scale=scale, zero_grad=True)
            self.d_flat.clean = True
            # --- generator: maximise log D(G(x)), through the discriminator just updated
            g_loss = bce_loss(self.disc(fake, frozen=True), 1.0)
            g_loss.backward(self._one)
        scale = sdist.allreduce_grad(self.asr_flat.grad[lo:hi])
        self.G_optim.clip_and_step(self.grad_clip, grad_scale=scale, zero_grad=True)
        self.asr_flat.clean = True         # (nothing behind the Listener was touched by this pass)
        settle_collector()
        return d_real, d_fake, g_loss

    _note = ASRTrainStep._note

    def finish(self):
        """Waits for the last iteration's words of BOTH optimizers; raises on a persistent time-out."""
        done = self.D_optim.poll(wait=True)
        if done is not None:
            self.last_done_d = done
            self.skipped_steps += int(done[1])
        self._note(self.G_optim.poll(wait=True))
        return self.last_done


class SAETrainStep:
    """One SAETrainer iteration (src/trainer.py:803-820) as a reusable object: the shared Listener, the speech
    autoencoder over its output and the raw frames, smooth-L1 against the input frames (:811-818), backward,
    gradient all-reduce, then Solver.step as the reference calls it -- norm, NaN guard and clip over the SPEECH
    AUTOENCODER's parameters only (:820), Adam (:789-794, conf/default.yaml:24-26) over them AND the Listener.
    The Listener's parameters stay where the ASR model's flat buffer has them (its first run): the other legs of
    the Seed loop train the same storage.  Batch-norm running statistics are per rank (the reference is
    single-device; they do not enter the training arithmetic)."""

    def __init__(self, asr_model, sae_model, opt=('Adam', 1e-4), grad_clip=5.0):
        if not next(sae_model.parameters()).is_cuda or not next(asr_model.parameters()).is_cuda:
            raise RuntimeError('SAETrainStep needs both models on the GPU (no CPU path)')
        if opt[0] != 'Adam':
            raise NotImplementedError('SAETrainStep: optimizer type %r (Adam has the kernel whose clipped range may '
                                      'differ from its update range)' % (opt[0],))
        self.asr, self.sae = asr_model, sae_model
        self.asr_flat = FlatParameters.of(asr_model)
        self.sae_flat = FlatParameters.of(sae_model)
        self.lo, self.hi = self.asr_flat.range_of(list(asr_model.encoder.parameters()))
        sdist.broadcast_flat(self.sae_flat.data)
        sdist.broadcast_flat(self.asr_flat.data)
        sdist.broadcast_buffers(sae_model)     # batch-norm running statistics start from rank 0's (then evolve per rank)
        self.optim = FusedAdam([(self.sae_flat.data, self.sae_flat.grad, True),
                                (self.asr_flat.data[self.lo:self.hi], self.asr_flat.grad[self.lo:self.hi], False)],
                               lr=opt[1], eps=1e-8)
        self.grad_clip = grad_clip
        # SSASR_SAE_OVERLAP: 0 the speech encoder behind the Listener on one stream, 1 / 2 see forward_loss (same results)
        self.overlap = int(os.environ.get('SSASR_SAE_OVERLAP', '1'))
        self._one = torch.ones((), device=self.sae_flat.data.device)
        self.last_pred = None          # [B, 8 T', F] of the most recent step (for the trainer's figures)
        self.last_done = None
        self.skipped_steps = 0

    def forward_loss(self, x, x_lens):
        """Listener and global speech encoder both read the fbanks and nothing of each other.  With `overlap` the speech
        encoder runs on the second stream -- 1: its forward still AFTER the Listener's, its backward (autograd runs a
        node on its forward's stream; created after the Listener's nodes it comes first in autograd's order) beside the
        Listener's BPTT; 2: its forward beside the Listener's too (measured: the first layer's recurrence then takes
        2.42 instead of 1.44 ms -- a loss).  The frame decoder waits for both.  A caller that runs forward_loss + backward by
        itself with overlap on must call ops.join_side_stream() before it reads sae_flat.grad (__call__ does)."""
        from .seed_ops import sae_loss
        if not (self.overlap and x.is_cuda):
            listener_out, _ = self.asr.encoder(x, x_lens)
            pred = self.sae(x, listener_out)
            return sae_loss(pred, x, max(x_lens)), pred
        cur, side = torch.cuda.current_stream(), ops.side_stream()
        ready = torch.cuda.Event()
        ready.record(cur)                                  # x (and the zeroed gradient buffers) are final here
        listener_out, _ = self.asr.encoder(x, x_lens)
        if self.overlap == 2:
            side.wait_event(ready)
        else:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            enc = self.sae.encoder(x)
        cur.wait_stream(side)
        enc.record_stream(cur)
        pred = self.sae.decode_frames(enc, listener_out)
        return sae_loss(pred, x, max(x_lens)), pred

    def __call__(self, x, x_lens):
        """x [B, T, F] padded fbanks on the GPU (T: the corpus' padding, >= max(x_lens)), x_lens host list
        (descending).  Returns the loss tensor."""
        self._note(self.optim.poll())
        ops.set_wgrad_listener(None)
        for flat in (self.sae_flat, self.asr_flat):
            if not flat.clean:
                flat.zero_grad()
        self.sae_flat.clean = self.asr_flat.clean = False
        with ops.shared_status_row(self.optim.status_row):
            loss, self.last_pred = self.forward_loss(x, x_lens)
            loss.backward(self._one)
        # the speech encoder's backward ran on the second stream and wrote its parameter gradients through sinks
        # (autograd's end-of-backward stream sync does not cover them): joined HERE, not left to the next reader
        ops.join_side_stream()
        scale = sdist.allreduce_grad(self.sae_flat.grad)
        sdist.allreduce_grad(self.asr_flat.grad[self.lo:self.hi])
        self.optim.clip_and_step(self.grad_clip, grad_scale=scale, zero_grad=True)
        self.sae_flat.clean = self.asr_flat.clean = True      # (nothing behind the Listener was touched)
        settle_collector()
        return loss

    _note = ASRTrainStep._note
    finish = ASRTrainStep.finish
