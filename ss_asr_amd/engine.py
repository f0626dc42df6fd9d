"""One ASR train step (the body of ASRTrainer.exec, src/trainer.py:415-438) as a
reusable object: forward, masked CE, backward, gradient all-reduce across
ranks, clip + NaN guard + Adadelta.  ASRTrainer.exec, bench.py, smoke() and the
tests all run THIS object, so that what is timed and checked is what trains."""
import torch

from . import dist as sdist
from . import ops
from .optim import FlatParameters, FusedAdadelta


class ASRTrainStep:
    def __init__(self, model, lr=1.0, eps=1e-8, rho=0.9, grad_clip=5.0):
        if not next(model.parameters()).is_cuda:
            raise RuntimeError('ASRTrainStep needs the model on the GPU (no CPU path)')
        self.model = model
        self.flat = FlatParameters(model)
        sdist.broadcast_flat(self.flat.data)
        self.optim = FusedAdadelta(self.flat, lr=lr, rho=rho, eps=eps)
        self.grad_clip = grad_clip
        # gradient all-reduce in two buckets, the large one overlapped with the last layer's BPTT
        self.reducer = sdist.GradReducer(self.flat, list(model.encoder.blstm_1.parameters()))
        ops.set_wgrad_listener(self.reducer.wgrad_enqueued)
        self._one = torch.ones((), device=self.flat.data.device)
        self._grads_clean = False      # True right after a step that zeroed them in its update kernel
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
        if not self._grads_clean:
            self.optim.zero_grad()
        self._grads_clean = False
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
        self._grads_clean = True
        return loss

    def _note(self, done):
        if done is not None:
            self.last_done = done
            self.skipped_steps += int(done[1])

    def finish(self):
        """Waits for the last step's words: returns its (grad_norm, skipped); raises on a timeout."""
        self._note(self.optim.poll(wait=True))
        return self.last_done


def label_geometry(y_cpu):
    """prepare_y's lengths on the host (src/ASRDataset.py:338): returns
    (y_lens, ans_len)."""
    y_lens = [int(v) + 1 for v in (y_cpu != 0).sum(-1)]
    return y_lens, max(y_lens) - 1
