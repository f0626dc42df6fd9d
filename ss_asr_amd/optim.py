"""Flat parameter/gradient storage and the fused clip + Adadelta step.

Solver.step in the reference (src/trainer.py:131-148) clips the global
gradient norm over 42 tensors, checks it for NaN on the host, and calls
torch.optim.Adadelta.step (src/trainer.py:401-403), i.e. ~10 passes over
41 MB in ~170 small kernels.  Here every parameter is a view into one flat
buffer, every gradient a view into another, and one C-ABI call
(ssasr_clip_adadelta) does norm, NaN guard, clip and update in three kernels
without a host round trip.  The same flat gradient buffer is what the data
parallel wrapper all-reduces (one RCCL call).
"""
import torch

from . import ops


class FlatParameters:
    """Re-homes a module's parameters (and their .grad) into two flat buffers."""

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError('module has no trainable parameters')
        dev, dt = params[0].device, params[0].dtype
        # keep every tensor 16-byte aligned inside the buffer
        offs, total = [], 0
        for p in params:
            if p.device != dev or p.dtype != dt:
                raise ValueError('all parameters must share one device and dtype')
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.params = params
        self.offsets = offs
        self.numel = total
        self.data = torch.zeros(total, device=dev, dtype=dt)
        self.grad = torch.zeros(total, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.data[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._ssasr_flat_grad = True     # ops may accumulate into p.grad from a side stream
        # True right after a kernel that left the whole gradient buffer zeroed (an update kernel with
        # zero_grad folded in); whoever accumulates into it clears the flag
        self.clean = True
        # one flat home per module: a second step object over the same module (the trainers of the Seed
        # loop share one ASR model, src/trainer.py:1126-1177) finds and reuses it
        module._ssasr_flat = self

    @classmethod
    def of(cls, module):
        """The module's flat home: the existing one (its parameters can only live in ONE pair of buffers)
        or a new one."""
        flat = getattr(module, '_ssasr_flat', None)
        if flat is not None and all(p.data_ptr() == flat.data.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets)):
            return flat
        return cls(module)

    def range_of(self, params):
        """(begin, end) float offsets of the smallest run of the buffer that holds `params` -- they must
        be consecutive tensors of it (e.g. everything of the ASR model behind its Listener)."""
        ids = {id(p) for p in params}
        idx = [k for k, p in enumerate(self.params) if id(p) in ids]
        if len(idx) != len(ids) or idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError('parameters are not a contiguous run of the flat buffer')
        last = idx[-1]
        return self.offsets[idx[0]], self.offsets[last] + (self.params[last].numel() + 3) // 4 * 4

    def zero_grad(self):
        self.grad.zero_()
        self.clean = True
        for p, o in zip(self.params, self.offsets):     # re-attach if something detached them
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedAdadelta(torch.optim.Optimizer):
    """torch.optim.Adadelta(lr, rho, eps, weight_decay=0) over FlatParameters,
    fused with clip_grad_norm_ and the NaN guard of Solver.step."""

    def __init__(self, flat, lr=1.0, rho=0.9, eps=1e-6, span=None):
        """span (begin, end): the optimizer owns only that run of the flat buffer (FlatParameters.range_of) --
        ADVTrainer's generator optimizer holds the Listener alone (src/trainer.py:940-943), a run of the ASR
        model's buffer; norm, NaN guard, clip and update then cover that run."""
        self.flat = flat
        lo, hi = span if span is not None else (0, flat.numel)
        self._data, self._grad = flat.data[lo:hi], flat.grad[lo:hi]
        owned = [p for p, o in zip(flat.params, flat.offsets) if lo <= o < hi]
        super().__init__(owned, dict(lr=lr, rho=rho, eps=eps))
        dev = flat.data.device
        self.square_avg = torch.zeros_like(self._data)
        self.acc_delta = torch.zeros_like(self._data)
        self._ws = ops.clip_adadelta_ws(hi - lo, dev)
        # per-step words that reach the host in ONE asynchronous copy: [grad_norm, skipped] written
        # by the update kernel, then the int32[8] status row of the step's persistent launches
        # (ops.shared_status_row; words 4 / 5 turn non-zero when a launch timed out)
        self._words = torch.zeros(2 + 8, device=dev)
        self.stats = self._words[:2]
        self.status_row = self._words[2:].view(torch.int32)
        self._host_words = torch.zeros(2 + 8, pin_memory=dev.type == 'cuda')
        self._pending = None

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def clip_and_step(self, max_norm=5.0, grad_scale=1.0, zero_grad=False):
        """zero_grad=True leaves the flat gradient zeroed (the next step's zero_grad()
        folded into the update kernel)."""
        ops.join_side_stream()       # weight gradients enqueued on the side stream
        g = self.param_groups[0]
        ops.clip_adadelta_(self._data, self._grad, self.square_avg, self.acc_delta,
                           self._ws, self.stats, grad_scale=grad_scale, max_norm=max_norm,
                           lr=g['lr'], rho=g['rho'], eps=g['eps'], zero_grad=zero_grad)
        # the norm / NaN flag / status row reach the host asynchronously; see poll()
        self._host_words.copy_(self._words, non_blocking=True)
        self._pending = torch.cuda.Event()
        self._pending.record()

    def step(self, closure=None):
        """Plain optimizer step (no clipping): max_norm = inf."""
        self.clip_and_step(max_norm=float('inf'))

    def poll(self, wait=False):
        """Returns (grad_norm, skipped) of the most recent finished step, or None when there is
        none (or, without `wait`, when its words have not arrived yet).  Raises if a persistent
        launch of that step reported a timeout in the shared status row: its results, and the
        update computed from them, are not to be trusted."""
        if self._pending is None:
            return None
        if wait:
            self._pending.synchronize()
        elif not self._pending.query():
            return None
        self._pending = None
        status = self._host_words[2:].view(torch.int32)
        if status.any():
            raise RuntimeError(ops.describe_status(status.tolist()))
        return float(self._host_words[0]), bool(self._host_words[1] != 0)


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) fused with Solver.step's clip and NaN
    guard, over parameters that live in SEVERAL flat buffers: TAETrainer's optimizer holds the text
    autoencoder and the ASR model's embed / attention / decoder / char_trans (src/trainer.py:633-641), and
    Solver.step is handed the text autoencoder's parameters alone (:676) -- only their norm is clipped.

    segments: [(data, grad, clipped)] flat float32 views (a whole FlatParameters or a run of one); exactly
    one of them is the clipped range.  State (exp_avg, exp_avg_sq, step count) lives here."""

    def __init__(self, segments, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.segments = [(d, g, bool(c)) for d, g, c in segments]
        clipped = [s for s in self.segments if s[2]]
        if len(clipped) != 1:
            raise ValueError('exactly one segment is the clipped range')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = clipped[0][0].device
        self.exp_avg = [torch.zeros_like(d) for d, _, _ in self.segments]
        self.exp_avg_sq = [torch.zeros_like(d) for d, _, _ in self.segments]
        self.state_step = torch.zeros(1, device=dev)
        self._ws = torch.empty(int(ops._lib.load().ssasr_adam_ws(clipped[0][1].numel())), device=dev)
        self._words = torch.zeros(2 + 8, device=dev)
        self.stats = self._words[:2]
        self.status_row = self._words[2:].view(torch.int32)
        self._host_words = torch.zeros(2 + 8, pin_memory=dev.type == 'cuda')
        self._pending = None

    def clip_and_step(self, max_norm=5.0, grad_scale=1.0, zero_grad=False):
        ops.join_side_stream()
        clip_grad = next(g for _, g, c in self.segments if c)
        ops.adam_prepare(clip_grad, self.state_step, self._ws, self.stats, grad_scale=grad_scale, max_norm=max_norm,
                         lr=self.lr, betas=self.betas)
        for (d, g, c), m, v in zip(self.segments, self.exp_avg, self.exp_avg_sq):
            ops.adam_update_(d, g, m, v, self._ws, self.stats, clipped=c, grad_scale=grad_scale, betas=self.betas,
                             eps=self.eps, zero_grad=zero_grad)
        self._host_words.copy_(self._words, non_blocking=True)
        self._pending = torch.cuda.Event()
        self._pending.record()

    poll = FusedAdadelta.poll
