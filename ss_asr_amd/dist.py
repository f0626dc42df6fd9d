"""Plain data parallelism for the ASR train step: one process per GPU, the
flat gradient summed across ranks once per step.

The reference is single-process; this is the one
parallel strategy the build adds.  Utterance batches shard naturally across
ranks; the only exchange is a SUM all-reduce of the flat fp32 gradient buffer
(10,269,874 values = 41 MB at the default sizes) on RCCL (torch.distributed
backend "nccl" on ROCm).  Averaging is folded into the clip + Adadelta kernel
as ``grad_scale = 1 / world_size`` so the gradient is not touched a second
time, and clipping / the NaN guard run after the reduction so every rank
takes the same branch.

GradReducer issues that sum as two collectives so that almost all of it
overlaps the backward pass: the first encoder layer's parameters sit at the
head of the flat buffer and are the last to receive their gradient; everything
behind them is complete when the second layer's weight-gradient GEMMs have
been enqueued, and is reduced from then on (on the second stream, behind
those GEMMs) while the first layer's BPTT still runs.  Only the head, 1.4 MB,
is reduced after the backward pass.
"""
import os

import torch
import torch.distributed as dist


_single = False      # a one-rank process group drives the collective path (SSASR_DIST_SINGLE=1)


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when
    they are set (torchrun); returns (rank, world_size, local_rank).
    SSASR_DIST_SINGLE=1 with a world of one still creates the process group and keeps the
    collective path on (broadcast, both all-reduces of GradReducer): a hardware rehearsal of
    the RCCL calls where only one GPU is at hand."""
    global _single
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not os.environ.get('SSASR_DIST_SINGLE'):
        return 0, 1, 0
    _single = world <= 1
    world = max(world, 1)
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('MASTER_PORT', '29511')
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        # SSASR_DIST_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run
        # (several ranks on one GPU; gloo reduces CUDA tensors through the host)
        backend = os.environ.get('SSASR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            # Co-residency with the persistent BPTT: RCCL runs a collective as ONE
            # 256-thread workgroup per channel with 19.7 KB of static LDS (rcclGenericKernel in this ROCm's
            # librccl, read from its gfx950 code object).  The BPTT keeps 128 CUs with ~156 KB of LDS each
            # (its 118 KB reservation), so channel workgroups cannot land beside it -- they share the other 128
            # CUs with the weight-gradient GEMMs -- and a BPTT launch that finds channel workgroups already
            # resident needs 128 CUs free of them.  Capping the channels at 32 keeps both true whatever RCCL's
            # tuner would pick (a 41 MB all-reduce hidden behind ~1 ms of recurrence does not need more); an
            # explicit NCCL_MAX_NCHANNELS in the environment wins.
            os.environ.setdefault('NCCL_MAX_NCHANNELS', '32')
            if os.environ.get('SSASR_RCCL_INFO'):
                # bench.py: let RCCL write its init report to a file of this rank's, so that the channel count it
                # GRANTED (not the cap asked for above) can be put into the bench line: rccl_channels()
                os.environ.setdefault('NCCL_DEBUG', 'INFO')
                os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH')
                os.environ.setdefault('NCCL_DEBUG_FILE', '/tmp/ssasr_rccl_%d.log' % os.getpid())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rccl_channels():
    """The channel count RCCL set this rank's communicator up with, read from the init report that
    SSASR_RCCL_INFO=1 made it write ('Channel 00/32 : ...' lines, or 'N coll channels'); None when no report
    exists (gloo, no SSASR_RCCL_INFO) or it names no channels.  Call after the first collective."""
    import re
    path = os.environ.get('NCCL_DEBUG_FILE', '')
    try:
        text = open(path).read()
    except OSError:
        return None
    m = re.findall(r'(\d+) coll channels', text)
    if m:
        return int(m[-1])
    m = re.findall(r'Channel \d+/(\d+)', text)
    return int(m[-1]) if m else None


def is_active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single)


def world_size():
    return dist.get_world_size() if is_active() else 1


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def broadcast_flat(flat_data):
    """Makes every rank start from rank 0's parameters."""
    if is_active():
        dist.broadcast(flat_data, src=0)


def barrier():
    """Every rank waits here for every other (no-op without a process group).  The trainers call it after rank
    0 has written a checkpoint: the next leg of the Seed loop reads that file on EVERY rank."""
    if is_active():
        dist.barrier()


def broadcast_module(module):
    """Rank 0's parameters AND buffers of `module` on every rank -- for modules that are read but not trained
    by a step object (ADVTrainStep's text encoder) and for batch-norm running statistics, which no flat
    parameter buffer holds."""
    if not is_active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=0)


def broadcast_buffers(*modules):
    """Rank 0's buffers (batch-norm running statistics) of the modules on every rank."""
    if is_active():
        for m in modules:
            for b in m.buffers():
                dist.broadcast(b.data, src=0)


def save_atomic(obj, path):
    """torch.save through a temporary file in the same directory + os.replace: a reader never sees a
    half-written archive (torch.save writes in place)."""
    tmp = '%s.tmp.%d' % (path, os.getpid())
    torch.save(obj, tmp)
    os.replace(tmp, path)


def allreduce_grad(flat_grad):
    """SUM all-reduce of the flat gradient; returns the scale (1/world) the
    optimizer kernel must apply."""
    if not is_active():
        return 1.0
    if flat_grad.is_cuda:
        from . import ops
        ops.join_side_stream()       # deferred weight-gradient GEMMs must have landed
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


class GradReducer:
    """Two-bucket all-reduce of FlatParameters.grad (see the module docstring).

    ``head_params``: the parameters whose backward comes last (the first encoder
    layer's); they must be a prefix of the flat buffer, else one collective is used.
    Call ``begin()`` before backward and ``finish()`` after it (returns the scale for
    the optimizer kernel); ``wgrad_enqueued`` is the listener for
    ``ops.set_wgrad_listener``.
    """

    def __init__(self, flat, head_params):
        ids = {id(p) for p in head_params}
        n = 0
        for p, o in zip(flat.params, flat.offsets):
            if id(p) in ids:
                n = max(n, o + (p.numel() + 3) // 4 * 4)
        if any(id(p) in ids for p in flat.params[len(ids):]) or n == 0 or n >= flat.numel:
            n = 0                      # head is not a proper prefix of the buffer: single collective
        self.flat, self.split = flat, n
        # overlap: the tail's collective is issued on the second stream while the first layer's BPTT still
        # runs (False, or SSASR_DDP_NO_OVERLAP in the environment: one collective after the backward pass);
        # skip: no collective at all -- bench.py's attribution run ("the step with the reduce replaced by
        # a no-op"), never a training mode
        self.overlap = not os.environ.get('SSASR_DDP_NO_OVERLAP')
        self.skip = False
        self.pending = None            # deferred tail gradients (float offsets in flat.grad) still awaited this step
        self.learned = None            # the set of deferred tail gradients, observed in the first step
        self.seen = set()
        self.work = None
        self._grad_ptr = flat.grad.data_ptr()      # the flat buffers are allocated once (checked every step)

    def _offsets(self, sinks):
        """Float offsets inside flat.grad of those gradient tensors `sinks` that are views of it."""
        base, n = self.flat.grad.data_ptr(), self.flat.grad.numel()
        if base != self._grad_ptr:
            raise RuntimeError('GradReducer: the flat gradient buffer was reallocated after the step object was built')
        offs = []
        for t in sinks:
            o = (t.data_ptr() - base) // 4
            if 0 <= o < n:             # (gradients of another model's buffer: another step object's business)
                offs.append(o)
        return offs

    def begin(self):
        self.work = None
        self.seen = set()
        self.pending = None
        if is_active() and self.split and self.learned and self.overlap and not self.skip:
            self.pending = set(self.learned)

    def wgrad_enqueued(self, sinks):
        """Listener for ops: the gradients `sinks` (tensors) have just been enqueued on the
        second stream.  Which gradients arrive that way is learned from the first step;
        from then on, when the last of the tail's has arrived, the tail is reduced
        right behind it."""
        ptrs = self._offsets(sinks)
        self.seen.update(ptrs)
        if self.pending is None:
            return
        self.pending.difference_update(ptrs)
        if not self.pending:
            self.pending = None
            self._reduce_tail()

    def _reduce_tail(self):
        tail = self.flat.grad[self.split:]
        if tail.is_cuda:
            from . import ops
            side = ops.side_stream()
            side.wait_stream(torch.cuda.current_stream())     # gradients accumulated by autograd on this stream
            with torch.cuda.stream(side):
                self.work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self.work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)

    def buckets(self):
        """Bytes of the collectives one step issues, in issue order."""
        n = self.flat.grad.numel()
        if self.split and self.overlap:
            return [4 * (n - self.split), 4 * self.split]
        return [4 * n]

    def finish(self):
        if not is_active():
            return 1.0
        if self.flat.grad.is_cuda:
            from . import ops
            ops.join_side_stream()
        if self.skip:
            return 1.0 / dist.get_world_size()
        if self.work is None:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM)
        else:
            self.work.wait()
            self.work = None
            dist.all_reduce(self.flat.grad[:self.split], op=dist.ReduceOp.SUM)
        if self.learned is None and self.split:
            self.learned = frozenset(o for o in self.seen if o >= self.split)
        self.pending = None
        return 1.0 / dist.get_world_size()
