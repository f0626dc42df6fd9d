"""Plain data parallelism for the ASR train step: one process per GPU, one
collective per step.

The reference is single-process; this is the one
parallel strategy the build adds.  Utterance batches shard naturally across
ranks; the only exchange is a SUM all-reduce of the flat fp32 gradient buffer
(10,269,874 values = 41 MB at the default sizes) issued once per step on RCCL
(torch.distributed backend "nccl" on ROCm).  Averaging is folded into the
clip + Adadelta kernel as ``grad_scale = 1 / world_size`` so the gradient is
not touched a second time, and clipping / the NaN guard run after the
reduction so every rank takes the same branch.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when
    they are set (torchrun); returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1, 0
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def is_active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_active() else 1


def broadcast_flat(flat_data):
    """Makes every rank start from rank 0's parameters."""
    if is_active():
        dist.broadcast(flat_data, src=0)


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
