"""World-size-2 checks of the data-parallel host logic on the gloo backend (CPU).
The arithmetic under test is the collective contract of ss_asr_amd/dist.py:
SUM all-reduce of the flat gradient, averaging folded into the optimizer as
grad_scale = 1/world, parameters broadcast from rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ss_asr_amd import dist as sdist
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.optim import FlatParameters
    r, w, _ = sdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and sdist.is_active() and sdist.world_size() == world

    torch.manual_seed(100 + rank)                      # ranks start from different weights
    model = ASR(50, 32, 32, 16, 12, 1.0)
    flat = FlatParameters(model)
    sdist.broadcast_flat(flat.data)
    ref = [torch.empty_like(flat.data) for _ in range(world)]
    dist.all_gather(ref, flat.data)
    assert all(torch.equal(ref[0], t) for t in ref)    # everyone holds rank 0's parameters

    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(flat.numel, generator=g)
    flat.grad.copy_(local)
    scale = sdist.allreduce_grad(flat.grad)
    assert scale == 1.0 / world
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    assert torch.allclose(flat.grad * scale, mean, atol=1e-6)
    # the clip norm every rank would compute is identical (same branch of the NaN guard)
    norms = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(norms, (flat.grad * scale).norm().reshape(1))
    assert all(torch.equal(norms[0], n) for n in norms)
    # parameter views still alias the flat buffer after the collective
    p = next(model.parameters())
    assert p.grad.data_ptr() == flat.grad.data_ptr()

    # two-bucket reduction (GradReducer): head = first encoder layer, tail reduced early
    red = sdist.GradReducer(flat, list(model.encoder.blstm_1.parameters()))
    assert 0 < red.split < flat.numel
    params = list(model.parameters())
    nhead = len(list(model.encoder.blstm_1.parameters()))
    deferred = [q.grad for q in params[nhead:] if q.dim() >= 1][3:]        # some tail gradients arrive "late"
    head_sinks = [q.grad for q in params[:nhead]]
    for step in range(3):
        g2 = torch.Generator().manual_seed(1000 * step + rank)
        local = torch.randn(flat.numel, generator=g2)
        red.begin()
        flat.grad.copy_(local)                       # stands for the gradients autograd wrote directly
        if step == 0:
            assert red.pending is None               # nothing learned yet: one collective
        else:
            assert red.pending is not None and red.work is None
        half = len(deferred) // 2
        red.wgrad_enqueued(deferred[:half])
        assert red.work is None
        red.wgrad_enqueued(deferred[half:])
        assert (red.work is not None) == (step > 0)  # tail goes out as soon as its last deferred part is in
        red.wgrad_enqueued(head_sinks)               # the head's own gradients arrive last
        sc = red.finish()
        assert sc == 1.0 / world and red.work is None
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(flat.grad, sum(gathered), atol=1e-5), step
    # An odd number of batches: both ranks run the same number of steps (the tail batch is
    # dropped), so every all-reduce has its peer and the shared step counter stays aligned.
    from ss_asr_amd.gpu_loader import rank_batches
    mine = rank_batches(5, rank, world)
    assert mine == [rank, 2 + rank]
    for _ in mine:
        t = torch.ones(4)
        dist.all_reduce(t)
        assert float(t[0]) == world

    # SURVEY.md 8(e) parity statement on stand-in steps: a rank's loss on its local batch equals
    # the single-process loss on that batch, and the averaged gradient equals the mean of the
    # per-rank single-process gradients (a linear model stands in for the HIP step, which has no
    # CPU path; the collective contract under test is dist.py's).
    torch.manual_seed(5)
    w0 = torch.randn(6, 3)
    data = [(torch.randn(4, 6, generator=torch.Generator().manual_seed(50 + r)),
             torch.randn(4, 3, generator=torch.Generator().manual_seed(60 + r))) for r in range(world)]

    def local_step(r):
        w = w0.clone().requires_grad_(True)
        xb, yb = data[r]
        loss = ((xb @ w - yb) ** 2).mean()
        loss.backward()
        return float(loss), w.grad.reshape(-1).clone()
    singles = [local_step(r) for r in range(world)]          # what one process would compute per batch
    my_loss, my_grad = local_step(rank)
    assert my_loss == singles[rank][0]
    buf = my_grad.clone()
    sc = sdist.allreduce_grad(buf)
    want = sum(g for _, g in singles) / world
    assert torch.allclose(buf * sc, want, atol=1e-6)

    if rank == 0:
        with open(out, 'w') as f:
            f.write('ok')
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world_size_two(tmp_path):
    out = os.path.join(str(tmp_path), 'done')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == 'ok'
