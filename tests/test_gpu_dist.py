"""The RCCL code path on hardware with ONE rank (the build has a 1-GPU lease; the 8-GPU runs are
the driver's): process-group init on the nccl (= RCCL) backend, parameter broadcast, and both
all-reduces of dist.GradReducer -- the early tail bucket on the second stream, the head after the
backward pass -- inside engine.ASRTrainStep.  With one rank every collective is the identity, so
losses and weights must equal the run without a process group (to the rounding of the atomically
accumulated weight gradients); what the test proves
is that the library loads, the collectives are issued on the right streams and nothing deadlocks.
Multi-rank arithmetic is covered by tests/test_ddp_cpu.py (gloo, world size 2)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, random, sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'oracle'))
import las_oracle as lo
import torch.distributed as dist
from ss_asr_amd import dist as sdist
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.synthetic import make_batch
rank, world, local = sdist.init_from_env()
calls = []
if sdist.is_active():
    assert dist.get_backend() == 'nccl'
    real = dist.all_reduce
    def counted(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = counted
torch.cuda.set_device(0)
model = ASR(50, 256, 256, 128, 80, 1.0)
lo.seeded_weights(model, 3)
model = model.to('cuda:0')
step = ASRTrainStep(model)
losses = []
for k in range(3):
    x, y, lens = make_batch(np.array([200 - 8 * k, 160, 96, 40]), np.array([14, 9, 6, 3]), 80, seed=40 + k)
    _, ans_len = label_geometry(y)
    random.seed(k)
    losses.append(float(step(x.cuda(), y.cuda(), lens, ans_len)))
norm, skipped = step.finish()
w = float(step.flat.data.double().abs().sum())
print('RESULT ' + json.dumps(dict(active=sdist.is_active(), losses=losses, norm=norm, wsum=w, calls=calls,
                                  numel=step.flat.numel, split=step.reducer.split)))
sdist.shutdown()
'''


def _run(single):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_SINGLE', 'SSASR_DIST_BACKEND'):
        env.pop(k, None)
    if single:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env.update(SSASR_DIST_SINGLE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=ROOT)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[7:])


@pytest.mark.timeout(900)
def test_rccl_single_rank_train_steps_equal_the_plain_run():
    plain = _run(False)
    rccl = _run(True)
    assert not plain['active'] and rccl['active']
    assert plain['calls'] == []
    n, split = rccl['numel'], rccl['split']
    assert 0 < split < n
    # step 1: one collective over the whole flat gradient (which gradients arrive by the second
    # stream is learned from it); steps 2, 3: the tail bucket early, the head after backward
    assert rccl['calls'] == [n, n - split, split, n - split, split], rccl['calls']
    # identity collectives: equal up to the run-to-run rounding of the split-K weight-gradient atomics
    assert max(abs(a - b) for a, b in zip(rccl['losses'], plain['losses'])) < 1e-5
    assert abs(rccl['norm'] - plain['norm']) < 1e-5 * max(1.0, plain['norm'])
    assert abs(rccl['wsum'] - plain['wsum']) < 1e-6 * plain['wsum']
