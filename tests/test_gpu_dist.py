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


def _run(single, no_overlap=False):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_SINGLE', 'SSASR_DIST_BACKEND', 'SSASR_DDP_NO_OVERLAP'):
        env.pop(k, None)
    if no_overlap:
        env['SSASR_DDP_NO_OVERLAP'] = '1'
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
    # the other mode (SSASR_DDP_NO_OVERLAP=1, what bench.py falls back to after a persistent time-out): ONE
    # collective over the whole buffer after every backward pass, same results
    late = _run(True, no_overlap=True)
    assert late['active'] and late['calls'] == [n, n, n], late['calls']
    assert max(abs(a - b) for a, b in zip(late['losses'], plain['losses'])) < 1e-5
    assert abs(late['wsum'] - plain['wsum']) < 1e-6 * plain['wsum']


def _bench(extra_env, *flags):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_BACKEND', 'SSASR_DDP_NO_OVERLAP', 'SSASR_DDP_FALLBACK'):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env.update(SSASR_DIST_SINGLE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **extra_env)
    env.setdefault('SSASR_DDP_FALLBACK_LOG', os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ssasr_bench_first_attempt_%d.log' % os.getpid()))
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-roofline',
                          '--no-config4', '--no-cpu-baseline', '--no-epoch'] + list(flags), env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0]), res.stderr


@pytest.mark.timeout(1200)
def test_bench_falls_back_to_the_late_all_reduce_after_a_persistent_time_out():
    """the first multi-rank run must not be able to come back empty.  bench.py on a one-rank
    RCCL process group (SSASR_DIST_SINGLE=1: the collective path is on) with a REAL missing producer injected
    for as long as the overlapped two-bucket mode is selected: (a) the rank catches the time-out its status row
    reports, every rank switches to one collective after the backward pass, the region is repeated and the
    line says which mode produced the number; (b) when the ranks die of the time-out instead, the parent --
    which never touches a GPU -- starts ONE fresh set of ranks with SSASR_DDP_NO_OVERLAP=1.  Both lines carry
    the collective's self-description."""
    clean, _ = _bench({})
    assert clean['config']['ddp_overlap'] is True and clean['config']['ddp_fallback'] is None
    col = clean['collective']
    assert col['backend'] == 'nccl' and col['world_size'] == 1 and col['bytes_per_step'] == sum(col['buckets'])
    assert len(col['buckets']) == 2 and col['allreduce_alone_ms']['reps'] == 20 and col['ms_per_step_without_reduce'] > 0
    inproc, err = _bench({'SSASR_TEST_DROP_TILE_IF_OVERLAP': '3'})
    assert inproc['config']['ddp_overlap'] is False and inproc['config']['ddp_fallback'].startswith('in-process'), inproc['config']
    assert 'timed out' in err and inproc['collective']['buckets'] == [inproc['collective']['bytes_per_step']]
    # (the timed-out steps were NaN-skipped by the optimizer kernel: the weights survive them)
    assert inproc['value'] > 0 and abs(inproc['final_loss'] - clean['final_loss']) < 0.5
    parent, err = _bench({'SSASR_TEST_DROP_TILE_IF_OVERLAP': '3', 'SSASR_TEST_BENCH_DIE_ON_TIMEOUT': '1'}, '--self-launch')
    assert parent['config']['ddp_overlap'] is False and parent['config']['ddp_fallback'] == 'parent', parent['config']
    assert err.count('starting 1 ranks') == 2 and 'starting fresh ranks with SSASR_DDP_NO_OVERLAP=1' in err
    # the first attempt's evidence survives the fallback and the line names it
    ev = parent['config']['ddp_fallback_evidence']
    assert ev and os.path.isfile(ev) and 'timed out' in open(ev).read()
    assert clean['config']['ddp_fallback_evidence'] is None
    assert clean['collective']['rccl_channels_granted'] is None or clean['collective']['rccl_channels_granted'] >= 1


# ---- two ranks on the one GPU, gloo between them, the REAL train step -------------
# the section-8(e) statement had only been asserted on a linear stand-in model.
# Here both ranks run engine.ASRTrainStep on cuda:0 (SSASR_DIST_BACKEND=gloo: RCCL cannot serve two
# ranks on one device), over FIVE batches -- an odd count, so the tail batch is dropped and both
# ranks run the same number of steps (two: the second goes through GradReducer's overlapped two-bucket path).  A third, single-process child computes what 8(e) says
# the ranks must see: the loss of each local batch on its own, and the weights after one
# Solver.step on the MEAN of the two gradients.
DDP_COMMON = r'''
import json, os, random, sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'oracle'))
import las_oracle as lo
from ss_asr_amd import dist as sdist, ops
from ss_asr_amd.asr import ASR
from ss_asr_amd.engine import ASRTrainStep, label_geometry
from ss_asr_amd.gpu_loader import rank_batches
from ss_asr_amd.synthetic import make_batch
DIMS = (50, 256, 256, 128, 80)
def batch(k):
    if os.environ.get('SSASR_TEST_LONG'):
        # BASELINE.json configs[3]'s shape: T' = 375 encoder frames -> blstm_4 runs as three column windows, the decode
        # loop takes its long-encoder form
        return make_batch(np.array([3000 - 160 * k, 2400, 1800 + 80 * k, 1520]), np.array([24, 18 + k, 12, 9]), 80, seed=170 + k)
    return make_batch(np.array([216 - 16 * k, 160, 96 + 8 * k, 40]), np.array([14, 9 + k, 6, 3]), 80, seed=70 + k)
def model():
    m = ASR(*DIMS, 1.0)
    lo.seeded_weights(m, 9)
    return m.to('cuda:0')
PROBE = ('encoder.blstm_1.layer.weight_hh_l0', 'encoder.blstm_4.weight_ih_l0_reverse', 'attention.phi.weight',
         'decoder.layer_1.weight_ih', 'char_trans.bias')
def report(m, **kw):
    p = dict(m.named_parameters())
    kw['probe'] = {n: p[n].detach().reshape(-1)[:64].double().cpu().tolist() for n in PROBE}
    kw['wsum'] = float(torch.cat([t.detach().reshape(-1) for t in m.parameters()]).double().abs().sum())
    print('RESULT ' + json.dumps(kw))
'''

DDP_RANK = DDP_COMMON + r'''
rank, world, local = sdist.init_from_env()
assert world == 2 and sdist.is_active()
import torch.distributed as dist
assert dist.get_backend() == 'gloo'
torch.cuda.set_device(0)
m = model()
step = ASRTrainStep(m)
mine = list(rank_batches(5, rank, world))          # five batches, two ranks: two steps each, the tail batch is dropped
# Test-only orchestration: the two ranks share ONE GPU here (in production each has its own), and two
# persistent launches of different processes may each get part of the chip and wait for ever for the
# rest.  A file lock keeps the ranks' forward + backward passes apart; it is released before the
# gradient all-reduce, where the ranks must meet.
import fcntl
lockf = open(os.environ['SSASR_TEST_LOCK'], 'w')
real_finish = step.reducer.finish
def finish_after_unlock():
    ops.join_side_stream()
    torch.cuda.synchronize()
    fcntl.flock(lockf, fcntl.LOCK_UN)
    return real_finish()
step.reducer.finish = finish_after_unlock
losses = []
for k in mine:
    x, y, lens = batch(k)
    _, ans_len = label_geometry(y)
    random.seed(0)
    fcntl.flock(lockf, fcntl.LOCK_EX)
    losses.append(float(step(x.cuda(), y.cuda(), lens, ans_len)))
norm, skipped = step.finish()
report(m, rank=rank, mine=mine, losses=losses, norm=norm, skipped=bool(skipped), split=step.reducer.split,
       overlapped=bool(step.reducer.learned) and step.reducer.overlap)
sdist.shutdown()
'''

DDP_SINGLE = DDP_COMMON + r'''
from ss_asr_amd.optim import FlatParameters, FusedAdadelta
torch.cuda.set_device(0)
m = model()
flat = FlatParameters(m)
optim = FusedAdadelta(flat, lr=1.0, eps=1e-8)
losses, norm = [], None
for rnd in range(2):                                   # round = one step of BOTH ranks: batches 2 rnd, 2 rnd + 1
    total = torch.zeros_like(flat.grad)
    for k in (2 * rnd, 2 * rnd + 1):
        x, y, lens = batch(k)
        _, ans_len = label_geometry(y)
        optim.zero_grad()
        random.seed(0)
        _, logits, _ = m(x.cuda(), ans_len, teacher=y.cuda(), state_len=lens)
        loss = ops.masked_ce_loss(logits, y.cuda(), ans_len)
        loss.backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        ops.check_persistent_status()
        losses.append(float(loss))
        total += flat.grad
    flat.grad.copy_(total)
    optim.clip_and_step(max_norm=5.0, grad_scale=0.5)   # Solver.step on the mean of the two gradients
    norm, skipped = optim.poll(wait=True)
    assert not skipped
report(m, losses=losses, norm=norm, skipped=False)
'''


def _child(script, env):
    return subprocess.Popen([sys.executable, '-c', script % dict(root=ROOT)], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)


def _result(proc, timeout):
    out, _ = proc.communicate(timeout=timeout)
    assert proc.returncode == 0, out[-3000:]
    return json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][-1][7:])


@pytest.mark.timeout(900)
@pytest.mark.parametrize('overlap,long_shape', [(True, False), (False, False), (True, True)])
def test_two_ranks_on_one_gpu_follow_the_section_8e_statement_on_the_real_step(tmp_path, overlap, long_shape):
    base = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_SINGLE', 'SSASR_DIST_BACKEND', 'SSASR_DDP_NO_OVERLAP',
              'SSASR_TEST_LONG'):
        base.pop(k, None)
    if long_shape:              # configs[3] under DDP: column windows + the long-encoder decode loop
        base['SSASR_TEST_LONG'] = '1'
    single = _result(_child(DDP_SINGLE, base), 420)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    lock = str(tmp_path / 'gpu.lock')
    for r in range(2):
        env = dict(base, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), SSASR_DIST_BACKEND='gloo', SSASR_TEST_LOCK=lock)
        if not overlap:
            env['SSASR_DDP_NO_OVERLAP'] = '1'          # the mode bench.py falls back to
        procs.append(_child(DDP_RANK, env))
    try:
        ranks = [_result(p, 600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ranks.sort(key=lambda d: d['rank'])
    # equal step counts with an odd batch count: batch 4 is nobody's
    assert [d['mine'] for d in ranks] == [[0, 2], [1, 3]]
    assert not single['skipped'] and not any(d['skipped'] for d in ranks)
    # the second step went through the two-bucket path: the tail of the flat gradient reduced on the second
    # stream while the first layer's BPTT ran, the head after the backward pass
    # (or, overlap off, through one collective after the backward pass: same arithmetic)
    assert all(d['overlapped'] == overlap and d['split'] > 0 for d in ranks)
    # rank-local loss == the single-process loss on that local batch (step 2: from the weights both ranks hold
    # after the first averaged update)
    for r in range(2):
        for stp in range(2):
            assert abs(ranks[r]['losses'][stp] - single['losses'][2 * stp + r]) < 5e-6, (r, stp, ranks[r]['losses'], single['losses'])
    # both ranks clipped the same, reduced gradient ...
    assert abs(ranks[0]['norm'] - ranks[1]['norm']) < 1e-6 * max(1.0, ranks[0]['norm'])
    assert abs(ranks[0]['norm'] - single['norm']) < 5e-5 * max(1.0, single['norm'])
    # ... and hold the weights of two Solver.steps on the means of the single-process gradients
    for r in range(2):
        assert abs(ranks[r]['wsum'] - single['wsum']) < 1e-6 * single['wsum']
        for n, want in single['probe'].items():
            got = ranks[r]['probe'][n]
            assert max(abs(a - b) for a, b in zip(got, want)) < 5e-5, (r, n)
    assert ranks[0]['probe'] == ranks[1]['probe']


# ---- the Seed loop's ADV and SAE legs under data parallelism: two gloo ranks on the one GPU -------------------------
# Each rank runs ONE iteration of engine.ADVTrainStep / engine.SAETrainStep on its own batch; a single-process child
# computes what plain data parallelism must give: every optimizer stepped once on the MEAN of the two ranks'
# gradients (for ADV: the discriminator first, then the generator pass of BOTH batches through the updated
# discriminator).  The ranks share the GPU, so a file lock keeps their persistent launches apart; it is released
# around every gradient all-reduce, where the ranks must meet.
SEED_COMMON = DDP_COMMON + r'''
from ss_asr_amd.discriminator import Discriminator
from ss_asr_amd.engine import ADVTrainStep, SAETrainStep
from ss_asr_amd.speech_autoencoder import SpeechAutoEncoder
from ss_asr_amd.text_autoencoder import TextAutoEncoder
LEG = os.environ['SSASR_TEST_LEG']
def seed_batch(k):
    return make_batch(np.array([208 - 16 * k, 160, 96 + 8 * k, 40]), np.array([14, 9 + k, 6, 3]), 80, seed=90 + k, pad_to=208)
def models():
    torch.manual_seed(0)
    asr = model()
    tae = TextAutoEncoder(50, 128, 256, 2); lo.seeded_tae_weights(tae, 10)
    if LEG == 'adv':
        other = Discriminator(512, 256); lo.seeded_generic_weights(other, 11)
    else:
        other = SpeechAutoEncoder(512, 80, [[1, 36], [5, 1], [3, 1]], [32, 64, 256], [[3, 1], [5, 1], [11, 40]])
        lo.seeded_generic_weights(other, 12)
    return asr, tae.to('cuda:0'), other.to('cuda:0')
def make_step(asr, tae, other):
    return ADVTrainStep(asr, tae, other) if LEG == 'adv' else SAETrainStep(asr, other)
def run(step, k):
    x, y, lens = seed_batch(k)
    return step(x.cuda(), lens, y.cuda()) if LEG == 'adv' else step(x.cuda(), lens)
def seed_report(asr, other, **kw):
    kw['other_wsum'] = float(torch.cat([t.detach().reshape(-1) for t in other.parameters()]).double().abs().sum())
    kw['other_probe'] = next(other.parameters()).detach().reshape(-1)[:64].double().cpu().tolist()
    report(asr, **kw)
'''

SEED_RANK = SEED_COMMON + r'''
import fcntl
rank, world, local = sdist.init_from_env()
assert world == 2 and sdist.is_active()
torch.cuda.set_device(0)
asr, tae, other = models()
step = make_step(asr, tae, other)
lockf = open(os.environ['SSASR_TEST_LOCK'], 'w')
real_ar = sdist.allreduce_grad
def meet(g):
    ops.join_side_stream(); torch.cuda.synchronize()
    fcntl.flock(lockf, fcntl.LOCK_UN)
    r = real_ar(g)
    torch.cuda.synchronize()
    fcntl.flock(lockf, fcntl.LOCK_EX)
    return r
sdist.allreduce_grad = meet
import ss_asr_amd.engine as eng
eng.sdist.allreduce_grad = meet
fcntl.flock(lockf, fcntl.LOCK_EX)
out = run(step, rank)
torch.cuda.synchronize()
fcntl.flock(lockf, fcntl.LOCK_UN)
norm, skipped = step.finish()
losses = [float(v) for v in out] if LEG == 'adv' else [float(out)]
seed_report(asr, other, rank=rank, losses=losses, norm=norm, skipped=bool(skipped))
sdist.shutdown()
'''

SEED_SINGLE = SEED_COMMON + r'''
from ss_asr_amd.seed_ops import bce_loss
torch.cuda.set_device(0)
asr, tae, other = models()
step = make_step(asr, tae, other)
one = torch.ones((), device='cuda:0')
def sync():
    ops.join_side_stream(); torch.cuda.synchronize(); ops.check_persistent_status()
losses = []
if LEG == 'sae':
    total_s, total_a = torch.zeros_like(step.sae_flat.grad), torch.zeros_like(step.asr_flat.grad)
    for k in (0, 1):
        step.sae_flat.zero_grad(); step.asr_flat.zero_grad()
        x, _, lens = seed_batch(k)
        loss, _ = step.forward_loss(x.cuda(), lens)
        loss.backward(one); sync()
        losses.append([float(loss)])
        total_s += step.sae_flat.grad; total_a += step.asr_flat.grad
    step.sae_flat.grad.copy_(total_s); step.asr_flat.grad.copy_(total_a)
    step.optim.clip_and_step(5.0, grad_scale=0.5)
    norm, skipped = step.optim.poll(wait=True)
else:
    # discriminator: both batches' two passes, one step on the mean
    fakes, total = [], torch.zeros_like(step.d_flat.grad)
    step.asr_flat.zero_grad()
    for k in (0, 1):
        step.d_flat.zero_grad()
        x, y, lens = seed_batch(k)
        real, fake = step.frames(x.cuda(), lens, y.cuda())
        d_real = bce_loss(step.disc(real), 0.9); d_real.backward(one)
        d_fake = bce_loss(step.disc(fake.detach()), 0.0); d_fake.backward(one)
        sync()
        losses.append([float(d_real), float(d_fake)])
        fakes.append(fake)
        total += step.d_flat.grad
    step.d_flat.grad.copy_(total)
    step.D_optim.clip_and_step(5.0, grad_scale=0.5)
    assert not step.D_optim.poll(wait=True)[1]
    # generator: both batches through the updated discriminator, one step on the mean
    for k in (0, 1):
        g = bce_loss(step.disc(fakes[k], frozen=True), 1.0)
        g.backward(one); sync()
        losses[k].append(float(g))
    step.G_optim.clip_and_step(5.0, grad_scale=0.5)
    norm, skipped = step.G_optim.poll(wait=True)
assert not skipped
seed_report(asr, other, losses=losses, norm=norm, skipped=False)
'''


@pytest.mark.timeout(900)
@pytest.mark.parametrize('leg', ['adv', 'sae'])
def test_two_ranks_run_the_seed_loops_other_legs_as_plain_data_parallelism(tmp_path, leg):
    base = dict(os.environ, SSASR_TEST_LEG=leg)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_SINGLE', 'SSASR_DIST_BACKEND', 'SSASR_DDP_NO_OVERLAP'):
        base.pop(k, None)
    single = _result(_child(SEED_SINGLE, base), 420)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    lock = str(tmp_path / 'gpu.lock')
    procs = [_child(SEED_RANK, dict(base, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                                    MASTER_PORT=str(port), SSASR_DIST_BACKEND='gloo', SSASR_TEST_LOCK=lock)) for r in range(2)]
    try:
        ranks = [_result(p, 600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ranks.sort(key=lambda d: d['rank'])
    assert not any(d['skipped'] for d in ranks)
    for r in range(2):
        # rank-local losses = the single-process losses on that batch (ADV: its generator loss is taken through the
        # discriminator updated with the MEAN gradient, as in the single-process run)
        assert max(abs(a - b) for a, b in zip(ranks[r]['losses'], single['losses'][r])) < 5e-6, (r, ranks[r]['losses'], single['losses'])
        assert abs(ranks[r]['norm'] - single['norm']) < 5e-5 * max(1.0, single['norm'])
        assert abs(ranks[r]['wsum'] - single['wsum']) < 1e-6 * single['wsum']
        assert abs(ranks[r]['other_wsum'] - single['other_wsum']) < 1e-6 * single['other_wsum']
        for n, want in single['probe'].items():
            assert max(abs(a - b) for a, b in zip(ranks[r]['probe'][n], want)) < 5e-5, (r, n)
        assert max(abs(a - b) for a, b in zip(ranks[r]['other_probe'], single['other_probe'])) < 5e-5
    assert ranks[0]['probe'] == ranks[1]['probe'] and ranks[0]['other_probe'] == ranks[1]['other_probe']


# --- the Seed loop's checkpoint hand-off under two ranks (rank 0 alone writes asr_1/2/3.cpt and tae.cpt in
# close(); every rank reads them in the next leg's set_model) ---------------------------------------------------
SEED_LOOP_RANK = r'''
import json, os, random, sys, types
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'oracle')); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
from test_gpu_trainer import _seed_config, _paras
from ss_asr_amd import dist as sdist, trainer as T
root = os.environ['SSASR_TEST_ROOT']
index = os.path.join(root, 'index.tsv')
conf = _seed_config(index, n_epochs=1)
conf['seed_train'] = {'super_its': 1}
random.seed(1); np.random.seed(1); torch.manual_seed(1 + int(os.environ['RANK']))      # ranks would build DIFFERENT fresh models
ends = []
for cls in (T.TAETrainer, T.ADVTrainer, T.SAETrainer):
    def wrap(orig, name):
        def close(self):
            mods = [self.asr_model] + [getattr(self, a) for a in ('text_autoenc', 'discriminator', 'speech_autoenc') if hasattr(self, a)]
            torch.cuda.synchronize()
            ends.append([name] + [float(sum(p.double().abs().sum() for p in m.parameters())) for m in mods] +
                        [float(sum(b.double().abs().sum() for b in m.buffers())) for m in mods])
            return orig(self)
        return close
    cls.close = wrap(cls.close, cls.__name__)
T.asr_seed_train(conf, _paras(root, 'seed2'))
print('RESULT ' + json.dumps(dict(rank=int(os.environ['RANK']), ends=ends)))
sdist.shutdown()
'''


@pytest.mark.timeout(1200)
def test_two_ranks_hand_the_seed_loops_model_from_leg_to_leg_through_rank_zeros_checkpoints(tmp_path):
    """One super-iteration of trainer.asr_seed_train on two ranks (gloo between them, both on the one GPU), each
    rank seeded differently: without the barrier after rank 0's save and the broadcasts of what a step object only
    reads (the text autoencoder, batch-norm buffers), rank 1 would build fresh models or read half-written files.
    At the end of every leg both ranks hold the same parameters and started from the same buffers."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_host_cpu import make_corpus
    root = str(tmp_path)
    make_corpus(root, n=32, t_max=32, feat=80, seed=7)
    base = dict(os.environ, SSASR_TEST_ROOT=root)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SSASR_DIST_SINGLE', 'SSASR_DIST_BACKEND', 'SSASR_DDP_NO_OVERLAP'):
        base.pop(k, None)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = [_child(SEED_LOOP_RANK, dict(base, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                                         MASTER_PORT=str(port), SSASR_DIST_BACKEND='gloo')) for r in range(2)]
    try:
        ranks = [_result(p, 900) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ranks.sort(key=lambda d: d['rank'])
    a, b = ranks[0]['ends'], ranks[1]['ends']
    assert [e[0] for e in a] == ['TAETrainer', 'ADVTrainer', 'SAETrainer'] == [e[0] for e in b]
    for ea, eb in zip(a, b):
        n = (len(ea) - 1) // 2
        for va, vb in zip(ea[1:1 + n], eb[1:1 + n]):            # parameters: equal on both ranks after the leg
            assert abs(va - vb) <= 1e-9 * max(1.0, abs(va)), (ea, eb)
    ckpdir = os.path.join(root, 'result', 'seed2')
    for f in ('asr_1.cpt', 'asr_2.cpt', 'asr_3.cpt', 'tae.cpt', 'adv.cpt', 'sae.cpt'):
        assert os.path.isfile(os.path.join(ckpdir, f)), f
        torch.load(os.path.join(ckpdir, f), map_location='cpu')                     # whole archives
    assert not [f for f in os.listdir(ckpdir) if '.tmp.' in f]
