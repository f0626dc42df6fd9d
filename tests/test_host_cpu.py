"""CPU-only checks (no GPU in the build container): the C-ABI library loads and
exports every symbol include/ssasr.h declares, the host-side mirrors of the
reference's data / trainer surface behave like the reference, and the product
path refuses to compute without a GPU instead of falling back."""
import ctypes
import json
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_class(decl):
    """'ptr' / 'i64' / 'i32' / 'f32' of one C parameter or field declaration."""
    decl = decl.strip()
    if '*' in decl:
        return 'ptr'
    base = re.sub(r'\bconst\b', '', decl).split()[0]
    return {'int64_t': 'i64', 'int': 'i32', 'int32_t': 'i32', 'float': 'f32'}[base]


def _ctypes_class(t):
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, 'contents'):
        return 'ptr'
    return {ctypes.c_int64: 'i64', ctypes.c_int: 'i32', ctypes.c_int32: 'i32', ctypes.c_float: 'f32'}[t]


def header_prototypes():
    """{name: (return class, [parameter classes])} of every function include/ssasr.h declares."""
    header = open(os.path.join(ROOT, 'include', 'ssasr.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    protos = {}
    for ret, name, params in re.findall(r'\b(int|int64_t)\s+(ssasr_\w+)\s*\(([^)]*)\)\s*;', header):
        params = params.strip()
        plist = [] if params in ('', 'void') else [_c_class(p) for p in params.split(',')]
        protos[name] = (_c_class(ret), plist)
    return protos, header


def test_library_exports_every_declared_symbol():
    from ss_asr_amd import _lib
    protos, header = header_prototypes()
    declared = set(protos)
    assert len(declared) >= 17
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.ssasr_abi_version() == _lib.ABI_VERSION
    # struct layouts bound by ctypes must match the C declarations field for field, name and class
    for cname, cls in (('ssasr_decoder', _lib.Decoder), ('ssasr_decoder_grads', _lib.DecoderGrads)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), header, re.S).group(1)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:
                names = re.sub(r'^(const\s+)?\w+\s*\**', '', decl, count=1).split(',')
                fields += [(f.strip().lstrip('*').strip(), _c_class(decl)) for f in names]
        assert fields == [(f[0], _ctypes_class(f[1])) for f in cls._fields_], cname


def test_ctypes_signatures_match_the_header_prototypes():
    """Per entry point: return class, parameter count and each parameter's class (pointer / int64 /
    int / float) in include/ssasr.h == the ctypes table the product binds (VERDICT r2, row b: a
    prototype that drifts from the definition passes a stream where an int is expected)."""
    from ss_asr_amd import _lib
    protos, _ = header_prototypes()
    mismatches = []
    for name, (res, args) in _lib.SIGNATURES.items():
        got = (_ctypes_class(res), [_ctypes_class(a) for a in args])
        if got != protos[name]:
            mismatches.append((name, protos[name], got))
    assert not mismatches, mismatches


def test_a_c_caller_builds_against_the_header(tmp_path):
    """A plain C program compiled from include/ssasr.h alone (gcc -Wall -Werror, no HIP headers)
    links the shared object, reads the ABI version and gets negative codes for argument errors --
    the header is what a C caller sees, so it has to agree with the library by itself."""
    import shutil
    import subprocess
    from ss_asr_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc on this box')
    _lib.load()
    exe = str(tmp_path / 'abi_smoke')
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', exe, '-L', libdir,
                    '-lssasr_hip', '-Wl,-rpath,' + libdir], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert 'abi %d' % _lib.ABI_VERSION in out.stdout


def integration_stub():
    """The ctypes stub INTEGRATION.md section 2 documents, executed: returns its namespace."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, re.S)
    block = [b for b in blocks if 'ssasr_attn_step_fwd.argtypes' in b]
    assert len(block) == 1
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                    # the stub loads "ss_asr_amd/libssasr_hip.so" relative to the repo root
    try:
        exec(compile(block[0], 'INTEGRATION.md', 'exec'), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_integration_stub_binds_the_header_prototype():
    """The documented drop-in stub runs against the built library and declares exactly the
    prototype of include/ssasr.h (a stale stub passes the stream as `ws`)."""
    ns = integration_stub()
    protos, _ = header_prototypes()
    fn = ns['lib'].ssasr_attn_step_fwd
    assert (_ctypes_class(fn.restype), [_ctypes_class(a) for a in fn.argtypes]) == protos['ssasr_attn_step_fwd']
    assert callable(ns['attention_step'])


def test_argument_errors_are_negative_and_need_no_gpu():
    from ss_asr_amd import _lib
    lib = _lib.load()
    assert lib.ssasr_bilstm_fwd(None, 0, 0, 0, 0, 0, 0, None, *([None] * 8), None, 0, 0, None,
                                None, None, None, None, 0, None, None) < 0
    assert lib.ssasr_decoder_fwd(None, None) < 0
    assert lib.ssasr_clip_adadelta_ws(10269874) == 1 + (10269874 + 4095) // 4096
    # diagnostic switches: known names round-trip, unknown names are refused
    assert _lib.set_option('SSASR_GEMM_TILE', 64) == 0 and _lib.set_option('SSASR_GEMM_TILE', 0) == 64
    # matrix products default to the split-bf16 form; 0 selects the fp32 MFMA instruction
    assert _lib.set_option('SSASR_GEMM_X6', 0) == 1 and _lib.set_option('SSASR_GEMM_X6', 1) == 0
    assert lib.ssasr_set_option(b'SSASR_NO_SUCH_SWITCH', 1) < 0


def test_ops_refuse_cpu_tensors():
    from ss_asr_amd import ops
    from ss_asr_amd.asr import ASR
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    model = ASR(50, 32, 32, 16, 12, 1.0)
    with pytest.raises(RuntimeError, match='no CPU path'):
        model(torch.randn(2, 8, 12), 3, teacher=torch.zeros(2, 5, dtype=torch.long),
              state_len=[8, 8])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'ss_asr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f), encoding='utf-8').read()
                assert 'las_oracle' not in text and 'oracle/' not in text, f


def test_seeded_init_equals_the_reference_initialisation(golden):
    """ASR(...) built after seeding the RNGs as src/train.py:58-61 does holds, bit for bit, the
    parameters the reference's own ASR.__init__ + init_parameters (src/asr.py:16-50, :175-212)
    produced under the same seed (captured as w0/* of the small_tf1 fixture)."""
    import random
    from ss_asr_amd.asr import ASR
    fx = golden('small_tf1')
    seed = int(fx['seed'])
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    model = ASR(*[int(v) for v in fx['dims']], float(fx['tf_rate']))
    sd = model.state_dict()
    keys = [k[3:] for k in fx.files if k.startswith('w0/')]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        assert np.array_equal(sd[k].numpy(), fx['w0/' + k]), k


def test_state_dict_and_seeded_init_match_the_oracle_model():
    """Same constructor-time RNG consumption and key set as the reference layout."""
    import las_oracle as lo
    from ss_asr_amd.asr import ASR
    torch.manual_seed(7)
    a = ASR(50, 32, 32, 16, 12, 0.9)
    torch.manual_seed(7)
    b = lo.OracleASR(50, 32, 32, 16, 12, 0.9)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    n = sum(p.numel() for p in ASR(50, 256, 256, 128, 80, 0.9).parameters())
    assert n == 10269874


def test_packed_sequence_contract_errors():
    from ss_asr_amd.asr import _check_lengths
    _check_lengths([5, 5, 3], 5)
    with pytest.raises(RuntimeError, match='decreasing'):
        _check_lengths([3, 5], 5)
    with pytest.raises(RuntimeError, match='greater than 0'):
        _check_lengths([3, 0], 5)


def test_flat_parameters_keep_values_and_alias_grads():
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.optim import FlatParameters
    model = ASR(50, 32, 32, 16, 12, 1.0)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatParameters(model)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    p = next(model.parameters())
    p.grad.add_(1.0)
    assert float(flat.grad.sum()) == p.numel()
    flat.zero_grad()
    assert float(p.grad.abs().sum()) == 0.0
    assert all(o % 4 == 0 for o in flat.offsets)


# ----------------------------------------------------------------- data ----
def make_corpus(tmp_path, n=16, t_max=40, feat=80, seed=1):
    """BASELINE.json configs[0] in miniature: .npy fbanks padded to the corpus
    maximum (float64, like src/preprocess.py:267) and the 6-column index."""
    from ss_asr_amd.preprocess import ALL_CHARS
    rng = np.random.default_rng(seed)
    lens = sorted(rng.integers(t_max // 2, t_max + 1, size=n).tolist(), reverse=True)
    lens[0] = t_max
    rows = []
    for i, l in enumerate(lens):
        fb = np.zeros((t_max, feat))
        fb[:l] = rng.standard_normal((l, feat)).astype(np.float32)
        path = os.path.join(tmp_path, 'u%03d.npy' % i)
        np.save(path, fb)
        text = '<' + ''.join(rng.choice(list(ALL_CHARS), size=int(rng.integers(3, 9)))) + '>'
        rows.append('\t'.join([text, path, str(len(text)), str(l), 'na', 'u%03d.wav' % i]))
    index = os.path.join(tmp_path, 'index.tsv')
    with open(index, 'w', encoding='utf-8') as f:
        f.write('\n'.join(rows) + '\n')
    return index, lens


def test_flat_parameters_are_shared_between_step_objects_and_give_contiguous_runs():
    """Host logic behind the Seed loop's shared ASR object: a module has ONE flat home (FlatParameters.of finds
    it again), and the parameters behind the Listener are one contiguous run of it -- what TAETrainStep's Adam
    steps inside the ASR model's own buffer."""
    from ss_asr_amd.asr import ASR
    from ss_asr_amd.optim import FlatParameters
    model = ASR(50, 32, 32, 16, 12, 1.0)
    flat = FlatParameters.of(model)
    assert FlatParameters.of(model) is flat and flat.clean
    shared = (list(model.attention.parameters()) + list(model.decoder.parameters()) + list(model.embed.parameters()) +
              list(model.char_trans.parameters()))
    lo_, hi_ = flat.range_of(shared)
    enc = sum((p.numel() + 3) // 4 * 4 for p in model.encoder.parameters())
    assert (lo_, hi_) == (enc, flat.numel)
    assert all(lo_ <= o < hi_ for p, o in zip(flat.params, flat.offsets) if any(p is q for q in shared))
    with pytest.raises(ValueError):
        flat.range_of([model.embed.weight, model.encoder.blstm_1.layer.weight_hh_l0])      # not a contiguous run
    # a module whose parameters were re-homed elsewhere gets a fresh flat home
    model.embed.weight.data = model.embed.weight.data.clone()
    assert FlatParameters.of(model) is not flat


def test_dataset_batches_lengths_and_mapper(tmp_path):
    from ss_asr_amd.ASRDataset import Mapper, load_asr_dataset, prepare_x, prepare_y
    index, lens = make_corpus(str(tmp_path), n=18)
    mapper, ds, loader = load_asr_dataset(index, batch_size=8, n_jobs=0)
    assert mapper.get_dim() == 50 and ds.get_feature_dim() == 80
    assert len(ds) == 2                      # 18 // 8, remainder dropped (src/ASRDataset.py:63)
    x, y = next(iter(loader))
    assert x.shape[:2] == (1, 8) and x.dtype == torch.float64
    xs, x_lens = prepare_x(x)
    assert xs.dtype == torch.float32 and x_lens == lens[:8]
    ys, y_lens = prepare_y(y)
    assert ys.dtype == torch.long and ys[:, 0].eq(0).all()
    assert y_lens == [int((row != 0).sum()) + 1 for row in ys]
    text = ds.get_text(0)
    assert Mapper().translate(ds.encode(text)) == text[1:-1]


def test_postprocess_metrics():
    from ss_asr_amd.ASRDataset import Mapper
    from ss_asr_amd.postprocess import calc_acc, calc_err, edit_distance, trim_eos
    assert edit_distance('kitten', 'sitting') == 3 and edit_distance([], ['a']) == 1
    assert trim_eos([5, 6, 1, 7]) == [5, 6, 1]
    label = torch.tensor([[4, 5, 6, 1, 0], [7, 8, 1, 0, 0]])
    logits = torch.nn.functional.one_hot(torch.tensor([[4, 5, 9, 1, 3], [7, 8, 1, 2, 2]]), 50).float()
    assert abs(calc_acc(logits, label) - (3 / 4 + 1.0) / 2) < 1e-9
    assert calc_err(logits, label, Mapper()) == pytest.approx((1.0 + 0.0) / 2)


@pytest.mark.parametrize('name', ['small_tf1', 'small_odd', 'small_greedy', 'full_b4', 'edge_b1', 'edge_short', 'full_b40', 'full_b16_t400'])
def test_calc_acc_equals_the_reference_value_on_reference_logits(golden, name):
    """`acc` in the fixtures is the REFERENCE's own calc_acc (src/postprocess.py:7-29, imported through
    oracle/ref_harness.py) on its own logits; the product's metric on those logits must give that number."""
    import torch
    from ss_asr_amd.postprocess import calc_acc
    fx = golden(name)
    ans_len = int(fx['ans_len'])
    logits = torch.from_numpy(fx['logits'])[:, :ans_len]
    label = torch.from_numpy(fx['y'])[:, 1:ans_len + 1]
    assert abs(calc_acc(logits, label) - float(fx['acc'])) < 1e-12


def test_tracker_format_and_resume(tmp_path):
    from ss_asr_amd.TrackerHandler import TrackerHandler
    path = os.path.join(str(tmp_path), 'tracker.json')
    tr = TrackerHandler(path, 'asr')
    tr.do_step(); tr.do_step(); tr.set_best(3.5)
    assert json.load(open(path)) == {'asr': {'best': 3.5, 'step': 2}}
    assert TrackerHandler(path, 'asr').step == 2


def test_trainer_plumbing_without_gpu(tmp_path):
    """Config 1 (16 utterances, batch 16, feature_dim 80) through ASRTrainer up to
    the first forward, which must refuse to run on the CPU."""
    from ss_asr_amd.trainer import ASRTrainer
    if torch.cuda.is_available():
        pytest.skip('covered by the gpu test')
    index, _ = make_corpus(str(tmp_path), n=16)
    config = {'asr': {'opt': {'type': 'Adadelta', 'learning_rate': 1.0},
                      'mdl': {'encoder_state_size': 32, 'mlp_out_size': 16,
                              'decoder_state_size': 32, 'tf_rate': 0.9, 'feature_dim': 80},
                      'train_index': index, 'valid_index': index, 'wer_step': 50,
                      'train_batch_size': 16, 'valid_batch_size': 16, 'n_epochs': 1,
                      'loader_jobs': 0}}
    paras = types.SimpleNamespace(name='t', logdir=os.path.join(str(tmp_path), 'runs'),
                                  ckpdir=os.path.join(str(tmp_path), 'result'), verbose=False,
                                  seed=1)
    tr = ASRTrainer(config, paras)
    tr.load_data()
    tr.set_model()
    assert len(tr.train_set) == 1 and tr.mapper.get_dim() == 50
    with pytest.raises(RuntimeError, match='no CPU path'):
        tr.exec()


def test_flat_names_serve_the_reference_entry_point():
    """src/train.py does `import trainer` and getattr(trainer, 'ASRTrainer')."""
    code = ("import ss_asr_amd.flat; import trainer, asr, ASRDataset, preprocess;"
            "assert trainer.ASRTrainer.__module__ == 'ss_asr_amd.trainer';"
            "assert asr.ASR and ASRDataset.load_asr_dataset and preprocess.TOKENS == '<>$'")
    import subprocess
    subprocess.run([sys.executable, '-c', code], check=True, cwd=ROOT)


def test_gpu_loader_plans_the_reference_batches_and_refuses_the_cpu(tmp_path):
    from ss_asr_amd.gpu_loader import GpuResidentLoader, plan_batches
    assert plan_batches(18, 8) == [0, 8]            # whole batches only (src/ASRDataset.py:63)
    assert plan_batches(7, 8) == []
    index, _ = make_corpus(str(tmp_path), n=9)
    with pytest.raises(RuntimeError, match='no CPU path'):
        GpuResidentLoader(index, 4, 'cpu')


# ------------------------------------------------------- drop-in boundary ----
REFERENCE_TRAIN = '/root/reference/src/train.py'


@pytest.mark.skipif(not os.path.isfile(REFERENCE_TRAIN),
                    reason='build-container only: the reference tree does not travel to the GPU box')
@pytest.mark.parametrize('kind', ['ASRTrainer', 'TAETrainer', 'AdvTrainer', 'SAETrainer', 'Seed'])
def test_reference_train_script_runs_unchanged_against_this_package(tmp_path, kind):
    """`python -m ss_asr_amd.run_reference <reference>/src/train.py ...`
    executes the reference's unmodified entry point with its bare imports (`import trainer`,
    src/train.py:9) bound to ss_asr_amd.  Without a GPU the run must get all the way through
    argument parsing, yaml, seeding, ASRTrainer construction, load_data and set_model, and stop at
    the first arithmetic with the product's "no CPU path" error -- never with an ImportError from
    the reference's own modules (librosa, editdistance, tensorboardX are absent here)."""
    import subprocess
    import yaml
    root = str(tmp_path)
    index, _ = make_corpus(root, n=16, t_max=24, feat=80, seed=2)
    conf = {'asr': {'opt': {'type': 'Adadelta', 'learning_rate': 1.0},
                    'mdl': {'encoder_state_size': 32, 'mlp_out_size': 16, 'decoder_state_size': 32,
                            'tf_rate': 0.9, 'feature_dim': 80},
                    'train_index': index, 'valid_index': index, 'wer_step': 1, 'train_batch_size': 16,
                    'valid_batch_size': 16, 'n_epochs': 1, 'loader_jobs': 0},
            # config 5's first leg and the Seed loop (src/train.py:66-70: `Seed` -> trainer.asr_seed_train,
            # else getattr(trainer, type)): conf/default.yaml:42-53 in miniature
            'tae': {'opt': {'type': 'Adam', 'learning_rate': 0.0001},
                    'mdl': {'state_size': 32, 'emb_dim': 8, 'num_layers': 2}, 'drop_rate': 0.1,
                    'train_index': index, 'valid_index': index, 'train_batch_size': 16, 'valid_batch_size': 16,
                    'n_epochs': 1, 'loader_jobs': 0},
            # the other two legs (conf/default.yaml:23-40, :62-82 in miniature; 24 frames leave [1, 40] for the last pooling)
            'adv': {'G_opt': {'type': 'Adadelta', 'learning_rate': 1.0}, 'D_opt': {'type': 'Adadelta', 'learning_rate': 1.0},
                    'mdl': {'hidden_dim': 16}, 'label_smoothing': 0.1, 'train_index': index, 'valid_index': index,
                    'train_batch_size': 16, 'valid_batch_size': 16, 'n_epochs': 1, 'loader_jobs': 0},
            'sae': {'opt': {'type': 'Adam', 'learning_rate': 0.0001},
                    'mdl': {'kernel_sizes': [[1, 36], [5, 1], [3, 1]], 'num_filters': [8, 8, 16],
                            'pool_kernel_sizes': [[3, 1], [1, 1], [2, 40]]},
                    'train_index': index, 'valid_index': index, 'train_batch_size': 16, 'valid_batch_size': 16,
                    'n_epochs': 1, 'loader_jobs': 0},
            'seed_train': {'super_its': 1}}
    conf_path = os.path.join(root, 'conf.yaml')
    with open(conf_path, 'w') as f:
        yaml.safe_dump(conf, f)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    res = subprocess.run([sys.executable, '-m', 'ss_asr_amd.run_reference', REFERENCE_TRAIN, kind, 'dropin',
                          conf_path, os.path.join(root, 'runs'), os.path.join(root, 'result')],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = res.stdout
    assert 'ModuleNotFoundError' not in out and 'ImportError' not in out, out[-2000:]
    if torch.cuda.is_available():
        assert res.returncode == 0, out[-2000:]
    else:
        assert res.returncode != 0 and 'no CPU path' in out, out[-2000:]
    # the trainer that ran was ours: it wrote its tracker where the reference's Solver would
    assert os.path.isdir(os.path.join(root, 'result', 'dropin'))


def test_flat_names_bind_the_product_modules():
    """Importing `trainer` / `asr` / `ASRDataset` the way the reference's scripts do, after `import ss_asr_amd.flat`,
    yields the product's modules (and the modules of the Seed loop's other legs)."""
    import subprocess
    code = ("import ss_asr_amd.flat; import trainer, asr, ASRDataset, discriminator, speech_autoencoder, text_autoencoder; "
            "import ss_asr_amd.trainer as t, ss_asr_amd.asr as a; "
            "assert trainer is t and asr is a and trainer.ASRTrainer is t.ASRTrainer and asr.Listener is a.Listener; "
            "assert hasattr(ASRDataset, 'load_asr_dataset') and hasattr(trainer, 'Solver') and hasattr(trainer, 'AdvTrainer'); "
            "assert hasattr(discriminator, 'Discriminator') and hasattr(speech_autoencoder, 'SpeechAutoEncoder'); print('bound')")
    res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=120)
    assert res.returncode == 0 and 'bound' in res.stdout, res.stdout[-2000:]


# ------------------------------------------------------- the Seed loop's extra modules: checkpoint compatibility ----
def test_discriminator_and_speech_autoencoder_state_dicts_are_the_references():
    """The parameter / buffer names of discriminator.Discriminator and speech_autoencoder.SpeechAutoEncoder against the
    names the REFERENCE's classes had when the trajectory fixtures were captured (`param_names` of
    adv_traj_full_b8 / sae_traj_full_b8 list the reference's own state_dict keys), in order, and their shapes against
    the oracle's modules: a checkpoint written by either side loads in the other (batch-norm buffers included)."""
    import las_oracle as lo
    from ss_asr_amd.discriminator import Discriminator
    from ss_asr_amd.speech_autoencoder import SpeechAutoEncoder
    gold = os.path.join(ROOT, 'tests', 'golden')
    adv = np.load(os.path.join(gold, 'adv_traj_full_b8.npz'))
    sae = np.load(os.path.join(gold, 'sae_traj_full_b8.npz'))
    d, d_ref = Discriminator(512, 256), lo.OracleDiscriminator(512, 256)
    assert sorted('disc.' + k for k in d.state_dict()) == [str(n) for n in adv['param_names'] if str(n).startswith('disc.')]
    cfg = ([list(map(int, k)) for k in sae['kernel_sizes']], [int(v) for v in sae['num_filters']],
           [list(map(int, k)) for k in sae['pool_kernel_sizes']])
    s, s_ref = SpeechAutoEncoder(512, 80, *cfg), lo.OracleSpeechAutoEncoder(512, 80, *cfg)
    assert sorted('sae.' + k for k in s.state_dict()) == [str(n) for n in sae['param_names'] if str(n).startswith('sae.')]
    for mine, ref in ((d, d_ref), (s, s_ref)):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
        mine.load_state_dict(b)                      # and back
        ref.load_state_dict(mine.state_dict())


def test_settle_collector_freezes_once_and_honours_its_switch(monkeypatch):
    """engine.settle_collector (called by every step object after the first train step): moves what is alive to the
    collector's permanent generation once per process; `again` repeats it; SSASR_GC_FREEZE=0 leaves the collector alone."""
    import gc
    from ss_asr_amd import engine
    gc.unfreeze()
    monkeypatch.setattr(engine, '_settled', False)
    monkeypatch.setenv('SSASR_GC_FREEZE', '0')
    engine.settle_collector()
    assert gc.get_freeze_count() == 0 and engine._settled is False
    monkeypatch.delenv('SSASR_GC_FREEZE')
    try:
        engine.settle_collector()
        n = gc.get_freeze_count()
        assert n > 1000 and engine._settled is True
        keep = [[i] for i in range(5000)]            # made after the freeze: tracked, not frozen
        engine.settle_collector()                    # a second call is a no-op ...
        assert gc.get_freeze_count() == n
        engine.settle_collector(again=True)          # ... unless asked for
        assert gc.get_freeze_count() >= n + 5000
        del keep
    finally:
        gc.unfreeze()
