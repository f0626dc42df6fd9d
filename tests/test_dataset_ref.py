"""SURVEY.md section 8 rows a16 / a17 / f2 against the REFERENCE run in the build container.

tests/golden/dataset_ref.npz was written by oracle/make_dataset_golden.py from the reference's own `ASRDataset`,
`Mapper`, `load_asr_dataset`, `prepare_x`, `prepare_y` (src/ASRDataset.py:25-340), `trim_eos`
(src/postprocess.py:62-72), `normalize_string` / `zero_pad` / `sort_index` (src/preprocess.py:225-269, :301-316)
and `TrackerHandler` (src/TrackerHandler.py:1-42) over the corpus of oracle/corpus_recipe.py; ref_small_asr.cpt
is a checkpoint the reference wrote.  This is integer / index / file-format work: everything is compared bit for
bit.  The tests rebuild the corpus from the recipe and run the PRODUCT's mirrors over it."""
import os

import numpy as np
import pytest
import torch

import corpus_recipe as cr
from conftest import GOLDEN


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(GOLDEN, 'dataset_ref.npz'), allow_pickle=False)


@pytest.fixture(scope='module')
def corpus(tmp_path_factory, fx):
    root = str(tmp_path_factory.mktemp('corpus'))
    assert list(fx['frames']) == cr.FRAMES and list(fx['texts']) == cr.TEXTS
    assert (int(fx['feat']), int(fx['seed'])) == (cr.FEAT, cr.SEED)
    return root, cr.write_corpus(root)


def same(a, b):
    """bit-equal arrays of the same dtype and shape"""
    a, b = np.asarray(a), np.asarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def test_loader_batches_equal_the_references(fx, corpus):
    from ss_asr_amd.ASRDataset import load_asr_dataset, prepare_x, prepare_y
    _, index = corpus
    mapper, ds, loader = load_asr_dataset(index, batch_size=8, n_jobs=0)
    assert len(ds) == int(fx['len']) == 2                      # 20 rows: the last 4 are dropped (:63)
    assert same(ds.batch_inds, fx['batch_inds'])
    assert (ds.get_feature_dim(), ds.get_char_dim(), ds.num_samples) == \
        (int(fx['feature_dim']), int(fx['char_dim']), int(fx['num_samples']))
    n = 0
    for b, (x, y) in enumerate(loader):
        assert same(x.numpy(), fx['b%d_x' % b]), b             # float64 [1, 8, 64, 12]: padded to the corpus maximum
        assert same(y.numpy(), fx['b%d_y' % b]), b             # float64 label rows padded with index 0
        px, x_lens = prepare_x(x)
        py, y_lens = prepare_y(y)
        assert same(px.numpy(), fx['b%d_px' % b]) and x_lens == list(fx['b%d_x_lens' % b])
        assert same(py.numpy(), fx['b%d_py' % b]) and y_lens == list(fx['b%d_y_lens' % b])
        assert all(type(v) is int for v in x_lens + y_lens)
        n += 1
    assert n == int(fx['n_batches_yielded'])
    assert [int(ds.get_framelength(i)) for i in range(ds.num_samples)] == list(fx['framelengths'])
    assert [ds.get_text(i) for i in range(ds.num_samples)] == list(fx['get_text'])
    assert same(ds.encode(ds.get_text(3)), fx['encode_3'])
    assert ds.decode(ds.encode(ds.get_text(3))) == str(fx['decode_3'][0])
    root = corpus[0]
    paths = [os.path.join(root, 'fbanks', 'u%03d.npy' % i) for i in (19, 0, 7)]
    assert same(ds.get_batched_fbanks_by_paths(paths), fx['by_paths'])


def test_batch_planning_for_other_batch_sizes(fx, corpus):
    from ss_asr_amd.ASRDataset import ASRDataset
    from ss_asr_amd.gpu_loader import plan_batches
    _, index = corpus
    for bs in (32, 20, 7):
        d = ASRDataset(index, bs)
        assert len(d) == int(fx['len_bs%d' % bs]) and same(d.batch_inds, fx['batch_inds_bs%d' % bs])
        assert plan_batches(d.num_samples, bs) == [int(v) for v in fx['batch_inds_bs%d' % bs][:-1]]
    d7 = ASRDataset(index, 7)
    x, y = d7[len(d7) - 1]
    assert same(y, fx['bs7_last_y']) and float(x.sum()) == float(fx['bs7_last_x_sum'])


@pytest.mark.parametrize('key,asc', [('unpadded_num_frames', True), ('unpadded_num_frames', False), ('s_len', True),
                                     ('s_len', False), ('normalized_text', True)])
def test_sort_key_orders_equal_pandas_sort_values(fx, corpus, key, asc):
    """src/ASRDataset.py:55-57: rows with equal keys come out in the order pandas' default sort leaves them."""
    from ss_asr_amd.ASRDataset import ASRDataset, prepare_x
    _, index = corpus
    d = ASRDataset(index, 6, sort_key=key, sort_ascending=asc)
    tag = 'sort_%s_%d' % (key, int(asc))
    assert [r['wav_fname'] for r in d._rows] == list(fx[tag])
    x, y = d[0]
    assert same(y, fx[tag + '_y0'])
    assert prepare_x(torch.from_numpy(x)[None])[1] == list(fx[tag + '_x0_lens'])


def test_text_only_and_noisy_label_batches(fx, corpus):
    """TAETrainer's loader (src/trainer.py:608-614): clean rows, and the noise model of :111-127 consuming
    numpy's global stream exactly as the reference does (no draw for '<' and '>')."""
    from ss_asr_amd.ASRDataset import load_asr_dataset
    _, index = corpus
    _, _, lt = load_asr_dataset(index, batch_size=8, n_jobs=0, text_only=True)
    for b, y in enumerate(lt):
        assert same(y.numpy(), fx['text_b%d' % b])
    np.random.seed(int(fx['noisy_np_seed']))
    _, dn, ln = load_asr_dataset(index, batch_size=8, n_jobs=0, text_only=True, drop_rate=float(fx['noisy_rate']))
    for b, (clean, noisy) in enumerate(ln):
        assert same(clean.numpy(), fx['noisy_b%d_clean' % b]) and same(noisy.numpy(), fx['noisy_b%d_noisy' % b])
    assert np.random.rand() == float(fx['noisy_next_rand'])
    np.random.seed(int(fx['get_text_drop_seed']))
    assert [dn.get_text(i, 0.5) for i in range(dn.num_samples)] == list(fx['get_text_drop'])


def test_mapper_and_trim_eos(fx):
    from ss_asr_amd.ASRDataset import Mapper
    from ss_asr_amd.postprocess import trim_eos
    m = Mapper()
    assert m.get_dim() == int(fx['mapper_dim'])
    assert ''.join(m.r_mapping[i] for i in range(m.get_dim())) == str(fx['mapper_chars'][0])
    seqs = [[int(v) for v in fx['seq%d' % k]] for k in range(int(fx['n_seqs']))]
    assert [m.translate(s) for s in seqs] == list(fx['translate'])
    assert [m.translate(torch.tensor(s, dtype=torch.long)) for s in seqs if s] == list(fx['translate_tensor'])
    assert [m.translate(np.array(s)) for s in seqs if s] == list(fx['translate_array'])
    assert m.translate(fx['b0_y'][0, 2]) == str(fx['translate_float_row'][0])
    for k, s in enumerate(seqs):
        got = trim_eos(s)
        assert got == [int(v) for v in fx['trim%d' % k]] and all(type(v) is int for v in got)
        assert trim_eos(torch.tensor(s, dtype=torch.long)) == got
    assert [m.ind_to_char(i) for i in (0, 1, 2, 3, 49)] == list(fx['ind_to_char'])
    assert [m.char_to_ind(c) for c in '<>$að?'] == list(fx['char_to_ind'])


def test_preprocess_text_padding_and_the_index_the_reference_writes(fx, corpus, tmp_path):
    from ss_asr_amd import preprocess as pre
    from ss_asr_amd.ASRDataset import load_index
    root, index = corpus
    raw = list(fx['normalize_in'])
    assert [pre.normalize_string(s)[0] for s in raw] == list(fx['normalize_out'])
    assert [pre.normalize_string(s)[1] for s in raw] == list(fx['normalize_len'])
    assert [pre.normalize_string(s, append_tokens=False)[0] for s in raw] == list(fx['normalize_bare'])
    assert same(pre.zero_pad(cr.utterance(19), 12), fx['zero_pad'])
    for key, asc in (('unpadded_num_frames', False), ('s_len', True)):
        # (1) the product's sort_index writes the reference's file, byte for byte
        dst = os.path.join(str(tmp_path), 'sorted_%s.tsv' % key)
        pre.sort_index(index, key, sort_ascending=asc, out_index=dst)
        ref_text = str(fx['sort_index_%s' % key][0])
        assert open(dst, encoding='utf-8').read().replace(root + os.sep, '') == ref_text
        # (2) the product's reader parses the file the REFERENCE wrote
        ref_file = os.path.join(str(tmp_path), 'ref_%s.tsv' % key)
        with open(ref_file, 'w', encoding='utf-8') as f:
            f.write(ref_text)
        rows = load_index(ref_file)
        assert [r['wav_fname'] for r in rows] == [line.split('\t')[5] for line in ref_text.splitlines()]
        by_name = {('u%03d.wav' % i): (t, n) for i, (t, n) in enumerate(zip(cr.TEXTS, cr.FRAMES))}
        assert all((r['normalized_text'], r['unpadded_num_frames']) == by_name[r['wav_fname']] for r in rows)
        assert all(type(r['s_len']) is int and r['s_len'] == len(r['normalized_text']) for r in rows)


def test_tracker_json_is_the_references_file_after_every_operation(fx, tmp_path):
    from ss_asr_amd.TrackerHandler import TrackerHandler
    path = os.path.join(str(tmp_path), 'tracker.json')
    files = []
    t = TrackerHandler(path, 'asr')
    files.append(open(path).read())
    t.do_step(); files.append(open(path).read())
    t.do_step(); files.append(open(path).read())
    t.set_best(3.25); files.append(open(path).read())
    t2 = TrackerHandler(path, 'tae')
    assert t2.step == 0 and t2.get_best() == 10000
    t2.do_step(); files.append(open(path).read())
    t3 = TrackerHandler(path, 'asr')
    assert [t3.step, t3.get_best()] == list(fx['tracker_resume'])
    t3.set_best(0.5); files.append(open(path).read())
    assert files == list(fx['tracker_files'])
    # and the product resumes from a file the reference wrote
    ref = os.path.join(str(tmp_path), 'ref_tracker.json')
    with open(ref, 'w') as f:
        f.write(str(fx['tracker_files'][3]))
    r = TrackerHandler(ref, 'asr')
    assert (r.step, r.get_best()) == (2, 3.25)


def test_a_checkpoint_the_reference_wrote_loads_into_the_product_model(fx, tmp_path):
    """`.cpt` = torch.save(model.state_dict()) (src/trainer.py:451, :545; read back at :164): the reference-written
    file loads strictly into the product's ASR, and what the product saves has the same keys, order, dtypes,
    shapes and values."""
    from ss_asr_amd.asr import ASR
    dims = [int(v) for v in fx['cpt_dims']]
    ref_sd = torch.load(os.path.join(GOLDEN, 'ref_small_asr.cpt'), weights_only=True)
    assert list(ref_sd) == list(fx['cpt_keys'])
    model = ASR(*dims, 1.0)
    assert list(model.state_dict()) == list(ref_sd)
    missing, unexpected = model.load_state_dict(ref_sd, strict=True)
    assert not missing and not unexpected
    out = os.path.join(str(tmp_path), 'asr.cpt')
    torch.save(model.state_dict(), out)
    back = torch.load(out, weights_only=True)
    assert list(back) == list(ref_sd)
    for k in ref_sd:
        assert back[k].dtype == ref_sd[k].dtype and back[k].shape == ref_sd[k].shape and torch.equal(back[k], ref_sd[k])


def test_oracle_reproduces_the_reference_logits_from_the_reference_checkpoint(fx):
    """The checker itself on this row: the oracle model, loaded from the reference-written checkpoint, gives the
    reference's logits on batch 1 of the corpus (greedy == teacher-forced here: eval mode, tf_rate 1)."""
    import las_oracle as lo
    dims = [int(v) for v in fx['cpt_dims']]
    ref = lo.OracleASR(*dims, 1.0)
    ref.load_state_dict(torch.load(os.path.join(GOLDEN, 'ref_small_asr.cpt'), weights_only=True), strict=True)
    x, y = torch.from_numpy(fx['b1_px']), torch.from_numpy(fx['b1_py'])
    ans_len = int(max(fx['b1_y_lens'])) - 1
    with torch.no_grad():
        enc_len, logits, att = ref(x, ans_len, teacher=y, state_len=[int(v) for v in fx['b1_x_lens']])
    assert list(enc_len) == list(fx['cpt_enc_len'])
    assert np.abs(logits.numpy() - fx['cpt_logits']).max() < 1e-5
    assert np.abs(att[0].numpy() - fx['cpt_att_row0']).max() < 2e-6
