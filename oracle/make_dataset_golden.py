"""Generates tests/golden/dataset_ref.npz and tests/golden/ref_small_asr.cpt by RUNNING THE REAL REFERENCE's
host-side data code on CPU (build container only; needs /root/reference, see oracle/ref_harness.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_dataset_golden.py

SURVEY.md section 8 rows a16 / a17 / f2 are integer / index / file-format work; their bar is bit-exact against
the reference.  What is captured, all of it from the reference's own functions run over the corpus of
oracle/corpus_recipe.py (written to a temporary directory):

* `load_asr_dataset` + `ASRDataset` (src/ASRDataset.py:25-226, :264-295): length, batch starts, every batch the
  DataLoader yields (fbanks in full, label rows in full, dtypes), the remainder drop, dataset-wide padding;
  `sort_key` orders (ties included); `text_only` batches; the noisy-label batches of `drop_rate > 0` under a
  fixed numpy seed (:111-127); accessors;
* `prepare_x` / `prepare_y` (:297-340) on those batches;
* `Mapper` (:228-262): its table, `translate` on sequences with and without '>', as lists / arrays / tensors;
* `postprocess.trim_eos` (src/postprocess.py:62-72);
* `preprocess.normalize_string`, `zero_pad`, `sort_index` (src/preprocess.py:225-269, :301-316): the index file
  the reference WRITES, byte for byte (paths made relative to the corpus root);
* `TrackerHandler` (src/TrackerHandler.py:1-42): the text of tracker.json after every operation;
* a checkpoint the reference writes, `torch.save(model.state_dict(), path)` (src/trainer.py:451, :545) of a
  seeded small `ASR`, with the reference's logits for a small batch: the `.cpt` format, pinned by a file.

The fixture is data only: arrays, strings the reference produced, the recipe.  `calc_err` is not captured (it
needs the `editdistance` package, absent here).
"""
import io
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import corpus_recipe as cr  # noqa: E402
from ref_harness import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def strs(seq):
    return np.array(list(seq), dtype=np.str_)


def main():
    asr_mod = import_reference()
    import ASRDataset as ds_mod                # reference modules (path set by import_reference)
    import postprocess as post_mod
    import preprocess as pre_mod
    import TrackerHandler as tr_mod
    for m in (ds_mod, post_mod, pre_mod, tr_mod):
        assert os.path.abspath(m.__file__).startswith('/root/reference/src'), m.__file__

    out = dict(frames=np.array(cr.FRAMES), feat=np.int64(cr.FEAT), seed=np.int64(cr.SEED), texts=strs(cr.TEXTS))
    with tempfile.TemporaryDirectory() as root:
        index = cr.write_corpus(root)

        # ---- rows a17 / a16: the loader's batches and prepare_x / prepare_y on them
        mapper, ds, loader = ds_mod.load_asr_dataset(index, batch_size=8, n_jobs=0)
        out['len'] = np.int64(len(ds))
        out['batch_inds'] = np.asarray(ds.batch_inds)
        out['feature_dim'] = np.int64(ds.get_feature_dim())
        out['char_dim'] = np.int64(ds.get_char_dim())
        out['num_samples'] = np.int64(ds.num_samples)
        for b, (x, y) in enumerate(loader):
            out['b%d_x' % b] = x.numpy()                                  # [1, 8, 64, 12] float64
            out['b%d_y' % b] = y.numpy()                                  # [1, 8, L] float64
            px, x_lens = ds_mod.prepare_x(x)
            py, y_lens = ds_mod.prepare_y(y)
            assert px.dtype == torch.float32 and py.dtype == torch.int64
            out['b%d_px' % b], out['b%d_x_lens' % b] = px.numpy(), np.array(x_lens)
            out['b%d_py' % b], out['b%d_y_lens' % b] = py.numpy(), np.array(y_lens)
        out['n_batches_yielded'] = np.int64(b + 1)
        out['framelengths'] = np.array([int(ds.get_framelength(i)) for i in range(ds.num_samples)])
        out['get_text'] = strs(ds.get_text(i) for i in range(ds.num_samples))
        out['encode_3'] = ds.encode(ds.get_text(3))
        out['decode_3'] = strs([ds.decode(ds.encode(ds.get_text(3)))])
        out['by_paths'] = ds.get_batched_fbanks_by_paths(
            [os.path.join(root, 'fbanks', 'u%03d.npy' % i) for i in (19, 0, 7)])
        # a batch size that does not divide the corpus and leaves no whole batch / exactly one
        for bs in (32, 20, 7):
            d = ds_mod.ASRDataset(index, bs)
            out['len_bs%d' % bs] = np.int64(len(d))
            out['batch_inds_bs%d' % bs] = np.asarray(d.batch_inds)
        d7 = ds_mod.ASRDataset(index, 7)
        out['bs7_last_y'] = d7[len(d7) - 1][1]
        out['bs7_last_x_sum'] = np.float64(d7[len(d7) - 1][0].sum())

        # ---- sort_key orders (pandas sort_values, ties included): the row order as wav names
        for key, asc in (('unpadded_num_frames', True), ('unpadded_num_frames', False), ('s_len', True),
                         ('s_len', False), ('normalized_text', True)):
            d = ds_mod.ASRDataset(index, 6, sort_key=key, sort_ascending=asc)
            tag = 'sort_%s_%d' % (key, int(asc))
            out[tag] = strs(d._frame['wav_fname'].tolist())
            out[tag + '_y0'] = d[0][1]
            out[tag + '_x0_lens'] = np.array(ds_mod.prepare_x(torch.from_numpy(d[0][0])[None])[1])

        # ---- text_only and the noise model (TAETrainer's loader, src/trainer.py:608-614)
        _, dt, lt = ds_mod.load_asr_dataset(index, batch_size=8, n_jobs=0, text_only=True)
        for b, y in enumerate(lt):
            out['text_b%d' % b] = y.numpy()
        np.random.seed(7)
        _, dn, ln = ds_mod.load_asr_dataset(index, batch_size=8, n_jobs=0, text_only=True, drop_rate=0.3)
        for b, (clean, noisy) in enumerate(ln):
            out['noisy_b%d_clean' % b], out['noisy_b%d_noisy' % b] = clean.numpy(), noisy.numpy()
        out['noisy_np_seed'] = np.int64(7)
        out['noisy_rate'] = np.float64(0.3)
        out['noisy_next_rand'] = np.float64(np.random.rand())            # the stream position afterwards
        np.random.seed(11)
        out['get_text_drop'] = strs(dn.get_text(i, 0.5) for i in range(dn.num_samples))
        out['get_text_drop_seed'] = np.int64(11)

        # ---- Mapper and trim_eos
        out['mapper_chars'] = strs([''.join(mapper.r_mapping[i] for i in range(mapper.get_dim()))])
        out['mapper_dim'] = np.int64(mapper.get_dim())
        seqs = [[0, 13, 5, 1, 7, 7], [0, 4, 4, 36, 4], [1], [0, 0, 2, 45, 46, 1, 1], [], [49, 48, 47, 3, 1]]
        out['translate'] = strs(mapper.translate(s) for s in seqs)
        out['translate_tensor'] = strs(mapper.translate(torch.tensor(s, dtype=torch.long)) for s in seqs if s)
        out['translate_array'] = strs(mapper.translate(np.array(s)) for s in seqs if s)
        out['translate_float_row'] = strs([mapper.translate(out['b0_y'][0, 2])])      # a float64 label row
        for k, s in enumerate(seqs):
            out['seq%d' % k] = np.array(s, dtype=np.int64)
            out['trim%d' % k] = np.array(post_mod.trim_eos(s), dtype=np.int64)
        out['n_seqs'] = np.int64(len(seqs))
        out['ind_to_char'] = strs(mapper.ind_to_char(i) for i in (0, 1, 2, 3, 49))
        out['char_to_ind'] = np.array([mapper.char_to_ind(c) for c in '<>$að?'])

        # ---- preprocess: normalize_string, zero_pad, sort_index (the index file the reference writes)
        raw = ['Halló  Heimur!', 'ÞETTA er\tpróf 12', ' a\nb ', 'wqz', '']
        norm = [pre_mod.normalize_string(s) for s in raw]
        out['normalize_in'] = strs(raw)
        out['normalize_out'] = strs(n[0] for n in norm)
        out['normalize_len'] = np.array([n[1] for n in norm])
        out['normalize_bare'] = strs(pre_mod.normalize_string(s, append_tokens=False)[0] for s in raw)
        pre_mod.N_DIMS = cr.FEAT            # the reference pads to its module constant (src/preprocess.py:30, :267)
        out['zero_pad'] = pre_mod.zero_pad(cr.utterance(19), 12)
        for key, asc in (('unpadded_num_frames', False), ('s_len', True)):
            dst = os.path.join(root, 'sorted_%s.tsv' % key)
            pre_mod.sort_index(index, key, sort_ascending=asc, out_index=dst)
            text = open(dst, encoding='utf-8').read().replace(root + os.sep, '')
            out['sort_index_%s' % key] = strs([text])

        # ---- TrackerHandler: tracker.json after every operation
        tpath = os.path.join(root, 'tracker.json')
        files = []
        t = tr_mod.TrackerHandler(tpath, 'asr')
        files.append(open(tpath).read())                     # created as "{}": nothing saved yet
        t.do_step(); files.append(open(tpath).read())
        t.do_step(); files.append(open(tpath).read())
        t.set_best(3.25); files.append(open(tpath).read())
        t2 = tr_mod.TrackerHandler(tpath, 'tae')             # a second module on the same file
        assert t2.step == 0 and t2.get_best() == 10000
        t2.do_step(); files.append(open(tpath).read())
        t3 = tr_mod.TrackerHandler(tpath, 'asr')             # resume
        out['tracker_resume'] = np.array([t3.step, t3.get_best()], dtype=np.float64)
        t3.set_best(0.5); files.append(open(tpath).read())
        out['tracker_files'] = strs(files)

    # ---- a checkpoint as the reference writes it (src/trainer.py:451, :545) + its logits on a small batch
    dims = (50, 32, 32, 16, 12)
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    model = asr_mod.ASR(*dims, 1.0)
    cpt = os.path.join(OUT, 'ref_small_asr.cpt')
    torch.save(model.state_dict(), cpt)
    x = torch.from_numpy(out['b1_px'])                       # batch 1 of the corpus: 60 .. 17 frames of 64
    y = torch.from_numpy(out['b1_py'])
    x_lens = [int(v) for v in out['b1_x_lens']]
    ans_len = int(max(out['b1_y_lens'])) - 1
    model.eval()
    with torch.no_grad():
        enc_len, logits, att = model(x, ans_len, teacher=y, state_len=x_lens)
    out['cpt_dims'] = np.array(dims)
    out['cpt_keys'] = strs(model.state_dict().keys())
    out['cpt_logits'] = logits.numpy()
    out['cpt_enc_len'] = np.array(enc_len)
    out['cpt_att_row0'] = att[0].numpy()

    path = os.path.join(OUT, 'dataset_ref.npz')
    np.savez_compressed(path, **out)
    print('%s: %d entries, %.1f KB; %s %.1f KB' % (os.path.basename(path), len(out), os.path.getsize(path) / 1024,
                                                   os.path.basename(cpt), os.path.getsize(cpt) / 1024))


if __name__ == '__main__':
    main()
