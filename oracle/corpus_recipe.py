"""Test infrastructure: the synthetic on-disk corpus behind tests/golden/dataset_ref.npz.

`oracle/make_dataset_golden.py` writes this corpus to a temporary directory and runs the REFERENCE's
`ASRDataset` / `Mapper` / `load_asr_dataset` / `prepare_x` / `prepare_y` (src/ASRDataset.py:25-340) over it;
the tests write the same corpus from the recipe the fixture carries and run the product's mirror over it.
The recipe is data: frame counts, transcripts, a seed.  Layout on disk is the reference's (src/preprocess.py:
47-59): one float64 .npy per utterance, zero-padded on the time axis to the corpus maximum, and a 6-column
tab-separated index without a header (src/ASRDataset.py:13-23).
"""
import os

import numpy as np

# 20 utterances: batches of 8 leave a remainder of 4 (src/ASRDataset.py:63); lengths are descending inside
# every batch of 8 (conf/README.md:16), one is odd, two are equal (a tie for the sort keys), the last batch's
# longest utterance is shorter than the corpus maximum (dataset-wide padding past the batch maximum)
FRAMES = [64, 61, 61, 56, 50, 47, 40, 33,
          60, 58, 52, 52, 44, 31, 24, 17,
          48, 48, 30, 9]
FEAT = 12
SEED = 20261005
TEXTS = [
    '<halló heimur>', '<þetta er prófun, já>', '<góðan daginn.>', '<hvað er klukkan?>',
    '<ég á 3 ketti>', '<æ, ó og ú>', '<$ óþekkt tákn $>', '<stutt>',
    '<ýmislegt að gera í dag>', '<veðrið er gott>', '<0123456789>', '<íslenska er falleg>',
    '<sjö, átta, níu>', '<já>', '<nei.>', '<x>',
    '<tvö orð>', '<eitt tvö þrjú fjögur>', '<ö>', '<.>',
]


def utterance(i, frames=FRAMES, feat=FEAT, seed=SEED):
    """Unpadded float32-valued frames of utterance i, [frames[i], feat]; no row sums to zero."""
    rng = np.random.default_rng(seed + i)
    return rng.standard_normal((frames[i], feat)).astype(np.float32)


def write_corpus(root, frames=FRAMES, texts=TEXTS, feat=FEAT, seed=SEED, name='index.tsv'):
    """Writes fbanks/u%03d.npy + index.tsv under `root`; returns the index path."""
    fdir = os.path.join(root, 'fbanks')
    os.makedirs(fdir, exist_ok=True)
    t_max = max(frames)
    lines = []
    for i, (n, text) in enumerate(zip(frames, texts)):
        padded = np.zeros([t_max, feat])                      # float64, as src/preprocess.py:267 leaves it
        padded[:n] = utterance(i, frames, feat, seed)
        path = os.path.join(fdir, 'u%03d.npy' % i)
        np.save(path, padded)
        # s_len = characters before the tokens were added + 2 (src/preprocess.py:225-251)
        lines.append('\t'.join([text, path, str(len(text)), str(n), 'u%03d.txt' % i, 'u%03d.wav' % i]))
    index = os.path.join(root, name)
    with open(index, 'w', encoding='utf-8') as f:
        f.write('\n'.join(lines) + '\n')
    return index
