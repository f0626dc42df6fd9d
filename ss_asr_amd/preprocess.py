"""Vocabulary constants and the padding helpers of the reference's offline
preprocessing (src/preprocess.py:17-33, :253-269) that the training path
imports.  The wav -> log-mel frontend itself lives in ss_asr_amd/frontend.py."""
import re

import numpy as np

CHARS = 'abcdefghijklmnoprstuvxy0123456789'
ICE_CHARS = 'áéíóúýæöþð'
SPECIAL_CHARS = ' .,?'
ALL_CHARS = CHARS + ICE_CHARS + SPECIAL_CHARS
SOS_TKN = '<'       # also the padding symbol of label rows (index 0)
EOS_TKN = '>'
UNK_TKN = '$'
TOKENS = SOS_TKN + EOS_TKN + UNK_TKN

N_DIMS = 40         # reference default number of mel bins (src/preprocess.py:30)
WIN_SIZE = 25       # ms
STRIDE = 10         # ms


def normalize_string(s, append_tokens=True):
    """src/preprocess.py:225-251: lower-case, collapse whitespace, map anything
    outside the alphabet to '$', wrap in '<' ... '>'.  Returns (text, length
    before mapping + 2)."""
    s = re.sub(r'\s+', ' ', s.lower())
    s_len = len(s) + 2
    s = re.sub(r"[^0-9{}]".format(CHARS + ICE_CHARS + SPECIAL_CHARS), UNK_TKN, s)
    if append_tokens:
        s = SOS_TKN + s + EOS_TKN
    return s, s_len


def zero_pad(fbank, max_len, n_dims=None):
    """src/preprocess.py:253-269, with the mel-bin count taken from the data
    instead of the hard-coded N_DIMS."""
    n_dims = fbank.shape[1] if n_dims is None else n_dims
    padded = np.zeros([max_len, n_dims])
    padded[:fbank.shape[0], :fbank.shape[1]] = fbank
    return padded


def sort_index(index, sort_key, sort_ascending=True, out_index=None):
    """src/preprocess.py:301-316: rewrites the index sorted by one column (in place unless `out_index`);
    same row order as the reference's pandas sort, ties included, same file bytes (tab separated, no header,
    minimal quoting, '\\n' line ends)."""
    import csv

    from .ASRDataset import COLUMNS, load_index, sort_rows
    rows = sort_rows(load_index(index), sort_key, sort_ascending)
    with open(out_index if out_index is not None else index, 'w', encoding='utf-8', newline='') as f:
        w = csv.writer(f, delimiter='\t', quoting=csv.QUOTE_MINIMAL, lineterminator='\n')
        for r in rows:
            w.writerow([r[c] for c in COLUMNS])
