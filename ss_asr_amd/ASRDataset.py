"""Index / fbank loading with the surface of src/ASRDataset.py: the 6-column
tab-separated index (:13-23), one .npy per utterance, a Dataset whose items
are whole batches (:206-226), Mapper (:228-262), load_asr_dataset (:264-295),
prepare_x / prepare_y (:297-340)."""
import csv

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .postprocess import trim_eos
from .preprocess import ALL_CHARS, EOS_TKN, SOS_TKN, TOKENS

COLUMNS = ['normalized_text', 'path_to_fbank', 's_len', 'unpadded_num_frames', 'text_fname',
           'wav_fname']


def load_index(path):
    """Rows of the index as dicts (ints converted).  Field quoting follows the reference's reader, pandas'
    read_csv defaults (src/ASRDataset.py:13-23): minimal quoting with '"', blank lines skipped."""
    rows = []
    with open(path, 'r', encoding='utf-8', newline='') as f:
        for rec in csv.reader(f, delimiter='\t', quoting=csv.QUOTE_MINIMAL):
            if not rec:
                continue
            row = dict(zip(COLUMNS, rec))
            row['s_len'] = int(row['s_len'])
            row['unpadded_num_frames'] = int(row['unpadded_num_frames'])
            rows.append(row)
    return rows


def sort_rows(rows, key, ascending=True):
    """The row order of `DataFrame.sort_values(by=[key], ascending=...)` (src/ASRDataset.py:55-57,
    src/preprocess.py:314), ties included: pandas sorts one column with numpy's default argsort (not a stable
    sort) and, for a descending order, sorts the reversed column and reverses the result."""
    vals = [r[key] for r in rows]
    col = np.array(vals, dtype=np.int64) if isinstance(vals[0], int) else np.array(vals, dtype=object)
    idx = np.arange(len(rows))
    if not ascending:
        col, idx = col[::-1], idx[::-1]
    order = idx[col.argsort(kind='quicksort')]
    if not ascending:
        order = order[::-1]
    return [rows[int(i)] for i in order]


class ASRDataset(Dataset):
    def __init__(self, tsv_file, batch_size=32, chars=TOKENS + ALL_CHARS, text_only=False,
                 sort_key='', sort_ascending=True, drop_rate=0.0):
        self.text_only = text_only
        self.chars = chars
        self.char2idx_dict = {c: i for i, c in enumerate(chars)}
        self.idx2char_dict = {i: c for i, c in enumerate(chars)}
        self._rows = load_index(tsv_file)
        if sort_key:
            self._rows = sort_rows(self._rows, sort_key, sort_ascending)
        self._feature_dim = self.get_fbank(0).shape[1]
        self.batch_size = batch_size
        self.num_samples = len(self._rows)
        # whole batches only: the remainder is dropped (src/ASRDataset.py:63)
        self.batch_inds = np.arange(0, self.num_samples + 1, self.batch_size)
        self.drop_rate = drop_rate

    def char2idx(self, char):
        return self.char2idx_dict[char]

    def idx2char(self, idx):
        return self.idx2char_dict[idx]

    def get_fbank(self, idx):
        return np.load(self._rows[idx]['path_to_fbank'])

    def get_fbank_by_path(self, path):
        return np.load(path)

    def _batch_range(self, start):
        return range(start, min(start + self.batch_size, self.num_samples))

    def get_batched_fbanks(self, start_idx):
        return np.stack([self.get_fbank(i) for i in self._batch_range(start_idx)], axis=0)

    def get_batched_fbanks_by_paths(self, paths):
        return np.stack([self.get_fbank_by_path(p) for p in paths])

    def get_text(self, idx, drop_rate=0.0):
        text = self._rows[idx]['normalized_text']
        if drop_rate > 0:
            return ''.join(c for c in text
                           if c in (EOS_TKN, SOS_TKN) or np.random.rand() > drop_rate)
        return text

    def get_batched_texts(self, start_idx, pad_token=SOS_TKN, drop_rate=0.0):
        enc = [self.encode(self.get_text(i, drop_rate)) for i in self._batch_range(start_idx)]
        out = np.zeros([self.batch_size, max(e.shape[0] for e in enc)]) + self.char2idx(pad_token)
        for i, e in enumerate(enc):
            out[i, :e.shape[0]] = e
        return out

    def encode(self, text):
        return np.array([self.char2idx(c) for c in text])

    def decode(self, inds):
        return ''.join(self.idx2char(int(i)) for i in inds)

    def get_framelength(self, idx):
        return self._rows[idx]['unpadded_num_frames']

    def get_batched_framelengths(self, start_idx):
        return [self.get_framelength(i) for i in self._batch_range(start_idx)]

    def get_feature_dim(self):
        return self._feature_dim

    def get_char_dim(self):
        return len(self.chars)

    def __len__(self):
        return len(self.batch_inds) - 1

    def __getitem__(self, idx):
        start = self.batch_inds[idx]
        if self.text_only:
            if self.drop_rate > 0:
                return (self.get_batched_texts(start),
                        self.get_batched_texts(start, drop_rate=self.drop_rate))
            return self.get_batched_texts(start)
        return self.get_batched_fbanks(start), self.get_batched_texts(start)


class Mapper:
    """Index <-> character translation (src/ASRDataset.py:228-262)."""

    def __init__(self, tokens=TOKENS + ALL_CHARS):
        self.mapping = {c: i for i, c in enumerate(tokens)}
        self.r_mapping = {i: c for c, i in self.mapping.items()}

    def get_dim(self):
        return len(self.mapping)

    def translate(self, seq):
        text = ''.join(self.r_mapping[c] for c in trim_eos(seq))
        return text.replace(SOS_TKN, '').replace(EOS_TKN, '')

    def ind_to_char(self, ind):
        return self.r_mapping[ind]

    def char_to_ind(self, char):
        return self.mapping[char]


def load_asr_dataset(path, batch_size=1, n_jobs=8, text_only=False, use_gpu=False, sort_key='',
                     sort_ascending=True, drop_rate=0.0):
    """-> (Mapper, ASRDataset, DataLoader); the loader's items are whole batches
    with a leading axis of 1 (src/ASRDataset.py:291-295)."""
    dataset = ASRDataset(path, batch_size, text_only=text_only, sort_key=sort_key,
                         sort_ascending=sort_ascending, drop_rate=drop_rate)
    return Mapper(), dataset, DataLoader(dataset, batch_size=1, num_workers=n_jobs,
                                         pin_memory=use_gpu)


def prepare_x(x, device=torch.device('cpu')):
    """[1, B, T, F] -> ([B, T, F] float32 on `device`, list of unpadded frame
    counts).  A frame counts when its feature sum is non-zero
    (src/ASRDataset.py:311-315); on the GPU the count runs as a HIP kernel and
    only B integers come back instead of the whole tensor."""
    x = x.squeeze(0).to(device=device, dtype=torch.float32)
    if x.is_cuda:
        from . import ops
        x_lens = ops.frame_lengths(x).cpu().tolist()
    else:
        x_lens = [int(v) for v in (x.sum(-1) != 0).sum(-1)]
    return x, x_lens


def prepare_y(y, device=torch.device('cpu')):
    """[1, B, L] -> ([B, L] int64 on `device`, label lengths = count(y != 0) + 1;
    src/ASRDataset.py:333-340).  Lengths are taken on the host copy so that no
    device round trip is needed."""
    y = y.squeeze(0)
    y_host = y if not y.is_cuda else y.cpu()
    y_lens = [int(v) + 1 for v in (y_host != 0).sum(-1)]
    return y.to(device=device, dtype=torch.long), y_lens
