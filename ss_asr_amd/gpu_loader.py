"""Device-resident, bucketed batch loader for the ASR corpus.

Replaces, for training on the GPU, the chain ASRDataset.__getitem__ ->
DataLoader -> prepare_x / prepare_y (src/ASRDataset.py:206-226, :264-340): the
whole corpus is read once (index.tsv + one .npy per utterance), its unpadded
frames are kept back to back in ONE device tensor (a 10 h corpus is ~1.2 GB of
the 288 GB), and a batch is assembled on the GPU by ``ssasr_gather_batch``.
Frame and label lengths are known on the host from load time, so a training
step has no host<->device round trip for its inputs (the reference recovers
lengths with a row sum over the batch tensor every step, :311-315).

Batches are the reference's: consecutive groups of ``batch_size`` rows of the
index, whole batches only (:63); the index is expected to be sorted so that
every batch is in decreasing frame-length order (conf/README.md:16).
``bucket=True`` instead sorts the whole corpus by length first.

Yields ``(x [B, T, F] float32, x_lens list, y [B, L] int64, y_lens list)`` with
T = max(x_lens) rounded up to a multiple of 8 (the pyramid's time reduction),
the values ``prepare_x`` / ``prepare_y`` return for the same batch.
"""
import numpy as np
import torch

from .ASRDataset import load_index
from .preprocess import ALL_CHARS, SOS_TKN, TOKENS


def plan_batches(n_rows, batch_size):
    """Start row of every whole batch (src/ASRDataset.py:63)."""
    return list(range(0, n_rows - batch_size + 1, batch_size))


def rank_batches(n_batches, rank, world):
    """Batch indices rank `rank` of `world` runs in one epoch: r, r + world, ... over the full
    rounds of `world` batches.  The tail that does not fill a round is dropped, so every rank runs
    n_batches // world steps: a rank alone in the gradient all-reduce would wait for ever, and
    unequal step counts would let validation / checkpoints fire at different steps per rank."""
    return [k * world + rank for k in range(n_batches // world)]


class GpuResidentLoader:
    def __init__(self, tsv_file, batch_size, device, chars=TOKENS + ALL_CHARS, rank=0, world=1,
                 bucket=False, time_multiple=8):
        rows = load_index(tsv_file)
        if bucket:
            rows.sort(key=lambda r: r['unpadded_num_frames'], reverse=True)
        self.rows = rows
        char2idx = {c: i for i, c in enumerate(chars)}
        starts = plan_batches(len(rows), batch_size)
        used = rows[:starts[-1] + batch_size if starts else 0]
        kept = []
        for r in used:
            a = np.load(r['path_to_fbank']).astype(np.float32, copy=False)
            n = int((a.sum(-1) != 0).sum())              # prepare_x's definition of a frame
            kept.append(a[:n])
        labels = [[char2idx[c] for c in r['normalized_text']] for r in used]
        self._setup(kept, labels, batch_size, device, rank, world, time_multiple, char2idx[SOS_TKN])

    @classmethod
    def from_arrays(cls, utterances, labels, batch_size, device, rank=0, world=1, time_multiple=8, pad=0):
        """The same loader over in-memory data: `utterances` = unpadded float32 [n_i, F] arrays in
        index order, `labels` = their character-id rows ('<' ... '>').  Used by bench.py, whose
        corpus is synthetic: the timed step then assembles its batch exactly as ASRTrainer's does."""
        self = cls.__new__(cls)
        self.rows = None
        n = len(plan_batches(len(utterances), batch_size)) * batch_size
        self._setup([np.asarray(u, dtype=np.float32) for u in utterances[:n]], [list(l) for l in labels[:n]],
                    batch_size, device, rank, world, time_multiple, pad)
        return self

    @classmethod
    def from_waveforms(cls, waves, sample_rate, texts, batch_size, device, n_mels=None, chars=TOKENS + ALL_CHARS,
                       rank=0, world=1, time_multiple=8):
        """The corpus built from WAVEFORMS on the GPU: all utterances go through the log-mel frontend in ONE
        batched call (frontend.log_fbank_batch -> ssasr_logmel_batch; src/preprocess.py:187-208 is what it
        replaces) and the frames stay on the device -- no .npy round trip, no host copy of the features.  `waves`: 1-D float arrays
        (or tensors) at `sample_rate`; `texts`: the normalised transcripts WITHOUT the '<' '>' tokens.
        Utterances are ordered by decreasing frame count first (the order the reference's index must
        have inside a batch, conf/README.md:16), then cut into whole batches as everywhere else.
        (The frontend's parity with librosa 0.6.3 is unpinned: DESIGN.md 4.6.)"""
        from . import _lib
        from .frontend import frontend_constants, log_fbank_batch
        from .preprocess import EOS_TKN, N_DIMS
        n_mels = n_mels or N_DIMS
        char2idx = {c: i for i, c in enumerate(chars)}
        # frame counts are known before anything is computed (1 + samples // hop): order and cut first, then ONE
        # batched frontend call over the kept utterances in that order (frontend.log_fbank_batch: three
        # launches for the whole corpus); the features stay where the call wrote them -- utterance i's frames
        # at rows first[i] .., with a few unused rows between utterances -- and the loader's offsets point there
        n_fft, hop = frontend_constants(sample_rate, n_mels, torch.device(device))[:2]
        nfr = [int(_lib.load().ssasr_logmel_frames(int(torch.as_tensor(w).numel()), n_fft, hop)) for w in waves]
        order = sorted(range(len(waves)), key=lambda i: -nfr[i])
        n = len(plan_batches(len(order), batch_size)) * batch_size
        order = order[:n]
        feats, first, frames = log_fbank_batch([waves[i] for i in order], sample_rate, n_mels, device=device)
        self = cls.__new__(cls)
        self.rows = None
        labels = [[char2idx[SOS_TKN]] + [char2idx[c] for c in texts[i]] + [char2idx[EOS_TKN]] for i in order]
        self._setup(None, labels, batch_size, device, rank, world, time_multiple, char2idx[SOS_TKN],
                    resident=(feats, first, frames))
        return self

    def _setup(self, kept, labels, batch_size, device, rank, world, time_multiple, pad, resident=None):
        """kept: the utterances' unpadded [n_i, F] arrays (uploaded back to back), or None with
        resident = (frames tensor on the device, first row of every utterance, its frame count)."""
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('GpuResidentLoader keeps the corpus on the GPU (no CPU path)')
        self.batch_size = batch_size
        self.rank, self.world = rank, world
        self.time_multiple = time_multiple
        if resident is not None:
            self.frames, first, lens = resident
            lens = [int(v) for v in lens]
            self.starts = plan_batches(len(lens), batch_size)
            self.feature_dim = int(self.frames.shape[1])
            offs = np.asarray(list(first) + [0], dtype=np.int64)
        else:
            self.starts = plan_batches(len(kept), batch_size)
            lens = [int(a.shape[0]) for a in kept]
            self.feature_dim = kept[0].shape[1] if kept else 0
            offs = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=offs[1:])
            # one upload; batches are gathered from it
            self.frames = torch.from_numpy(np.concatenate(kept, axis=0)).to(self.device) if kept else None
        self.x_lens = lens
        self.offsets = torch.from_numpy(offs[:-1].copy()).to(self.device)
        self.lens_dev = torch.tensor(lens, dtype=torch.int32, device=self.device)
        # labels: padded with <sos> like ASRDataset.get_batched_texts
        self.y, self.y_lens = [], []
        for s in self.starts:
            enc = labels[s:s + batch_size]
            L = max(len(e) for e in enc)
            y = np.full((batch_size, L), pad, dtype=np.int64)
            for i, e in enumerate(enc):
                y[i, :len(e)] = e
            self.y.append(torch.from_numpy(y).to(self.device))
            self.y_lens.append([int(v) + 1 for v in (y != 0).sum(-1)])      # prepare_y

    def __len__(self):
        return len(self.starts)

    def bytes_resident(self):
        return 0 if self.frames is None else self.frames.numel() * 4

    def batch(self, b):
        from . import ops
        s = self.starts[b]
        lens = self.x_lens[s:s + self.batch_size]
        m = self.time_multiple
        T = (max(max(lens), 1) + m - 1) // m * m
        x = ops.gather_batch(self.frames, self.offsets[s:s + self.batch_size],
                             self.lens_dev[s:s + self.batch_size], T)
        return x, list(lens), self.y[b], list(self.y_lens[b])

    def __iter__(self):
        for b in rank_batches(len(self.starts), self.rank, self.world):
            yield (b,) + self.batch(b)
