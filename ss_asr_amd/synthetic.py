"""Synthetic Malromur-shaped corpora: zero-padded
N(0,1) log-mel frames and '<' chars '>' label rows, bucketed by length so that
every batch is sorted by decreasing frame count, which is what the reference's
index order guarantees (conf/README.md:16, src/asr.py:413)."""
import numpy as np
import torch

VOCAB = 50          # len(TOKENS + ALL_CHARS), src/preprocess.py:17-27


def corpus_lengths(n_utts, seed, lo=100, hi=800, mean=450.0, sigma=0.4):
    rng = np.random.default_rng(seed)
    frames = np.clip(rng.lognormal(np.log(mean), sigma, size=n_utts), lo, hi).astype(np.int64)
    chars = np.clip(frames // 10, 5, 80).astype(np.int64)
    order = np.argsort(-frames, kind='stable')
    return frames[order], chars[order]


def make_batch(frames, chars, feat_dim, seed, device='cpu', pad_to=None):
    """frames / chars: per-utterance lengths, frames sorted descending.
    Returns x [B, T, F] float32 (zero rows past each length), y [B, L] int64
    (0 = '<' = pad, 1 = '>'), lens list[int]."""
    rng = np.random.default_rng(seed)
    b = len(frames)
    t = int(pad_to or frames[0])
    x = rng.standard_normal((b, t, feat_dim), dtype=np.float32)
    for i, l in enumerate(frames):
        x[i, int(l):] = 0
    width = int(chars.max()) + 2
    y = np.zeros((b, width), dtype=np.int64)
    for i, l in enumerate(chars):
        y[i, 1:1 + int(l)] = rng.integers(3, VOCAB, size=int(l))
        y[i, 1 + int(l)] = 1
    return (torch.from_numpy(x).to(device), torch.from_numpy(y).to(device),
            [int(v) for v in frames])


def config2_batches(n_batches, batch_size=32, feat_dim=80, n_utts=8000, seed=1, device='cpu',
                    rank=0, hi=800):
    """`n_batches` batches spread evenly over the length-sorted ~10 h corpus of
    BASELINE.json configs[1] (8,000 utterances, <= 800 frames).  Ranks of a data-parallel run
    hold replicas of that corpus (configs[2]: "same corpus replicated x8"): the same utterance
    lengths and bucket picks on every rank -- so that no rank waits in the gradient all-reduce
    for one that drew longer utterances -- with rank-specific frame and label values."""
    frames, chars = corpus_lengths(n_utts, seed, hi=hi, lo=min(100, hi),
                                   mean=min(450.0, hi * 0.5625))
    total = n_utts // batch_size
    picks = np.linspace(0, total - 1, n_batches).astype(int)
    out = []
    for j, bi in enumerate(picks):
        sl = slice(bi * batch_size, (bi + 1) * batch_size)
        out.append(make_batch(frames[sl], chars[sl], feat_dim, seed * 1000 + j + 100000 * rank,
                              device))
    return out


def config4_batch(batch_size=32, feat_dim=80, seed=4, lo=1500, hi=3000):
    """One batch of BASELINE.json configs[3]'s shape: frame lengths U{lo..hi} sorted
    descending with the longest at `hi`, characters U{lo/10..hi/10}."""
    rng = np.random.default_rng(seed)
    frames = np.sort(rng.integers(lo, hi + 1, size=batch_size))[::-1].copy()
    frames[0] = hi
    chars = rng.integers(lo // 10, hi // 10 + 1, size=batch_size)
    return make_batch(frames, chars, feat_dim, seed * 1000 + 7)
