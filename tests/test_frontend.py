"""Frontend (log-mel) tests.  The oracle for this kernel is a float64 numpy
restatement of the librosa algorithm the reference calls; parity with the
reference itself is UNPINNED (librosa absent, no sample fbanks in the
reference) -- see oracle/frontend_oracle.py."""
import numpy as np
import pytest
import torch

import frontend_oracle as fo


def test_host_constants_match_the_independent_oracle():
    from ss_asr_amd.frontend import mel_filters
    for sr, n_mels in ((22050, 40), (22050, 80), (16000, 80)):
        n_fft = int(sr * 0.025)
        a, b = mel_filters(sr, n_fft, n_mels), fo.mel_basis(sr, n_fft, n_mels)
        assert a.shape == (n_mels, n_fft // 2 + 1)
        np.testing.assert_allclose(a, b, atol=1e-12)
        assert (a.sum(1) > 0).all()           # no empty filter at these sizes


def signals(sr):
    rng = np.random.default_rng(3)
    t = np.arange(int(0.73 * sr)) / sr
    chirp = np.sin(2 * np.pi * (200 + 3000 * t) * t) * 0.5
    return {'sine440': 0.8 * np.sin(2 * np.pi * 440 * t), 'chirp': chirp,
            'noise': rng.standard_normal(len(t)) * 0.1,
            'short': rng.standard_normal(300) * 0.3,           # shorter than one window
            'mix': 0.3 * np.sin(2 * np.pi * 1234.5 * t) + 0.01 * rng.standard_normal(len(t))}


@pytest.mark.gpu
@pytest.mark.parametrize('sr,n_mels', [(22050, 40), (22050, 80)])
def test_log_fbank_matches_oracle(sr, n_mels):
    from ss_asr_amd.frontend import log_fbank
    for name, y in signals(sr).items():
        want = fo.log_fbank(y, sr, n_mels)
        got = log_fbank(y.astype(np.float32), sr, n_mels).cpu().numpy()
        assert got.shape == want.shape, name            # (1 + (len + 2 (n_fft // 2) - n_fft) // hop frames)
        assert got.dtype == np.float32
        # fp32 DFT by GEMM against a float64 FFT: compare where the band energy is
        # above the fp32 noise floor of the frame; everywhere bounded in linear power
        ref_pow, got_pow = np.exp(want), np.exp(got.astype(np.float64))
        floor = 1e-5 * ref_pow.max(axis=1, keepdims=True)
        np.testing.assert_allclose(got_pow, ref_pow, rtol=2e-3, atol=float(floor.max()), err_msg=name)
        strong = ref_pow > 1e-3 * ref_pow.max()
        assert np.abs(got - want)[strong].max() < 2e-3, name


@pytest.mark.gpu
def test_log_fbank_feeds_the_listener_contract():
    """Output is [frames, mel] float32 with no all-zero frame for real audio
    (prepare_x counts frames by non-zero feature sums, src/ASRDataset.py:314)."""
    from ss_asr_amd.frontend import log_fbank
    y = signals(22050)['noise'].astype(np.float32)
    fb = log_fbank(y, 22050, 80)
    assert fb.dtype == torch.float32 and fb.is_cuda
    assert int((fb.sum(-1) != 0).sum()) == fb.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize('sr,n_mels', [(22050, 80), (16000, 80), (16000, 40)])
def test_batched_log_fbank_matches_the_oracle_and_the_per_utterance_form(sr, n_mels):
    """ssasr_logmel_batch (three launches for a LIST of waveforms: hop-aligned reflect layout, the DFT read from
    it as overlapping rows with the power formed in the product's epilogue, mel + log) against the float64
    oracle on every utterance of a ragged list -- incl. one shorter than a window, one of exactly k * hop
    samples and one of a single hop -- with the tolerances of the per-utterance test, and against the
    per-utterance kernel path; rows between utterances are ignored, rows of utterances do not see each other."""
    from ss_asr_amd.frontend import log_fbank, log_fbank_batch
    hop = int(sr * 0.010)
    sig = signals(sr)
    rng = np.random.default_rng(11)
    waves = [sig['sine440'], sig['short'], sig['chirp'], sig['noise'][:7 * hop], sig['mix'], rng.standard_normal(hop) * 0.2,
             sig['noise']]
    feats, first, frames = log_fbank_batch([w.astype(np.float32) for w in waves], sr, n_mels)
    assert feats.is_cuda and feats.dtype == torch.float32 and feats.shape[1] == n_mels
    for i, y in enumerate(waves):
        assert frames[i] == 1 + (len(y) + 2 * (int(sr * 0.025) // 2) - int(sr * 0.025)) // hop      # librosa's framing
        assert i == 0 or first[i] >= first[i - 1] + frames[i - 1]
        got = feats[first[i]:first[i] + frames[i]].cpu().numpy()
        want = fo.log_fbank(y, sr, n_mels)
        assert got.shape == want.shape, (i, got.shape, want.shape)
        ref_pow, got_pow = np.exp(want), np.exp(got.astype(np.float64))
        floor = 1e-5 * ref_pow.max(axis=1, keepdims=True)
        np.testing.assert_allclose(got_pow, ref_pow, rtol=2e-3, atol=float(floor.max()), err_msg=str(i))
        strong = ref_pow > 1e-3 * ref_pow.max()
        assert np.abs(got - want)[strong].max() < 2e-3, i
        single = log_fbank(y.astype(np.float32), sr, n_mels).cpu().numpy()
        assert np.abs(np.exp(got.astype(np.float64)) - np.exp(single.astype(np.float64))).max() <= 2e-3 * ref_pow.max() + float(floor.max())
    # the same utterance alone and inside a batch: bit-identical rows (no leakage between neighbours)
    alone, f0, n0 = log_fbank_batch([waves[2].astype(np.float32)], sr, n_mels)
    assert torch.equal(alone[f0[0]:f0[0] + n0[0]], feats[first[2]:first[2] + frames[2]])
