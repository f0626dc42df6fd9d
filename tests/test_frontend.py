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
        assert got.shape == want.shape == (1 + len(y) // int(sr * 0.010), n_mels), name
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
