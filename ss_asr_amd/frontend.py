"""log_fbank on the GPU: the frontend of src/preprocess.py:187-208.

The reference calls librosa 0.6.3 (`melspectrogram(y, sr, n_mels, n_fft=ws,
hop_length=st)` with ws = int(0.025 sr), st = int(0.010 sr), then
`log(S + eps)` and a transpose to [frames, mel]).  librosa is not vendored in
the reference and not installed here, so its documented defaults are restated:
centred STFT with reflect padding, periodic Hann window of n_fft samples,
power 2, Slaney mel filters (htk=False, area normalised), fmin 0, fmax sr/2.
The constant matrices (window, real DFT basis, mel filters) are built once per
(sample_rate, n_mels) on the host in float64 and kept on the device; all
per-sample arithmetic runs in ssasr_logmel (csrc/frontend.hip + the MFMA GEMM).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .preprocess import N_DIMS, STRIDE, WIN_SIZE

_constants = {}


def _hz_to_mel(f):
    """Slaney scale (librosa hz_to_mel, htk=False)."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep,
                    mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filters(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=1)."""
    nb = n_fft // 2 + 1
    fftfreqs = np.linspace(0.0, sr / 2.0, nb)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights


def frontend_constants(sample_rate, n_mels, device):
    """(n_fft, hop, window, dft_basis, mel_basis) on `device`, cached."""
    key = (int(sample_rate), int(n_mels), str(device))
    if key not in _constants:
        n_fft = int(sample_rate * 0.001 * WIN_SIZE)
        hop = int(sample_rate * 0.001 * STRIDE)
        nb = n_fft // 2 + 1
        kp, nbp = (n_fft + 3) // 4 * 4, (nb + 3) // 4 * 4
        k = np.arange(n_fft, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)            # periodic Hann
        ang = 2.0 * np.pi * np.outer(np.arange(nb, dtype=np.float64), k) / n_fft
        basis = np.zeros((2 * nb, kp))
        basis[:nb, :n_fft] = np.cos(ang)
        basis[nb:, :n_fft] = np.sin(ang)
        mel = np.zeros((n_mels, nbp))
        mel[:, :nb] = mel_filters(sample_rate, n_fft, n_mels)
        to = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
        _constants[key] = (n_fft, hop, to(window), to(basis), to(mel))
    return _constants[key]


def log_fbank(y, sample_rate, n_mels=N_DIMS):
    """[frames, n_mels] float32 log-mel filterbank of waveform `y` (numpy array
    or tensor, any device) on the GPU; frames = 1 + len(y) // hop."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError('ss_asr_amd.frontend needs an MI355X (no CPU path)')
    wav = torch.as_tensor(y, dtype=torch.float32)
    if not wav.is_cuda:
        wav = wav.cuda()
    wav = wav.contiguous().view(-1)
    dev = wav.device
    n_fft, hop, window, basis, mel = frontend_constants(sample_rate, n_mels, dev)
    n = wav.numel()
    frames = int(lib.ssasr_logmel_frames(n, hop))
    nb = n_fft // 2 + 1
    f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    ws_frames, ws_spec, ws_power = f(frames, basis.shape[1]), f(frames, 2 * nb), f(frames, mel.shape[1])
    out = f(frames, n_mels)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.ssasr_logmel(p(wav), n, n_fft, hop, n_mels, p(window), p(basis), p(mel), p(ws_frames),
                           p(ws_spec), p(ws_power), p(out),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'ssasr_logmel')
    return out
