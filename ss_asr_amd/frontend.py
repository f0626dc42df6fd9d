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
    """(n_fft, hop, window, dft_basis, mel_basis, windowed interleaved dft_basis) on `device`, cached."""
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
        # the batched form's basis: the window folded in (float64, rounded once) and the cos / sin rows of a bin
        # interleaved, so that the DFT product's epilogue holds (re, im) of one bin in one lane
        basis_w = np.zeros((2 * nb, kp))
        basis_w[0::2, :n_fft] = np.cos(ang) * window[None, :]
        basis_w[1::2, :n_fft] = np.sin(ang) * window[None, :]
        to = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
        _constants[key] = (n_fft, hop, to(window), to(basis), to(mel), to(basis_w))
    return _constants[key]


def log_fbank(y, sample_rate, n_mels=N_DIMS):
    """[frames, n_mels] float32 log-mel filterbank of waveform `y` (numpy array
    or tensor, any device) on the GPU; frames = 1 + (len(y) + 2 * (n_fft // 2) - n_fft) // hop
    (librosa's centred framing: 1 + len(y) // hop for an even window)."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError('ss_asr_amd.frontend needs an MI355X (no CPU path)')
    wav = torch.as_tensor(y, dtype=torch.float32)
    if not wav.is_cuda:
        wav = wav.cuda()
    wav = wav.contiguous().view(-1)
    dev = wav.device
    n_fft, hop, window, basis, mel, _ = frontend_constants(sample_rate, n_mels, dev)
    n = wav.numel()
    frames = int(lib.ssasr_logmel_frames(n, n_fft, hop))
    nb = n_fft // 2 + 1
    f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    ws_frames, ws_spec, ws_power = f(frames, basis.shape[1]), f(frames, 2 * nb), f(frames, mel.shape[1])
    out = f(frames, n_mels)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.ssasr_logmel(p(wav), n, n_fft, hop, n_mels, p(window), p(basis), p(mel), p(ws_frames),
                           p(ws_spec), p(ws_power), p(out),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'ssasr_logmel')
    return out


def log_fbank_batch(waves, sample_rate, n_mels=N_DIMS, device=None):
    """log_fbank of a list of waveforms in ONE ssasr_logmel_batch call (three launches for the whole list).
    Returns (feats [total_rows, n_mels] float32 on the GPU, first_rows, frames): utterance i's features are
    feats[first_rows[i] : first_rows[i] + frames[i]] (frames[i] as for log_fbank); the rows between
    utterances belong to no frame.  The waveforms go up in one host->device copy (host arrays) or are
    concatenated on the device (tensors)."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError('ss_asr_amd.frontend needs an MI355X (no CPU path)')
    if len(waves) == 0:
        raise ValueError('log_fbank_batch: no waveforms')
    on_dev = all(torch.is_tensor(w) and w.is_cuda for w in waves)
    dev = waves[0].device if on_dev else torch.device(device or 'cuda')
    lens = [int(w.numel() if torch.is_tensor(w) else np.asarray(w).size) for w in waves]
    if min(lens) <= 0:
        raise ValueError('log_fbank_batch: empty waveform')
    n_fft, hop, _, _, mel, basis_w = frontend_constants(sample_rate, n_mels, dev)
    rows = [int(lib.ssasr_logmel_batch_rows(n, n_fft, hop)) for n in lens]
    frames = [int(lib.ssasr_logmel_frames(n, n_fft, hop)) for n in lens]
    first, offs = np.zeros(len(lens), dtype=np.int64), np.zeros(len(lens), dtype=np.int64)
    np.cumsum(rows[:-1], out=first[1:])
    np.cumsum(lens[:-1], out=offs[1:])
    total_rows = int(sum(rows))
    if on_dev:
        wav = torch.cat([w.reshape(-1).to(torch.float32) for w in waves])
    else:
        host = torch.empty(sum(lens), dtype=torch.float32, pin_memory=True)
        for o, n, w in zip(offs, lens, waves):
            host[o:o + n] = torch.as_tensor(np.asarray(w, dtype=np.float32).reshape(-1)) if not torch.is_tensor(w) else w.reshape(-1).float()
        wav = host.to(dev, non_blocking=True)
    utt = torch.from_numpy(np.stack([offs, np.asarray(lens, dtype=np.int64), first], axis=1).copy()).to(dev)
    f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    ws_wave, ws_power = f(total_rows * hop + n_fft), f(total_rows, mel.shape[1])
    out = f(total_rows, n_mels)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.ssasr_logmel_batch(p(wav), p(utt), len(lens), max(lens), total_rows, n_fft, hop, n_mels, p(basis_w), p(mel),
                                 p(ws_wave), p(ws_power), p(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
          'ssasr_logmel_batch')
    return out, [int(v) for v in first], frames
