"""CPU oracle for the log-mel frontend  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's log_fbank (src/preprocess.py:187-208) calls
librosa 0.6.3, which is neither vendored in the reference nor installed in
this image, and the reference holds no sample fbanks.  This file restates
librosa's published algorithm for melspectrogram with the arguments the
reference passes (n_fft = int(0.025 sr), hop = int(0.010 sr), defaults
otherwise: center=True, pad_mode='reflect', periodic Hann, power 2, Slaney mel,
norm=1, fmin=0, fmax=sr/2) in float64 numpy, written independently of
ss_asr_amd/frontend.py (FFT-based, loop-built filters)."""
import numpy as np


def hz_to_mel(f):
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 1000.0 / f_sp + np.log(f / 1000.0) / (np.log(6.4) / 27.0)
    return f / f_sp


def mel_to_hz(m):
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    if m >= min_log_mel:
        return 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - min_log_mel))
    return f_sp * m


def mel_basis(sr, n_fft, n_mels):
    nb = n_fft // 2 + 1
    freqs = [i * (sr / 2.0) / (nb - 1) for i in range(nb)]
    lo, hi = hz_to_mel(0.0), hz_to_mel(sr / 2.0)
    pts = [mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    w = np.zeros((n_mels, nb))
    for m in range(n_mels):
        left, centre, right = pts[m], pts[m + 1], pts[m + 2]
        for k, f in enumerate(freqs):
            up = (f - left) / (centre - left)
            down = (right - f) / (right - centre)
            w[m, k] = max(0.0, min(up, down)) * 2.0 / (right - left)
    return w


def log_fbank(y, sr, n_mels):
    """float64 [frames, n_mels]; src/preprocess.py:194-206."""
    y = np.asarray(y, dtype=np.float64)
    n_fft, hop = int(sr * 0.025), int(sr * 0.010)
    padded = np.pad(y, n_fft // 2, mode='reflect')
    frames = 1 + (len(padded) - n_fft) // hop
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    spec = np.empty((frames, n_fft // 2 + 1))
    for f in range(frames):
        seg = padded[f * hop:f * hop + n_fft] * window
        spec[f] = np.abs(np.fft.rfft(seg)) ** 2
    mel = spec @ mel_basis(sr, n_fft, n_mels).T
    return np.log(mel + np.finfo(float).eps)
