"""Throughput of the log-mel frontend: 4.5 s utterances at 16 kHz (the
corpus mean, src/preprocess.py:318), 25 ms / 10 ms frames, 80 mels, waveforms resident on the
GPU.  Prints utterances/s, the real-time factor and algorithmic flops.  (Accuracy against
the CPU restatement is tests/test_frontend.py's job.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ss_asr_amd import frontend

sr, secs, n_mels = 16000, 4.5, 80
rng = np.random.default_rng(0)
wavs = [torch.from_numpy(rng.standard_normal(int(sr * secs)).astype(np.float32)).cuda() for _ in range(16)]
for w in wavs[:4]:
    frontend.log_fbank(w, sr, n_mels)
torch.cuda.synchronize()
N = 400
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    out = frontend.log_fbank(wavs[i % 16], sr, n_mels)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
n_fft, hop = 400, 160
frames = 1 + int(sr * secs) // hop
nb = n_fft // 2 + 1
flops = 2.0 * frames * n_fft * 2 * nb + 2.0 * frames * nb * n_mels
print('GPU: %.3f ms per %.1f s utterance = %.0f utterances/s = %.0f x real time; %d frames; %.2f GFLOP -> %.1f TFLOP/s'
      % (ms, secs, 1e3 / ms, secs * 1e3 / ms, frames, flops / 1e9, flops / (ms * 1e-3) / 1e12))
