"""Throughput of the log-mel frontend, both forms: bench.frontend_roofline -- 32 utterances of
4.5 s at 16 kHz (the corpus mean, src/preprocess.py:318), 25 ms / 10 ms frames, 80 mels, waveforms resident on
the GPU: ONE ssasr_logmel_batch call against 32 ssasr_logmel calls.  (Accuracy against the CPU restatement is
tests/test_frontend.py's job.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
print(json.dumps(bench.frontend_roofline(torch.device('cuda', 0)), indent=1))
