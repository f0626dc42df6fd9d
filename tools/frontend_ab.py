import json, os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo') else os.getcwd())
import torch, bench
from ss_asr_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
for wide in (1, 0):
    lib.ssasr_set_option(b'SSASR_GEMM_WIDE', wide)
    for sr in (16000, 22050, 16000, 22050):
        r = bench.frontend_roofline(dev, sr=sr)
        print('wide', wide, 'sr', sr, 'us_per_batch', r['us_per_batch'], 'TF', r['achieved'], 'utt/s', r['utterances_per_sec'], r['shape']['n_fft'], r['shape']['rows'])
