"""Standalone attention kernel (energy + masked softmax + context) at several encoder lengths:
HIP-graph replay timed with HIP events, algorithmic bytes / time against the 8 TB/s HBM peak."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
for T in [int(v) for v in (sys.argv[1:] or ['100', '188', '375'])]:
    r = bench.attention_roofline(dev, T=T)
    print('T=%4d  %-32s %7.2f us  %7.1f GB/s  frac %.3f  (%.2f MB)' % (
        T, r['kernel'], r['us_per_launch'], r['achieved'], r['frac'], r['bytes_per_launch'] / 1e6), flush=True)
