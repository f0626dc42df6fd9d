"""Workload for the rocprofv3 --pmc passes behind `roofline.traffic`: one
forward + BPTT call of an encoder layer at the bench's roofline shape and a few
attention steps.  Run once per counter (FETCH_SIZE and WRITE_SIZE do not fit
one pass):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d out_f -- python3 tools/pmc_layer.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 tools/pmc_layer.py
    python3 tools/pmc_to_json.py out_f out_w > profiles/r02_traffic.json
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

dev = torch.device('cuda', 0)
bench.recurrence_roofline(dev, reps=int(os.environ.get('SSASR_ROOFLINE_REPS', '1')))   # 5 = what bench.py times
bench.attention_roofline(dev, T=375, iters=4)      # split-T kernel (BASELINE configs[3]'s longest encoder output)
bench.attention_roofline(dev, T=100, iters=4)      # the training shape
torch.cuda.synchronize()
