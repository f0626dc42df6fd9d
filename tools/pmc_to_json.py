"""Turns two rocprofv3 --pmc runs of tools/pmc_layer.py (FETCH_SIZE, WRITE_SIZE)
into profiles/r0N_traffic.json: HBM-side bytes per launch of the kernels that
bench.py prices.  MI355X_MICROARCH.md, HBM section: FETCH_SIZE is in units of
64 B requests tallied for 128 B requests on gfx950 (so kB * 2 for wide reads;
the exchange loads and the saved-activation loads here are 16 B per lane), and
WRITE_SIZE reads exact for 16-B-per-lane stores."""
import csv
import glob
import json
import os
import sys

KERNELS = {'bptt': 'lstm_enc_bwd_rs_kernel', 'fwd': 'lstm_enc_fwd_persistent_kernel',
           'attention': 'attn_step_fwd_fast_kernel', 'attention_split': 'attn_step_fwd_split_kernel'}


def collect(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            for key, pat in KERNELS.items():
                if pat in r['Kernel_Name']:
                    per.setdefault(key, []).append(float(r['Counter_Value']))
    return per


fetch = collect(sys.argv[1], 'FETCH_SIZE')
write = collect(sys.argv[2], 'WRITE_SIZE')
out = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/pmc_layer.py',
       'correction': 'FETCH_SIZE kB x 2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE kB as is'}
for key in KERNELS:
    f = fetch.get(key, [])
    w = write.get(key, [])
    if not f or not w:
        continue
    # last launch of each kernel = steady state (first launches warm the caches)
    fb = f[-1] * 1024 * 2
    wb = w[-1] * 1024
    out[key + '_fetch_bytes'] = int(fb)
    out[key + '_write_bytes'] = int(wb)
    out[key + '_bytes_per_launch'] = int(fb + wb)
print(json.dumps(out, indent=1))
