"""HBM-side bytes per launch of named kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs of the same command, --kernel-trace only beside them):
    python tools/pmc_kernels.py <fetch dir> <write dir> <launches to keep> <kernel name pattern> ...
Corrections as in tools/pmc_to_json.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE kB x 2, WRITE_SIZE kB as is.
The LAST `launches to keep` launches of each kernel are reported in launch order (one step's worth: steady state)."""
import csv
import glob
import json
import os
import sys


def collect(d, counter, pats):
    per = {p: [] for p in pats}
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r.get('Counter_Name') == counter]
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    for r in rows:
        for p in pats:
            if p in r['Kernel_Name']:
                per[p].append(float(r['Counter_Value']))
    return per


fetch_dir, write_dir, keep = sys.argv[1], sys.argv[2], int(sys.argv[3])
pats = sys.argv[4:]
fetch, write = collect(fetch_dir, 'FETCH_SIZE', pats), collect(write_dir, 'WRITE_SIZE', pats)
out = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)',
       'correction': 'FETCH_SIZE kB x 2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE kB as is'}
for p in pats:
    f, w = fetch[p][-keep:], write[p][-keep:]
    out[p] = [{'fetch_bytes': int(a * 2048), 'write_bytes': int(b * 1024), 'bytes': int(a * 2048 + b * 1024)} for a, b in zip(f, w)]
print(json.dumps(out, indent=1))
