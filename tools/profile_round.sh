#!/bin/bash
# The three GPU runs behind profiles/<tag>_*: 7 train steps under rocprofv3 --kernel-trace --stats, the
# launches bench.py prices for its roofline objects under the same, and the bench line itself.
# On the GPU box:  bash tools/profile_round.sh   then, here:
#   python tools/summarize_profile.py gpurun_out/prof_h <tag>_final 7
#   python tools/summarize_profile.py gpurun_out/prof_r <tag>_roofline 1
#   grep '^{' gpurun_out/bench_final.log > profiles/<tag>_bench.json
# (PMC traffic: see tools/pmc_layer.py)
set -x
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_h $R/gpurun_out/prof_r
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_h.log 2>&1 &&
SSASR_ROOFLINE_REPS=5 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r -- python3 $R/tools/pmc_layer.py > $R/gpurun_out/prof_r.log 2>&1 &&
cd $R && timeout -k 10 300 python3 bench.py > gpurun_out/bench_final.log 2>&1
