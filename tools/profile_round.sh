#!/bin/bash
# The three GPU runs behind profiles/<tag>_*: 7 train steps under rocprofv3 --kernel-trace --stats, the
# launches bench.py prices for its roofline objects under the same, and the bench line itself.
# On the GPU box:  bash tools/profile_round.sh   then, here:
#   python tools/summarize_profile.py gpurun_out/prof_h <tag>_final 7
#   python tools/summarize_profile.py gpurun_out/prof_r <tag>_roofline 1
#   grep '^{' gpurun_out/bench_final.log > profiles/<tag>_bench.json
#   cp gpurun_out/traffic.json profiles/<tag>_traffic.json   (PMC passes: tools/pmc_layer.py)
# (config 4: rocprofv3 --kernel-trace --stats -- python3 tools/config4_step.py 5 -> summarize_profile.py ... r03_config4 5)
set -x
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_h $R/gpurun_out/prof_r
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-config4 --no-epoch --no-bf16-variant > $R/gpurun_out/prof_h.log 2>&1 &&
SSASR_ROOFLINE_REPS=5 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r -- python3 $R/tools/pmc_layer.py > $R/gpurun_out/prof_r.log 2>&1 &&
cd /tmp && rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/tools/pmc_layer.py > $R/gpurun_out/pmc_f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/pmc_layer.py > $R/gpurun_out/pmc_w.log 2>&1 &&
python3 $R/tools/pmc_to_json.py $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w > $R/gpurun_out/traffic.json
cd $R && timeout -k 10 400 python3 bench.py > gpurun_out/bench_final.log 2>&1
tail -2 gpurun_out/bench_final.log | cut -c1-300
