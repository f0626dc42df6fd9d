#!/bin/bash
# rocprofv3 kernel stats of a few train steps: tools/prof_step.sh <tag> [ENV=VAL ...]  -> gpurun_out/prof_<tag>/ + top lines
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-config4 --no-epoch > $R/gpurun_out/prof_$tag.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$tag.log
find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec head -14 {} \; | cut -c1-150
