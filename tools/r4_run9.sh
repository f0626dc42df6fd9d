set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t9.log; tail -3 gpurun_out/r4_t9.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; tail -1 gpurun_out/r4_smoke.log
bash tools/ab_env.sh SSASR_FWD_OVERLAP 1 0 3 470 > gpurun_out/r4_ab_fwd_overlap.log 2>&1; cat gpurun_out/r4_ab_fwd_overlap.log
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -- python3 $R/tools/fixed_step.py 470 6 > $R/gpurun_out/prof_tl.log 2>&1
cd $R && python tools/timeline.py $(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1) > gpurun_out/r4_timeline_overlap.txt 2>&1; head -24 gpurun_out/r4_timeline_overlap.txt | cut -c1-150
rm -rf gpurun_out/prof_tl
