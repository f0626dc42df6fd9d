set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t3.log; tail -3 gpurun_out/r4_t3.log
python tools/frontend_bench.py > gpurun_out/r4_frontend.json 2>gpurun_out/r4_frontend.err; cat gpurun_out/r4_frontend.json | head -30
bash tools/ab_env.sh SSASR_WGRAD_FUSED 1 0 3 470 > gpurun_out/r4_ab_wgrad2.log 2>&1; cat gpurun_out/r4_ab_wgrad2.log
for v in 1 0; do
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_tl$v && SSASR_WGRAD_FUSED=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl$v -- python3 $R/tools/fixed_step.py 470 6 > $R/gpurun_out/prof_tl$v.log 2>&1
cd $R && python tools/timeline.py $(find gpurun_out/prof_tl$v -name "*kernel_trace.csv" | head -1) > gpurun_out/r4_timeline_fused$v.txt 2>&1; tail -3 gpurun_out/r4_timeline_fused$v.txt
rm -rf gpurun_out/prof_tl$v
done
