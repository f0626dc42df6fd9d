set -x
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "segmented or backward or reproducible or missing or times_out or forward_matches or integration" > gpurun_out/r4_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t2.log; tail -3 gpurun_out/r4_t2.log
bash tools/ab_env.sh SSASR_WGRAD_FUSED 1 0 3 470 > gpurun_out/r4_ab_wgrad.log 2>&1; cat gpurun_out/r4_ab_wgrad.log
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -- python3 $GRAFT_REPO_ROOT/tools/fixed_step.py 470 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_tl.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1) > gpurun_out/r4_timeline_a.txt 2>&1; tail -4 gpurun_out/r4_timeline_a.txt
