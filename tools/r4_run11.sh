set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_fe && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fe -- python3 $R/tools/frontend_bench.py > $R/gpurun_out/prof_fe.log 2>&1
cd $R && python tools/summarize_profile.py gpurun_out/prof_fe r04_frontend 1 > gpurun_out/r4_fe_summary.txt 2>&1; head -16 gpurun_out/r4_fe_summary.txt
rm -rf gpurun_out/prof_fe/*/*kernel_trace.csv
