set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_c4 && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/tools/config4_step.py 5 > $R/gpurun_out/prof_c4.log 2>&1
cd $R && python tools/summarize_profile.py gpurun_out/prof_c4 r04_config4 5 > gpurun_out/r4_c4_summary.txt 2>&1; head -30 gpurun_out/r4_c4_summary.txt
rm -rf gpurun_out/prof_c4/*/*kernel_trace.csv
NCCL_DEBUG=INFO SSASR_DIST_SINGLE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-roofline --no-config4 --no-cpu-baseline --no-epoch > gpurun_out/r4_nccl_debug.log 2>&1; grep -i "NCCL INFO" gpurun_out/r4_nccl_debug.log | grep -i "channel\|nranks\|Init COMPLETE\|version\|MAX_NCHANNELS" | head -30
