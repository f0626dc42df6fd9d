R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
  rm -rf $R/gpurun_out/tl_$w
  SSASR_GEMM_WIDE=$w timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$w -- python3 $R/tools/fixed_step.py ${FRAMES:-800} 6 > $R/gpurun_out/tl_$w.log 2>&1
  f=$(find $R/gpurun_out/tl_$w -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/timeline.py $f > $R/gpurun_out/timeline_${FRAMES:-800}_wide$w.txt
  rm -rf $R/gpurun_out/tl_$w
done
