set -e
mkdir -p gpurun_out
L=gpurun_out/ab_batch.log
: > $L
run() { timeout -k 10 200 python tools/ab_option.py "$@" 2>&1 | grep -v amdgpu.ids >> $L; }
run SSASR_LAST_SEG_PCT 60 40 470 20 5
run SSASR_LAST_SEG_PCT 60 80 470 20 5
run SSASR_LAST_SEG_PCT 60 100 470 20 5
run SSASR_TAIL_INLINE 1 0 470 20 5
run SSASR_BPTT_RESERVE_KB 118 0 470 20 5
run SSASR_WGRAD_FUSED 1 0 470 20 5
run SSASR_GEMM_KCAT 1 0 470 20 5
run ops:bptt_segments 4 3 470 20 5
run ops:bptt_segments 4 5 470 20 5
run ops:bptt_segments 4 6 470 20 5
cat $L
