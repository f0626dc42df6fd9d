set -x
R=$GRAFT_REPO_ROOT
bash tools/profile_round.sh > gpurun_out/r4_profile_round.log 2>&1; tail -3 gpurun_out/r4_profile_round.log | cut -c1-400
for v in 0 1; do SSASR_NO_WINDOWS=$v python tools/config4_step.py 6 2>&1 | grep "last 3"; done > gpurun_out/r4_ab_windows.log; for v in 0 1; do SSASR_NO_WINDOWS=$v python tools/config4_step.py 6 2>&1 | grep "last 3"; done >> gpurun_out/r4_ab_windows.log; cat gpurun_out/r4_ab_windows.log
