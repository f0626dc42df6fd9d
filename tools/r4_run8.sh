set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t8.log; tail -3 gpurun_out/r4_t8.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; tail -1 gpurun_out/r4_smoke.log
SOAK_STEPS=3000 SOAK_LONG=300 timeout -k 10 400 python tools/soak.py > gpurun_out/r4_soak.txt 2>&1; tail -3 gpurun_out/r4_soak.txt
