set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t10.log; tail -3 gpurun_out/r4_t10.log
timeout -k 10 600 python bench.py > gpurun_out/r4_b10.log 2>&1; tail -c 300 gpurun_out/r4_b10.log
SOAK_STEPS=3000 SOAK_LONG=300 timeout -k 10 400 python tools/soak.py > gpurun_out/r4_soak.txt 2>&1; tail -3 gpurun_out/r4_soak.txt
