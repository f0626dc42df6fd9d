set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t4.log; tail -3 gpurun_out/r4_t4.log
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r4_b4.log 2>&1; tail -c 400 gpurun_out/r4_b4.log
