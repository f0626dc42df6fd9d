set -x
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "packed or segmented or sequence_major" > gpurun_out/r4_t12.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t12.log; tail -3 gpurun_out/r4_t12.log
