set -x
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sequence_major or packed or segmented" > gpurun_out/r4_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t5.log; tail -3 gpurun_out/r4_t5.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_ctc.py -m gpu -x -q -k "long or config4 or joint" > gpurun_out/r4_t5b.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t5b.log; tail -3 gpurun_out/r4_t5b.log
python tools/config4_step.py 6 > gpurun_out/r4_c4.log 2>&1; tail -3 gpurun_out/r4_c4.log
python tools/fwd_neighbour.py > gpurun_out/r4_fwd_neighbour.txt 2>&1; cat gpurun_out/r4_fwd_neighbour.txt
