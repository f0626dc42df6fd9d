# Socket power and shader clock (rocm-smi) while a workload runs: the train step, then the wide fp32 GEMM back to back.
mkdir -p gpurun_out
L=gpurun_out/power.log; : > $L
sample() { rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' ' >> $L; echo >> $L; }
echo "== idle" >> $L; sample
SOAK_STEPS=3500 python tools/soak.py > gpurun_out/power_soak.txt 2>&1 &
PID=$!
sleep 12
echo "== train steps (tools/soak.py)" >> $L
for i in 1 2 3 4; do sample; sleep 1.5; done
wait $PID
python tools/gemm_wide.py 6000 "big NT" 256 > gpurun_out/power_gemm.txt 2>&1 &
PID=$!
sleep 7
echo "== 4096^3 NT on the wide stream-K kernel, back to back (tools/gemm_wide.py 6000 'big NT' 256)" >> $L
for i in 1 2 3 4 5 6; do sample; sleep 1; done
wait $PID
tail -n 1 gpurun_out/power_gemm.txt >> $L
python tools/gemm_wide.py 6000 "big NT" 128 > gpurun_out/power_gemm.txt 2>&1 &
PID=$!
sleep 7
echo "== the same on the 128 x 128 tile kernel" >> $L
for i in 1 2 3 4 5 6; do sample; sleep 1; done
wait $PID
tail -n 1 gpurun_out/power_gemm.txt >> $L
cat $L
