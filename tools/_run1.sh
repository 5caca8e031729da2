mkdir -p gpurun_out/r3f
for v in h1 h1r; do
PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_$v.so NONE 2>&1 | grep kernel | tee -a gpurun_out/r3f/t2.txt
PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_$v.so NONE rfi 2>&1 | grep kernel | tee -a gpurun_out/r3f/t2.txt
done
