mkdir -p gpurun_out/r3d
KSP_RING_TRACE=/tmp/ring.trace PAD=16 N=3 W=60 timeout -k 10 120 python tools/time_fused.py build/variants/lib_trace.so NONE 2>&1 | grep kernel
python tools/trace_ring.py /tmp/ring.trace > gpurun_out/r3d/trace_cur.txt 2>&1; grep -v "^wg" gpurun_out/r3d/trace_cur.txt
