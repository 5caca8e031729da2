for i in 1 2; do
for e in "KSP_RING_MEMSET=1" "KSP_RING_ZERO=1"; do
env $e PAD=16 timeout -k 10 120 python tools/time_fused.py NONE rfi 2>&1 | grep kernel | sed "s/^/$e /"
done; done
