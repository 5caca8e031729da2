mkdir -p gpurun_out/r3d
for v in nost; do
PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_$v.so NONE 2>&1 | grep kernel | tee -a gpurun_out/r3d/ml.txt
PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_$v.so NONE rfi 2>&1 | grep kernel | tee -a gpurun_out/r3d/ml.txt
done
cp build/variants/lib_nost.so /tmp/lib_full.so
timeout -k 10 600 python - <<'PY' 2>&1 | tail -3
import os, sys
sys.path.insert(0, os.getcwd())
from katsdpsigproc_amd import _lib
_lib.load("/tmp/lib_full.so")
import pytest
sys.exit(pytest.main(["tests/test_gpu_flagger.py", "-m", "gpu", "-x", "-q"]))
PY
