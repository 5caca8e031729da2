mkdir -p gpurun_out/r3g
for e in "" "KSP_RING_MEMSET=1"; do env $e PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_nost.so NONE 2>&1 | grep kernel; env $e PAD=16 timeout -k 10 120 python tools/time_fused.py build/variants/lib_nost.so NONE rfi 2>&1 | grep kernel; done
cp build/variants/lib_nost.so /tmp/lib_full.so
timeout -k 10 600 python - <<'PY' 2>&1 | tail -3
import os, sys
sys.path.insert(0, os.getcwd())
from katsdpsigproc_amd import _lib
_lib.load("/tmp/lib_full.so")
import pytest
sys.exit(pytest.main(["tests/test_gpu_flagger.py", "-m", "gpu", "-x", "-q"]))
PY
