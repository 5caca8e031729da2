#!/usr/bin/env python3
"""Diagnostic: percentile5 (config 2) and the noise estimators (config 3) timings."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import _lib
if os.environ.get('KSP_LIB'): _lib.load(os.path.abspath(os.environ['KSP_LIB']))
from katsdpsigproc_amd import accel, percentile
from katsdpsigproc_amd.rfi import device
ctx = accel.create_some_context(False); q = ctx.create_command_queue()
rs = np.random.RandomState(1)
def timeit(fn, reps=200):
    for _ in range(100): fn()
    q.finish(); a = q.enqueue_marker()
    for _ in range(reps): fn()
    b = q.enqueue_marker(); q.finish(); return b.time_since(a) / reps
n = 4096
src = np.abs(rs.standard_normal((n, n))).astype(np.float32)
op = percentile.Percentile5Template(ctx, n, is_amplitude=True).instantiate(q, (n, n))
op.ensure_all_bound(); op.buffer("src").set(q, src)
t = timeit(op); print(f"percentile5 4096x4096: {1e3*t:.4f} ms {4*n*n/t/1e9:.0f} GB/s", flush=True)
C, B = 4096, 8192
dev = rs.standard_normal((B, C)).astype(np.float32)
ne = device.NoiseEstMADTDeviceTemplate(ctx, 10240).instantiate(q, C, B)
ne.ensure_all_bound(); ne.buffer("deviations").set(q, dev)
t = timeit(ne); print(f"madnz_t 4096x8192: {1e3*t:.4f} ms {4*C*B/t/1e9:.0f} GB/s", flush=True)
for method in (0, 1):
    ne = device.NoiseEstMADDeviceTemplate(ctx, tuning={"method": method}).instantiate(q, C, B)
    ne.ensure_all_bound(); ne.buffer("deviations").set(q, np.ascontiguousarray(dev.T))
    t = timeit(ne); print(f"madnz method {method} 4096x8192: {1e3*t:.4f} ms {4*C*B/t/1e9:.0f} GB/s", flush=True)
