#!/usr/bin/env python3
"""Diagnostic: standalone median filter (config 3 shape) per channel split."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import _lib
if os.environ.get('KSP_LIB'): _lib.load(os.path.abspath(os.environ['KSP_LIB']))
from katsdpsigproc_amd import accel
from katsdpsigproc_amd.rfi import device
ctx = accel.create_some_context(False); q = ctx.create_command_queue()
C, B = 4096, 8192
rs = np.random.RandomState(1)
vis = (rs.standard_normal((C, B)).astype(np.float32) + 1j * rs.standard_normal((C, B)).astype(np.float32)).astype(np.complex64)
for csplit in (16, 32, 64):
    fn = device.BackgroundMedianFilterDeviceTemplate(ctx, 13, tuning={"csplit": csplit}).instantiate(q, C, B)
    fn.ensure_all_bound(); fn.buffer("vis").set(q, vis)
    for _ in range(150): fn()
    q.finish()
    a = q.enqueue_marker()
    for _ in range(100): fn()
    b = q.enqueue_marker(); q.finish()
    t = b.time_since(a) / 100
    print(f"csplit {csplit:4d}: {1e3*t:.4f} ms  {12*C*B/t/1e9:.0f} GB/s", flush=True)
