#!/usr/bin/env python3
"""Diagnostic: the fused flagger against the kernel-per-stage sequence for every median
width (4096 channels x 8192 baselines of noise, SumThreshold with 4 windows; the 4-baseline
kernel: KSP_FUSED_RING=0 is set here so that width 13 is comparable).
usage: tools/time_widths.py [sequence]"""
import os
import sys

import numpy as np

os.environ["KSP_FUSED_RING"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
C, B = 4096, 8192
rs = np.random.RandomState(1)
vis = (rs.standard_normal((C, B)) + 1j * rs.standard_normal((C, B))).astype(np.complex64)
for width in range(3, 32, 2):
    line = "width %2d" % width
    for fused in (True, False) if "sequence" in sys.argv[1:] else (True,):
        t = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(ctx, width),
            device.NoiseEstMADTDeviceTemplate(ctx, 10240),
            device.ThresholdSumDeviceTemplate(ctx), fused=fused,
            tuning={"vis_pad": 16} if fused else None)
        fn = t.instantiate(q, C, B, threshold_args={"n_sigma": 11.0})
        fn.ensure_all_bound()
        fn.buffer("vis").set(q, vis)
        for _ in range(20):
            fn()
        q.finish()
        a = q.enqueue_marker()
        for _ in range(30):
            fn()
        b = q.enqueue_marker()
        q.finish()
        ms = 1e3 * b.time_since(a) / 30
        line += "   %s %.4f ms (%.3f of 8 TB/s)" % ("fused" if fused else "sequence", ms, 9e-6 * C * B / ms / 8000)
    print(line, flush=True)
