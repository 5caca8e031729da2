#!/usr/bin/env python3
"""Diagnostic: time the fused flagger against the kernel-per-stage sequence for SumThreshold
with n_windows = 4 .. 8 (4096 channels x 8192 baselines; clean noise and interference on
1/16 of the samples).  usage: tools/time_windows.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import inputs  # noqa: E402
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
C, B = 4096, 8192
clean = inputs.generate_data(C, B, seed=1)
dirty = inputs.add_rfi_sparse(clean, seed=3)
for nw in (4, 5, 6, 8):
    for fused in (True, False):
        t = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
            device.NoiseEstMADTDeviceTemplate(ctx, 10240),
            device.ThresholdSumDeviceTemplate(ctx, n_windows=nw), fused=fused,
            tuning={"vis_pad": 16} if fused else None)
        fn = t.instantiate(q, C, B, threshold_args={"n_sigma": 11.0})
        fn.ensure_all_bound()
        for name, vis in (("clean", clean), ("rfi", dirty)):
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
            print("n_windows %d %-8s %-5s %.4f ms  (%.0f GB/s algorithmic, %.3f of 8 TB/s)"
                  % (nw, "fused" if fused else "sequence", name, ms, 9e-6 * C * B / ms,
                     9e-6 * C * B / ms / 8000))
