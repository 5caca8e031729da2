#!/usr/bin/env python3
"""Diagnostic: the fused flagger (4-baseline kernel: KSP_FUSED_RING=0 is set here) over
n_windows and n_sigma -- the lower the thresholds, the more windows hold samples that could
make them fire -- on 4096 channels x 8192 baselines of noise; with `sequence` the
kernel-per-stage sequence as well.  usage: tools/time_windows.py [sequence]"""
import os
import sys

import numpy as np

os.environ["KSP_FUSED_RING"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import inputs  # noqa: E402
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
C, B = 4096, 8192
clean = inputs.generate_data(C, B, seed=1)
for nw in (4, 5, 6, 8):
    for fused in (True, False) if "sequence" in sys.argv[1:] else (True,):
        t = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
            device.NoiseEstMADTDeviceTemplate(ctx, 10240),
            device.ThresholdSumDeviceTemplate(ctx, n_windows=nw), fused=fused,
            tuning={"vis_pad": 16} if fused else None)
        for n_sigma in (1000.0, 11.0, 6.0):
            fn = t.instantiate(q, C, B, threshold_args={"n_sigma": n_sigma})
            fn.ensure_all_bound()
            fn.buffer("vis").set(q, clean)
            for _ in range(20):
                fn()
            q.finish()
            a = q.enqueue_marker()
            for _ in range(30):
                fn()
            b = q.enqueue_marker()
            q.finish()
            ms = 1e3 * b.time_since(a) / 30
            print("n_windows %d %-8s n_sigma %6.1f  %.4f ms  (%.3f of 8 TB/s at 9 B/sample), %d flagged"
                  % (nw, "fused" if fused else "sequence", n_sigma, ms, 9e-6 * C * B / ms / 8000,
                     int(fn.buffer("flags").get(q).sum())), flush=True)
