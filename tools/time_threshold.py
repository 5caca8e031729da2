#!/usr/bin/env python3
"""Diagnostic: standalone SumThreshold (transposed layout, 4096 channels x 8192 baselines) on
deviations without and with interference (1/16 of the samples at 50-70 sigma)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import _lib  # noqa: E402

if os.environ.get("KSP_LIB"):
    _lib.load(os.path.abspath(os.environ["KSP_LIB"]))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
C, B = 4096, 8192
rs = np.random.RandomState(1)
dev = rs.standard_normal((B, C)).astype(np.float32)
for label in ("clean", "rfi"):
    if label == "rfi":
        hit = rs.random_sample(dev.shape) < 1 / 16
        dev[hit] += (rs.random_sample(int(hit.sum())) * 20 + 50).astype(np.float32)
    for vt in (8, 16):
        fn = device.ThresholdSumDeviceTemplate(ctx, tuning={"wgsx": 256, "vt": vt}).instantiate(q, C, B, n_sigma=11.0)
        fn.ensure_all_bound()
        fn.buffer("deviations").set(q, dev)
        fn.buffer("noise").set(q, np.ones(B, np.float32))
        for _ in range(60):
            fn()
        q.finish()
        a = q.enqueue_marker()
        for _ in range(60):
            fn()
        b = q.enqueue_marker()
        q.finish()
        t = b.time_since(a) / 60
        flagged = np.count_nonzero(fn.buffer("flags").get(q)) / dev.size
        print(f"{label:5s} vt {vt:2d}: {1e3 * t:.4f} ms  {5 * C * B / t / 1e9:.0f} GB/s  flagged {flagged:.4f}", flush=True)
