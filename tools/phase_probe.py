#!/usr/bin/env python3
"""Diagnostic: time the fused flagger with the kernel stopped after each phase
(KSP_FUSED_DEBUG_STOP), to see where a strip spends its time. Not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

channels = int(os.environ.get("CH", 4096))
baselines = int(os.environ.get("BL", 32768))
ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
use_flags = getattr(device.BackgroundFlags, os.environ.get("FLAGS", "NONE"))
t = device.FlaggerDeviceTemplate(
    device.BackgroundMedianFilterDeviceTemplate(ctx, 13, use_flags=use_flags),
    device.NoiseEstMADTDeviceTemplate(ctx, 10240),
    device.ThresholdSumDeviceTemplate(ctx), keep_deviations=False)
fn = t.instantiate(q, channels, baselines, threshold_args={"n_sigma": 11.0})
fn.ensure_all_bound()
rs = np.random.RandomState(1)
vis = (rs.standard_normal((channels, baselines)).astype(np.float32)
       + 1j * rs.standard_normal((channels, baselines)).astype(np.float32)).astype(np.complex64)
fn.buffer("vis").set(q, vis)
if use_flags == device.BackgroundFlags.CHANNEL:
    fn.buffer("input_flags").set(q, (rs.random_sample(channels) < 1 / 16).astype(np.uint8))
elif use_flags == device.BackgroundFlags.FULL:
    fn.buffer("input_flags").set(q, (rs.random_sample((channels, baselines)) < 1 / 16).astype(np.uint8))
names = {11: "load-noamp", 1: "load", 2: "+median", 31: "+keys", 32: "+bitsearch", 33: "+gather", 34: "+rank", 35: "+below", 36: "+noise64", 3: "+mad", 4: "+threshold", 0: "full"}
for stop in (1, 11, 1, 2, 31, 32, 33, 3, 4, 0):
    os.environ["KSP_FUSED_DEBUG_STOP"] = str(stop)
    fn(); q.finish()
    a = q.enqueue_marker()
    for _ in range(5):
        fn()
    b = q.enqueue_marker()
    q.finish()
    print(f"stop={stop} {names[stop]:>12}: {1e3 * b.time_since(a) / 5:.3f} ms", flush=True)
