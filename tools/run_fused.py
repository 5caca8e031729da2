#!/usr/bin/env python3
"""Diagnostic: run the fused flagger a few times on the benchmark shape (for rocprofv3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import accel
from katsdpsigproc_amd.rfi import device
channels = int(os.environ.get("CH", 4096)); baselines = int(os.environ.get("BL", 32768))
ctx = accel.create_some_context(False); q = ctx.create_command_queue()
t = device.FlaggerDeviceTemplate(device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
    device.NoiseEstMADTDeviceTemplate(ctx, 10240), device.ThresholdSumDeviceTemplate(ctx), keep_deviations=False)
fn = t.instantiate(q, channels, baselines, threshold_args={"n_sigma": 11.0}); fn.ensure_all_bound()
rs = np.random.RandomState(1)
vis = np.empty((channels, baselines), np.complex64)
vis.real = rs.standard_normal((channels, baselines)).astype(np.float32)
vis.imag = rs.standard_normal((channels, baselines)).astype(np.float32)
fn.buffer("vis").set(q, vis)
for _ in range(int(os.environ.get("N", 3))):
    fn()
q.finish()
