#!/usr/bin/env python3
"""Diagnostic: does the row stride of `vis` matter to the fused kernel? Times the benchmark
shape with the visibilities in buffers padded by PAD elements per row (default list)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

channels, baselines = 4096, int(os.environ.get("BL", 32768))
ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
t = device.FlaggerDeviceTemplate(
    device.BackgroundMedianFilterDeviceTemplate(ctx, 13), device.NoiseEstMADTDeviceTemplate(ctx, 16384),
    device.ThresholdSumDeviceTemplate(ctx), fused=True)
fn = t.instantiate(q, channels, baselines, threshold_args={"n_sigma": 11.0})
fn.ensure_all_bound()
rs = np.random.RandomState(1)
block = (rs.standard_normal((channels, 4096)).astype(np.float32)
         + 1j * rs.standard_normal((channels, 4096)).astype(np.float32)).astype(np.complex64)
vis = np.tile(block, (1, baselines // 4096))
pads = [int(x) for x in sys.argv[1:]] or [0, 16, 32, 64, 128, 256, 512, 2048]
for rep in range(2):
    for pad in pads:
        buf = accel.DeviceArray(ctx, (channels, baselines), np.complex64, (channels, baselines + pad))
        buf.set(q, vis)
        fn.slots["vis"].buffer = buf  # (past the slot's padding check: this is the experiment)
        for _ in range(60):
            fn()
        q.finish()
        ev = []
        for _ in range(100):
            ev.append(fn.profile_next_run())
            fn()
        q.finish()
        k = [1e3 * b.time_since(a) for a, b in ev]
        print(f"pad {pad:5d} elements: kernel mean {np.mean(k):.4f} min {np.min(k):.4f} ms", flush=True)
        del buf
