#!/usr/bin/env python3
"""Diagnostic: run the fused flagger a few times on the benchmark shape (for rocprofv3).
Environment: CH, BL (shape), N (launches), FLAGS=NONE|CHANNEL|FULL, RFI=1 (inject interference),
PAD (row padding of vis in elements instead of the autotuned one), KSP_LIB (another library build)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import _lib  # noqa: E402

if os.environ.get("KSP_LIB"):
    _lib.load(os.path.abspath(os.environ["KSP_LIB"]))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

channels = int(os.environ.get("CH", 4096))
baselines = int(os.environ.get("BL", 32768))
mode = os.environ.get("FLAGS", "NONE")
ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
t = device.FlaggerDeviceTemplate(
    device.BackgroundMedianFilterDeviceTemplate(ctx, 13, use_flags=device.BackgroundFlags[mode],
                                                tuning={"csplit": 0}),
    device.NoiseEstMADTDeviceTemplate(ctx, 10240),
    device.ThresholdSumDeviceTemplate(ctx, tuning={"vt": 0}), fused=True,
    tuning={"vis_pad": int(os.environ["PAD"])} if "PAD" in os.environ else None)
fn = t.instantiate(q, channels, baselines, threshold_args={"n_sigma": 11.0})
fn.ensure_all_bound()
rs = np.random.RandomState(1)
tile = min(baselines, 4096)
block = (rs.standard_normal((channels, tile)).astype(np.float32)
         + 1j * rs.standard_normal((channels, tile)).astype(np.float32)).astype(np.complex64)
if os.environ.get("RFI") == "1":
    hit = rs.random_sample(block.shape) < 1 / 16
    n = int(hit.sum())
    block[hit] += ((rs.random_sample(n) * 20 + 50) * np.exp(2j * np.pi * rs.random_sample(n))).astype(np.complex64)
fn.buffer("vis").set(q, np.tile(block, (1, -(-baselines // tile)))[:, :baselines])
if mode == "CHANNEL":
    fn.buffer("input_flags").set(q, (np.random.RandomState(2).random_sample(channels) < 1 / 16).astype(np.uint8))
elif mode == "FULL":
    fn.buffer("input_flags").set(q, (rs.random_sample((channels, baselines)) < 1 / 16).astype(np.uint8))
for _ in range(int(os.environ.get("N", 3))):
    fn()
q.finish()
