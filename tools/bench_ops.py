#!/usr/bin/env python3
"""Timing of the standalone operations on the BASELINE.json bring-up shapes (configs 2 and 3)
and of the reference-shaped five-kernel flagger sequence, with their algorithmic bytes
(SURVEY.md section 8(d)). Diagnostic companion of bench.py (which times config 4 only).
Prints one line per operation: ms, algorithmic GB/s, fraction of the 8 TB/s roofline."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import accel, maskedsum, percentile, transpose  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
rs = np.random.RandomState(1)


def timeit(fn, reps=10):
    fn()
    q.finish()
    a = q.enqueue_marker()
    for _ in range(reps):
        fn()
    b = q.enqueue_marker()
    q.finish()
    return b.time_since(a) / reps


def report(name, seconds, nbytes):
    gbs = nbytes / seconds / 1e9
    print(f"{name:58s} {1e3 * seconds:8.3f} ms  {gbs:8.1f} GB/s  {gbs / 8000:6.1%} of roofline", flush=True)


def randc(shape):
    return (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)


# ---- config 2: transpose + percentile5, 4096 x 4096 float32 -------------------------------
n = 4096
src = np.abs(rs.standard_normal((n, n))).astype(np.float32)
op = transpose.TransposeTemplate(ctx, np.float32, "float").instantiate(q, (n, n))
op.ensure_all_bound()
op.buffer("src").set(q, src)
report("transpose 4096x4096 f32 (8 B/element)", timeit(op), 8 * n * n)
op = percentile.Percentile5Template(ctx, n, is_amplitude=True).instantiate(q, (n, n))
op.ensure_all_bound()
op.buffer("src").set(q, src)
report("percentile5 4096x4096 f32 (4 B/element)", timeit(op), 4 * n * n)
op = maskedsum.MaskedSumTemplate(ctx).instantiate(q, (n, n))
op.ensure_all_bound()
op.buffer("src").set(q, randc((n, n)))
op.buffer("mask").set(q, (rs.random_sample(n) < 0.5).astype(np.float32))
report("maskedsum 4096x4096 c64 (8 B/element)", timeit(op), 8 * n * n)

# ---- config 3: background (width 13) + NoiseEstMAD, 4096 ch x 8192 bl ----------------------
C, B = 4096, 8192
vis = randc((C, B))
bg = device.BackgroundMedianFilterDeviceTemplate(ctx, 13).instantiate(q, C, B)
bg.ensure_all_bound()
bg.buffer("vis").set(q, vis)
t_bg = timeit(bg)
report("background median filter 4096x8192 c64 (12 B/sample)", t_bg, 12 * C * B)
dev = bg.buffer("deviations").get(q)
for name, tmpl, data in (
    ("NoiseEstMAD (channel-major)", device.NoiseEstMADDeviceTemplate(ctx), dev),
    ("NoiseEstMADT (baseline-major)", device.NoiseEstMADTDeviceTemplate(ctx, 10240), dev.T.copy()),
):
    ne = tmpl.instantiate(q, C, B)
    ne.ensure_all_bound()
    ne.buffer("deviations").set(q, data)
    t = timeit(ne)
    report(f"{name} 4096x8192 f32 (4 B/sample)", t, 4 * C * B)
noise = ne.buffer("noise").get(q)
for name, tmpl in (
    ("ThresholdSum (baseline-major)", device.ThresholdSumDeviceTemplate(ctx)),
    ("ThresholdSimple (channel-major)", device.ThresholdSimpleDeviceTemplate(ctx, transposed=False)),
):
    th = tmpl.instantiate(q, C, B, 11.0)
    th.ensure_all_bound()
    th.buffer("deviations").set(q, dev.T.copy() if tmpl.transposed else dev)
    th.buffer("noise").set(q, noise)
    report(f"{name} 4096x8192 (5 B/sample)", timeit(th), 5 * C * B)

# ---- the reference-shaped sequence vs the fused kernel, 4096 x 8192 ------------------------
for fused in (False, True):
    t = device.FlaggerDeviceTemplate(
        device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
        device.NoiseEstMADTDeviceTemplate(ctx, 10240),
        device.ThresholdSumDeviceTemplate(ctx), fused=fused, keep_deviations=not fused)
    fn = t.instantiate(q, C, B, threshold_args={"n_sigma": 11.0})
    fn.ensure_all_bound()
    fn.buffer("vis").set(q, vis)
    report(f"full flagger 4096x8192, {'fused kernel' if fused else 'five-kernel sequence'} (9 B/sample)",
           timeit(fn), 9 * C * B)
