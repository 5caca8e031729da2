#!/usr/bin/env python3
"""Diagnostic: time the fused flagger kernel of one library build on the benchmark shape.

    usage: [PAD=n] tools/time_fused.py [path/to/lib.so] [NONE|CHANNEL|FULL] [rfi] [dev] [each]
    (PAD: row padding of vis in elements instead of the autotuned one)

Prints the kernel's mean / min duration (HIP events around the kernel itself) and the
device time per step (zero-fill + kernel). Input: tiled standard-normal block (cheap to
make; timing does not depend on the exact values), optionally with 1/16 interference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from katsdpsigproc_amd import _lib  # noqa: E402

args = sys.argv[1:]
lib = next((a for a in args if a.endswith(".so")), None)
if lib:
    _lib.load(os.path.abspath(lib))
from katsdpsigproc_amd import accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

mode = next((a for a in args if a in ("NONE", "CHANNEL", "FULL")), "NONE")
channels = int(os.environ.get("CH", 4096))
baselines = int(os.environ.get("BL", 32768))
steps = int(os.environ.get("N", 20))
ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
use_flags = getattr(device.BackgroundFlags, mode)
t = device.FlaggerDeviceTemplate(
    device.BackgroundMedianFilterDeviceTemplate(ctx, 13, use_flags=use_flags),
    device.NoiseEstMADTDeviceTemplate(ctx, 16384),
    device.ThresholdSumDeviceTemplate(ctx), fused=True, keep_deviations="dev" in args,
    tuning={"vis_pad": int(os.environ["PAD"])} if "PAD" in os.environ else None)
fn = t.instantiate(q, channels, baselines, threshold_args={"n_sigma": 11.0})
fn.ensure_all_bound()
rs = np.random.RandomState(1)
tile = min(baselines, 4096)
block = (rs.standard_normal((channels, tile)).astype(np.float32)
         + 1j * rs.standard_normal((channels, tile)).astype(np.float32)).astype(np.complex64)
if "rfi" in args:
    hit = rs.random_sample(block.shape) < 1 / 16
    n = int(hit.sum())
    block[hit] += ((rs.random_sample(n) * 20 + 50) * np.exp(2j * np.pi * rs.random_sample(n))).astype(np.complex64)
vis = np.tile(block, (1, -(-baselines // tile)))[:, :baselines]
fn.buffer("vis").set(q, vis)
if mode == "CHANNEL":
    fn.buffer("input_flags").set(q, (np.random.RandomState(2).random_sample(channels) < 1 / 16).astype(np.uint8))
elif mode == "FULL":
    fn.buffer("input_flags").set(q, (rs.random_sample((channels, baselines)) < 1 / 16).astype(np.uint8))
# (the device reaches its sustained clocks only after some tens of milliseconds of load:
# launches right after an idle period take up to 15 % longer)
for _ in range(int(os.environ.get("W", 60))):
    fn()
q.finish()
marks = [q.enqueue_marker()]
events = []
tracing = "KSP_FUSED_DEBUG_TRACE" in os.environ  # (KSP_DIAG builds: no events then)
for _ in range(steps):
    if not tracing:
        events.append(fn.profile_next_run())
    fn()
    marks.append(q.enqueue_marker())
q.finish()
k = [1e3 * b.time_since(a) for a, b in events] or [0.0]
s = [1e3 * b.time_since(a) for a, b in zip(marks[:-1], marks[1:])]
flagged = np.count_nonzero(fn.buffer("flags").get(q)) / (channels * baselines)
print(f"{os.path.basename(lib) if lib else 'product':28s} {mode:7s} {'rfi' if 'rfi' in args else 'clean':5s} "
      f"kernel mean {np.mean(k):.4f} min {np.min(k):.4f} max {np.max(k):.4f} ms; step {np.mean(s):.4f} ms; "
      f"flagged {flagged:.4f}", flush=True)
if "each" in args:  # the individual launches, in order (drift or scatter?)
    print(" ".join(f"{x:.3f}" for x in k))
