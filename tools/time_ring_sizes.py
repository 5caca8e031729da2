#!/usr/bin/env python3
"""Diagnostic: the persistent ring kernel against the 4-baseline kernel (KSP_FUSED_RING=1 / 0
in child processes) over the number of baselines, 4096 channels, clean noise.
usage: tools/time_ring_sizes.py [baselines ...]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def measure(sizes):
    from katsdpsigproc_amd import accel
    from katsdpsigproc_amd.rfi import device

    ctx = accel.create_some_context(False)
    q = ctx.create_command_queue()
    rs = np.random.RandomState(1)
    for B in sizes:
        t = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
            device.NoiseEstMADTDeviceTemplate(ctx, 10240),
            device.ThresholdSumDeviceTemplate(ctx), fused=True, tuning={"vis_pad": 16})
        fn = t.instantiate(q, 4096, B, threshold_args={"n_sigma": 11.0})
        fn.ensure_all_bound()
        vis = (rs.standard_normal((4096, B)) + 1j * rs.standard_normal((4096, B))).astype(np.complex64)
        fn.buffer("vis").set(q, vis)
        for _ in range(30):
            fn()
        q.finish()
        a = q.enqueue_marker()
        for _ in range(50):
            fn()
        b = q.enqueue_marker()
        q.finish()
        ms = 1e3 * b.time_since(a) / 50
        print("%s baselines %6d  %.4f ms  %.3f of 8 TB/s" % (
            "ring kernel      " if os.environ.get("KSP_FUSED_RING") == "1" else "4-baseline kernel",
            B, ms, 9e-6 * 4096 * B / ms / 8000), flush=True)


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768]
    if os.environ.get("KSP_RING_SIZES_CHILD"):
        measure(sizes)
    else:
        for no_ring in (False, True):
            env = dict(os.environ, KSP_RING_SIZES_CHILD="1", KSP_FUSED_RING="0" if no_ring else "1")
            subprocess.run([sys.executable, __file__] + [str(s) for s in sizes], env=env, check=True)
