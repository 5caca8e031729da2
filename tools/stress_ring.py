#!/usr/bin/env python3
"""Diagnostic: the persistent ring kernel against the 4-baseline kernel on the same input, bit
for bit, over many launches and inputs at sizes where the ring's prefetch, its ticket counters
and the counted waits of its LDS-DMA are all in play (the parity tests compare both with the
oracle, but on arrays of a few strips).  usage: tools/stress_ring.py [rounds] [baselines]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from katsdpsigproc_amd import _lib, accel  # noqa: E402
from katsdpsigproc_amd.rfi import device  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
C = 4096
ctx = accel.create_some_context(False)
q = ctx.create_command_queue()
fns = {}
for pad in (16, 0, 48):
    t = device.FlaggerDeviceTemplate(
        device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
        device.NoiseEstMADTDeviceTemplate(ctx, 10240),
        device.ThresholdSumDeviceTemplate(ctx), fused=True, tuning={"vis_pad": pad})
    fn = t.instantiate(q, C, B, threshold_args={"n_sigma": 11.0})
    fn.ensure_all_bound()
    fns[pad] = fn
bad = 0
for r in range(rounds):
    rs = np.random.RandomState(100 + r)
    tile = 2048
    block = (rs.standard_normal((C, tile)) + 1j * rs.standard_normal((C, tile))).astype(np.complex64)
    block *= np.float32(10.0 ** rs.uniform(-3, 3))
    kind = r % 4
    if kind >= 1:  # strong interference on a fraction of the samples
        hit = rs.random_sample(block.shape) < (1 / 16 if kind == 1 else 1 / 200)
        block[hit] *= rs.uniform(20, 80)
    if kind == 3:  # weak broad-band interference: the window sums decide
        for b in range(0, tile, 3):
            c = rs.randint(0, C - 16)
            block[c:c + rs.randint(2, 12), b] *= rs.uniform(2.0, 5.0)
    vis = np.tile(block, (1, -(-B // tile)))[:, :B]
    # (every tile shifted by its index: strips differ)
    for k in range(1, B // tile):
        vis[:, k * tile:(k + 1) * tile] = np.roll(block, 7 * k, axis=0)
    for pad, fn in fns.items():
        fn.buffer("vis").set(q, vis)
        out = {}
        for ring in ("1", "0"):
            _lib.call("ksp_flagger_fused_ring_mode", 1 if ring == "1" else -1)
            for rep in range(3 if ring == "1" else 1):
                fn.buffer("flags").set(q, np.full((C, B), 255, np.uint8))
                fn()
                path = _lib.call("ksp_flagger_fused_last_path")
                assert bool(path & 4) == (ring == "1"), path
                got = (fn.buffer("flags").get(q), fn.buffer("noise").get(q))
                if ring in out:
                    same = np.array_equal(got[0], out[ring][0]) and np.array_equal(got[1], out[ring][1], equal_nan=True)
                    if not same:
                        bad += 1
                        print("ROUND %d pad %d: the ring kernel's repeat %d differs from its first run" % (r, pad, rep))
                out[ring] = got
        same = np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1], equal_nan=True)
        if not same:
            bad += 1
            d = np.argwhere(out["1"][0] != out["0"][0])
            print("ROUND %d pad %d: ring and 4-baseline kernels differ: %d flags (first at %s), %d noise values"
                  % (r, pad, len(d), d[:1].tolist(), int((out["1"][1] != out["0"][1]).sum())))
        print("round %d (kind %d) pad %2d: flagged %.4f  %s" % (r, kind, pad, float((out["1"][0] != 0).mean()),
                                                              "same" if same else "DIFFERENT"), flush=True)
_lib.call("ksp_flagger_fused_ring_mode", 0)
print("stress_ring: %d mismatches" % bad)
sys.exit(1 if bad else 0)
