#!/usr/bin/env python3
"""Diagnostic: summarise the per-strip phase stamps of a -DRING_TRACE build of the ring kernel
(KSP_RING_TRACE=<file> while it runs).  usage: tools/trace_ring.py <file>
Stamps per (workgroup, wavefront, strip): 0 strip start, 1 deviations done (ring consumed),
2 MAD done, 3 thresholds and flags done; [7] = strip number + 1."""
import sys

import numpy as np

STRIPS = 24
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, STRIPS, 16).astype(np.int64)
valid = t[..., 7] > 0
print("strips per workgroup: min %d max %d" % (valid[:, 0].sum(1).min(), valid[:, 0].sum(1).max()))
s0, s1, s2, s3 = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
for name, d in (("stream (ring, |z|, median)", s1 - s0), ("MAD", s2 - s1), ("thresholds + flags", s3 - s2),
                ("strip", s3 - s0)):
    v = d[valid]
    print("%-28s mean %8.0f  median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (name, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
for name, d in (("  of it: waiting for DMA", t[..., 5]), ("  of it: waiting at barriers", t[..., 6])):
    v = d[valid]
    print("%-28s mean %8.0f  median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (name, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
# inside the MAD (fused_common.h, mad_noise stamps 8 .. 14)
names = ["keys", "bit planes (transpose)", "key search", "candidate list", "exact recomputation", "ranking", "rest"]
prev = s1
for i, name in enumerate(names):
    cur = t[..., 8 + i]
    ok = valid & (cur > 0) & (prev > 0)
    if ok.any():
        v = (cur - prev)[ok]
        print("    MAD: %-24s mean %7.0f  median %7.0f cycles" % (name, v.mean(), np.median(v)))
    prev = np.where(cur > 0, cur, prev)
# gap between strips of a wavefront (barriers, ticket)
gap = s0[:, :, 1:] - s3[:, :, :-1]
v = gap[valid[:, :, 1:]]
print("%-28s mean %8.0f  median %8.0f cycles" % ("between strips", v.mean(), np.median(v)))
# per-strip-index profile for workgroup 0..7, wavefront 0
for wg in (0, 1, 100):
    n = int(valid[wg, 0].sum())
    print("wg %3d wave 0:" % wg, " ".join("%d/%d/%d" % (s1[wg, 0, i] - s0[wg, 0, i], s2[wg, 0, i] - s1[wg, 0, i], s3[wg, 0, i] - s2[wg, 0, i]) for i in range(n)))
# shader clock: cycles per 100 MHz tick between the first and the last strip start of a wavefront
rt = t[..., 4]
n = valid.sum(2)
clk = []
for wg in range(256):
    for w in range(8):
        k = int(n[wg, w])
        if k >= 3 and rt[wg, w, k - 1] > rt[wg, w, 0]:
            clk.append((s0[wg, w, k - 1] - s0[wg, w, 0]) / (rt[wg, w, k - 1] - rt[wg, w, 0]) * 100.0)
if clk:
    print("shader clock while the kernel runs: median %.0f MHz (min %.0f, max %.0f)" % (np.median(clk), min(clk), max(clk)))
    first = np.array([rt[wg, 0, 0] for wg in range(256)])
    last = np.array([rt[wg, 0, int(n[wg, 0]) - 1] for wg in range(256)])
    cnt = np.array([int(n[wg, 0]) for wg in range(256)])
    base = first.min()
    print("first strip start after the earliest one: median %.1f us, max %.1f us" % (np.median(first - base) / 100.0, (first - base).max() / 100.0))
    print("last strip START after the earliest first start: min %.1f median %.1f max %.1f us" % ((last - base).min() / 100.0, np.median(last - base) / 100.0, (last - base).max() / 100.0))
    per = (last - first) / np.maximum(cnt - 1, 1) / 100.0
    print("mean time per strip of a workgroup: min %.2f median %.2f max %.2f us" % (per.min(), np.median(per), per.max()))
    for x in range(8):
        sel = np.arange(256) % 8 == x
        print("  workgroups %d mod 8: strips %.1f, per strip %.2f us, last start %.1f us" % (x, cnt[sel].mean(), per[sel].mean(), np.median((last - base)[sel]) / 100.0))
