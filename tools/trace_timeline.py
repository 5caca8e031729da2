#!/usr/bin/env python3
"""Diagnostic: chip-wide timeline from a KSP_DIAG_REALTIME trace (100 MHz stamps): how many
workgroups are loading / computing and the aggregate load rate over the launch."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
t = a[:, :7]
ok = t[:, 0] > 0
t = t[ok]
t = (t - t[:, 0].min()) * 10e-3  # microseconds
span = t[:, 6].max()
bytes_per_wave = float(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 8 * 4 / 4  # strip bytes / 4 waves
nb = 60
edges = np.linspace(0, span, nb + 1)
print(f"waves {len(t)}, span {span:.1f} us")
print("   t(us)  loading  computing  load GB/s")
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    mid = 0.5 * (lo + hi)
    loading = np.count_nonzero((t[:, 0] <= mid) & (mid < t[:, 1]))
    computing = np.count_nonzero((t[:, 2] <= mid) & (mid < t[:, 6]))
    # bytes: each wave's load spread uniformly over its load phase
    ov = np.clip(np.minimum(t[:, 1], hi) - np.maximum(t[:, 0], lo), 0, None)
    dur = np.maximum(t[:, 1] - t[:, 0], 1e-3)
    gb = (ov / dur).sum() * bytes_per_wave / ((hi - lo) * 1e-6) / 1e9
    print(f"{mid:8.1f} {loading:8d} {computing:9d} {gb:10.0f}")
d = np.diff(t, axis=1)
print("phase means (us): load %.2f barrier %.2f median %.2f mad %.2f thr %.2f write %.2f; life %.2f" % (
    *d.mean(0), (t[:, 6] - t[:, 0]).mean()))
