#!/usr/bin/env python3
"""Diagnostic: read the per-wavefront phase time stamps written by the fused kernel
(KSP_FUSED_DEBUG_TRACE=<file>) and summarise how co-resident workgroups interleave."""
import sys
import numpy as np

a16 = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
a = a16[:, :8]
t = a[:, :7].astype(np.int64)
hw = a[:, 7]
ok = t[:, 0] > 0
t, hw = t[ok], hw[ok]
t0 = t[:, 0].min()
t -= t0
names = ["load", "barrier", "median", "mad", "threshold", "write"]
d = np.diff(t, axis=1)
print("waves:", len(t), " kernel span (cycles):", t[:, 6].max())
for i, nme in enumerate(names):
    print(f"{nme:10s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
if a16[:, 8:12].any():
    okm = ok & (a16[:, 8] > 0) & (a16[:, 11] > 0)
    m = a16[okm].astype(np.int64)
    print("MAD split (mean cycles): keys+zeros %.0f, transpose %.0f, bit search %.0f, gather %.0f, rank+exact %.0f" % (
        (m[:, 8] - m[:, 3]).mean(), (m[:, 9] - m[:, 8]).mean(), (m[:, 10] - m[:, 9]).mean(),
        (m[:, 11] - m[:, 10]).mean(), (m[:, 4] - m[:, 11]).mean()))
if a16[:, 12:14].any():
    okm = ok & (a16[:, 12] > 0) & (a16[:, 13] > 0) & (a16[:, 11] > 0)
    m = a16[okm].astype(np.int64)
    print("rank+exact split: exact_dev %.0f, rank %.0f, rest of MAD %.0f" % (
        (m[:, 12] - m[:, 11]).mean(), (m[:, 13] - m[:, 12]).mean(), (m[:, 4] - m[:, 13]).mean()))
if a16[:, 14].any():
    okm = ok & (a16[:, 13] > 0) & (a16[:, 14] > 0)
    m = a16[okm].astype(np.int64)
    print("after the ranking: to the end of mad_noise %.0f, from there to the MAD stamp of the kernel %.0f" % (
        (m[:, 14] - m[:, 13]).mean(), (m[:, 4] - m[:, 14]).mean()))
print("wave lifetime mean", (t[:, 6] - t[:, 0]).mean())
# per CU: fraction of time in which k waves are in the load phase
hwid = (hw & 0xffffffff).astype(np.int64)
xcc = (hw >> 32).astype(np.int64) & 0xf
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 0x7
simd = (hwid >> 4) & 3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs:", len(np.unique(key)))
# timeline for one CU
k0 = np.unique(key)[len(np.unique(key)) // 2]
sel = np.where(key == k0)[0]
order = sel[np.argsort(t[sel, 0])]
print(f"CU {k0}: {len(sel)} waves; start, load_end, median_end, mad_end, end, simd")
for i in order[:40]:
    print("  ", t[i, 0], t[i, 1], t[i, 3], t[i, 4], t[i, 6], simd[i])
# overlap statistic: for each CU, total time with >=1 wave loading and >=1 wave computing
span = t[:, 6].max()
grid = np.linspace(0, span, 4000)
both = only_load = only_comp = idle = 0
for k in np.unique(key)[::16]:
    s = key == k
    L = ((t[s, 0][:, None] <= grid) & (grid < t[s, 1][:, None])).sum(0)
    Cc = ((t[s, 2][:, None] <= grid) & (grid < t[s, 6][:, None])).sum(0)
    both += ((L > 0) & (Cc > 0)).sum(); only_load += ((L > 0) & (Cc == 0)).sum()
    only_comp += ((L == 0) & (Cc > 0)).sum(); idle += ((L == 0) & (Cc == 0)).sum()
tot = both + only_load + only_comp + idle
print(f"CU time: load+compute {both/tot:.2f}, load only {only_load/tot:.2f}, compute only {only_comp/tot:.2f}, idle {idle/tot:.2f}")
# workgroup slot numbers (HW_ID bits 16-19) of the waves of that CU, in start order
tg = (hwid >> 16) & 0xf
wave_slot = hwid & 0xf
print("tg_id / wave slot of the first 16 waves of the CU:", [(int(tg[i]), int(wave_slot[i])) for i in order[:16]])
first = t[:, 0] < np.percentile(t[:, 0], 5)
print("tg_id histogram of the earliest 5% of waves:", np.bincount(tg[first], minlength=16))
