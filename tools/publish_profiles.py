#!/usr/bin/env python3
"""Copy a profile set collected by tools/collect_profiles.sh from gpurun_out/ into profiles/
(the tracked, judged location) and refresh profiles/hbm_traffic.json.

    usage: tools/publish_profiles.py <tag> <version>      e.g. r02a v1
"""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ver = sys.argv[1], sys.argv[2]
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
rnd = tag[:3]
for name in ("bench_kernel_stats.csv", "bench_default_kernel_stats.csv", "kernel_stats_NONE.csv",
             "kernel_stats_CHANNEL.csv", "kernel_stats_L8192.csv", "kernel_stats_L10240.csv",
             "kernel_stats_RFI.csv"):
    if not os.path.exists(os.path.join(src, name)):
        continue
    shutil.copy(os.path.join(src, name),
                os.path.join(root, "profiles", f"{rnd}_{name[:-4]}_{ver}.csv"))
shutil.copy(os.path.join(src, "pmc_summary.json"),
            os.path.join(root, "profiles", f"{rnd}_pmc_summary_{ver}.json"))
s = json.load(open(os.path.join(src, "pmc_summary.json")))
entries = []
for mode in ("NONE", "CHANNEL"):
    f = s[mode]["fused"]
    # KB -> B, x2: gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM section)
    fetch = f["FETCH_SIZE"] * 1024 * 2
    write = f["WRITE_SIZE"] * 1024
    entries.append({
        "channels": 4096, "baselines": 32768, "use_flags": mode,
        "kernel": f.get("__kernel__", "flagger kernel"),
        "hbm_bytes_per_launch": round(fetch + write),
        "fetch_bytes": round(fetch), "write_bytes": round(write),
        # (without input flags the persistent kernel zero-fills the flags itself: its own
        # WRITE_SIZE then holds the 128 MiB and there is no fill kernel)
        "fill_kernel_write_bytes": round(s[mode]["fill"]["WRITE_SIZE"] * 1024) if "fill" in s[mode] else 0,
        "source": f"profiles/{rnd}_pmc_summary_{ver}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                  "passes; FETCH_SIZE doubled per the guide's gfx950 correction; cross-check "
                  f"TCC_EA0_RDREQ_sum x 128 B = {round(f['TCC_EA0_RDREQ_sum'] * 128)})",
    })
    hit = f["TCC_HIT_sum"] / (f["TCC_HIT_sum"] + f["TCC_MISS_sum"])
    print(f"{mode}: HBM {entries[-1]['hbm_bytes_per_launch'] / 1e9:.3f} GB/launch; L2 hit rate {hit:.3f}; "
          f"VALU instr {f['SQ_INSTS_VALU']:.4g} ({f['SQ_INSTS_VALU'] * 64 / (4096 * 32768):.1f} per sample); "
          f"VALU active {f['SQ_ACTIVE_INST_VALU'] * 4 / (f['SQ_BUSY_CYCLES'] / 32 * 1024):.2f} of SQ busy")
for case, (channels, baselines) in (("L8192", (8192, 4096)), ("L10240", (10240, 3276)), ("RFI", (4096, 32768))):
    if case not in s or "fused" not in s[case]:
        continue
    f = s[case]["fused"]
    print(f"{case}: {f.get('__kernel__')}: FETCH {f['FETCH_SIZE'] * 2048 / 1e9:.3f} GB, WRITE "
          f"{f['WRITE_SIZE'] * 1024 / 1e9:.3f} GB, VALU {f['SQ_INSTS_VALU'] * 64 / (channels * baselines):.1f} per sample, "
          f"VALU active {f['SQ_ACTIVE_INST_VALU'] * 4 / (f['SQ_BUSY_CYCLES'] / 32 * 1024):.2f} of SQ busy")
json.dump(entries, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
for name in ("bench_kernel_stats.csv", "kernel_stats_NONE.csv", "kernel_stats_CHANNEL.csv"):
    print(name)
    print(open(os.path.join(src, name)).read())
