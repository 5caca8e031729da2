#!/bin/bash
# Collect the round's judged profiles on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (kernel durations)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hits / SQ counters) of the fused
#      kernel, for use_flags NONE (config 4) and CHANNEL (the per-GPU launch of config 5)
# Results land in gpurun_out/profiles_<tag>/ ; tools/publish_profiles.py copies what
# should be judged into profiles/.      usage: tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KATSDPSIGPROC_TUNE_DB=/tmp/ksp_tune_$TAG.db
# (fill the tuning cache first: the profiled runs then launch nothing but what they measure)
python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --preheat 0 > $OUT/bench_warm_cache.log 2>&1
FLAGS=CHANNEL N=1 python3 $R/tools/run_fused.py >> $OUT/bench_warm_cache.log 2>&1
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# the default command as the driver runs it (adds the RFI-laden variant and the bring-up
# shapes of configs 2 and 3: their kernels' durations; the fused kernel's average mixes shapes)
rm -rf /tmp/ktd
rocprofv3 --kernel-trace --stats -d /tmp/ktd -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_default_under_rocprof.log 2>&1
cp $(find /tmp/ktd -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
for MODE in NONE CHANNEL; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rm -rf /tmp/pm_$i
    FLAGS=$MODE N=3 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$i -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pm_$i.log 2>&1
    cp $(find /tmp/pm_$i -name "*counter_collection.csv" | head -1) $OUT/pmc_${MODE}_pass$i.csv
  done
  # (300 launches: the first ~40 after idle run at rising clocks, see bench.py)
  rm -rf /tmp/kt_$MODE
  FLAGS=$MODE N=300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$MODE -o k --output-format csv -- python3 $R/tools/run_fused.py > /tmp/kt_$MODE.log 2>&1
  cp $(find /tmp/kt_$MODE -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$MODE.csv
done
# the long-band kernels (reference presets kat7 / big, scripts/rfiflagtest.py:190-195: 8192 and
# 10240 channels; the number of samples of config 3), the ring kernel on an RFI-laden block, and
# the 4-baseline kernel with 8 windows: instruction and traffic counters, kernel durations
for CASE in "L8192 8192 4096 0" "L10240 10240 3276 0" "RFI 4096 32768 1"; do
  set -- $CASE
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    rm -rf /tmp/pl_$i
    CH=$2 BL=$3 RFI=$4 N=3 rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$i -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pl_$i.log 2>&1
    cp $(find /tmp/pl_$i -name "*counter_collection.csv" | head -1) $OUT/pmc_$1_pass$i.csv
  done
  rm -rf /tmp/kl
  CH=$2 BL=$3 RFI=$4 N=200 rocprofv3 --kernel-trace --stats -d /tmp/kl -o k --output-format csv -- python3 $R/tools/run_fused.py > /tmp/kl.log 2>&1
  cp $(find /tmp/kl -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$1.csv
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summary = {}
for mode in ("NONE", "CHANNEL", "L8192", "L10240", "RFI"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(out + f"/pmc_{mode}_pass*.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = "fused" if ("flagger_ring" in name or "flagger_fused" in name or "flagger_long" in name) else ("fill" if "fillBuffer" in name else None)
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k].setdefault("__kernel__", []).append(name.split("(")[0])
    summary[mode] = {k: {c: (sum(v) / len(v) if c != "__kernel__" else sorted(set(v))[0]) for c, v in d.items()}
                     for k, d in acc.items()}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
cat $OUT/bench_kernel_stats.csv
tail -1 $OUT/bench_under_rocprof.log
