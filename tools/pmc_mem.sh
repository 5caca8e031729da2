#!/bin/bash
# Diagnostic: L2 / memory-side counters of the fused kernel. Output: gpurun_out/pmc_mem.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mem.txt
: > $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  N=2 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$i -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pm_$i.log 2>&1 || { tail -5 /tmp/pm_$i.log >> $OUT; continue; }
  python3 - $i >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(f"/tmp/pm_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flagger_fused" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
done
cat $OUT
