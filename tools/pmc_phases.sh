#!/bin/bash
# Diagnostic: dynamic instruction counters per phase of the fused kernel
# (KSP_FUSED_DEBUG_STOP = 1 load, 2 +median, 3 +mad, 0 full). Output: gpurun_out/pmc_phases.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_phases.txt
: > $OUT
for stop in ${STOPS:-1 2 3 0}; do
  export KSP_FUSED_DEBUG_STOP=$stop N=2
  rm -rf /tmp/pmc_$stop
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/pmc_$stop -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pmc_$stop.log 2>&1 || { tail -5 /tmp/pmc_$stop.log; exit 1; }
  echo "== stop=$stop" >> $OUT
  python3 - $stop >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(f"/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flagger_fused" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
cat $OUT
