#!/bin/bash
# Diagnostic: dynamic instruction counters of the ring kernel for library variants
# (build/variants/lib_<name>.so, e.g. the RING_STOP builds). usage: LIBS="stop2 stop3 stop0" tools/pmc_ring.sh
# Output: gpurun_out/pmc_ring.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ring.txt
: > $OUT
export N=3 PAD=${PAD:-16}
for name in ${LIBS:-stop2 stop3 stop0}; do
  export KSP_LIB=$R/build/variants/lib_$name.so
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD"; do
    rm -rf /tmp/pmc_ring
    rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_ring -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pmc_ring.log 2>&1 || { tail -5 /tmp/pmc_ring.log; exit 1; }
    echo "== $name" >> $OUT
    python3 - >> $OUT <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_ring/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flagger_ring" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
  done
done
cat $OUT
