#!/bin/bash
# Diagnostic: LDS and wait counters of the ring kernel (product library).
# Output: gpurun_out/pmc_ring_lds.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ring_lds.txt
: > $OUT
export N=3 PAD=${PAD:-16}
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32"; do
  rm -rf /tmp/pmc_ring
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_ring -o p --output-format csv -- python3 $R/tools/run_fused.py > /tmp/pmc_ring.log 2>&1 || { tail -5 /tmp/pmc_ring.log >> $OUT; continue; }
  python3 - >> $OUT <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_ring/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flagger_ring" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
done
cat $OUT
