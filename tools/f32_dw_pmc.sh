#!/bin/bash
# SQ counters of the fp32 weight-gradient kernel at C2's size (tools/f32_dw_fused_probe.py), a few per pass, --pmc only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${ROUND:-r04b}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
wc -l $OUT/sq_counters.txt
n=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  n=$((n+1)); rm -rf /tmp/dwpmc_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/dwpmc_$n -- python3 $R/tools/f32_dw_fused_probe.py > /tmp/dwpmc_$n.log 2>&1 || { echo "pass $n failed"; tail -3 /tmp/dwpmc_$n.log; continue; }
  python3 - "$(find /tmp/dwpmc_$n -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "f32_dw_fused8" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s per launch %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
