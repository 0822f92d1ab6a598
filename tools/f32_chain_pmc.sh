#!/bin/bash
# SQ counters of the fp32 chain kernel (forward + loss + backward data in one launch) at C2's size, a few per pass, --pmc only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$((n+1)); rm -rf /tmp/chpmc_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/chpmc_$n -- python3 $R/tools/f32_chain_probe.py --rows 176584 --iters 10 --no-gemm --shapes 5:1:128x2 > /tmp/chpmc_$n.log 2>&1 || { echo "pass $n failed"; tail -3 /tmp/chpmc_$n.log; continue; }
  python3 - "$(find /tmp/chpmc_$n -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_f32_chain_kernel<128, true>" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s per launch %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
