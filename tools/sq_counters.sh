#!/bin/bash
# SQ occupancy / stall counters of one kernel probe, four counters per pass (each pass its own run):
#   bash tools/sq_counters.sh bwd_chain_probe mlp_bwd_chain_kernel     -> gpurun_out/sq_<probe>.json
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PROBE=${1:-bwd_chain_probe}
KERNEL=${2:-mlp_bwd_chain_kernel}
OUT=$R/gpurun_out/sq_$PROBE
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/$PROBE.py --rows 4194304 --iters 3 > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$KERNEL" > $R/gpurun_out/sq_$PROBE.json <<'PY'
import csv, glob, json, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(acc.items())}
json.dump(res, sys.stdout, indent=1)
PY
rm -rf $OUT
echo done
