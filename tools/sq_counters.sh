#!/bin/bash
# SQ occupancy / stall counters of one kernel probe, four counters per pass (each pass its own run):
#   bash tools/sq_counters.sh bwd_chain_probe mlp_bwd_chain_kernel     -> gpurun_out/sq_<probe>.json
#   PROBE_ARGS="--no-gemm --shapes 5:1:128x2 --rows 1048576 --iters 3" bash tools/sq_counters.sh f32_chain_probe "mlp_f32_res_kernel<128, 2, true>" "mlp_f32_res_kernel<128, 2, false>"
#   (several kernel-name patterns: one table per pattern)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PROBE=${1:-bwd_chain_probe}
KERNEL=${2:-mlp_bwd_chain_kernel}
PROBE_ARGS=${PROBE_ARGS:---rows 4194304 --iters 3}
shift; shift
OUT=$R/gpurun_out/sq_$PROBE
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/$PROBE.py $PROBE_ARGS > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$KERNEL" "$@" > $R/gpurun_out/sq_$PROBE.json <<'PY'
import csv, glob, json, sys, collections
out, kerns = sys.argv[1], sys.argv[2:]
acc = {k: collections.defaultdict(list) for k in kerns}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for kern in kerns:
            if kern in r["Kernel_Name"]:
                acc[kern][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {kern: {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(a.items())} for kern, a in acc.items()}
json.dump(res if len(kerns) > 1 else res[kerns[0]], sys.stdout, indent=1)
PY
rm -rf $OUT
echo done
