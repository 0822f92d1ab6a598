#!/bin/bash
# HBM traffic of the fp32 chain learner's two kernels (5-128-128-1, 1,048,576 rows): FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes (read bytes = 2 x FETCH_SIZE on gfx950, MI355X_MICROARCH.md HBM section) against the algorithmic bytes per row.
# Round 5 (the resident 16-row kernel in front of the 8-wave weight-gradient job):
#   TAG=r05 CHAIN_KERNEL="mlp_f32_res_kernel<128, 2, true>" DW_KERNEL="mlp_f32_dw_fused8_kernel<128" bash tools/profile_f32_pmc.sh
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${TAG:-r03}
export CHAIN_KERNEL=${CHAIN_KERNEL:-mlp_f32_chain_kernel<128, true>}
export DW_KERNEL=${DW_KERNEL:-mlp_f32_dw_kernel<128>}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--rows 1048576 --iters 5 --no-gemm --shapes 5:1:128x2"
rm -rf /tmp/f32pmc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f32pmc/trace -- python3 $R/tools/f32_chain_probe.py $ARGS > /tmp/f32pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/f32pmc/fetch -- python3 $R/tools/f32_chain_probe.py $ARGS > /tmp/f32pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/f32pmc/write -- python3 $R/tools/f32_chain_probe.py $ARGS > /tmp/f32pmc.log 2>&1
python3 - /tmp/f32pmc > $OUT/${TAG}_f32_chain_pmc.json <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
rows = 1048576
def mean(kern, counter, sub):
    v = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
stats = {}
for f in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/f32_chain_probe.py --rows 1048576 --iters 5 --no-gemm --shapes 5:1:128x2",
       "rows": rows, "kernels": {}}
# algorithmic bytes per row, 5-128-128-1 with rebuilt operands: forward + loss + backward writes the second activation and the bottom
# dZ (512 B each), d loss / d output (16) and the top mask bits (16), reads the padded input (32) and three loss inputs (12);
# the weight gradients read those two matrices, the input, d loss / d output and the mask bits (+ the slabs, below)
for name, kern, bpr in (("forward_backward", os.environ["CHAIN_KERNEL"], 512 + 512 + 16 + 16 + 32 + 12),
                        ("weight_grad", os.environ["DW_KERNEL"], 512 + 512 + 32 + 16 + 16),
                        ("weight_grad_reduction", "mlp_f32_dw_finish_kernel", 0)):
    fetch, nf = mean(kern, "FETCH_SIZE", "fetch")
    write, nw = mean(kern, "WRITE_SIZE", "write")
    k = {"kernel": kern, "FETCH_SIZE_KiB_per_launch_raw": fetch, "WRITE_SIZE_KiB_per_launch_raw": write, "launches": [nf, nw]}
    if fetch is not None and write is not None:
        rd, wr = 2 * fetch * 1024, write * 1024
        k.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr)
        if bpr:
            k.update(algorithmic_bytes_per_row=bpr, algorithmic_bytes_per_launch=bpr * rows, traffic_over_algorithmic=(rd + wr) / (bpr * rows))
    for n_, (us, calls) in stats.items():
        if kern in n_:
            k.update(rocprof_avg_us=us, rocprof_calls=calls)
    out["kernels"][name] = k
out["note"] = ("weight_grad also writes 512 slabs x 85 KB = 43.5 MB per launch and the reduction reads them (the windows it needs: 35.6 MB): "
               "both appear in the measured traffic, not in the per-row figure")
json.dump(out, sys.stdout, indent=1)
PY
cat $OUT/${TAG}_f32_chain_pmc.json | head -60
