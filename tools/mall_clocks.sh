#!/bin/bash
# effective shader clock (GRBM_GUI_ACTIVE summed over 8 XCDs / 8 / duration) of the update's three kernels, product build against
# the cache-resident-window probe builds (tools/mall_window_probe.sh) -> gpurun_out/r03/mall_clocks.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03/mall_clocks.txt
mkdir -p $(dirname $OUT); : > $OUT
cd /tmp && export TMPDIR=/tmp
for lib in product w64k_plain w16k_plain; do
  if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
  for p in dw bwd_chain fwd_chain; do
    extra=""; [ $p = fwd_chain ] && extra="--fused-head"; [ $p = dw ] && extra="--no-gemm"
    rm -rf /tmp/clk_$p
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$p -- python3 $R/tools/${p}_probe.py --rows 4194304 --iters 6 $extra > /tmp/clk_$p.log 2>&1
    python3 - /tmp/clk_$p $p $lib >> $OUT <<'PY'
import csv, glob, sys, collections
root, p, lib = sys.argv[1], sys.argv[2], sys.argv[3]
key = {"dw": "dw_kernel<256>", "bwd_chain": "mlp_bwd_chain_kernel<256", "fwd_chain": "mlp_fwd_chain_kernel<256, 8, true"}[p]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
us = sum(dur) / max(len(dur), 1)
print(lib, p, "launches", len(dur), "us", round(us), "clock_GHz", round(m.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3, 3), flush=True)
PY
  done
done
cat $OUT
