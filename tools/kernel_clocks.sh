#!/bin/bash
# effective shader clock of the three update kernels: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for p in dw bwd_chain fwd_chain; do
  rm -rf /tmp/clk_$p
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$p -- python3 $R/tools/${p}_probe.py --rows 4194304 --iters 6 > /tmp/clk_$p.log 2>&1
  python3 - /tmp/clk_$p $p <<'PY'
import csv, glob, sys, collections
root, p = sys.argv[1], sys.argv[2]
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
print(p, "launches", len(dur), "us", round(us), "clock_GHz", round(m.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3, 3), flush=True)
PY
done
