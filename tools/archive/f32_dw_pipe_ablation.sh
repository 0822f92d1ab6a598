#!/bin/bash
# Timing-only ablations of the one-barrier 8-wave weight-gradient job (probe builds -DTG_F32DW_ABLATE=3/4/5 in scratch/, results
# meaningless): what the stage loop costs without the rebuild, without the products, without the DMA, with a third of the rebuild's vector instructions gone (6).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in product:"" norebuild:$R/scratch/libtg_p8abl3.so noproducts:$R/scratch/libtg_p8abl4.so nodma:$R/scratch/libtg_p8abl5.so lessvalu:$R/scratch/libtg_p8abl6.so; do
  name=${v%%:*}; lib=${v#*:}
  rm -rf /tmp/dwa_$name
  if [ -n "$lib" ]; then export TG_NATIVE_LIB=$lib; else unset TG_NATIVE_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dwa_$name -- python3 $R/tools/f32_dw_fused_probe.py > /tmp/dwa_$name.log 2>&1
  python3 - "$name" "$(find /tmp/dwa_$name -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "f32_dw_fused8" in r["Name"]:
        print(sys.argv[1], "avg %.1f us  min %.1f  max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
