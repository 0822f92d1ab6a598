#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats) of the fp32 weight-gradient launch pair at C2's size, one line per variant:
# the one-barrier 8-wave job, the two-barrier 8-wave job, the 4-wave job.  (tools/f32_dw_fused_probe.py's event timing is host-bound
# below ~114 us per call.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "pipe8:TG_F32DW_PIPE=1" "fused8:TG_F32DW_PIPE=0" "fused4:TG_F32DW_FUSED8=0"; do
  name=${v%%:*}; kv=${v#*:}
  rm -rf /tmp/dwk_$name
  export $kv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dwk_$name -- python3 $R/tools/f32_dw_fused_probe.py > /tmp/dwk_$name.log 2>&1
  unset ${kv%%=*}
  python3 - "$name" "$(find /tmp/dwk_$name -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "f32_dw" in r["Name"]:
        print(sys.argv[1], r["Name"][:60], "calls", r["Calls"], "avg %.1f us  min %.1f  max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
