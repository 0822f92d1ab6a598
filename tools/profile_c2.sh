#!/bin/bash
# per-kernel time of the C2 step (CartPole GRPO, 4,096 envs, fp32 5-128-128-1): rocprofv3 kernel trace of bench.py --config c2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2prof -- python3 $R/bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/bench_c2_under_rocprof.err
cp $(find /tmp/c2prof -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python3 - $OUT/bench_c2_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (25 steps incl. warmup): %.1f us" % (tot / 25e3))
for r in rows[:45]:
    print("%8.1f us/step %6d calls %9.1f avg_ns  %s" % (float(r["TotalDurationNs"]) / 25e3, int(r["Calls"]), float(r["AverageNs"]), r["Name"][:110]))
PY
