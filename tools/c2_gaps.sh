#!/bin/bash
# GPU idle time inside the C2 step: rocprofv3 kernel trace of bench.py --config c2, busy time vs wall span of the timed steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c2gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2gaps -- python3 $R/bench.py --config ${CONFIG:-c2} --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-launch-events > $OUT/c2_gaps_bench.json 2> $OUT/c2_gaps.err
python3 - "$(find /tmp/c2gaps -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps: split at the fused rollout kernel
starts = [i for i, e in enumerate(ev) if "fused_rollout" in e[2]]
print("rollout launches", len(starts))
tot_span = tot_busy = 0
gaps = {}
for a, b in zip(starts[6:-1], starts[7:]):
    seg = ev[a:b]
    span = seg[-1][1] - seg[0][0] + 0
    # span to the next rollout's start
    span = ev[b][0] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    tot_span += span; tot_busy += busy
    prev_end = seg[0][1]
    for e in seg[1:] + [ev[b]]:
        g = e[0] - prev_end
        if g > 5000:
            key = e[2][:60]
            gaps[key] = gaps.get(key, 0) + g
        prev_end = max(prev_end, e[1])
n = len(starts) - 7
print("steps %d: span %.3f ms/step, kernels busy %.3f ms/step, idle %.3f ms/step" % (n, tot_span / n / 1e6, tot_busy / n / 1e6, (tot_span - tot_busy) / n / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:14]:
    print("  idle before %-60s %.1f us/step" % (k, v / n / 1e3))
a, b = starts[12], starts[13]
t0 = ev[a][0]
prev_end = ev[a][0]
print("one step, kernel by kernel (start us, gap us, duration us):")
for e in ev[a:b + 1]:
    g = (e[0] - prev_end) / 1e3
    d = (e[1] - e[0]) / 1e3
    if g > float(__import__("os").environ.get("GAP_US", "4")) or d > float(__import__("os").environ.get("DUR_US", "20")):
        print("  %8.1f  gap %6.1f  dur %7.1f  %s" % ((e[0] - t0) / 1e3, g, d, e[2][:70]))
    prev_end = max(prev_end, e[1])
PY
