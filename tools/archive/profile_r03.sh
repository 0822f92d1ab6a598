#!/bin/bash
# Round-3 evidence (same recipe as profile_r03.sh): each learner kernel's probe at 2^22 rows under rocprofv3 --kernel-trace --stats and with the HBM counters
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes; read bytes = 2 x FETCH_SIZE on gfx950, MI355X_MICROARCH.md HBM section).
#   tools/profile_r03.sh [probe ...]     -> gpurun_out/r03_<probe>_{kernel_stats.csv,pmc.json,.json}
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBES=${@:-dw_probe bwd_chain_probe fwd_chain_probe}
for probe in $PROBES; do
  case $probe in
    dw_probe) args="--rows 4194304 --iters 5 --no-gemm"; kern="dw_kernel"; bpr=3184 ;;
    bwd_chain_probe) args="--rows 4194304 --iters 5"; kern="mlp_bwd_chain_kernel"; bpr=1776 ;;
    fwd_chain_probe) args="--rows 4194304 --iters 5 --fused-head"; kern="mlp_fwd_chain_kernel<256, 8, true"; bpr=1800 ;;
  esac
  python3 $R/tools/$probe.py $args > $OUT/r03_$probe.json 2> $OUT/r03_$probe.err
  rm -rf $OUT/p_$probe
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$probe/trace -- python3 $R/tools/$probe.py $args > $OUT/p_$probe.log 2>&1
  cp "$(find $OUT/p_$probe/trace -name '*kernel_stats.csv' | head -1)" $OUT/r03_${probe}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p_$probe/fetch -- python3 $R/tools/$probe.py $args > $OUT/p_$probe.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p_$probe/write -- python3 $R/tools/$probe.py $args > $OUT/p_$probe.log 2>&1
  python3 - "$OUT/p_$probe" "$kern" "$bpr" "$probe $args" "$OUT/r03_${probe}_kernel_stats.csv" > $OUT/r03_${probe}_pmc.json <<'PY'
import csv, glob, json, sys
root, kern, bpr, cmd, stats = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = 4194304
def mean(counter, sub):
    v = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
fetch, nf = mean("FETCH_SIZE", "fetch")
write, nw = mean("WRITE_SIZE", "write")
avg_us = calls = None
for r in csv.DictReader(open(stats)):
    if kern in r["Name"]:
        avg_us, calls = float(r["AverageNs"]) / 1e3, int(r["Calls"])
        break
out = {"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/{cmd}", "rows": rows, "kernel": kern,
       "FETCH_SIZE_KiB_per_launch_raw": fetch, "launches_fetch": nf, "WRITE_SIZE_KiB_per_launch_raw": write, "launches_write": nw}
if fetch is not None and write is not None:
    rd, wr = 2 * fetch * 1024, write * 1024                       # FETCH_SIZE counts 64 B per 128-B request on gfx950
    out.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
               algorithmic_bytes_per_launch=bpr * rows, traffic_over_algorithmic=(rd + wr) / (bpr * rows))
out.update(rocprof_avg_us=avg_us, rocprof_calls=calls)
json.dump(out, sys.stdout, indent=1)
PY
  rm -rf $OUT/p_$probe $OUT/p_$probe.log
  echo "$probe: $(cat $OUT/r03_$probe.json | cut -c1-400)"
  grep -E "traffic_over|rocprof_avg" $OUT/r03_${probe}_pmc.json
done
