#!/bin/bash
# r03: the stand-alone dynamics kernel (tg_rollout_step, 16-B mean rows) at 65,536 and 4,194,304 envs: duration (kernel trace),
# then FETCH_SIZE and WRITE_SIZE in their own passes -> gpurun_out/r03/step_kernel_<n>_{kernel_stats.csv,pmc.json}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 65536 4194304; do
  rm -rf /tmp/step_$n /tmp/step_${n}_fetch /tmp/step_${n}_write
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_$n -- python3 $R/tools/step_kernel_probe.py $n > $OUT/step_$n.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/step_${n}_fetch -- python3 $R/tools/step_kernel_probe.py $n > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/step_${n}_write -- python3 $R/tools/step_kernel_probe.py $n > /dev/null 2>&1
  cp $(find /tmp/step_$n -name "*kernel_stats.csv" | head -1) $OUT/step_kernel_${n}_kernel_stats.csv
  python3 - $n /tmp/step_${n}_fetch /tmp/step_${n}_write > $OUT/step_kernel_${n}_pmc.json <<'PY'
import csv, glob, json, sys
n, fetch_dir, write_dir = int(sys.argv[1]), sys.argv[2], sys.argv[3]
def col(d, name):
    v = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return v
f, w = col(fetch_dir, "FETCH_SIZE"), col(write_dir, "WRITE_SIZE")
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
mf, mw = sum(f) / len(f), sum(w) / len(w)
print(json.dumps({"n_envs": n, "FETCH_SIZE": {"launches": len(f), "mean_KB": mf, "min_KB": min(f), "max_KB": max(f)},
                  "WRITE_SIZE": {"launches": len(w), "mean_KB": mw, "min_KB": min(w), "max_KB": max(w)},
                  "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/step_kernel_probe.py (16-B mean rows); "
                          "MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request on gfx950 -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact",
                  "hbm_bytes_per_launch": (2 * mf + mw) * 1024.0,
                  "algorithmic_bytes_per_launch": 189 * n,
                  "expected_bytes_per_launch": (80 + 16 + 4 + 80 + 16 + 4 + 1) * n}, indent=1))
PY
done
tail -2 $OUT/step_65536.log $OUT/step_4194304.log
