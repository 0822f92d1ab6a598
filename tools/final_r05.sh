#!/bin/bash
# End-of-round evidence, round 5 (outputs under gpurun_out/r05/ with the names they are committed with in profiles/):
#   part "bench": smoke, the driver-style bench line (C3 headline + roofline.other_configs = C2 / C4 / C5), the same command under
#                 rocprofv3 --kernel-trace --stats, the launch counts of one C3 step;
#   part "fp32":  C3 with the reference's own precision (--policy-dtype fp32: the H = 256 fp32 chain learner), plain and under rocprofv3
#                 (the top of the kernel table must hold no library GEMM), the stand-alone probe against hipBLASLt;
#   part "ppo":   PPO at the reference factory's size, this tree against the unpacked round-4 tree (scratch/r04_tree);
#   part "pmc":   HBM traffic of the two H = 256 fp32 kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
# usage: tools/final_r05.sh [bench|fp32|ppo|pmc ...]   (default: all)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
PARTS=${@:-bench fp32 ppo pmc}
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has bench; then
  timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
  timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_final_bench.json 2> $OUT/r05_final_bench.err || exit 1
  echo "bench done"
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/bench_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fixed-work --other-configs off > $OUT/r05_final_bench_under_rocprof.json 2> $OUT/bench_prof.err
  cp "$(find /tmp/bench_prof -name '*kernel_stats.csv' | head -1)" $OUT/r05_final_bench_kernel_stats.csv
  python3 - "$(find /tmp/bench_prof -name '*kernel_trace.csv' | head -1)" > $OUT/r05_c3_launch_counts.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "fused_rollout_kernel" in e[1]]
a, b = starts[-4], starts[-3]            # the last full step of the timed region: from one fused rollout launch to the next
seg = [n for _, n in ev[a:b]]
c = collections.Counter("at::native" if "at::native" in n else ("rocclr" if "rocclr" in n else ("tg::" if "tg::" in n else "other")) for n in seg)
print("one C3 step (rollout launch to rollout launch):", len(seg), "launches:", dict(c))
top = collections.Counter(n[:90] for n in seg if "tg::" not in n)
for k, v in top.most_common(20):
    print("  %3d  %s" % (v, k))
PY
  head -8 $OUT/r05_c3_launch_counts.txt
  cd $R
fi

if has fp32; then
  timeout -k 10 500 python3 bench.py --policy-dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r05_bench_fp32_policy.json 2> $OUT/r05_bench_fp32_policy.err || exit 1
  python3 -c "import json; d=json.load(open('$OUT/r05_bench_fp32_policy.json')); print('fp32 policy:', round(d['value']/1e6,3), 'M env-steps/s,', round(d['update_ns_per_valid_row'],1), 'ns of update per valid row')"
  timeout -k 10 300 python3 tools/f32_h256_probe.py --rows 1048576 --iters 10 > $OUT/r05_f32_h256_probe.jsonl 2> $OUT/probe.err || exit 1
  timeout -k 10 300 python3 tools/f32_h256_probe.py --rows 1048576 --iters 10 --out-dim 1 >> $OUT/r05_f32_h256_probe.jsonl 2>> $OUT/probe.err || exit 1
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/bench_prof_f32
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof_f32 -- python3 $R/bench.py --policy-dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-work > $OUT/r05_bench_fp32_policy_under_rocprof.json 2> $OUT/bench_prof_f32.err
  cp "$(find /tmp/bench_prof_f32 -name '*kernel_stats.csv' | head -1)" $OUT/r05_bench_fp32_policy_kernel_stats.csv
  head -8 $OUT/r05_bench_fp32_policy_kernel_stats.csv
  cd $R
fi

if has ppo; then
  for tree in $R/scratch/r04_tree $R; do
    for mode in "" "--gae"; do
      timeout -k 10 200 python3 tools/ppo_factory_epoch.py --tree $tree --epochs 60 $mode 2>> $OUT/ppo.err
    done
  done | tee $OUT/r05_ppo_factory_epoch.jsonl
fi

if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/f32_h256_probe.py --rows 1048576 --iters 2 --wide-only > /dev/null 2> $OUT/pmc_$c.err
    cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $OUT/r05_f32_wide_pmc_$c.csv
  done
  python3 - $OUT > $OUT/r05_f32_wide_pmc.json <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/r05_f32_wide_pmc_{c}.csv")):
        if r["Counter_Name"] == c and "mlp_f32_wide" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        res[k][c] = sum(v) / len(v)
rows = 1048576
o = {"rows": rows, "note": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes), KiB per launch averaged over the probe's launches; read bytes = 2 x FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM)", "kernels": {}}
for k, v in res.items():
    rd, wr = 2 * v.get("FETCH_SIZE", 0.0) * 1024, v.get("WRITE_SIZE", 0.0) * 1024
    o["kernels"][k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr, "bytes_per_row": (rd + wr) / rows}
print(json.dumps(o, indent=1))
PY
  cat $OUT/r05_f32_wide_pmc.json | head -30
  cd $R
fi
echo done
