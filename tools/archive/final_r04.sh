#!/bin/bash
# End-of-round evidence, round 4: smoke, the driver-style bench line (with the in-kernel clocks, socket power and the sustained
# matrix rate in it), C2 / C4 / C5 / fp32-policy bench lines, the C3 and C2 benches under rocprofv3 --kernel-trace --stats, the
# launch-by-launch trace of a C2 step, the count of torch (at::native) launches per C3 step.  Outputs under gpurun_out/r04/ with the
# names they are committed with in profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_final_bench.json 2> $OUT/r04_final_bench.err || exit 1
echo "bench done"
timeout -k 10 300 python3 bench.py --config c2 --steps 40 --warmup 10 > $OUT/r04_bench_c2.json 2> $OUT/r04_bench_c2.err || exit 1
for c in c4 c5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 5 > $OUT/r04_bench_$c.json 2> $OUT/r04_bench_$c.err || exit 1
done
timeout -k 10 400 python3 bench.py --policy-dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r04_bench_fp32_policy.json 2> $OUT/r04_bench_fp32_policy.err || exit 1
echo "configs done"
# same-box A/Bs of the round's switches on C2
for v in "TG_NATIVE_PREPARE=0" "TG_FOLD_OLD_LOGP=0" "TG_ADAM_PUSH=0" "TG_TRUST_VERSION_KEYS=1" "TG_ADAM_RIDER=0" "TG_F32DW_PIPE=0" "TG_F32DW_FUSED8=0" \
         "TG_NATIVE_PREPARE=0 TG_FOLD_OLD_LOGP=0 TG_ADAM_PUSH=0 TG_TRUST_VERSION_KEYS=1 TG_ADAM_RIDER=0 TG_F32DW_FUSED8=0"; do
  env $v timeout -k 10 200 python3 bench.py --config c2 --steps 60 --warmup 10 --no-launch-events --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 [$v]', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms')"
done | tee $OUT/r04_c2_switches_same_box.txt
timeout -k 10 200 python3 bench.py --config c2 --steps 60 --warmup 10 --no-launch-events --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 [product]', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],3), 'ms')" | tee -a $OUT/r04_c2_switches_same_box.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fixed-work > $OUT/r04_final_bench_under_rocprof.json 2> $OUT/bench_prof.err
cp "$(find /tmp/bench_prof -name '*kernel_stats.csv' | head -1)" $OUT/r04_final_bench_kernel_stats.csv
python3 - "$(find /tmp/bench_prof -name '*kernel_trace.csv' | head -1)" > $OUT/r04_c3_launch_counts.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "fused_rollout_kernel" in e[1]]
# the last full step of the timed region: from one fused rollout launch to the next
a, b = starts[-4], starts[-3]
seg = [n for _, n in ev[a:b]]
c = collections.Counter("at::native" if "at::native" in n else ("rocclr" if "rocclr" in n else ("tg::" if "tg::" in n else "other")) for n in seg)
print("one C3 step (rollout launch to rollout launch):", len(seg), "launches:", dict(c))
top = collections.Counter(n[:70] for n in seg if "at::native" in n)
for k, v in top.most_common(12):
    print("  %3d  %s" % (v, k))
PY
cat $OUT/r04_c3_launch_counts.txt | head -5
rm -rf /tmp/bench_prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof_c2 -- python3 $R/bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04_bench_c2_under_rocprof.json 2> $OUT/bench_prof_c2.err
cp "$(find /tmp/bench_prof_c2 -name '*kernel_stats.csv' | head -1)" $OUT/r04_bench_c2_kernel_stats.csv
GAP_US=-1 DUR_US=-1 CONFIG=c2 bash $R/tools/c2_gaps.sh > $OUT/r04_c2_gaps.txt 2>&1; head -12 $OUT/r04_c2_gaps.txt
echo done
