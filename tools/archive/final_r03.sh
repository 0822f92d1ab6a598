#!/bin/bash
# End-of-round evidence, round 3: smoke, the driver-style bench line, the same command under rocprofv3 --kernel-trace --stats,
# C2 / C4 / C5 bench lines.  Outputs under gpurun_out/r03/ with the names they are committed with in profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_final_bench.json 2> $OUT/r03_final_bench.err || exit 1
echo "bench done"
for c in c2 c4 c5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 5 > $OUT/r03_bench_$c.json 2> $OUT/r03_bench_$c.err || exit 1
done
echo "configs done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fixed-work > $OUT/r03_final_bench_under_rocprof.json 2> $OUT/bench_prof.err
cp "$(find /tmp/bench_prof -name '*kernel_stats.csv' | head -1)" $OUT/r03_final_bench_kernel_stats.csv
echo done
