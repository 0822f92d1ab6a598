#!/bin/bash
# Round 3, second half (fused fp32 weight-gradient job, sampled launch events): C2 evidence on ONE box -- the bench line, round 2's
# path on the same box, the per-kernel stats of the same command, the chain-learner probe with and without the rebuilt operands.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 > $OUT/r03_bench_c2.json 2> $OUT/r03_bench_c2.err || exit 1
TG_F32_CHAIN=0 TG_FUSED_ADAM=0 timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_bench_c2_r02_path_same_box.json 2>/dev/null || exit 1
TG_F32_RECOMPUTE=0 timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_bench_c2_stored_operands_same_box.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --event-every 1 > $OUT/r03_bench_c2_all_launches_timed_same_box.json 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/f32_chain_probe.py --iters 30 > $OUT/r03_f32_chain_probe.jsonl 2>/dev/null || exit 1
TG_F32_RECOMPUTE=0 timeout -k 10 200 python3 tools/f32_chain_probe.py --iters 30 --no-gemm > $OUT/r03_f32_chain_probe_stored_operands.jsonl 2>/dev/null || exit 1
for f in r03_bench_c2 r03_bench_c2_r02_path_same_box r03_bench_c2_stored_operands_same_box r03_bench_c2_all_launches_timed_same_box; do
  python3 -c "import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms')"
done
TG_ALWAYS_REBUILD=1 timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_bench_c2_always_rebuild_same_box.json 2>/dev/null || exit 1
python3 -c "import json; d=json.load(open('$OUT/r03_bench_c2_always_rebuild_same_box.json')); print('always_rebuild', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms')"
bash $R/tools/profile_c2.sh | head -12
cp $OUT/bench_c2_kernel_stats.csv $OUT/r03_bench_c2_kernel_stats.csv
bash $R/tools/c2_gaps.sh > $OUT/r03_c2_gaps.txt 2>&1; head -12 $OUT/r03_c2_gaps.txt
