#!/bin/bash
# End-of-round evidence with the final kernels: the plain bench line, the same command under rocprofv3
# (--kernel-trace --stats), and the fp32 fused rollout probe under rocprofv3.  Summaries land in gpurun_out/final/
# under the names they are committed with in profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/r01_final_bench.json 2> $OUT/bench.err
echo "bench done" && tail -c 400 $OUT/r01_final_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r01_final_bench_under_rocprof.json 2> $OUT/bench_prof.err
cp "$(find $OUT/bench_prof -name '*kernel_stats.csv' | head -1)" $OUT/r01_final_bench_kernel_stats.csv
echo "bench profile done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32_prof -- python3 $R/tools/fused_f32_probe.py --iters 2 > $OUT/r01_fused_f32_probe_under_rocprof.jsonl 2> $OUT/f32_prof.err
cp "$(find $OUT/f32_prof -name '*kernel_stats.csv' | head -1)" $OUT/r01_fused_f32_probe_kernel_stats.csv
rm -rf $OUT/bench_prof $OUT/f32_prof
echo done
