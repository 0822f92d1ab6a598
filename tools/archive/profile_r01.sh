#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun; outputs under gpurun_out/, summaries are
# then copied into profiles/ by hand).  Counter passes are separate from the timing pass.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. whole bench: per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
# 2. dynamics kernel alone at 65,536 and 4,194,304 envs: time, then FETCH_SIZE and WRITE_SIZE in their own passes
for n in 65536 4194304; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_$n -- python3 $R/tools/step_kernel_probe.py $n > $OUT/step_$n.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/step_${n}_fetch -- python3 $R/tools/step_kernel_probe.py $n > $OUT/step_${n}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/step_${n}_write -- python3 $R/tools/step_kernel_probe.py $n > $OUT/step_${n}_write.log 2>&1
done
echo done
