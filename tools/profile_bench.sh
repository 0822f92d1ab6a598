#!/bin/bash
# The bench under rocprofv3 (--kernel-trace --stats): per-kernel time of the whole step.
#   tools/profile_bench.sh <tag> [bench args]    -> gpurun_out/<tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
shift || true
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err
cp "$(find $OUT/${TAG}_prof -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
head -25 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-200
