#!/bin/bash
# Profiling recipe for the learner after tg_dx_relu_bias: whole-bench kernel times, then the kernel's HBM
# counters (FETCH_SIZE and WRITE_SIZE in separate passes) on tools/dx_kernel_probe.py at the learner's chunk size.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_dx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/probe -- python3 $R/tools/dx_kernel_probe.py --rows 4194304 --iters 5 > $OUT/probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -- python3 $R/tools/dx_kernel_probe.py --rows 4194304 --iters 5 > $OUT/probe_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/probe_write -- python3 $R/tools/dx_kernel_probe.py --rows 4194304 --iters 5 > $OUT/probe_write.log 2>&1
echo done
