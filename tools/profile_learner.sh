#!/bin/bash
# Profiling recipe for the learner kernels (tg_mlp_forward_chain, tg_mlp_backward_chain, tg_dx_relu_bias): whole-bench kernel times, then each
# kernel's probe under --kernel-trace and its HBM counters (FETCH_SIZE and WRITE_SIZE in separate passes).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_learner
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
for probe in dx_kernel_probe fwd_chain_probe bwd_chain_probe; do
  extra=""
  if [ $probe = dx_kernel_probe ]; then extra="--bits"; fi      # the learner feeds the kernel 1-bit ReLU masks
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$probe -- python3 $R/tools/$probe.py --rows 4194304 --iters 5 $extra > $OUT/$probe.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${probe}_fetch -- python3 $R/tools/$probe.py --rows 4194304 --iters 5 $extra > $OUT/${probe}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${probe}_write -- python3 $R/tools/$probe.py --rows 4194304 --iters 5 $extra > $OUT/${probe}_write.log 2>&1
done
echo done
