#!/bin/bash
# Socket power and shader clock (rocm-smi) while ONE update kernel runs in a loop: which of them sit on the package power limit.
#   bash tools/kernel_power.sh      -> three lines per kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# (iteration counts sized so that the samples fall into the kernel's own loop: the forward probe times the training pass first)
for p in "dw_probe 3000 --no-gemm" "bwd_chain_probe 9000" "fwd_chain_probe 7000"; do
  set -- $p
  (timeout -k 10 90 python3 $R/tools/$1.py --rows 4194304 --iters $2 $3 > /dev/null 2>&1 &)
  sleep 12
  for i in 1 2 3; do
    echo "$1: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power|Socket Power" | sed 's/clock level//' | tr '\n' ' ' | cut -c1-220)"
    sleep 1.5
  done
  wait
  sleep 8
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
