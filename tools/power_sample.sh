#!/bin/bash
# Socket power and shader clock while the bench runs (the update sits at the 1400 W package limit):
#   bash tools/power_sample.sh        -> 8 samples of rocm-smi during a 25-step bench run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
(timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --steps 25 --warmup 1 > /dev/null 2>&1 &)
sleep 22
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed "s/clock level//" | tr "\n" " " | cut -c1-300
  echo
  sleep 1.5
done
wait
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
