#!/bin/bash
# The update's three kernels at 2^22 rows per launch, product build against probe builds whose row addresses are folded into a
# cache-resident window (mfma_ring.hpp: TG_PROBE_ROW_WINDOW).  One JSON line per (build, kernel); two passes in ABAB order.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03/mall_window.jsonl
mkdir -p $(dirname $OUT); : > $OUT
for pass in 1 2; do
  for lib in product w64k_plain w16k_plain w64k_nt nowin_plain; do
    if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
    for probe in "fwd_chain_probe.py --fused-head --rows 4194304 --iters 20" "bwd_chain_probe.py --rows 4194304 --iters 20" "dw_probe.py --rows 4194304 --iters 20 --no-gemm"; do
      echo -n "{\"pass\": $pass, \"lib\": \"$lib\", \"probe\": \"${probe%% *}\", \"result\": " >> $OUT
      timeout -k 10 120 python3 $R/tools/$probe 2>/dev/null | tail -1 >> $OUT || exit 1
      echo "}" >> $OUT
    done
  done
done
echo done
