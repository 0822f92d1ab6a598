#!/bin/bash
# Where the resident fp32 H = 128 kernel's time goes (5-128-128-1 at C2's row count and at 2^20 rows): the product build against
# timing-only probe builds (-DTG_F32R_ABLATE bits: 1 = no activation / dZ / mask stores, 2 = no matrix products in the H x H tiles,
# 4 = no LDS reads of their weights, 8 = no head / loss arithmetic).  Build them first: tools/build_probe_libs.sh.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in product f32rabl1 f32rabl2 f32rabl4 f32rabl6 f32rabl8 f32rabl9 f32rabl15 product; do
  if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
  python3 tools/f32_chain_probe.py --no-gemm --shapes 5:1:128x2 --rows 176584 1048576 --iters 30 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib: rows %8d  no-grad forward %6.1f us, forward + loss + backward %6.1f us' % (d['rows'], d['forward_us'], d['forward_backward_us']))"
done
