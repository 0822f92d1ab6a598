#!/bin/bash
# How much of the bf16 forward chain's time its vector instructions are: the product build against probe builds with part of them
# removed (tools/build_probe_libs.sh `chainvalu1..3`: 1 = no ReLU mask bits (14 of ~40 vector instructions per 32-feature block),
# 2 = no clamp in the ReLU pack (8), 3 = both); tg_mlp_forward_chain_loss (the learner's training pass) and the no-grad pass at 2^22 rows.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in product chainvalu1 chainvalu2 chainvalu3 product; do
  if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
  a=$(python3 tools/fwd_chain_probe.py --rows 4194304 --iters 5 --fused-head 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())[0]; print('%.1f us (%.0f GB/s)' % (d['chain_keep_us'], d['chain_keep_GBps']))")
  b=$(python3 tools/fwd_chain_probe.py --rows 4194304 --iters 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read())[0]; print('keep %.1f us, no-grad %.1f us' % (d['chain_keep_us'], d['chain_nokeep_us']))")
  echo "$lib: training pass $a; plain chain $b"
done
