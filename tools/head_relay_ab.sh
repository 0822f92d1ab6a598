#!/bin/bash
# What the bf16 forward chain's head hand-off costs: the learner's training pass (tg_mlp_forward_chain_loss, 20-256x5-4,
# 2^22 rows) in the product build, without the relay, and with the relay minus its eight workgroup barriers; then the plain training
# pass that stores FOUR activations and has no head (tg_mlp_forward_chain, keep): the yardstick the verdict quoted.  Same box, same process order.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  for lib in product headrelay1 headrelay2; do
    if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
    python3 tools/fwd_chain_probe.py --rows 4194304 --iters 20 --fused-head 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[') or l.startswith('{'):
        d=json.loads(l); d=d[0] if isinstance(d,list) else d
        print('$lib rep $rep: forward + loss head %.1f us' % d['chain_keep_us'])"
  done
  unset TG_NATIVE_LIB
  python3 tools/fwd_chain_probe.py --rows 4194304 --iters 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[') or l.startswith('{'):
        d=json.loads(l); d=d[0] if isinstance(d,list) else d
        print('product rep $rep: plain training pass (all activations stored, no head) %.1f us' % d['chain_keep_us'])"
done
