#!/bin/bash
# Where the fp32 H = 256 chain kernel's time goes: the product build against timing-only probe builds (-DTG_F32W_ABLATE bits: 1 = no block
# barrier, 2 = no weight DMA inside the rounds, 4 = no activation / dZ stores), tg_mlp_f32w_forward_backward at 2^20 rows of 20-256x5-4.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in product f32wabl1 f32wabl2 f32wabl4 f32wabl7 product; do
  if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
  python3 tools/f32_h256_probe.py --rows 1048576 --iters 5 --wide-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib: chain %.3f ms (%.1f TFLOP/s), no-grad forward %.3f ms' % (d['chain_ms'], 1126.6e9/d['chain_ms']/1e9 if False else (2*(20*256+4*65536+1024)+2*(4*65536+1024))*1048576/d['chain_ms']/1e9, d['nograd_ms']))"
done
# ... and of the wide weight-gradient job (four 256 x 256 layers at 2^20 rows): bit 3 = no stage barrier, bit 4 = no DMA inside the stage loop
for lib in product f32wabl8 f32wabl16 f32wabl24 product; do
  if [ $lib = product ]; then unset TG_NATIVE_LIB; else export TG_NATIVE_LIB=$R/scratch/libtg_$lib.so; fi
  echo -n "$lib: "; H=256 JOBS=mm,mm,mm,mm python3 tools/f32_dw_jobs_probe.py 2>&1 | grep -v amdgpu.ids
done
