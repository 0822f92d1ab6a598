#!/bin/bash
# Probe builds of the library (timing-only / diagnostic variants; never the product) into scratch/, selected at run time with
# TG_NATIVE_LIB=<path>.  tools/mall_window_probe.sh and tools/mall_clocks.sh expect the first four.
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/scratch
build() { make -C $R/trajopt-grpo_amd/csrc -j8 OUT=../../scratch/libtg_$1.so OBJDIR=../../build/obj_$1 EXTRA="$2" 2>&1 | grep -E "error|warning:" ; ls -la $R/scratch/libtg_$1.so; }
build w64k_plain  "-DTG_PROBE_ROW_WINDOW=65536 -DTG_ACT_STORE_NT=0 -DTG_DW_LOAD_AUX=0"     # update kernels' streams folded into a 64 K-row window
build w16k_plain  "-DTG_PROBE_ROW_WINDOW=16384 -DTG_ACT_STORE_NT=0 -DTG_DW_LOAD_AUX=0"
build w64k_nt     "-DTG_PROBE_ROW_WINDOW=65536"                                            # ... with the product's non-temporal policies
build nowin_plain "-DTG_ACT_STORE_NT=0 -DTG_DW_LOAD_AUX=0"                                 # default cache policies, no window
build dwstamps    "-DTG_F32DW_STAMPS=1"                                                    # fp32 weight-gradient kernel with s_memtime stamps
build fusedbound  "-DTG_ABLATE_FUSED_CHAIN=1"                                            # upper bound of a fused bf16 forward + loss + backward chain kernel
build tiledstore  "-DTG_TILED_STORE=1"                                                   # chain kernels store their tiles untransposed into a tiled layout (timing only)
for v in 3 4 5 6; do build p8abl$v "-DTG_F32DW_ABLATE=$v"; done                            # one-barrier 8-wave fp32 weight-gradient job: no rebuild / no products / no DMA / a third of the rebuild's vector instructions gone (tools/f32_dw_pipe_ablation.sh)
for v in 1 2; do build headrelay$v "-DTG_ABLATE_HEAD_RELAY=$v"; done                       # forward chain's head hand-off: removed / without its barriers (tools/head_relay_ab.sh)
for v in 1 2 4 7 8 16 24; do build f32wabl$v "-DTG_F32W_ABLATE=$v"; done                          # fp32 H = 256 chain kernel: no barrier / no DMA / no stores / none of the three (tools/f32_wide_ablation.sh)
for v in 1 2 4 6 8 9 15; do build f32rabl$v "-DTG_F32R_ABLATE=$v"; done                        # fp32 resident H = 128 kernel: no stores / no products / no weight reads / neither / no head + loss (tools/f32_res_ablation.sh)
for v in 1 2 3; do build chainvalu$v "-DTG_ABLATE_CHAIN_VALU=$v"; done                               # bf16 chain kernels with part of their vector instructions removed (sensitivity probe)
