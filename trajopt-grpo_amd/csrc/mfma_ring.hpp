// Shared pieces of the persistent MFMA-chain kernels (fused_rollout.hip, mlp_fwd_chain.hip): vector types and the
// weight ring that streams A fragments L2 -> LDS by LDS-DMA.
//
// One weight block = the A fragments of one 32-row output tile for all k-steps (hidden layers and head), or of all
// output tiles of the first layer (K padded to 32 = 2 k-steps): always KS KiB = KS pieces of 1 KiB (one
// wave-instruction each).  The block stream is the same every pass (L2-resident) and flows into a ring of D slots with
// P = D - 1 blocks in flight (`global_load_lds_dwordx4`, no VGPR staging: the stream needs ~15-30 GB/s per CU, far more
// bytes in flight than two register sets can hold).  Per block: a COUNTED `s_waitcnt vmcnt` (this wave's pieces of
// the block have landed; the younger blocks stay in flight), a raw `s_barrier` (everyone's pieces have landed, and
// everyone has finished reading the slot that is about to be refilled), then the DMA for block +P is issued.
// `__syncthreads()` would drain the ring (its fence waits vmcnt(0)).
//
// Two compiler behaviours to design around (hipcc, ROCm 7.2):
//   * vector-memory operations retire in issue order, stores included: the counted wait must allow for every
//     vector-memory operation issued behind the block it waits for (the caller passes that count);
//   * an LDS read WITHOUT alias-scope metadata is made to wait for every outstanding LDS-DMA (vmcnt(0)): LDS reads
//     inside the ring loop go through inlined helpers with `__restrict__` parameters (which attaches the metadata),
//     or are opaque inline assembly.
#pragma once
#include "tg_common.hpp"

namespace tg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// Cache policy of the chain kernels' activation / dZ stores.  Default: non-temporal (written once, read by a later kernel from
// beyond the caches anyway at the learner's 4 M-row chunks).  -DTG_ACT_STORE_NT=0 builds the default-policy form for the
// cache-residency probe (tools/mall_probe.py).
#ifndef TG_ACT_STORE_NT
#define TG_ACT_STORE_NT 1
#endif
// Probe builds only (tools/mall_probe.py --window): -DTG_PROBE_ROW_WINDOW=W (a power of two) folds every row-indexed global
// address of the chain / weight-gradient kernels into the first W rows of its buffer.  The results are meaningless; the
// instruction stream and the bytes moved are those of the real launch, but the streams stay cache-resident: what the update
// would cost if its activation / dZ round trips were served on-die.  -DTG_DW_LOAD_AUX=0: default cache policy for the
// weight-gradient kernel's panel loads (the product build reads them non-temporal).
#ifndef TG_PROBE_ROW_WINDOW
#define TG_PROBE_ROW_WINDOW 0
#endif
#ifndef TG_DW_LOAD_AUX
#define TG_DW_LOAD_AUX 2
#endif
// Probe builds only (tools/build_probe_libs.sh `fusedbound`): -DTG_ABLATE_FUSED_CHAIN=1 takes out of the bf16 chain kernels exactly
// the memory traffic a single forward + loss + backward kernel would not have -- the forward chain's mask-bit and d loss / d output
// stores, the backward chain's loads of them and its second read of the input row.  Results are meaningless; the timing is an
// upper bound on what that fusion could gain, before its own costs.
#ifndef TG_TILED_STORE
#define TG_TILED_STORE 0
#endif
#ifndef TG_ABLATE_FUSED_CHAIN
#define TG_ABLATE_FUSED_CHAIN 0
#endif
// Probe builds only (tools/build_probe_libs.sh `headrelay1` / `headrelay2`): what the forward chain's head hand-off -- the top
// activation relayed block by block through shared LDS tiles behind eight workgroup barriers, so that the head's weight gradient
// is contracted on chip -- costs: 1 = the relay removed, 2 = the relay without its barriers.  Results meaningless.
#ifndef TG_ABLATE_HEAD_RELAY
#define TG_ABLATE_HEAD_RELAY 0
#endif
// Probe builds only (`chainvalu1..3`): how much of the bf16 chain kernels' time their vector instructions are -- bit 0: the forward
// chain's ReLU mask bits not formed (7 instructions per 8 activations), bit 1: ReLU's clamp of the packed pair left out (1 per pair).
// Results meaningless.  (profiles/r05_chain_valu_sensitivity.md)
#ifndef TG_ABLATE_CHAIN_VALU
#define TG_ABLATE_CHAIN_VALU 0
#endif
__device__ static inline int64_t mem_row(int64_t r) {
#if TG_PROBE_ROW_WINDOW
    return r & (int64_t)(TG_PROBE_ROW_WINDOW - 1);
#else
    return r;
#endif
}
typedef unsigned int act_u32x4 __attribute__((ext_vector_type(4)));
__device__ static inline void act_store16(act_u32x4 v, act_u32x4* p) {
#if TG_ACT_STORE_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ReLU + bf16 pack of 8 accumulators: round first (v_cvt_pk_bf16_f32, 2 per instruction), then clamp the PACKED halves
// with v_pk_max_i16(x, 0) -- a negative bf16 (and -0) is a negative int16.  relu(round(x)) == round(relu(x)); 4 + 4
// instructions instead of 8 v_med3_f32 + 4 conversions (A/B on one box: rollout 8.3 -> 8.1 ms, chain with stores -1.5 %).
__device__ static inline uint32_t relu_pack_bf16x2(float a, float b) {
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = __builtin_convertvector(f32x2{a, b}, bf16x2);          // ONE v_cvt_pk_bf16_f32
#if TG_ABLATE_CHAIN_VALU & 2
    return __builtin_bit_cast(uint32_t, p);                                 // (probe build: no clamp)
#endif
    const i16x2 zero = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, p), zero));
}
__device__ static inline bf16x8 relu_pack_bf16(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    const uint4 u = {relu_pack_bf16x2(a0, a1), relu_pack_bf16x2(a2, a3), relu_pack_bf16x2(a4, a5), relu_pack_bf16x2(a6, a7)};
    return __builtin_bit_cast(bf16x8, u);
}

// Masked epilogue of one 32-feature block of a backward-data product (tg_mlp_backward_chain; tg_mlp_weight_grad's kind RH rebuilds
// the top layer's dZ with the same instructions) for one of the lane's two rows: round the 2 x 4 accumulators pairwise (dword d =
// features 2 d, 2 d + 1 of the lane's 8) and multiply each 16-bit half by its keep bit (v_pk_mul_lo_u16).  `wsh` = the block
// pair's mask word already shifted right by the lane's nibble 4 (g & 1): feature pair d of block mt is bit (mt & 1) * 8 + d
// (even feature) and 16 + that (odd feature).
__device__ static inline bf16x8 masked_pack(const f32x4& lo, const f32x4& hi, uint32_t wsh, int mt) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t wk = wsh >> ((mt & 1) * 8);
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[2 * d], v[2 * d + 1]}, bf16x2));
        const u16x2 keep = __builtin_bit_cast(u16x2, (wk >> d) & 0x00010001u);
        o[d] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, pk) * keep);
    }
    return __builtin_bit_cast(bf16x8, uint4{o[0], o[1], o[2], o[3]});
}

template <int KS, int WPW>
__device__ static inline void ring_dma_block(const uint4* __restrict__ gblock, uint4* __restrict__ slot, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < KS / WPW; ++q) {
        const int piece = q * WPW + wave;
        __builtin_amdgcn_global_load_lds(gblock + piece * 64 + lane, (lds_void*)(slot + piece * 64), 16, 0, 0);
    }
}

// Consume the next block.  Expects in scope: wfrag, ring, n_blocks, wave, lane, the ring state (pre_pos, pre_slot,
// cur_slot) and the constants KS, WPW, D; declares `cur` (the block's fragments).  WAITN: see above.
#define TG_RING_WAIT(WAITN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
#define TG_RING_NEXT                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");                                                                         \
    ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);    \
    pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;                                                 \
    pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;                                                     \
    const uint4* cur = ring + cur_slot * KS * 64;                                                          \
    cur_slot = (cur_slot + 1 == D) ? 0 : cur_slot + 1;
#define TG_RING_ADVANCE(WAITN) TG_RING_WAIT(WAITN) TG_RING_NEXT

}  // namespace tg
