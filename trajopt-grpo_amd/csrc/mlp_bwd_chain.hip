// Backward-DATA pass of the reference's ReLU MLP (the dZ of every hidden layer, from the head down) for 10^6..10^7 rows
// in ONE persistent launch -- the mirror image of mlp_fwd_chain.hip.  Under torch autograd (algorithms/*.py
// `loss.backward()` through models/neural_network.py:48-66) every layer's dA is written, re-read, masked and written
// again; with tg_head_bwd_relu_bias + tg_dx_relu_bias each dZ is still re-read by the layer below (1.06 KB per row and
// layer).  Here a row's gradient stays in registers from the head to the first hidden layer:
//
//     dZ_top     = (dOut . W_head)    * (a_top  > 0)          dOut = d loss / d head output, [rows][8] bf16, zero padded
//     dZ_below   = (dZ . W_layer)     * (a_below > 0)         for every hidden-to-hidden layer, top down
//     partial[wg][layer][f] = sum over the workgroup's rows of dZ_layer[.][f]        (bias gradients)
//
// read 16 B (dOut) + 32 B of ReLU mask bits per layer, write 512 B per layer: 2.7 KB per row at 5 x 256 instead of 4.8.
// The dZ are written because the weight gradients (dW = dZ^T A, batched GEMMs) need them.
//
// Same machinery as the forward chain: transposed product with v_mfma_f32_16x16x32_bf16 (the shape the chip clocks highest
// on under its power limit), 2 x 16 rows per wave, the packed accumulators of a layer ARE the B operand of the layer below
// (keep-mask multiply instead of bias + ReLU), W^T streams L2 -> LDS through the LDS-DMA ring (mlp.FragmentStream(layout=
// "chain", transposed=True), head first).  The mask bits of a layer (1 KiB per wave) arrive by LDS-DMA two layers ahead,
// dOut one round ahead; neither is counted in the ring's waits (more operations behind a block only make its wait
// stricter).  Every second output block issues exactly four stores per wave, always (rows past the end are clamped to the
// last row and rewrite it with identical bytes), so the counted wait is the same at every site.  The top layer's dZ need not
// be written (dz[0] == NULL): it is a function of dOut and the mask bits, and tg_mlp_weight_grad (kind RH) rebuilds it on chip.
// Bias gradients: tg_mlp_weight_grad forms them inside its contraction; for callers that still ask for per-workgroup column
// sums (d_partial) a separate reduction kernel runs over the dZ just written (dz_colsum_kernel below).
#include "mfma_ring.hpp"

namespace tg {

constexpr int kBwdMaxLayers = 6;
TG_CLOCK_PROBE_VAR(g_probe_bwd_chain, attach_probe_bwd_chain)
struct BwdChainPtrs {
    uint16_t* dz[kBwdMaxLayers];            // outputs, top hidden layer first: bf16 [rows][H]
    const uint32_t* mask[kBwdMaxLayers];    // ReLU mask bits of the same layers (tg_mlp_forward_chain): u32 [rows][H/32]
    const uint4* x;                         // kFuse0: the net input, bf16 [rows][32] (column 31 = 1: the bias gradient rides along)
    float* w0_slabs;                        // kFuse0: f32 [grid][2][H][32] partial first-layer weight gradients
};

// ---- first layer's weight gradient inside the chain (kFuse0) ----
// dW0 = dZ1^T . x contracts over rows, which sit on LANES in this kernel, and a wave's own share of it (H x 32 fp32) would be 128
// registers.  So the eight waves of a workgroup split the OUTPUT: after every block of the last layer each wave leaves its
// 32 rows x 32 features (2 KiB, masked and rounded exactly as it would have been stored) in a shared LDS tile; one block later
// -- behind that block's ring barrier -- wave i multiplies the 16 features (i >> 1 & 1) x 16 inputs (i & 1) of that block over
// the rows of four of the eight waves (i >> 2), fragments through the transposing LDS read as in mlp_dw.hip: 4 MFMAs per wave and
// block (+ 12 % in the last layer), 4 accumulator registers per block.  dZ1 is never written (-512 B per row), the weight-gradient
// kernel loses its HX job (-576 B per row), and db0 is column 31 of dW0 because the caller put ones there.
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
__device__ static inline bf16x8 tr_frag16(const char* __restrict__ lo_p, const char* __restrict__ hi_p) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)lo_p);
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)hi_p);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ static inline void lds_store16(char* __restrict__ p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }


// dZ stores (as the forward chain's activation stores): after two 32-feature blocks a lane (col, g) holds, for each of its two
// rows, 2 x 16 B of the row's 128-B line; the wave transposes the 32 x 128 B through its LDS staging area (chunks XOR-swizzled
// by the row: conflict-free both ways) and each of its 4 store instructions writes 8 WHOLE 128-B lines, non-temporal
// (written once, read by the weight-gradient kernel).  `__restrict__`: alias scope (mfma_ring.hpp).
__device__ static inline void store_pair(uint4* __restrict__ st, uint16_t* __restrict__ g, int64_t row0, int64_t rows, int ld,
                                         int lane, const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
#if TG_TILED_STORE
    // Probe build (-DTG_TILED_STORE=1): the registers go out as they stand into a TILED layout [32-row tile][16-B feature chunk]
    // [row][16 B] -- every instruction writes four 256-B runs (whole 128-B lines), no LDS transpose.  Consumers do not read this
    // layout: timing only (what the epilogue's transpose costs).
    {
        const int col = lane & 15, grp = lane >> 4;
        const int64_t tile = row0 >> 5;
        // (g = buffer + first column of the block pair; buffers are 512-B aligned, ld a power of two: recover both)
        const int colofs = (int)(((uintptr_t)g >> 1) & (uintptr_t)(ld - 1));
        uint16_t* base = g - colofs;
        const int chunk0 = colofs / 8;                                                     // 16-B chunk of the pair's first block
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int row = 16 * c + col;
            uint16_t* ta = base + ((tile * (ld / 8) + chunk0 + grp) * 32 + row) * 8;
            uint16_t* tb = base + ((tile * (ld / 8) + chunk0 + 4 + grp) * 32 + row) * 8;
            act_store16(__builtin_bit_cast(act_u32x4, a[c]), reinterpret_cast<act_u32x4*>(ta));
            act_store16(__builtin_bit_cast(act_u32x4, b[c]), reinterpret_cast<act_u32x4*>(tb));
        }
        return;
    }
#endif
    const int col = lane & 15, grp = lane >> 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = 16 * c + col, sw = n & 7;
        st[n * 8 + ((grp + 0) ^ sw)] = __builtin_bit_cast(uint4, a[c]);
        st[n * 8 + ((grp + 4) ^ sw)] = __builtin_bit_cast(uint4, b[c]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3), ch = lane & 7;
        const uint4 v = st[r * 8 + (ch ^ (r & 7))];
        int64_t row = row0 + r;
        row = row < rows ? row : rows - 1; row = mem_row(row);
        act_store16(act_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<act_u32x4*>(g + row * ld + 8 * ch));
    }
}

__device__ static inline uint4 lds_read_b128_opaque(const uint4* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint4*)p;
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

template <int H, int WPW, bool kFuse0>
__global__ __launch_bounds__(64 * WPW, 2) void mlp_bwd_chain_kernel(const uint4* __restrict__ dzh, const uint4* __restrict__ wfrag,
                                                                    int32_t n_layers, int64_t rows, BwdChainPtrs ptrs) {
    constexpr int MT = H / 32, KS = H / 16, K8 = H / 32;
    // ring of 3 slots, 2 blocks in flight (4 / 3 measured the same; the LDS goes to the store staging instead).  Behind
    // the block a wait is for: the DMA of the one later block and the stores of the last two blocks, one of them odd
    // (4 stores; the head block's 16 only add to that).
    constexpr int D = 3, P = D - 1;
    constexpr int kWaitN = (P - 1) * (KS / WPW) + 4;
    static_assert(KS % WPW == 0, "every wave moves the same number of 1-KiB pieces per block");
    extern __shared__ uint4 lds[];
    uint4* ring = lds;                                                  // D * KS * 64 uint4
    uint4* dzs = lds + D * KS * 64;                                     // WPW waves * 64 uint4 (lanes 0..31 used)
    uint4* mks = dzs + WPW * 64;                                        // WPW waves * 3 buffers * 64 uint4
    uint4* stage = mks + WPW * 3 * 64 + (threadIdx.x >> 6) * (32 * 8);   // per wave: 32 rows x 128 B (store_pair)
    // kFuse0: the same 32 KiB seen as shared tiles T[2][WPW][2 KiB] (the last layer does not store), and behind them the
    // workgroup's input rows X[2][WPW][2 KiB] (double-buffered over rounds)
    char* tiles = reinterpret_cast<char*>(mks + WPW * 3 * 64);
    char* xtiles = tiles + 2 * WPW * 2048;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, col = lane & 15, nib = 4 * (grp & 1);
    const bool store_top = ptrs.dz[0] != nullptr;                      // null: tg_mlp_weight_grad (kind RH) rebuilds the top layer's dZ
    const int64_t n_rounds = (rows + 32 * WPW - 1) / (32 * WPW);
    const int n_blocks = (n_layers - 1) * MT + 1;
    constexpr int WPL = MT / 2;                                         // mask words per half-row and layer

    uint4* my_dzs = dzs + wave * 64;
    uint4* my_mks = mks + wave * 3 * 64;

    // kFuse0: this wave's 32 input rows of `round` (64 B each) into X[round parity][wave], in the tile image [row / 4][row % 4][64 B]
    // with the 16-B chunk c of a row in slot c ^ (row / 4 & 3) (bank-conflict-free for the transposing reads AND plain: mlp_dw.hip)
    [[maybe_unused]] auto dma_x0 = [&](int64_t round, int par) {
#if TG_ABLATE_FUSED_CHAIN
        return;
#endif
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            int64_t r = round * (32 * WPW) + wave * 32 + 16 * pc + (lane >> 2);
            r = r < rows ? r : rows - 1; r = mem_row(r);
            const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
            __builtin_amdgcn_global_load_lds(ptrs.x + r * 4 + chunk, (lds_void*)(xtiles + (par * WPW + wave) * 2048 + pc * 1024), 16, 0, 0);
        }
    };
    auto dma_dzh = [&](int64_t round) {
#if TG_ABLATE_FUSED_CHAIN
        return;
#endif
        if (lane < 32) {
            int64_t r = round * (32 * WPW) + wave * 32 + lane;
            r = r < rows ? r : rows - 1; r = mem_row(r);
            __builtin_amdgcn_global_load_lds(dzh + r, (lds_void*)my_dzs, 16, 0, 0);
        }
    };
    // mask bits of layer j for the 32 rows of `round`, row-major in LDS.  H = 256: lane L fetches the 16 B of (row L>>1, half
    // L&1); H = 128: a row's two halves are 16 B together and lanes 0..31 fetch one row each.
    static_assert(MT == 8 || MT == 4, "the mask staging moves 16 B per lane (H = 256 or 128)");
    auto dma_mask = [&](int64_t round, int j, int buf) {
#if TG_ABLATE_FUSED_CHAIN
        return;
#endif
        if constexpr (MT == 8) {
            int64_t r = round * (32 * WPW) + wave * 32 + (lane >> 1);
            r = r < rows ? r : rows - 1; r = mem_row(r);
            __builtin_amdgcn_global_load_lds(ptrs.mask[j] + r * MT + (lane & 1) * WPL, (lds_void*)(my_mks + buf * 64), 16, 0, 0);
        } else if (lane < 32) {
            int64_t r = round * (32 * WPW) + wave * 32 + lane;
            r = r < rows ? r : rows - 1; r = mem_row(r);
            __builtin_amdgcn_global_load_lds(ptrs.mask[j] + r * MT, (lds_void*)(my_mks + buf * 64), 16, 0, 0);
        }
    };

    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
    int mseq = 0;                                   // running layer number of this workgroup; its masks sit in buffer mseq % 3
#if TG_ABLATE_FUSED_CHAIN
    // (the probe build loads none of them: give the staging areas operands that look like data -- half the mask bits set, small
    // gradients, unit inputs -- zeros would multiply for free and let the package clock up)
    for (int q = threadIdx.x; q < WPW * 64; q += 64 * WPW) dzs[q] = uint4{0x3C003C00u, 0xBC003C00u, 0u, 0u};
    for (int q = threadIdx.x; q < WPW * 3 * 64; q += 64 * WPW) mks[q] = uint4{0xA5A5C3C3u, 0x5A5A3C3Cu, 0x0FF0F00Fu, 0x33CC55AAu};
    if constexpr (kFuse0)
        for (int q = threadIdx.x; q < 2 * WPW * 128; q += 64 * WPW) reinterpret_cast<uint4*>(xtiles)[q] = uint4{0x3F803F80u, 0x3F80BF80u, 0x3F003F00u, 0x3F803F80u};
    __syncthreads();
#endif
    TG_CLOCK_PROBE_BEGIN(g_probe_bwd_chain)
    dma_dzh(blockIdx.x);
    if constexpr (kFuse0) dma_x0(blockIdx.x, 0);
    dma_mask(blockIdx.x, 0, 0);
    dma_mask(blockIdx.x, 1, 1);
    for (int b0 = 0; b0 < P; ++b0) {
        ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);
        pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;
        pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the counted waits assume stores behind every block; none yet

    // the masks of layer seq + 2 go out when layer seq starts: at least one full layer (MT blocks) before their use
    auto prefetch_mask = [&](int64_t round, int j) {
        int j2 = j + 2;
        int64_t r2 = round;
        if (j2 >= n_layers) { j2 -= n_layers; r2 += gridDim.x; }
        dma_mask(r2, j2, (mseq + 2) % 3);
    };
    // the lane's WPL words of (row 16 c + col, half g >> 1), already shifted right by its nibble
    auto mask_words = [&](uint32_t (&mw)[2][WPL]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int r = 16 * c + col;
            if constexpr (MT == 8) {
                const uint4 v = lds_read_b128_opaque(my_mks + (mseq % 3) * 64 + r * 2 + (grp >> 1));
                mw[c][0] = v.x >> nib; mw[c][1] = v.y >> nib; mw[c][2] = v.z >> nib; mw[c][3] = v.w >> nib;
            } else {
                const uint4 v = lds_read_b128_opaque(my_mks + (mseq % 3) * 64 + r);
                mw[c][0] = ((grp >> 1) ? v.z : v.x) >> nib; mw[c][1] = ((grp >> 1) ? v.w : v.y) >> nib;
            }
        }
    };

    // kFuse0: this wave's share of dW0 (see BwdChainPtrs): block b's 16 features (wave >> 1 & 1) x 16 inputs (wave & 1), contracted
    // over the rows of waves 4 (wave >> 2) .. + 3
    [[maybe_unused]] f32x4 acc0[MT];
    if constexpr (kFuse0) {
#pragma unroll
        for (int b = 0; b < MT; ++b) acc0[b] = f32x4{};
    }
    [[maybe_unused]] const int q4 = (lane >> 2) & 3, p4 = lane & 3;
    [[maybe_unused]] auto frag_off = [&](int t, int hi) {              // lane part of a 16-column fragment read of a 2-KiB tile
        const int quad = 2 * grp + hi;
        return quad * 256 + q4 * 64 + (((2 * t + (p4 >> 1)) ^ (quad & 3)) * 16) + (p4 & 1) * 8;
    };
    [[maybe_unused]] const int a_lo = frag_off((wave >> 1) & 1, 0), a_hi = frag_off((wave >> 1) & 1, 1);
    [[maybe_unused]] const int b_lo = frag_off(wave & 1, 0), b_hi = frag_off(wave & 1, 1);
    [[maybe_unused]] auto owner_work = [&](int b) {
#pragma unroll
        for (int ks = 0; ks < WPW / 2; ++ks) {
            const int v = (WPW / 2) * (wave >> 2) + ks;
            const char* tb = tiles + ((b & 1) * WPW + v) * 2048;
            const bf16x8 fa = tr_frag16(tb + a_lo, tb + a_hi);
            const char* xb = xtiles + v * 2048;
            const bf16x8 fb = tr_frag16(xb + b_lo, xb + b_hi);      // (kept in registers across the round's blocks it spills: slower)
            acc0[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc0[b], 0, 0, 0);
        }
    };

    for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t row0 = round * (32 * WPW) + wave * 32;
        bf16x8 xin[2][K8], xout[2][K8];
        uint32_t mw[2][WPL];

        // ---- head: dZ_top^T = W_head^T . dOut^T; one block holds all MT output blocks (K padded to 32: one k-step) ----
        {
            // (kFuse0: the last layer of the previous round stored nothing: only the one later DMA lies behind this block)
            if constexpr (kFuse0) { TG_RING_WAIT((P - 1) * (KS / WPW)) } else { TG_RING_WAIT(kWaitN) }
            TG_RING_NEXT
            if constexpr (kFuse0) {
                if (round != (int64_t)blockIdx.x) dma_x0(round, 0);   // everyone is past the previous round's last tile reads (the barrier above)
            }
            bf16x8 x0[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4 gq = lds_read_b128_opaque(my_dzs + 16 * c + col);
                x0[c] = grp ? bf16x8{} : __builtin_bit_cast(bf16x8, gq);          // k = 8 g + j: outputs 0..7 sit in the g = 0 lanes
            }
            dma_dzh(round + gridDim.x);
            prefetch_mask(round, 0);
            mask_words(mw);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(mt * 2 + f) * 64 + lane]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x0[c], f32x4{}, 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) xout[c][mt] = masked_pack(acc[0][c], acc[1][c], mw[c][mt >> 1], mt);
                if ((mt & 1) && store_top) {
                    const bf16x8 pa[2] = {xout[0][mt - 1], xout[1][mt - 1]}, pb[2] = {xout[0][mt], xout[1][mt]};
                    store_pair(stage, ptrs.dz[0] + 32 * (mt - 1), row0, rows, H, lane, pa, pb);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < K8; ++ks) xin[c][ks] = xout[c][ks];
            ++mseq;
        }
        // ---- hidden layers, top down: dZ_below^T = W^T . dZ^T, one block per 32 output features ----
        for (int j = 1; j < n_layers; ++j) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // (without the head block's stores nothing but the one later DMA lies behind the second block of the first layer;
                // kFuse0: nor behind its first block -- the previous round ended without stores -- nor from the third block of the
                // last layer on, which writes LDS tiles instead of dZ)
                const bool fused_layer = kFuse0 && j == n_layers - 1;
                if constexpr (kFuse0) {
                    if (fused_layer) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's tile writes are done
                }
                if ((mt == 1 && j == 1 && !store_top) || (kFuse0 && mt == 0 && j == 1 && !store_top) || (fused_layer && mt >= 2)) {
                    TG_RING_WAIT((P - 1) * (KS / WPW))
                } else {
                    TG_RING_WAIT(kWaitN)
                }
                TG_RING_NEXT

                if (mt == 0) {
                    prefetch_mask(round, j);
                    mask_words(mw);
                }
                f32x4 acc[2][2] = {};
                // scheduling fences around the block's products: left alone, hipcc threads the pack / mask arithmetic of the
                // neighbouring blocks through the MFMA sequence (s_nop-padded); kept apart the kernel is 2.4 % faster (same-box A/B)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < K8; ++ks)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(ks * 2 + f) * 64 + lane]);
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xin[c][ks], acc[f][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kFuse0) {
                    // the tiles of block mt - 1 are complete behind this block's barrier; their products go behind this block's own
                    // (in front of them the fragment reads' latency is exposed: same-box A/B 2522 -> 2513 us per 2^22 rows)
                    if (fused_layer && mt >= 1) owner_work(mt - 1);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) xout[c][mt] = masked_pack(acc[0][c], acc[1][c], mw[c][mt >> 1], mt);
                if (kFuse0 && fused_layer) {
                    // the block's 32 rows x 32 features into T[mt & 1][wave]; rows past the end (clamped duplicates of the last row)
                    // as zeros: they must not be counted twice in the contraction
                    char* tw = tiles + ((mt & 1) * WPW + wave) * 2048;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int row = 16 * c + col;
                        uint4 o = __builtin_bit_cast(uint4, xout[c][mt]);
                        if (row0 + row >= rows) o = uint4{0u, 0u, 0u, 0u};
                        lds_store16(tw + (row >> 2) * 256 + (row & 3) * 64 + ((grp ^ ((row >> 2) & 3)) * 16), o);
                    }
                } else if (mt & 1) {
                    const bf16x8 pa[2] = {xout[0][mt - 1], xout[1][mt - 1]}, pb[2] = {xout[0][mt], xout[1][mt]};
                    store_pair(stage, ptrs.dz[j] + 32 * (mt - 1), row0, rows, H, lane, pa, pb);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < K8; ++ks) xin[c][ks] = xout[c][ks];
            ++mseq;
        }
        if constexpr (kFuse0) {                      // the last block's tiles: one plain barrier, then their products
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            owner_work(MT - 1);
        }
    }
    if constexpr (kFuse0) {
        // this wave's 16 x 16 tiles of the workgroup's partial dW0: slab [workgroup][K half][H][32]
        float* slab = ptrs.w0_slabs + ((int64_t)blockIdx.x * 2 + (wave >> 2)) * H * 32;
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(32 * b + 16 * ((wave >> 1) & 1) + 4 * grp + r) * 32 + 16 * (wave & 1) + col] = acc0[b][r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
    TG_CLOCK_PROBE_END(g_probe_bwd_chain)
}

// Per-workgroup column sums of a dZ matrix (the bias gradient), for callers of tg_mlp_backward_chain that pass d_partial:
// workgroup b sums rows b, b + grid, ... ; thread (rg, cq) sums the 8 columns 8 cq .. of its rows rg, rg + RG, ...; the RG row
// groups are added in a fixed order: deterministic.  partial[b][layer][H].
template <int H>
__global__ __launch_bounds__(256) void dz_colsum_kernel(const uint16_t* __restrict__ dz, int64_t rows, int32_t layer, int32_t n_layers,
                                                        float* __restrict__ partial) {
    constexpr int CQ = H / 8, RG = 256 / CQ;
    __shared__ float red[RG][H];
    const int cq = threadIdx.x % CQ, rg = threadIdx.x / CQ;
    float s[8] = {};
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < rows; r += (int64_t)gridDim.x * RG) {
        const uint4 v = *reinterpret_cast<const uint4*>(dz + r * H + 8 * cq);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[2 * k] += __uint_as_float(w[k] << 16);
            s[2 * k + 1] += __uint_as_float(w[k] & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rg][8 * cq + k] = s[k];
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) t += red[g][c];
        partial[((int64_t)blockIdx.x * n_layers + layer) * H + c] = t;
    }
}

static int bwd_chain_blocks() { return device_cus(); }

}  // namespace tg

using namespace tg;

template <int H>
static int launch_bwd_chain(const void* d_dout8, const void* d_wfrag, int32_t n_hidden_layers, int64_t rows, void* const* d_dz,
                            const void* const* d_masks, float* d_partial, const void* d_x, float* d_w0_slabs, hipStream_t st) {
    constexpr int WPW = 8, KS = H / 16;
    const int grid_max = bwd_chain_blocks();
    if (rows == 0) {
        if (!d_partial) return TG_OK;
        hipError_t e = hipMemsetAsync(d_partial, 0, (size_t)grid_max * n_hidden_layers * H * sizeof(float), st);
        return e == hipSuccess ? TG_OK : set_error(TG_ERR_HIP, "tg_mlp_backward_chain: memset failed (%s)", hipGetErrorString(e));
    }
    BwdChainPtrs ptrs{};
    for (int j = 0; j < n_hidden_layers; ++j) {
        TG_REQUIRE((d_dz[j] || (j == 0 && !d_partial) || (j == n_hidden_layers - 1 && d_x)) && d_masks[j],
                   "tg_mlp_backward_chain: buffer %d is null", j);
        ptrs.dz[j] = (uint16_t*)d_dz[j];
        ptrs.mask[j] = (const uint32_t*)d_masks[j];
    }
    ptrs.x = (const uint4*)d_x;
    ptrs.w0_slabs = d_w0_slabs;
    const size_t shmem = (size_t)3 * KS * 1024 + (size_t)WPW * 32 * 128 + (size_t)WPW * 1024 + (size_t)WPW * 3 * 1024 +
                         (d_x ? (size_t)WPW * 2048 : 0);
    const int64_t n_rounds = ceil_div(rows, (int64_t)32 * WPW);
    const unsigned grid = (unsigned)(n_rounds < grid_max ? n_rounds : grid_max);
    if (d_x) {
        auto kern = mlp_bwd_chain_kernel<H, WPW, true>;
        static LdsOptIn opt_in;
        if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_backward_chain_w0")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPW), shmem, st, (const uint4*)d_dout8, (const uint4*)d_wfrag, n_hidden_layers,
                           rows, ptrs);
    } else {
        auto kern = mlp_bwd_chain_kernel<H, WPW, false>;
        static LdsOptIn opt_in;
        if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_backward_chain")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPW), shmem, st, (const uint4*)d_dout8, (const uint4*)d_wfrag, n_hidden_layers,
                           rows, ptrs);
    }
    TG_LAUNCH_CHECK("tg_mlp_backward_chain");
    if (d_partial) {                     // the legacy bias-gradient contract: column sums of what was just written
        for (int j = 0; j < n_hidden_layers; ++j)
            hipLaunchKernelGGL(dz_colsum_kernel<H>, dim3((unsigned)grid_max), dim3(256), 0, st, (const uint16_t*)d_dz[j], rows, j,
                               n_hidden_layers, d_partial);
        TG_LAUNCH_CHECK("tg_mlp_backward_chain (column sums)");
    }
    return TG_OK;
}

extern "C" {

int tg_mlp_backward_chain_blocks(void) { return bwd_chain_blocks(); }

int tg_mlp_backward_chain(const void* d_dout8, const void* d_wfrag, int32_t hidden, int32_t n_hidden_layers, int64_t rows,
                          void* const* d_dz, const void* const* d_masks, float* d_partial, void* stream) {
    TG_REQUIRE(d_dout8 && d_wfrag && d_dz && d_masks, "tg_mlp_backward_chain: null pointer");
    TG_REQUIRE(hidden == 256 || hidden == 128, "tg_mlp_backward_chain: hidden width %d unsupported (128, 256)", hidden);
    TG_REQUIRE(n_hidden_layers >= 3 && n_hidden_layers <= kBwdMaxLayers, "tg_mlp_backward_chain: %d hidden layers outside 3..%d",
               n_hidden_layers, kBwdMaxLayers);
    TG_REQUIRE(rows >= 0, "tg_mlp_backward_chain: negative row count");
    hipStream_t st = (hipStream_t)stream;
    return hidden == 256 ? launch_bwd_chain<256>(d_dout8, d_wfrag, n_hidden_layers, rows, d_dz, d_masks, d_partial, nullptr, nullptr, st)
                         : launch_bwd_chain<128>(d_dout8, d_wfrag, n_hidden_layers, rows, d_dz, d_masks, d_partial, nullptr, nullptr, st);
}

int tg_mlp_backward_chain_w0(const void* d_dout8, const void* d_wfrag, int32_t hidden, int32_t n_hidden_layers, int64_t rows,
                             void* const* d_dz, const void* const* d_masks, const void* d_x, float* d_w0_slabs, int64_t slab_floats,
                             int32_t* n_slabs, void* stream) {
    TG_REQUIRE(d_dout8 && d_wfrag && d_dz && d_masks && d_x && d_w0_slabs && n_slabs, "tg_mlp_backward_chain_w0: null pointer");
    TG_REQUIRE(hidden == 256 || hidden == 128, "tg_mlp_backward_chain_w0: hidden width %d unsupported (128, 256)", hidden);
    TG_REQUIRE(n_hidden_layers >= 3 && n_hidden_layers <= kBwdMaxLayers, "tg_mlp_backward_chain_w0: %d hidden layers outside 3..%d",
               n_hidden_layers, kBwdMaxLayers);
    TG_REQUIRE(rows >= 0, "tg_mlp_backward_chain_w0: negative row count");
    const int64_t n_rounds = ceil_div(rows, (int64_t)32 * 8);
    const int grid = (int)(n_rounds < bwd_chain_blocks() ? n_rounds : bwd_chain_blocks());
    *n_slabs = 2 * grid;
    TG_REQUIRE(slab_floats >= (int64_t)2 * bwd_chain_blocks() * hidden * 32, "tg_mlp_backward_chain_w0: slab buffer of %lld floats < %lld",
               (long long)slab_floats, (long long)2 * bwd_chain_blocks() * hidden * 32);
    if (rows == 0) return TG_OK;
    hipStream_t st = (hipStream_t)stream;
    return hidden == 256 ? launch_bwd_chain<256>(d_dout8, d_wfrag, n_hidden_layers, rows, d_dz, d_masks, nullptr, d_x, d_w0_slabs, st)
                         : launch_bwd_chain<128>(d_dout8, d_wfrag, n_hidden_layers, rows, d_dz, d_masks, nullptr, d_x, d_w0_slabs, st);
}

}  // extern "C"
