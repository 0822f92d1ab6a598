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
// Same machinery as the forward chain: transposed product with v_mfma_f32_32x32x16_bf16, 32 rows per wave, the
// accumulator tile of a layer IS the B operand of the layer below (keep-mask AND instead of bias + ReLU), W^T streams
// L2 -> LDS through the LDS-DMA ring (mlp.FragmentStream(layout="chain") of the transposed weights, head first).
// The mask bits of a layer (1 KiB per wave) arrive by LDS-DMA two layers ahead, dOut one round ahead; neither is counted
// in the ring's waits (more operations behind a block only make its wait stricter).  Every second output tile issues
// exactly four stores per wave, always (rows past the end are clamped to the last row and rewrite it with identical
// bytes), so the counted wait is the same at every site.
// Bias gradients: the column sums of a tile over the wave's 32 rows are formed by letting the matrix core transpose the
// tile (column_sums below) and accumulated in a per-wave LDS table, which the workgroup adds up in a fixed order at
// the end: deterministic.
#include "mfma_ring.hpp"

namespace tg {

constexpr int kBwdMaxLayers = 6;        // 8 KiB of LDS per layer for the per-wave bias tables: 112 + 8 n <= 160 KiB
struct BwdChainPtrs {
    uint16_t* dz[kBwdMaxLayers];            // outputs, top hidden layer first: bf16 [rows][H]
    const uint32_t* mask[kBwdMaxLayers];    // ReLU mask bits of the same layers (tg_mlp_forward_chain): u32 [rows][H/32]
};

// dZ stores (as the forward chain's activation stores): after two output tiles a lane (n, h) holds 4 x 16 B of row n's
// 128-B line; the wave transposes the 32 x 128 B through its LDS staging area (XOR-swizzled 16-B chunks: conflict-free
// both ways) and each of its 4 store instructions writes 8 WHOLE 128-B lines, non-temporal (written once, read by the
// weight-gradient GEMM).  As 32-B pieces the same bytes took 5 % longer.  `__restrict__`: alias scope (mfma_ring.hpp).
__device__ static inline void store_pair(uint4* __restrict__ st, uint16_t* __restrict__ g, int64_t row0, int64_t rows, int ld,
                                         int lane, bf16x8 a_lo, bf16x8 a_hi, bf16x8 b_lo, bf16x8 b_hi) {
    const int n = lane & 31, sw = (n >> 1) & 7, c0 = 2 * (lane >> 5);
    uint4* w = st + n * 8;
    w[(c0 + 0) ^ sw] = __builtin_bit_cast(uint4, a_lo);
    w[(c0 + 1) ^ sw] = __builtin_bit_cast(uint4, a_hi);
    w[(c0 + 4) ^ sw] = __builtin_bit_cast(uint4, b_lo);
    w[(c0 + 5) ^ sw] = __builtin_bit_cast(uint4, b_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3), c = lane & 7;
        const uint4 v = st[r * 8 + (c ^ ((r >> 1) & 7))];
        int64_t row = row0 + r;
        row = row < rows ? row : rows - 1;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(g + row * ld + 8 * c));
    }
}

// Column sums of a tile over the wave's 32 rows (the bias gradient).  Rows sit on lanes, so a lane-wise reduction costs
// 5 DPP additions per value (80 per tile: the kernel became VALU-bound).  Instead the matrix core transposes: with the
// masked outputs as the A operand (lane = row, in-lane = 8 features) and a 0/1 selection matrix as B,
// T[row][n'] = A[row][k = n' - shift] lands with the ROWS in the accumulator registers of lane n' -- 15 in-lane additions
// sum them, one cross-half exchange adds the two row subsets.  The tile's two operands (`lo`: slots -> n' 0..15, `hi`:
// slots -> n' 16..31) accumulate into ONE transposed tile, so the additions run once per tile.
__device__ static inline bf16x8 selection_operand(int lane, int shift) {
    const int np = (lane & 31) - shift, hb = lane >> 5;
    bf16x8 b = {};
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (8 * hb + j == np) ? (__bf16)1.0f : (__bf16)0.0f;
    return b;
}
// -> lane n' < 32 (either half) holds the sum over the wave's 32 rows of slot n' of `lo` (n' < 16) or slot n' - 16 of `hi`
__device__ static inline float column_sums(bf16x8 lo, bf16x8 hi, bf16x8 sel_lo, bf16x8 sel_hi) {
    f32x16 t = {};
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, sel_lo, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, sel_hi, t, 0, 0, 0);
    float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    s += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // own 16 rows + the other half's 16 rows, same order in both
}

// per-wave bias-gradient table in LDS (`__restrict__`: alias scope, see mfma_ring.hpp)
__device__ static inline void bias_accumulate(float* __restrict__ slot, float v, bool writer) {
    if (writer) *slot += v;
}

__device__ static inline uint4 lds_read_b128_opaque(const uint4* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint4*)p;
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

// Masked epilogue of one 32-feature tile: round the accumulators pairwise, AND with the keep-masks from the layer's
// mask word `w` (feature r of this lane: bit (mt&1)*8 + (r>>1) + 16 (r&1)), hand the packed halves on and add their
// column sums to the wave's bias table (`own` = all ones for a lane with a row of its own, else 0; `full` = every lane
// of the wave has one, the case in all rounds but the last).
struct SelPair { bf16x8 lo, hi; };
template <int H, bool kBias>
__device__ static inline void masked_tile(const f32x16& acc, uint32_t w, int mt, bf16x8& lo, bf16x8& hi, float* __restrict__ btab,
                                          bool writer, uint32_t own, bool full, const SelPair& sel) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const uint32_t wk = w >> ((mt & 1) * 8);
    uint32_t o[8];
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[2 * k], acc[2 * k + 1]}, bf16x2));
        // the keep bits of the pair, one per 16-bit half (0 / 1), applied by ONE packed integer multiply (v_pk_mul_lo_u16):
        // 3 instructions per pair instead of shift, and, multiply to 0xFFFF, and
        const u16x2 keep = __builtin_bit_cast(u16x2, (wk >> k) & 0x00010001u);
        o[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, pk) * keep);
    }
    lo = __builtin_bit_cast(bf16x8, uint4{o[0], o[1], o[2], o[3]});
    hi = __builtin_bit_cast(bf16x8, uint4{o[4], o[5], o[6], o[7]});
    if constexpr (!kBias) return;                     // the bias gradients come out of tg_mlp_weight_grad's contraction
    // operand slot k = 8 h + j of `lo` is feature 16 h + j, of `hi` feature 16 h + 8 + j: lane n' (< 16: slot n' of lo,
    // 16..31: slot n' - 16 of hi) gets their sums, i.e. table entry 32 mt + 16 ((n'&15)>>3) + (n'&7) + 8 (n'>>4)
    float s;
    if (full) {
        s = column_sums(lo, hi, sel.lo, sel.hi);
    } else {                                          // a lane clamped onto the last row must not count it again
        s = column_sums(__builtin_bit_cast(bf16x8, uint4{o[0] & own, o[1] & own, o[2] & own, o[3] & own}),
                        __builtin_bit_cast(bf16x8, uint4{o[4] & own, o[5] & own, o[6] & own, o[7] & own}), sel.lo, sel.hi);
    }
    bias_accumulate(btab + 32 * mt, s, writer);
}

template <int H, int WPW, bool kBias>
__global__ __launch_bounds__(64 * WPW, 2) void mlp_bwd_chain_kernel(const uint4* __restrict__ dzh, const uint4* __restrict__ wfrag,
                                                                    int32_t n_layers, int64_t rows, BwdChainPtrs ptrs,
                                                                    float* __restrict__ partial) {
    constexpr int MT = H / 32, KS = H / 16;
    // ring of 3 slots, 2 blocks in flight (4 / 3 measured the same; the LDS goes to the store staging instead).  Behind
    // the block a wait is for: the DMA of the one later block and the stores of the last two tiles, one of them odd
    // (4 stores; the head block's 16 only add to that).
    constexpr int D = 3, P = D - 1;
    constexpr int kWaitN = (P - 1) * (KS / WPW) + 4;
    static_assert(KS % WPW == 0, "every wave moves the same number of 1-KiB pieces per block");
    extern __shared__ uint4 lds[];
    uint4* ring = lds;                                                  // D * KS * 64 uint4
    uint4* dzs = lds + D * KS * 64;                                     // WPW waves * 64 uint4 (lanes 0..31 used)
    uint4* mks = dzs + WPW * 64;                                        // WPW waves * 3 buffers * 64 uint4
    uint4* stage = mks + WPW * 3 * 64 + (threadIdx.x >> 6) * (32 * 8);   // per wave: 32 rows x 128 B (store_pair)
    float* bacc = reinterpret_cast<float*>(mks + WPW * 3 * 64 + WPW * 32 * 8);   // WPW waves * n_layers * H floats

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int64_t n_rounds = (rows + 32 * WPW - 1) / (32 * WPW);
    const int n_blocks = (n_layers - 1) * MT + 1;
    constexpr int WPL = MT / 2;                                         // mask words per lane and layer

    if constexpr (kBias) {
        for (int q = threadIdx.x; q < WPW * n_layers * H; q += 64 * WPW) bacc[q] = 0.f;
    }
    __syncthreads();

    uint4* my_dzs = dzs + wave * 64;
    uint4* my_mks = mks + wave * 3 * 64;
    // this lane's column-sum entry (lanes 0..31: slots of the tile's first operand in lanes 0..15, of its second in 16..31)
    float* my_bacc = bacc + wave * n_layers * H + 16 * ((col & 15) >> 3) + (col & 7) + 8 * (col >> 4);
    const bool writer = lane < 32;
    const SelPair sel = {selection_operand(lane, 0), selection_operand(lane, 16)};

    auto dma_dzh = [&](int64_t round) {
        if (lane < 32) {
            int64_t r = round * (32 * WPW) + wave * 32 + lane;
            r = r < rows ? r : rows - 1;
            __builtin_amdgcn_global_load_lds(dzh + r, (lds_void*)my_dzs, 16, 0, 0);
        }
    };
    // mask bits of layer j for the 32 rows of `round`.  H = 256: lane L fetches the 16 B of (row L>>1, lane half L&1);
    // H = 128: a row's two halves are 16 B together and lanes 0..31 fetch one row each.
    static_assert(MT == 8 || MT == 4, "the mask staging moves 16 B per lane (H = 256 or 128)");
    auto dma_mask = [&](int64_t round, int j, int buf) {
        if constexpr (MT == 8) {
            int64_t r = round * (32 * WPW) + wave * 32 + (lane >> 1);
            r = r < rows ? r : rows - 1;
            __builtin_amdgcn_global_load_lds(ptrs.mask[j] + r * MT + (lane & 1) * WPL, (lds_void*)(my_mks + buf * 64), 16, 0, 0);
        } else if (lane < 32) {
            int64_t r = round * (32 * WPW) + wave * 32 + lane;
            r = r < rows ? r : rows - 1;
            __builtin_amdgcn_global_load_lds(ptrs.mask[j] + r * MT, (lds_void*)(my_mks + buf * 64), 16, 0, 0);
        }
    };

    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
    int mseq = 0;                                   // running layer number of this workgroup; its masks sit in buffer mseq % 3
    dma_dzh(blockIdx.x);
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
    auto mask_words = [&](uint32_t (&mw)[WPL]) {
        if constexpr (MT == 8) {
            const uint4 v = lds_read_b128_opaque(my_mks + (mseq % 3) * 64 + col * 2 + h);
            mw[0] = v.x; mw[1] = v.y; mw[2] = v.z; mw[3] = v.w;
        } else {
            const uint4 v = lds_read_b128_opaque(my_mks + (mseq % 3) * 64 + col);
            mw[0] = h ? v.z : v.x; mw[1] = h ? v.w : v.y;
        }
    };

    for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t row0 = round * (32 * WPW) + wave * 32;
        int64_t row = row0 + col;
        const uint32_t own = row < rows ? 0xFFFFFFFFu : 0u;
        const bool full = row0 + 32 <= rows;          // wave-uniform: every lane has a row of its own
        row = row < rows ? row : rows - 1;            // clamped rows recompute and rewrite the last row (identical bytes)
        bf16x8 xin[KS], xout[KS];
        uint32_t mw[WPL];

        // ---- head: dZ_top^T = W_head^T . dOut^T; one block holds all MT output tiles (K padded to 32: 2 k-steps) ----
        {
            TG_RING_ADVANCE(kWaitN)
            const uint4 g = lds_read_b128_opaque(my_dzs + col);
            xin[0] = h ? bf16x8{} : __builtin_bit_cast(bf16x8, g);            // k = 8h + j: outputs 0..7 sit in the h = 0 lanes
            xin[1] = bf16x8{};
            dma_dzh(round + gridDim.x);
            prefetch_mask(round, 0);
            mask_words(mw);
            float* bt = my_bacc;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 acc = {};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(mt * 2 + ks) * 64 + lane]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[ks], acc, 0, 0, 0);
                }
                masked_tile<H, kBias>(acc, mw[mt >> 1], mt, xout[2 * mt], xout[2 * mt + 1], bt, writer, own, full, sel);
                if (mt & 1)
                    store_pair(stage, ptrs.dz[0] + 32 * (mt - 1), row0, rows, H, lane, xout[2 * mt - 2], xout[2 * mt - 1], xout[2 * mt],
                               xout[2 * mt + 1]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xin[ks] = xout[ks];
            ++mseq;
        }
        // ---- hidden layers, top down: dZ_below^T = W^T . dZ^T, one block per 32-feature output tile ----
        for (int j = 1; j < n_layers; ++j) {
            float* bt = my_bacc + j * H;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                TG_RING_ADVANCE(kWaitN)
                if (mt == 0) {
                    prefetch_mask(round, j);
                    mask_words(mw);
                }
                f32x16 acc = {};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[ks], acc, 0, 0, 0);
                }
                masked_tile<H, kBias>(acc, mw[mt >> 1], mt, xout[2 * mt], xout[2 * mt + 1], bt, writer, own, full, sel);
                if (mt & 1)
                    store_pair(stage, ptrs.dz[j] + 32 * (mt - 1), row0, rows, H, lane, xout[2 * mt - 2], xout[2 * mt - 1], xout[2 * mt],
                               xout[2 * mt + 1]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xin[ks] = xout[ks];
            ++mseq;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
    __syncthreads();
    if constexpr (!kBias) return;
    for (int c = threadIdx.x; c < n_layers * H; c += 64 * WPW) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPW; ++w) s += bacc[w * n_layers * H + c];
        partial[(int64_t)blockIdx.x * n_layers * H + c] = s;
    }
}

static int bwd_chain_blocks() { return device_cus(); }

}  // namespace tg

using namespace tg;

template <int H, bool kBias>
static int launch_bwd_chain(const void* d_dout8, const void* d_wfrag, int32_t n_hidden_layers, int64_t rows, void* const* d_dz,
                            const void* const* d_masks, float* d_partial, hipStream_t st) {
    constexpr int WPW = 8, KS = H / 16;
    const int grid_max = bwd_chain_blocks();
    const size_t partial_bytes = (size_t)grid_max * n_hidden_layers * H * sizeof(float);
    if (rows == 0) {
        if (!kBias) return TG_OK;
        hipError_t e = hipMemsetAsync(d_partial, 0, partial_bytes, st);
        return e == hipSuccess ? TG_OK : set_error(TG_ERR_HIP, "tg_mlp_backward_chain: memset failed (%s)", hipGetErrorString(e));
    }
    BwdChainPtrs ptrs{};
    for (int j = 0; j < n_hidden_layers; ++j) {
        TG_REQUIRE(d_dz[j] && d_masks[j], "tg_mlp_backward_chain: buffer %d is null", j);
        ptrs.dz[j] = (uint16_t*)d_dz[j];
        ptrs.mask[j] = (const uint32_t*)d_masks[j];
    }
    const size_t shmem = (size_t)3 * KS * 1024 + (size_t)WPW * 32 * 128 + (size_t)WPW * 1024 + (size_t)WPW * 3 * 1024 +
                         (kBias ? (size_t)WPW * n_hidden_layers * H * sizeof(float) : 0);
    auto kern = mlp_bwd_chain_kernel<H, WPW, kBias>;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_backward_chain")) return rc;
    const int64_t n_rounds = ceil_div(rows, (int64_t)32 * WPW);
    const unsigned grid = (unsigned)(n_rounds < grid_max ? n_rounds : grid_max);
    if (kBias && (int)grid < grid_max) {      // workgroups that do not run leave their partial rows untouched: clear them
        hipError_t e = hipMemsetAsync(d_partial, 0, partial_bytes, st);
        if (e != hipSuccess) return set_error(TG_ERR_HIP, "tg_mlp_backward_chain: memset failed (%s)", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPW), shmem, st, (const uint4*)d_dout8, (const uint4*)d_wfrag, n_hidden_layers, rows,
                       ptrs, d_partial);
    TG_LAUNCH_CHECK("tg_mlp_backward_chain");
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
#define TG_BWD_ARGS d_dout8, d_wfrag, n_hidden_layers, rows, d_dz, d_masks, d_partial, st
    if (d_partial) return hidden == 256 ? launch_bwd_chain<256, true>(TG_BWD_ARGS) : launch_bwd_chain<128, true>(TG_BWD_ARGS);
    return hidden == 256 ? launch_bwd_chain<256, false>(TG_BWD_ARGS) : launch_bwd_chain<128, false>(TG_BWD_ARGS);
#undef TG_BWD_ARGS
}

}  // extern "C"
