// The learner's update in the REFERENCE'S OWN PRECISION (fp32) at the reference's own net sizes -- Linear(S<=32, H) ReLU
// [Linear(H, H) ReLU]{0..3} Linear(H, A<=4), H in {64, 128}: what pipelines/cartpole_pipeline_grpo.py:54-76,
// cartpole_pipeline_ppo.py:54-79 and BASELINE configs[1] build (models/neural_network.py:67-77) -- as TWO launches per net and
// update instead of ~60 (per-layer hipBLASLt GEMMs + glue kernels; 94 % of C2's step in round 2):
//
//   tg_mlp_f32_forward_backward   forward pass, loss head (algorithms/ppo.py:159-179, grpo.py:122-140: the arithmetic of
//                                 loss_kernels.hip) and backward-DATA pass of a row in one go: a wave owns 32 rows from the
//                                 input to d loss / d first-layer pre-activation; activations and dZ are only WRITTEN (for the
//                                 weight gradients), the ReLU masks never leave the chip;
//   tg_mlp_f32_weight_grad        every weight and bias gradient of the net (what `loss.backward()` leaves in .grad,
//                                 algorithms/ppo.py:181-183) in one launch + a fixed-order slab reduction.
//   tg_mlp_f32_forward            the no-grad pass (old log-probs, values, per-step rollout): forward only.
//
// All products are v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 155 TFLOP/s chip-wide = 1/16 of the bf16 rate): these
// kernels are bound by the fp32 matrix pipe, not by HBM.  Transposed form Y^T = W . X^T as in the bf16 chain kernels: the 32
// columns of a tile are 32 rows of the batch, so a row never leaves its wave, and lane (row j, half h) of an accumulator tile
// holds features F(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15 -- register r of the two halves IS the B operand of MFMA step
// (tile, r) of the next layer (k pair F(r, 0), F(r, 1)); the k order this implies is folded into the weight stream
// (mlp.F32ChainStream).  The backward pass is the same machine on W^T with a mask multiply instead of bias + ReLU.
// Weights: the first layer, biases and head stay in LDS; the H x H layers stream L2 -> registers -> LDS in 32-feature blocks
// (H/2 MFMA steps x 64 lanes x 4 B = 16 KiB at H = 128) through two buffers, one plain workgroup barrier per block: a block is
// 64 MFMAs x 64 cycles per wave, so nothing about the staging needs to be clever.
#include <stdlib.h>

#include "tg_common.hpp"
#include "adam_update.hpp"
#include "f32_loss.hpp"
#include "f32_dw.hpp"

#ifndef TG_F32DW_STAMPS
#define TG_F32DW_STAMPS 0          /* diagnostic build: s_memtime stamps around the phases of the wide job's stage loop (never in the product) */
#endif
#ifndef TG_F32DW_ABLATE
#define TG_F32DW_ABLATE 0          /* timing-only probe builds of the wide weight-gradient job: 1 = no products, 2 = no stream;
                                      of the one-barrier 8-wave job: 3 = no rebuild inside the loop, 4 = no products, 5 = no DMA inside the loop */
#endif

namespace tg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kF32MaxHidden = 4;

struct F32Net {
    const uint4* w0;        // first layer, fragment order [H/32 tiles][k2/4][64 lanes] x 16 B (4 consecutive MFMA steps per lane)
    const float* bias;      // [n_hh + 1][H] hidden biases, natural order
    const float* wh;        // [4][H] head weights (rows >= A zero), natural order
    const float* bh;        // [4] head bias
    const uint4* blocks;    // H x H layers: forward blocks [n_hh][H/32][H/8][64] x 16 B, then the backward (transposed) blocks, top layer first
    int32_t n_hh;           // H x H layers (hidden layers - 1)
    int32_t k2;             // first-layer k pairs = padded input width / 2 (multiple of 4, <= 16)
};

struct F32ChainArgs {
    const float* x;         // [rows][2 k2] f32, zero padded
    int64_t rows;
    F32Net net;
    float* acts[kF32MaxHidden];   // kTrain: post-ReLU outputs of the hidden layers, f32 [rows][H]
    float* dz[kF32MaxHidden];     // kTrain: d loss / d pre-activation of the hidden layers, f32 [rows][H]
    float* out;             // !kTrain: head output f32 [rows][4]
    uint32_t* top_mask;     // kTrain, optional: the top hidden layer's ReLU mask bits, u32 [rows][4] (per row: [lane half h][MT / 2 words];
                            // feature 32 mt + 8 q + 4 h + low is bit low + 4 q + 16 (mt & 1) of word mt >> 1 of half h)
    int32_t resident;       // the whole H x H block stream fits the LDS beside the tables: loaded once, no ring, no block barriers
    F32Loss loss;
};

__device__ static inline f32x16 bias_rows(const float* __restrict__ b) {     // b = table + 32 tile + 4 h
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(b + 8 * q);
        acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
    }
    return acc;
}

// the lane's 16 values of tile `mt` (features 32 mt + 8 q + 4 h + 0..3) to row-major [rows][H]
template <int H>
__device__ static inline void store_tile(float* __restrict__ g, int64_t row, int mt, int h, const f32x16& v) {
    float* p = g + row * H + 32 * mt + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(p + 8 * q) = float4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

TG_CLOCK_PROBE_VAR(g_probe_f32_chain, attach_probe_f32_chain)          // the training pass (kTrain) only
TG_CLOCK_PROBE_VAR(g_probe_f32_dw, attach_probe_f32_dw)
int attach_probe_f32(int which, void* d_probe) { return which == 0 ? attach_probe_f32_chain(d_probe) : attach_probe_f32_dw(d_probe); }

template <int H, bool kTrain>
__global__ __launch_bounds__(512, 2) void mlp_f32_chain_kernel(F32ChainArgs a) {
    constexpr int MT = H / 32;                  // 32-feature tiles per layer
    constexpr int G4 = H / 8;                   // groups of 4 MFMA steps per H x H block
    constexpr int BLK = G4 * 64;                // uint4 per block
    constexpr int SQ = BLK / 512;               // uint4 per thread and block (2 at H = 128, 1 at H = 64)
    static_assert(BLK % 512 == 0, "a block is a whole number of 16-B pieces per thread");
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const F32Net& net = a.net;
    const int n_hh = net.n_hh, K2 = net.k2;
    const int n_stream = n_hh * MT * (kTrain ? 2 : 1);              // blocks per round (the same sequence every round)
    // (a LOCAL copy: writing into the by-value argument struct itself sends all of it to scratch -- 272 B per lane and 23 more
    // registers in this kernel, 9 % of a small PPO epoch, measured in round 5)
    F32Loss L = a.loss;
    if constexpr (kTrain) f32_loss_from_device(L);
    const bool resident = a.resident != 0;
    // LDS (f32_chain_lds() on the host computes the same sizes): the block ring (2 blocks) or the whole resident stream, the
    // first layer's fragments, bias / head tables, the ReLU mask bits of the (n_hh + 1) hidden layers, the loss-sum scratch
    uint4* ring = lds;
    uint4* w0s = ring + (resident ? n_stream : 2) * BLK;            // MT x (k2 / 4) x 64
    float* bias_s = reinterpret_cast<float*>(w0s + MT * (K2 / 4) * 64);    // kF32MaxHidden x H
    float* wh_s = bias_s + kF32MaxHidden * H;                       // 4 x H
    float* bh_s = wh_s + 4 * H;                                     // 4 (+ 12 pad)
    uint32_t* bits_s = reinterpret_cast<uint32_t*>(bh_s + 16);      // kTrain: [layer][wave][MT / 2 words][64 lanes]: ReLU masks
    double* red_s = reinterpret_cast<double*>(bits_s + (n_hh + 1) * 8 * (MT / 2) * 64);   // 8 waves x 4

    const int64_t rows = a.rows;
    const int64_t n_rounds = (rows + 255) / 256;
    if constexpr (kTrain) { TG_CLOCK_PROBE_BEGIN(g_probe_f32_chain) }

    // ---- resident tables ----
    for (int q = tid; q < MT * (K2 / 4) * 64; q += 512) w0s[q] = net.w0[q];
    for (int q = tid; q < (n_hh + 1) * H; q += 512) bias_s[q] = net.bias[q];
    for (int q = tid; q < 4 * H; q += 512) wh_s[q] = net.wh[q];
    if (tid < 4) bh_s[tid] = net.bh[tid];

    // ---- block stream: resident in LDS when it fits (C2's 5-128-128-1: 128 KiB), else two LDS buffers with register staging one
    // block ahead ----
    static_assert(SQ == 1 || SQ == 2, "one or two 16-B pieces per thread and block");
    uint4 stg0 = {}, stg1 = {};                                     // (named, not an array: hipcc put a 2-element array in scratch)
    int pos = 0, buf = 0;
    auto load_block = [&]() {
        const uint4* src = net.blocks + (int64_t)pos * BLK + tid;
        stg0 = src[0];
        if constexpr (SQ == 2) stg1 = src[512];
        pos = pos + 1 == n_stream ? 0 : pos + 1;
    };
    auto write_block = [&](int b) {
        uint4* dst = ring + b * BLK + tid;
        dst[0] = stg0;
        if constexpr (SQ == 2) dst[512] = stg1;
    };
    if (resident) {
        for (int q = tid; q < n_stream * BLK; q += 512) ring[q] = net.blocks[q];
    } else if (n_stream > 0) {
        load_block();
        write_block(0);
        load_block();
    }
    __syncthreads();
    // a block's life: BEGIN hands the next block (in registers since the previous BEGIN) to the idle buffer and requests the one
    // after; the products read `cur`; END is the workgroup barrier that retires `cur` and publishes the buffer just written
    // The stores of a hidden tile are DEFERRED to the next block's BEGIN, in front of that block's stream loads: vector-memory
    // operations retire in order, so the wait for those loads (one block later) then implies stores that have had a whole block
    // to complete, instead of stores issued a moment ago (+25 % on the kernel when each block waited for its own stores).
    // Resident stream: a block is just an offset -- no staging, no barrier, and the stores need no deferral because nothing in the
    // loop waits on the vector-memory queue any more (they drain behind the next round's input loads).
#define TG_F32_BLOCK_BEGIN                                  \
    const uint4* cur;                                       \
    if (resident) {                                         \
        cur = ring + pos * BLK;                             \
        pos = pos + 1 == n_stream ? 0 : pos + 1;            \
    } else {                                                \
        write_block(buf ^ 1);                               \
        flush_store();                                      \
        load_block();                                       \
        cur = ring + buf * BLK;                             \
    }
    // (not __syncthreads(): its fence would also wait for the activation / dZ stores issued a moment ago -- a memory round trip
    // per block; the barrier only orders LDS traffic: this wave's ring writes have landed, everyone's reads of `cur` are done)
#define TG_F32_BLOCK_END                                        \
    if (!resident) {                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
        buf ^= 1;                                               \
    }

    // one 32-feature output tile against a whole H-wide operand: H / 2 MFMA steps; step (mt, t) multiplies the block's k pair
    // F(t, 0 / 1) of input tile mt = register t of the two lane halves
    auto tile_products = [&](const uint4* __restrict__ cur, const f32x16 (&xin)[MT], f32x16 acc) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 w4 = cur[(4 * mt + q) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.x), xin[mt][4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.y), xin[mt][4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.z), xin[mt][4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.w), xin[mt][4 * q + 3], acc, 0, 0, 0);
            }
        return acc;
    };
    // ReLU in place + its mask as 16 bits
    auto relu_bits = [&](f32x16& v) {
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            m |= (v[r] > 0.0f ? 1u : 0u) << r;
            v[r] = fmaxf(v[r], 0.0f);
        }
        return m;
    };
    auto bits_slot = [&](int layer, int word) { return bits_s + ((layer * 8 + wave) * (MT / 2) + word) * 64 + lane; };

    double s_surr = 0.0, s_crit = 0.0, s_kl = 0.0, s_cnt = 0.0;
    f32x16 pend_v = {};
    float* pend_g = nullptr;                            // (wave-uniform) destination of the deferred tile, or null
    int pend_mt = 0;
    int64_t pend_row = 0;
    bool pend_valid = false;
    auto flush_store = [&]() {
        if (pend_g != nullptr) {
            if (pend_valid) store_tile<H>(pend_g, pend_row, pend_mt, h, pend_v);
            pend_g = nullptr;
        }
    };

    for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t row = round * 256 + wave * 32 + j;
        const bool valid = row < rows;
        auto defer_store = [&](float* g, int mt, const f32x16& v) {
            if (resident) {
                if (valid) store_tile<H>(g, row, mt, h, v);
            } else {
                pend_g = g; pend_mt = mt; pend_v = v; pend_row = row; pend_valid = valid;
            }
        };
        const int64_t rowc = valid ? row : rows - 1;
        f32x16 xin[MT], xout[MT];

        // ---- layer 0: K = 2 k2 <= 32 inputs; lane half h holds x[h k2 .. h k2 + k2) (step t pairs columns t and k2 + t) ----
        {
            float xr[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = float4{0.f, 0.f, 0.f, 0.f};
                if (4 * q < K2) v = *reinterpret_cast<const float4*>(a.x + rowc * (2 * K2) + h * K2 + 4 * q);
                xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                f32x16 acc = bias_rows(bias_s + 32 * mo + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * q < K2) {
                        const uint4 w4 = w0s[(mo * (K2 / 4) + q) * 64 + lane];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.x), xr[4 * q + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.y), xr[4 * q + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.z), xr[4 * q + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w4.w), xr[4 * q + 3], acc, 0, 0, 0);
                    }
                const uint32_t m = relu_bits(acc);
                xin[mo] = acc;
                if constexpr (kTrain) {
                    if (mo & 1) *bits_slot(0, mo >> 1) |= m << 16; else *bits_slot(0, mo >> 1) = m;
                    if (valid && a.acts[0] != nullptr) store_tile<H>(a.acts[0], row, mo, h, acc);      // (null: the weight-gradient job recomputes it)
                }
            }
        }
        // ---- hidden H x H layers: one streamed block per 32 output features ----
        for (int l = 1; l <= n_hh; ++l) {
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                TG_F32_BLOCK_BEGIN
                f32x16 acc = tile_products(cur, xin, bias_rows(bias_s + l * H + 32 * mo + 4 * h));
                const uint32_t m = relu_bits(acc);
                xout[mo] = acc;
                if constexpr (kTrain) {
                    if (mo & 1) *bits_slot(l, mo >> 1) |= m << 16; else *bits_slot(l, mo >> 1) = m;
                    defer_store(a.acts[l], mo, acc);
                }
                TG_F32_BLOCK_END
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xin[mt] = xout[mt];
        }
        // ---- head: <= 4 outputs as fp32 dot products over the lane's features, the two halves added by one exchange ----
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        const int A = kTrain ? L.A : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < A) {
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w = *reinterpret_cast<const float4*>(wh_s + k * H + 32 * mt + 8 * q + 4 * h);
                        s = fmaf(xin[mt][4 * q + 0], w.x, s);
                        s = fmaf(xin[mt][4 * q + 1], w.y, s);
                        s = fmaf(xin[mt][4 * q + 2], w.z, s);
                        s = fmaf(xin[mt][4 * q + 3], w.w, s);
                    }
                // fixed order: half 0 + half 1 (both halves end up with the same bits)
                const float other = __shfl_xor(s, 32, 64);
                o[k] = (h ? other + s : s + other) + bh_s[k];
            }
        if constexpr (!kTrain) {
            if (valid && h == 0) *reinterpret_cast<float4*>(a.out + row * 4) = float4{o[0], o[1], o[2], o[3]};
        } else {
            // ---- loss head (loss_kernels.hip::surrogate_loss_kernel, same arithmetic), evaluated by both lane halves ----
            float g[4], c_surr, c_crit, c_kl;
            f32_loss_row(L, o, row, rowc, valid, h == 0, g, c_surr, c_crit, c_kl);
            if (valid && h == 0) {
                s_surr += (double)c_surr; s_crit += (double)c_crit; s_kl += (double)c_kl; s_cnt += 1.0;
                *reinterpret_cast<float4*>(L.dout4 + row * 4) = float4{g[0], g[1], g[2], g[3]};
            }
            // ---- backward: dZ_top = (g . W_head) * (a_top > 0), then dZ_below = (W^T . dZ) * mask per layer, top down ----
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint32_t mw = *bits_slot(n_hh, mt >> 1) >> (16 * (mt & 1));
                f32x16 d;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 s = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < L.A) {
                            const float4 w = *reinterpret_cast<const float4*>(wh_s + k * H + 32 * mt + 8 * q + 4 * h);
                            s.x = fmaf(g[k], w.x, s.x); s.y = fmaf(g[k], w.y, s.y); s.z = fmaf(g[k], w.z, s.z); s.w = fmaf(g[k], w.w, s.w);
                        }
                    d[4 * q + 0] = (mw >> (4 * q + 0)) & 1u ? s.x : 0.f;
                    d[4 * q + 1] = (mw >> (4 * q + 1)) & 1u ? s.y : 0.f;
                    d[4 * q + 2] = (mw >> (4 * q + 2)) & 1u ? s.z : 0.f;
                    d[4 * q + 3] = (mw >> (4 * q + 3)) & 1u ? s.w : 0.f;
                }
                xin[mt] = d;
                if (valid && a.dz[n_hh] != nullptr) store_tile<H>(a.dz[n_hh], row, mt, h, d);         // (null: recomputed from g and the mask bits)
            }
            if (valid && a.top_mask != nullptr) {
#pragma unroll
                for (int w = 0; w < MT / 2; ++w) a.top_mask[row * 4 + h * (MT / 2) + w] = *bits_slot(n_hh, w);
            }
            for (int l = n_hh; l >= 1; --l) {
#pragma unroll
                for (int ko = 0; ko < MT; ++ko) {
                    TG_F32_BLOCK_BEGIN
                    f32x16 acc = tile_products(cur, xin, f32x16{});
                    const uint32_t mw = *bits_slot(l - 1, ko >> 1) >> (16 * (ko & 1));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = (mw >> r) & 1u ? acc[r] : 0.f;
                    xout[ko] = acc;
                    defer_store(a.dz[l - 1], ko, acc);
                    TG_F32_BLOCK_END
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xin[mt] = xout[mt];
            }
        }
    }
    if constexpr (kTrain) flush_store();
#undef TG_F32_BLOCK_BEGIN
#undef TG_F32_BLOCK_END
    if constexpr (kTrain) {
        // loss sums: lanes -> wave (fixed shuffle tree) -> workgroup (waves in order): deterministic
        double v[4] = {s_surr, s_crit, s_kl, s_cnt};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red_s[wave * 4 + k] = v[k];
        }
        __syncthreads();
        if (tid < 4) {
            double t = 0.0;
            for (int w = 0; w < 8; ++w) t += red_s[w * 4 + tid];
            L.work[(int64_t)blockIdx.x * 4 + tid] = t;
        }
    }
    if constexpr (kTrain) { TG_CLOCK_PROBE_END(g_probe_f32_chain) }
}

template <int H>
static size_t f32_chain_lds(int n_hh, int k2, int ring_blocks) {
    constexpr int MT = H / 32, BLK = (H / 8) * 64;
    return (size_t)ring_blocks * BLK * 16 + (size_t)MT * (k2 / 4) * 64 * 16 + (size_t)(kF32MaxHidden * H + 4 * H + 16) * 4 +
           (size_t)(n_hh + 1) * 8 * (MT / 2) * 64 * 4 + 8 * 4 * 8;
}

static int f32_chain_grid(int64_t rows) {
    const int64_t n_rounds = ceil_div(rows, 256);
    const int cus = device_cus();
    return (int)(n_rounds < cus ? n_rounds : cus);
}

template <int H, bool kTrain>
static int launch_f32_chain(const F32ChainArgs& args_in, hipStream_t st) {
    auto kern = mlp_f32_chain_kernel<H, kTrain>;
    F32ChainArgs args = args_in;
    const int n_stream = args.net.n_hh * (H / 32) * (kTrain ? 2 : 1);
    args.resident = n_stream > 0 && f32_chain_lds<H>(args.net.n_hh, args.net.k2, n_stream) <= 160 * 1024;
    const size_t shmem = f32_chain_lds<H>(args.net.n_hh, args.net.k2, args.resident ? n_stream : 2);
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32_forward")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)f32_chain_grid(args.rows)), dim3(512), shmem, st, args);
    TG_LAUNCH_CHECK("tg_mlp_f32_forward");
    return TG_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradients: dW_l = dZ_l^T . A_{l-1} (A_{-1} = the input), db_l = column sums of dZ_l, head: dW_h = g^T . A_top, db_h.
// The contraction runs over ROWS: A operand lane (i, kk) = P[row 2 s + kk][m0 + i], B operand lane (j, kk) = Q[row 2 s + kk][n0 + j]
// straight out of row-major LDS panels (one conflict-free ds_read_b32 per operand and MFMA).  A workgroup (4 waves) owns ONE
// job (layer) for a share of the 32-row stages and keeps that layer's whole gradient in accumulator registers; panels come in
// through registers one stage ahead (the products of a stage take 64 MFMAs x 64 cycles per wave: nothing to hide behind).
// Bias sums and the head's gradient (4 x H) are fp32 vector arithmetic beside the matrix work.
// ------------------------------------------------------------------------------------------------------------------------
// (kF32DwMaxJobs, F32DwJob, F32DwArgs: f32_dw.hpp -- shared with the H = 256 job of mlp_f32_wide.hip)

// Stages flow HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`: no staging registers) through a ring of kSlots slots with
// kSlots - 1 stages in flight, counted `s_waitcnt vmcnt` + ONE raw `s_barrier` per stage (mfma_ring.hpp's discipline: every
// wave issues the same number of DMA instructions per stage -- stages past the end re-read the last row, whose products are
// masked -- and every LDS read in the loop goes through a `__restrict__` helper, or hipcc drains the ring before it).
//   wide job (H x H layer): stage = SRW rows of P and of Q (16 KiB), 4 slots;
//   light job (first layer: P = dZ_0, narrow Q = the input rows; head: wide Q = the top activation, narrow P = d loss / d
//   output): stage = SRL rows of the wide operand (16 KiB) + the narrow operand as a zero-padded [SRL][32] image (4-8 KiB), 3 slots.
__device__ uint4 g_f32_zero16;
// Its address, held in a register pair by the kernel that uses it (`const uint4* zero16 = f32_zero16_addr();` at the top): written
// as `&g_f32_zero16` at the use, hipcc re-loads the symbol's address through the GOT in front of every LDS-DMA -- a scalar memory
// round trip per piece inside the stage loops (found in round 5 on the H = 256 jobs: ~15 % of a stage).
__device__ static inline const uint4* f32_zero16_addr() {
    const uint4* z = &g_f32_zero16;
    asm volatile("" : "+s"(z));
    return z;
}
#if TG_F32DW_STAMPS
__device__ unsigned long long g_f32_stamps[4096 * 4];      // per wave: cycles in [wait + bias][arrive][products + reads], stages
__device__ unsigned long long g_f32_stamps3[4096 * 12];     // fused job, per wave: cycles in [wait + barrier][issue][phase 1][barrier][operand reads][products], stages
__device__ unsigned long long g_f32_stamps2[4096 * 6];     // per wave: s_memtime at entry / loop start / loop end / exit, s_memrealtime at entry / exit
#endif                             // 16 zero bytes: the source of an image's padding lanes

typedef __attribute__((address_space(3))) void f32_lds_void;
__device__ static inline float lds_f(const float* __restrict__ p) { return *p; }
__device__ static inline float4 lds_f4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }
__device__ static inline uint4 lds_u4(const uint32_t* __restrict__ p) { return *reinterpret_cast<const uint4*>(p); }
__device__ static inline void lds_st(float* __restrict__ p, float v) { *p = v; }
__device__ static inline uint32_t lds_u(const uint32_t* __restrict__ p) { return *p; }

template <int H>
struct F32DwGeom {
    static constexpr int SRW = H == 128 ? 16 : 32;          // rows per stage of a wide job (P + Q = 16 KiB)
    static constexpr int SRL = H == 128 ? 32 : 64;          // rows per stage of a light job (wide operand = 16 KiB)
    static constexpr int LPR = H / 4;                       // lanes (16 B each) per wide row
    static constexpr int RPP = 64 / LPR;                    // wide rows per 1-KiB piece
    static constexpr int WIDE_SLOT = 16384, LIGHT_SLOT = 16384 + SRL * 128;
    static constexpr int WIDE_SLOTS = 4, LIGHT_SLOTS = 3;
    static constexpr int NG_WIDE = 4, NG_LIGHT = 4 + SRL / 32;      // DMA instructions per wave and stage
    static constexpr int LDS_PLAIN = WIDE_SLOT * WIDE_SLOTS > LIGHT_SLOT * LIGHT_SLOTS ? WIDE_SLOT * WIDE_SLOTS : LIGHT_SLOT * LIGHT_SLOTS;
    static constexpr int LDS_MAX = 79 * 1024;               // two workgroups per CU (160 KiB)
};

// `rows_in` rows x H floats from `g` (row-major) starting at row r0 into a linear LDS panel: this wave's 4 pieces of the 16
// (kZeroTail: rows past the end arrive as zeros instead of as re-reads of the last row)
template <int H, bool kZeroTail = false>
__device__ static inline void f32_dma_wide(const float* __restrict__ g, int64_t r0, int64_t rows, char* panel, int first_piece,
                                           int wave, int lane, const uint4* zero16 = nullptr) {
    using G = F32DwGeom<H>;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int piece = first_piece + 2 * wave + t;       // 8 pieces per 8-KiB half: waves take 2 each
        int64_t r = r0 + (int64_t)(piece - first_piece) * G::RPP + lane / G::LPR;
        const uint4* src = reinterpret_cast<const uint4*>(g + (r < rows ? r : rows - 1) * H) + lane % G::LPR;
        if constexpr (kZeroTail) src = r < rows ? src : zero16;
        __builtin_amdgcn_global_load_lds(src, (f32_lds_void*)(panel + piece * 1024), 16, 0, 0);
    }
}

// Wide job whose operands are REBUILT on chip instead of streamed, with the net's two light gradients riding on it (the fp32
// sibling of tg_mlp_weight_grad's kinds HR / RH).  The first hidden activation is a function of the 32..128-B input row, the top
// layer's dZ of the 16-B d loss / d output row and 16 B of mask bits -- so tg_mlp_f32_forward_backward need not write them (512 B
// per row each at H = 128) and this job need not read them.  What the job streams instead are the wide operands of the two light
// gradients, which then need no workgroups (and no matrix pipe idling beside their byte streams) of their own:
//   kRecP (top layer's job):    P = (g . W_head) * mask rebuilt;  rider: the head's gradient  dW_h = g^T . A_top, db_h = sum g
//   kRecQ (second layer's job): Q = relu(W0 x + b0) rebuilt;      rider: the first layer's    dW_0 = dZ_0^T . x,  db_0 = sum dZ_0
// (a net with two hidden layers has ONE such job carrying both).  Per stage of SRW rows: the ring slot holds the two streamed
// panels + the input rows as a zero-padded [SRW][32] image + the g | mask rows; phase 1: thread = (feature f, 8 rows) fills the
// rebuilt panel(s) -- fp32 FMA chains, k ascending -- and does the riders' vector work (bias sums, head products) on the values
// passing through its registers; barrier; phase 2: the matrix products, the first layer's on one extra tile per wave.
template <int H, bool kRecP, bool kRecQ>
struct F32FusedGeom {
    static constexpr int SR = F32DwGeom<H>::SRW;
    static constexpr int PANEL = SR * H * 4;                                 // 8 KiB
    static constexpr int OFF_P = 0;                                         // streamed P (if not rebuilt)
    static constexpr int OFF_Q = OFF_P + (kRecP ? 0 : PANEL);               // streamed Q (if not rebuilt)
    static constexpr int OFF_AT = OFF_Q + (kRecQ ? 0 : PANEL);              // head rider: the top activation
    static constexpr int OFF_Z0 = OFF_AT + (kRecP ? PANEL : 0);             // first-layer rider: the bottom dZ
    static constexpr int OFF_X = OFF_Z0 + (kRecQ ? PANEL : 0);              // input rows, [SR][32] floats
    static constexpr int OFF_G = OFF_X + (kRecQ ? SR * 128 : 0);            // g rows [32][4] floats, then mask rows [32][4] words
    static constexpr int SLOT = OFF_G + (kRecP ? 1024 : 0);
    static constexpr int N_SMALL = (kRecQ ? SR / 8 : 0) + (kRecP ? 1 : 0);  // 1-KiB pieces of the small operands
    static constexpr int SPW = (N_SMALL + 3) / 4;                           // ... per wave and stage (a wave may repeat a piece)
    static constexpr int NG = 4 + SPW;                                      // DMA instructions per wave and stage (always two panels)
    static constexpr int REC_PANELS = ((kRecP ? 1 : 0) + (kRecQ ? 1 : 0)) * PANEL;
    static constexpr int lds_bytes(int slots, int in_pad) { return slots * SLOT + REC_PANELS + (kRecQ ? H * (in_pad + 4) * 4 : 0); }
    static constexpr int SLAB = H * H + H + (kRecQ ? H * 32 + H : 0) + (kRecP ? 4 * H + 4 : 0);
};

template <int H, bool kRecP, bool kRecQ>
__device__ static void f32_dw_fused(const F32DwJob& job, int64_t rows, char* lds_c, float* __restrict__ ws) {
    const uint4* zero16 = f32_zero16_addr();
    using F = F32FusedGeom<H, kRecP, kRecQ>;
    constexpr int MT = H / 32, TW = MT >= 4 ? 2 : 1;
    constexpr int SR = F::SR, NG = F::NG;
#if TG_F32DW_STAMPS
    const unsigned long long fs_entry = __builtin_amdgcn_s_memtime(), fs_rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long fs_loop0 = 0, fs_loop1 = 0;
#endif
    constexpr int KS = 4 / MT;                                          // the first layer's tile: k-steps split over KS waves (H = 64: 2)
    static_assert(SR / (256 / H) == 8, "a thread rebuilds 8 rows of one feature per stage");
    const int D = job.ring_slots;                                       // 2 or 3 (host: what fits 79 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int my = (int)blockIdx.x - job.first_block, nb = job.n_blocks;
    const int64_t n_st = (rows + SR - 1) / SR;
    const int f = tid % H;                                              // this thread's feature ...
    const int rb = (wave * 64 / H) * 8;                                 // ... and first row of the stage (wave-uniform)
    float* Pp = reinterpret_cast<float*>(lds_c + D * F::SLOT);          // rebuilt panels [SR][H], behind the ring
    float* Qp = Pp + (kRecP ? SR * H : 0);
    float* w0_s = Qp + (kRecQ ? SR * H : 0);                            // [H][in_pad + 4]
    const int wstride = job.in_pad + 4;
    const int thin_f4 = job.in_pad / 4;
    float b0v = 0.f, whv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kRecQ) {
        for (int q = tid; q < H * job.in_pad; q += 256) {
            const int ff = q / job.in_pad, k = q - ff * job.in_pad;
            w0_s[ff * wstride + k] = k < job.in_dim ? job.w0[ff * job.in_dim + k] : 0.f;
        }
        b0v = job.b0[f];
    }
    if constexpr (kRecP) {
#pragma unroll
        for (int a = 0; a < 4; ++a) whv[a] = a < job.act_dim ? job.wh[a * H + f] : 0.f;
    }
    // feature f = 32 mt + 8 q + 4 hh + low is bit low + 4 q + 16 (mt & 1) of word hh * (MT / 2) + (mt >> 1) of its row's mask
    const int m_word = ((f >> 2) & 1) * (MT / 2) + (f >> 6), m_shift = (f & 3) + 4 * ((f & 31) >> 3) + 16 * ((f >> 5) & 1);

    auto issue = [&](int64_t sg, int slot) {
        char* sb = lds_c + slot * F::SLOT;
        const int64_t r0 = sg * SR;
        // (P-side operands arrive as zeros past the last row: no product or sum needs masking)
        if constexpr (!kRecP) f32_dma_wide<H, true>(job.p, r0, rows, sb + F::OFF_P, 0, wave, lane, zero16);
        if constexpr (!kRecQ) f32_dma_wide<H>(job.q, r0, rows, sb + F::OFF_Q, 0, wave, lane);
        if constexpr (kRecP) f32_dma_wide<H>(job.a_top, r0, rows, sb + F::OFF_AT, 0, wave, lane);
        if constexpr (kRecQ) f32_dma_wide<H, true>(job.dz0, r0, rows, sb + F::OFF_Z0, 0, wave, lane, zero16);
#pragma unroll
        for (int t = 0; t < F::SPW; ++t) {
            const int pc = (wave + 4 * t) % F::N_SMALL;
            if (kRecQ && pc < SR / 8) {
                // the input rows as a zero-padded [SR][32 floats] image: 8 lanes per row, 8 rows per piece
                int64_t r = r0 + pc * 8 + (lane >> 3);
                r = r < rows ? r : rows - 1;
                const int c4 = lane & 7;
                const uint4* src = c4 < thin_f4 ? reinterpret_cast<const uint4*>(job.q + r * job.in_pad) + c4 : zero16;
                __builtin_amdgcn_global_load_lds(src, (f32_lds_void*)(sb + F::OFF_X + pc * 1024), 16, 0, 0);
            } else {
                // lanes [0, 32): the stage's g rows (16 B each; zeros past the last row); lanes [32, 64): its mask rows
                const int rr = lane & 31;
                const int64_t r = r0 + (rr < SR ? rr : SR - 1);
                const int64_t rc = r < rows ? r : rows - 1;
                const uint4* src = lane < 32 ? (r < rows ? reinterpret_cast<const uint4*>(job.p) + rc : zero16)
                                             : reinterpret_cast<const uint4*>(job.mask) + rc;
                __builtin_amdgcn_global_load_lds(src, (f32_lds_void*)(sb + F::OFF_G), 16, 0, 0);
            }
        }
    };
    f32x16 acc[TW][TW], acc0 = f32x16{};
#pragma unroll
    for (int x = 0; x < TW; ++x)
#pragma unroll
        for (int y = 0; y < TW; ++y) acc[x][y] = f32x16{};
    float bsum = 0.f, b0sum = 0.f, hacc[4] = {0.f, 0.f, 0.f, 0.f}, gsum[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t sg_issue = my;
    int slot_issue = 0, slot = 0;
    __syncthreads();                                                    // the table is in place (ordinary stores: before any DMA)
#pragma unroll 1
    for (int t = 0; t < D - 1; ++t) {
        issue(sg_issue, slot_issue);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
    }
    const int tile0 = wave % MT, ks0 = wave / MT;                       // first-layer rider: this wave's tile and k-step phase
#if TG_F32DW_STAMPS
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
    fs_loop0 = __builtin_amdgcn_s_memtime();
#define TG_FSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - st_t; st_t = now_; }
#else
#define TG_FSTAMP(k)
#endif
#pragma unroll 1
    for (int64_t sg = my; sg < n_st; sg += nb) {
#if TG_F32DW_STAMPS
        unsigned long long st_t = __builtin_amdgcn_s_memtime();
        st[6] += 1;
#endif
        if (D == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // the stage has landed for every wave; every wave is done with the previous stage's slot and panels
        asm volatile("" ::: "memory");
        TG_FSTAMP(0)
        issue(sg_issue, slot_issue);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        char* sb = lds_c + slot * F::SLOT;
        slot = slot + 1 == D ? 0 : slot + 1;
        const float* X = reinterpret_cast<const float*>(sb + F::OFF_X);
        const float* Gm = reinterpret_cast<const float*>(sb + F::OFF_G);
        const uint32_t* Mm = reinterpret_cast<const uint32_t*>(sb + F::OFF_G + 512);
        const float* AT = reinterpret_cast<const float*>(sb + F::OFF_AT);
        const float* Z0 = reinterpret_cast<const float*>(sb + F::OFF_Z0);
        TG_FSTAMP(1)
        // ---- phase 1: rebuild, and the riders' vector work ----
        if constexpr (kRecQ) {
            // a0[row][f] = relu(b0[f] + sum_k W0[f][k] x[row][k]), k ascending
            float z[8], o[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) z[r] = lds_f(Z0 + (rb + r) * H + f);
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = b0v;
            for (int k4 = 0; k4 < thin_f4; ++k4) {
                const float4 w = lds_f4(w0_s + f * wstride + 4 * k4);
                float4 xv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) xv[r] = lds_f4(X + (rb + r) * 32 + 4 * k4);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    o[r] = fmaf(w.x, xv[r].x, o[r]); o[r] = fmaf(w.y, xv[r].y, o[r]); o[r] = fmaf(w.z, xv[r].z, o[r]); o[r] = fmaf(w.w, xv[r].w, o[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_st(Qp + (rb + r) * H + f, fmaxf(o[r], 0.f));
#pragma unroll
            for (int r = 0; r < 8; ++r) b0sum += z[r];
        }
        if constexpr (kRecP) {
            // dZ_top[row][f] = (sum_a g[row][a] W_head[a][f]) * bit(row, f), as the chain kernel forms it
            float at[8];
            float4 g4[8];
            uint32_t mw[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                at[r] = lds_f(AT + (rb + r) * H + f);
                g4[r] = lds_f4(Gm + 4 * (rb + r));
                mw[r] = lds_u(Mm + 4 * (rb + r) + m_word);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t word = mw[r];
                float v = whv[0] * g4[r].x;
                v = fmaf(whv[1], g4[r].y, v); v = fmaf(whv[2], g4[r].z, v); v = fmaf(whv[3], g4[r].w, v);
                const float pv = ((word >> m_shift) & 1u) != 0u ? v : 0.f;
                lds_st(Pp + (rb + r) * H + f, pv);
                bsum += pv;
                hacc[0] = fmaf(g4[r].x, at[r], hacc[0]); hacc[1] = fmaf(g4[r].y, at[r], hacc[1]);
                hacc[2] = fmaf(g4[r].z, at[r], hacc[2]); hacc[3] = fmaf(g4[r].w, at[r], hacc[3]);
                gsum[0] += g4[r].x; gsum[1] += g4[r].y; gsum[2] += g4[r].z; gsum[3] += g4[r].w;
            }
        } else {
            const float* Ps = reinterpret_cast<const float*>(sb + F::OFF_P);
            float pz[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) pz[r] = lds_f(Ps + (rb + r) * H + f);
#pragma unroll
            for (int r = 0; r < 8; ++r) bsum += pz[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this thread's panel writes have landed ...
        TG_FSTAMP(2)
        __builtin_amdgcn_s_barrier();                               // ... and everyone's
        asm volatile("" ::: "memory");
        TG_FSTAMP(3)
        // ---- phase 2: products (every operand in registers first: one exposed LDS latency per stage) ----
        const float* Pa = kRecP ? Pp : reinterpret_cast<const float*>(sb + F::OFF_P);
        const float* Qa = kRecQ ? Qp : reinterpret_cast<const float*>(sb + F::OFF_Q);
        float av[SR / 2][TW], bv[SR / 2][TW];
#pragma unroll
        for (int s = 0; s < SR / 2; ++s)
#pragma unroll
            for (int x = 0; x < TW; ++x) {
                av[s][x] = lds_f(Pa + (2 * s + kk) * H + 32 * (TW * wm + x) + i);
                bv[s][x] = lds_f(Qa + (2 * s + kk) * H + 32 * (TW * wn + x) + i);
            }
        float za[SR / 2 / KS], xb[SR / 2 / KS];
        if constexpr (kRecQ) {
#pragma unroll
            for (int s = 0; s < SR / 2 / KS; ++s) {
                const int row = 2 * (KS * s + ks0) + kk;
                za[s] = lds_f(Z0 + row * H + 32 * tile0 + i);
                xb[s] = lds_f(X + row * 32 + i);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < SR / 2; ++s)
#pragma unroll
            for (int x = 0; x < TW; ++x) asm volatile("" : "+v"(av[s][x]), "+v"(bv[s][x]));     // (the products stay behind the wait)
        TG_FSTAMP(4)
#pragma unroll
        for (int s = 0; s < SR / 2; ++s) {
#pragma unroll
            for (int x = 0; x < TW; ++x)
#pragma unroll
                for (int y = 0; y < TW; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][x], bv[s][y], acc[x][y], 0, 0, 0);
            if constexpr (kRecQ) {                                  // the rider's step between the wide steps: its chain is one accumulator
                if (s % KS == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(za[s / KS], xb[s / KS], acc0, 0, 0, 0);
            }
        }
        TG_FSTAMP(5)
    }
#if TG_F32DW_STAMPS
    fs_loop1 = __builtin_amdgcn_s_memtime();
#endif
#undef TG_FSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // no LDS-DMA may outlive the workgroup's LDS allocation
    __syncthreads();
    // ---- the row groups' partial sums (and, H = 64, the k-step phases' partial tiles) meet in LDS, added in a fixed order ----
    constexpr int RG = 256 / H;
    float* red = reinterpret_cast<float*>(lds_c);                       // [6][256] floats, [4][4] (g sums per wave), then [KS - 1][MT][16][64]
    red[tid] = bsum;
    red[256 + tid] = b0sum;
#pragma unroll
    for (int a = 0; a < 4; ++a) red[(2 + a) * 256 + tid] = hacc[a];
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) red[6 * 256 + wave * 4 + a] = gsum[a];    // (every thread of a row group holds the same row sums)
    }
    float* tiles = red + 7 * 256;
    if (kRecQ && KS > 1 && ks0 > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tiles[(((ks0 - 1) * MT + tile0) * 16 + r) * 64 + lane] = acc0[r];
    }
    __syncthreads();
    float* slab = ws + job.slab_off + (int64_t)my * job.slab_len;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int x = 0; x < TW; ++x)
#pragma unroll
        for (int y = 0; y < TW; ++y) {
            const int m0 = 32 * (TW * wm + x), n0 = 32 * (TW * wn + y);
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * H + n0 + col] = acc[x][y][r];
        }
    float* s1 = slab + H * H + H;                                       // first-layer rider: [H][32] + [H]
    float* s2 = s1 + (kRecQ ? H * 32 + H : 0);                          // head rider: [4][H] + [4]
    if (tid < H) {
        float t[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            t[q] = red[q * 256 + tid];
#pragma unroll
            for (int g = 1; g < RG; ++g) t[q] += red[q * 256 + g * H + tid];
        }
        slab[H * H + tid] = t[0];
        if constexpr (kRecQ) s1[H * 32 + tid] = t[1];
        if constexpr (kRecP) {
#pragma unroll
            for (int a = 0; a < 4; ++a) s2[a * H + tid] = t[2 + a];
            if (tid < 4) {
                // one wave per row group (the first of each group's waves): groups added in order
                float gs = 0.f;
#pragma unroll
                for (int g = 0; g < RG; ++g) gs += red[6 * 256 + (g * (H / 64)) * 4 + tid];
                s2[4 * H + tid] = gs;
            }
        }
    }
    if constexpr (kRecQ) {
        if (ks0 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc0[r];
#pragma unroll
                for (int k = 1; k < KS; ++k) v += tiles[(((k - 1) * MT + tile0) * 16 + r) * 64 + lane];
                s1[(32 * tile0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = v;
            }
        }
    }
#if TG_F32DW_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_f32_stamps3 + ((size_t)blockIdx.x * 4 + wave) * 12;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = st[k];
        o[7] = fs_entry; o[8] = fs_loop0; o[9] = fs_loop1; o[10] = __builtin_amdgcn_s_memtime(); o[11] = __builtin_amdgcn_s_memrealtime() - fs_rt0;
    }
#endif
}

template <int H>
__global__ __launch_bounds__(256, 2) void mlp_f32_dw_kernel(F32DwArgs args, int64_t rows, float* __restrict__ ws) {
    const uint4* zero16 = f32_zero16_addr();
    using G = F32DwGeom<H>;
    constexpr int MT = H / 32;
    constexpr int TW = MT >= 4 ? 2 : 1;                 // a wave's block of output tiles is TW x TW (H = 128: 2 x 2; H = 64: 1 x 1)
    extern __shared__ uint4 lds[];
    char* lds_c = reinterpret_cast<char*>(lds);
    TG_CLOCK_PROBE_BEGIN(g_probe_f32_dw)
#if TG_F32DW_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime(), st_rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_loop0 = 0, st_loop1 = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    int jb = 0;
#pragma unroll
    for (int t = 1; t < kF32DwMaxJobs; ++t)
        if (t < args.n_jobs && (int)blockIdx.x >= args.job[t].first_block) jb = t;
    const F32DwJob job = args.job[jb];
    const int my = (int)blockIdx.x - job.first_block, nb = job.n_blocks;
    const bool head = job.kind == F32DW_HEAD;
    const int N = job.n;                                // Q columns
    const bool narrow = !head && N <= 32;               // first-layer job: Q = the padded input rows
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TW][TW];
#pragma unroll
    for (int x = 0; x < TW; ++x)
#pragma unroll
        for (int y = 0; y < TW; ++y) acc[x][y] = f32x16{};
    float bsum = 0.f;                                   // bias gradient of one column (see the roles below)
    float hacc[4] = {0.f, 0.f, 0.f, 0.f};               // head: dW_h[a][tid]

    if (!head && !narrow && job.recompute != 0) {
        if (job.recompute == 3) f32_dw_fused<H, true, true>(job, rows, lds_c, ws);
        else if (job.recompute == 2) f32_dw_fused<H, true, false>(job, rows, lds_c, ws);
        else f32_dw_fused<H, false, true>(job, rows, lds_c, ws);
        TG_CLOCK_PROBE_END(g_probe_f32_dw)
        return;
    }
    if (!head && !narrow) {
        // ================= wide job: dW = P^T Q over H x H, SRW rows per stage =================
        constexpr int SR = G::SRW, D = G::WIDE_SLOTS, P_ = D - 1, NG = G::NG_WIDE;
        const int64_t n_st = (rows + SR - 1) / SR;
        auto issue = [&](int64_t sg, int slot) {
            char* sb = lds_c + slot * G::WIDE_SLOT;
            f32_dma_wide<H>(job.p, sg * SR, rows, sb, 0, wave, lane);
            f32_dma_wide<H>(job.q, sg * SR, rows, sb, 8, wave, lane);
        };
        int64_t sg_issue = my;
        int slot_issue = 0, slot = 0;
#pragma unroll 1
        for (int t = 0; t < P_; ++t) {
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        }
        // Software pipeline over stages: a stage's operands (SR / 2 x 2 TW registers) and bias terms are read from LDS while the
        // PREVIOUS stage's products run -- an MFMA leaves the wave ~56 issue cycles, and read back to back in front of the
        // products, wait, barrier, DMA issue and bias sums were 1,350 exposed cycles per 2,048 cycles of products.
        constexpr int RG = 256 / H, RPG = SR / RG;                          // bias sums: row groups, rows per group
        const int cb = tid % H, rg = tid / H;
        float av[SR / 2][TW], bv[SR / 2][TW], an[SR / 2][TW], bn[SR / 2][TW], bt[RPG];
        auto arrive = [&]() {                                               // the next stage of the ring is complete; refill the freed slot
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P_ - 1) * NG) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#if TG_F32DW_ABLATE != 2                                    /* probe build 2: no stream (the ring keeps its prologue's data) */
            issue(sg_issue, slot_issue);
#endif
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        };
        int64_t sg = my;
        if (sg < n_st) {
            arrive();
            const float* P = reinterpret_cast<const float*>(lds_c + slot * G::WIDE_SLOT);
            const float* Q = P + SR * H;
#pragma unroll
            for (int s = 0; s < SR / 2; ++s)
#pragma unroll
                for (int x = 0; x < TW; ++x) {
                    an[s][x] = lds_f(P + (2 * s + kk) * H + 32 * (TW * wm + x) + i);
                    bn[s][x] = lds_f(Q + (2 * s + kk) * H + 32 * (TW * wn + x) + i);
                }
#pragma unroll
            for (int r = 0; r < RPG; ++r) bt[r] = lds_f(P + (rg * RPG + r) * H + cb);
            slot = slot + 1 == D ? 0 : slot + 1;
        }
#if TG_F32DW_STAMPS
        unsigned long long st_a = 0, st_b = 0, st_c = 0, st_n = 0;
        st_loop0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
        for (; sg < n_st; sg += nb) {
            const int64_t r0 = sg * SR;
#if TG_F32DW_STAMPS
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
            // (every read of the previous stage's slot has landed in registers before this wave passes the barrier in arrive())
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const int nr = rows - r0 < SR ? (int)(rows - r0) : SR;
#pragma unroll
                for (int r = 0; r < RPG; ++r) bsum += rg * RPG + r < nr ? bt[r] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < SR / 2; ++s) {
                const bool ok = r0 + 2 * s + kk < rows;             // rows past the end are clamped re-reads: their products are zeroed
#pragma unroll
                for (int x = 0; x < TW; ++x) {
                    av[s][x] = ok ? an[s][x] : 0.f;
                    bv[s][x] = bn[s][x];
                }
            }
            const bool more = sg + nb < n_st;
#if TG_F32DW_STAMPS
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
            if (more) arrive();
#if TG_F32DW_STAMPS
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
            const float* P = reinterpret_cast<const float*>(lds_c + slot * G::WIDE_SLOT);
            const float* Q = P + SR * H;
#pragma unroll
            for (int s = 0; s < SR / 2; ++s) {
#pragma unroll
                for (int x = 0; x < TW; ++x)
#pragma unroll
                    for (int y = 0; y < TW; ++y) {
#if TG_F32DW_ABLATE == 1                                    /* probe build: no products (operands kept live) */
                        asm volatile("" :: "v"(av[s][x]), "v"(bv[s][y]));
#else
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][x], bv[s][y], acc[x][y], 0, 0, 0);
#endif
                    }
                if (more) {                                         // the next stage's step-s operands, in the shadow of these products
#pragma unroll
                    for (int x = 0; x < TW; ++x) {
                        an[s][x] = lds_f(P + (2 * s + kk) * H + 32 * (TW * wm + x) + i);
                        bn[s][x] = lds_f(Q + (2 * s + kk) * H + 32 * (TW * wn + x) + i);
                    }
                    if (s < RPG) bt[s] = lds_f(P + (rg * RPG + s) * H + cb);
                }
            }
            static_assert(RPG <= SR / 2, "the bias reads ride on the steps");
            if (more) slot = slot + 1 == D ? 0 : slot + 1;
#if TG_F32DW_STAMPS
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            st_a += t1 - t0; st_b += t2 - t1; st_c += t3 - t2; st_n += 1;
#endif
        }
#if TG_F32DW_STAMPS
        st_loop1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x < 1024) {
            unsigned long long* o = g_f32_stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
            o[0] = st_a; o[1] = st_b; o[2] = st_c; o[3] = st_n;
        }
#endif
    } else {
        // ================= light job: one wide operand, SRL rows per stage =================
        constexpr int SR = G::SRL, D = G::LIGHT_SLOTS, P_ = D - 1, NG = G::NG_LIGHT;
        const int64_t n_st = (rows + SR - 1) / SR;
        const float* wide = head ? job.q : job.p;       // head: the top activation; first layer: the bottom dZ
        const float* thin = head ? job.p : job.q;       // head: g [rows][4]; first layer: x [rows][N]
        const int thin_f4 = head ? 1 : N / 4;           // float4 per row of the narrow operand
        auto issue = [&](int64_t sg, int slot) {
            char* sb = lds_c + slot * G::LIGHT_SLOT;
            const int64_t r0 = sg * SR;
            f32_dma_wide<H>(wide, r0, rows, sb, 0, wave, lane);
            f32_dma_wide<H>(wide, r0 + SR / 2, rows, sb, 8, wave, lane);
            // the narrow operand as a zero-padded [SR][32 floats] image: 8 lanes per row, 8 rows per piece, SR / 8 pieces
#pragma unroll
            for (int t = 0; t < SR / 32; ++t) {
                const int piece = (SR / 32) * wave + t;
                int64_t r = r0 + piece * 8 + (lane >> 3);
                r = r < rows ? r : rows - 1;
                const int c4 = lane & 7;
                const uint4* src = c4 < thin_f4 ? reinterpret_cast<const uint4*>(thin + r * (4 * thin_f4)) + c4 : zero16;
                __builtin_amdgcn_global_load_lds(src, (f32_lds_void*)(sb + 16384 + piece * 1024), 16, 0, 0);
            }
        };
        int64_t sg_issue = my;
        int slot_issue = 0, slot = 0;
#pragma unroll 1
        for (int t = 0; t < P_; ++t) {
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        }
#pragma unroll 1
        for (int64_t sg = my; sg < n_st; sg += nb) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P_ - 1) * NG) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
            const float* W = reinterpret_cast<const float*>(lds_c + slot * G::LIGHT_SLOT);
            const float* T = W + 4096;                  // the narrow image [SR][32]
            slot = slot + 1 == D ? 0 : slot + 1;
            const int64_t r0 = sg * SR;
            const int nr = rows - r0 < SR ? (int)(rows - r0) : SR;
            if (head) {
                // (one column per thread, a plain loop over the stage's rows: spreading the rows over all 256 threads with the loads
                // issued together measured SLOWER here -- 167 vs 104 us for 1 M rows: the job is byte-bound, and the extra threads'
                // LDS traffic competes with the co-resident workgroup)
                if (tid < H) {
                    for (int r = 0; r < nr; ++r) {
                        const float4 g4 = lds_f4(T + 32 * r);
                        const float qv = lds_f(W + r * H + tid);
                        hacc[0] = fmaf(g4.x, qv, hacc[0]); hacc[1] = fmaf(g4.y, qv, hacc[1]);
                        hacc[2] = fmaf(g4.z, qv, hacc[2]); hacc[3] = fmaf(g4.w, qv, hacc[3]);
                    }
                } else if (tid < H + 4) {
                    for (int r = 0; r < nr; ++r) bsum += lds_f(T + 32 * r + (tid - H));
                }
            } else {
                if (wave < MT) {
#pragma unroll 8
                    for (int s = 0; s < SR / 2; ++s) {
                        float av = lds_f(W + (2 * s + kk) * H + 32 * wave + i);
                        av = 2 * s + kk < nr ? av : 0.f;
                        const float bv = lds_f(T + (2 * s + kk) * 32 + i);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0][0], 0, 0, 0);
                    }
                }
                if (tid >= 256 - H) {                   // (the waves without a tile when H = 64)
                    for (int r = 0; r < nr; ++r) bsum += lds_f(W + r * H + (tid - (256 - H)));
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation

    // ---- the wide job's row groups' partial bias sums meet in LDS and are added in group order ----
    if (!head && !narrow) {
        constexpr int RG = 256 / H;
        float* red = reinterpret_cast<float*>(lds_c);                       // [256]
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < H) {
            float t = red[tid];
#pragma unroll
            for (int g = 1; g < RG; ++g) t += red[g * H + tid];
            bsum = t;
        }
    }

    // ---- this workgroup's slab ----
    float* slab = ws + job.slab_off + (int64_t)my * job.slab_len;
    const int col = lane & 31, hh = lane >> 5;
    if (head) {
        if (tid < H) {
#pragma unroll
            for (int k = 0; k < 4; ++k) slab[k * H + tid] = hacc[k];
        } else if (tid < H + 4) {
            slab[4 * H + (tid - H)] = bsum;
        }
    } else if (narrow) {
        if (wave < MT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = acc[0][0][r];
        }
        if (tid >= 256 - H) slab[H * 32 + (tid - (256 - H))] = bsum;
    } else {
#pragma unroll
        for (int x = 0; x < TW; ++x)
#pragma unroll
            for (int y = 0; y < TW; ++y) {
                const int m0 = 32 * (TW * wm + x), n0 = 32 * (TW * wn + y);
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * H + n0 + col] = acc[x][y][r];
            }
        if (tid < H) slab[H * H + tid] = bsum;
    }
    TG_CLOCK_PROBE_END(g_probe_f32_dw)
#if TG_F32DW_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_f32_stamps2 + ((size_t)blockIdx.x * 4 + wave) * 6;
        o[0] = st_entry; o[1] = st_loop0; o[2] = st_loop1; o[3] = __builtin_amdgcn_s_memtime(); o[4] = st_rt0; o[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// The fused job (both operands rebuilt, both riders) as ONE 8-wave workgroup per slot, for nets whose only H x H layer is this job
// (C2's 5-128-128-1: pipelines/cartpole_pipeline_grpo.py:54-76).  f32_dw_fused's 4-wave workgroup keeps the layer's 128 x 128
// gradient in 64 accumulator registers per thread and a stage costs each wave a ~4,500-cycle chain of LDS round trips (DMA landed ->
// small rows read -> FMAs -> panel written -> barrier -> operands read) in front of 2,560 cycles of products; with two such waves per
// SIMD the chains overlap the CU-mate's products by a third only (profiles/r03_f32_dw_fused_job.md).  Here the same stage is spread
// over 512 threads: a thread rebuilds 4 rows of its feature instead of 8, a wave owns a 32 x 64 strip of the gradient (32
// accumulator registers, 16 + 4 products per stage), and a SIMD holds FOUR waves (two workgroups per CU as before) whose chains and
// products interleave.  Same slab layout as f32_dw_fused<H, true, true> (the reduction launch does not know which kernel ran).
//   kPipe: one barrier per stage instead of two.  Two ring slots and TWO sets of rebuilt panels: an iteration rebuilds stage t + 1
// (and does its riders) in the same instruction stream as the products of stage t, whose panels the previous iteration wrote -- the
// barrier at the top of the iteration (the next stage's DMA has landed) is also the one that publishes them.  The last iteration
// rebuilds a stage past the end: its P-side operands arrive as zeros, so every sum it touches gets zeros.
//   kLean (with kPipe, padded width 8): the net has at most 5 inputs and ONE output (CartPole, Pendulum: C2) -- the terms that are
// identically zero (input columns 5..7, head rows 1..3) are not computed: a third of the rebuild's vector instructions, which is
// what the stage's time follows (tools/f32_dw_pipe_ablation.sh).  Same bits: fma(0, x, o) == o.
template <int H, int IN_PAD, bool kPipe, bool kLean = false>
__global__ __launch_bounds__(512, 4) void mlp_f32_dw_fused8_kernel(F32DwJob job, int64_t rows, float* __restrict__ ws) {
    const uint4* zero16 = f32_zero16_addr();
    static_assert(!kLean || (kPipe && IN_PAD == 8), "the lean form is a specialisation of the one-barrier job at padded width 8");
    static_assert(H == 128, "one wave per 32 x 64 strip of a 128 x 128 gradient");
    static_assert(IN_PAD % 8 == 0 && IN_PAD >= 8 && IN_PAD <= 32, "padded input width");
    using F = F32FusedGeom<H, true, true>;
    constexpr int SR = F::SR, NW = 8, NG = 3;                           // DMA instructions per wave and stage: one piece of each wide panel + one small piece
    constexpr int RG = 512 / H, RPT = SR / RG;                          // row groups of the rebuild (4), rows per thread (4)
    constexpr int MT = H / 32, KS = NW / MT;                            // the first layer's 4 tiles: k-steps split over KS = 2 waves
    static_assert(SR == 16 && RPT == 4 && F::N_SMALL == 3, "stage geometry");
    // The first layer's gradient dW_0 = dZ_0^T x.  As a product it is one more 32 x 32 tile per wave and k-step pair -- a fifth of the
    // matrix pipe's time for a tile of which in_dim of 32 columns are real.  At padded width 8 the rebuild's thread (feature f, 4
    // rows) already holds dZ_0[row][f] and x[row][0..8) in registers: 8 FMAs per row into 8 accumulators instead (kVecRider).
    constexpr bool kVecRider = kPipe && IN_PAD == 8;
    extern __shared__ uint4 lds[];
    char* lds_c = reinterpret_cast<char*>(lds);
    TG_CLOCK_PROBE_BEGIN(g_probe_f32_dw)
#if TG_F32DW_STAMPS
    const unsigned long long fs_entry = __builtin_amdgcn_s_memtime(), fs_rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long fs_loop0 = 0, fs_loop1 = 0, st[7] = {0, 0, 0, 0, 0, 0, 0};
#define TG_FSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - st_t; st_t = now_; }
#else
#define TG_FSTAMP(k)
#endif
    const int D = kPipe ? 2 : job.ring_slots;                           // 2 or 3 (host: what fits 79 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                            // strip: P features [32 wm, +32) x Q features [64 wn, +64)
    const int my = (int)blockIdx.x, nb = (int)gridDim.x;
    const int64_t n_st = (rows + SR - 1) / SR;
    const int f = tid % H;                                              // this thread's feature ...
    const int rb = (wave >> 1) * RPT;                                   // ... and first row of the stage (wave-uniform)
    float* Pp = reinterpret_cast<float*>(lds_c + D * F::SLOT);          // rebuilt panels [SR][H] (kPipe: two sets of P | Q), behind the ring
    float* Qp = Pp + SR * H;
    float* w0_s = Pp + (kPipe ? 4 : 2) * SR * H;                        // [H][in_pad + 4]
    constexpr int wstride = IN_PAD + 4, thin_f4 = IN_PAD / 4;
    float b0v, whv[4];
    auto load_weights = [&]() {                                         // the first layer's table into LDS, this thread's bias and head weights
        for (int q = tid; q < H * IN_PAD; q += 512) {
            const int ff = q / IN_PAD, k = q - ff * IN_PAD;
            w0_s[ff * wstride + k] = k < job.in_dim ? job.w0[ff * job.in_dim + k] : 0.f;
        }
        b0v = job.b0[f];
#pragma unroll
        for (int a = 0; a < 4; ++a) whv[a] = a < job.act_dim ? job.wh[a * H + f] : 0.f;
    };
    if constexpr (!kPipe) load_weights();
    // feature f = 32 mt + 8 q + 4 hh + low is bit low + 4 q + 16 (mt & 1) of word hh * (MT / 2) + (mt >> 1) of its row's mask
    const int m_word = ((f >> 2) & 1) * (MT / 2) + (f >> 6), m_shift = (f & 3) + 4 * ((f & 31) >> 3) + 16 * ((f >> 5) & 1);

    // Source addresses as a wave-uniform 64-bit base (the stage's first row, clamped into the buffer) + a 32-bit lane offset: what
    // a lane keeps across stages is two small integers, not five 64-bit pointers (at 128 registers per wave those were spilled, and
    // a scratch reload inside the loop is a vector-memory operation in front of which the ring's counted waits would drain).
    const int lane_hi = lane >> 5, lane_c16 = (lane & 31) * 16;
    auto issue = [&](int64_t sg, int slot) {
        char* sb = lds_c + slot * F::SLOT;
        const int64_t r0 = sg * SR;
        const int64_t rbase = r0 < rows ? r0 : rows - 1;                // (stages past the end re-read the last row)
        const int64_t left = rows - 1 - rbase;                          // rows of the buffer beyond rbase
        const int lim = left < SR ? (int)left : SR;                     // a lane's row offset is clamped to [0, lim]
        const int n_valid = r0 < rows ? (rows - r0 < SR ? (int)(rows - r0) : SR) : 0;      // rows of this stage that exist
        // a wide panel = 8 pieces of 2 rows: wave w brings piece w of each (the bottom dZ arrives as zeros past the last row: no
        // product or sum needs masking)
        const int rr = 2 * wave + lane_hi;
        const uint32_t ow = (uint32_t)(rr < lim ? rr : lim) * (H * 4) + lane_c16;
        const char* at_b = reinterpret_cast<const char*>(job.a_top + rbase * H);
        const char* z0_b = reinterpret_cast<const char*>(job.dz0 + rbase * H);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(at_b + ow), (f32_lds_void*)(sb + F::OFF_AT + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(rr < n_valid ? reinterpret_cast<const uint4*>(z0_b + ow) : zero16,
                                         (f32_lds_void*)(sb + F::OFF_Z0 + wave * 1024), 16, 0, 0);
        const int pc = wave % F::N_SMALL;                               // (waves 3..7 repeat a piece: every wave issues NG instructions)
        if (pc < SR / 8) {
            // the input rows as a zero-padded [SR][32 floats] image: 8 lanes per row, 8 rows per piece
            const int rx = pc * 8 + (lane >> 3), c4 = lane & 7;
            const char* x_b = reinterpret_cast<const char*>(job.q + rbase * IN_PAD);
            const uint32_t ox = (uint32_t)(rx < lim ? rx : lim) * (IN_PAD * 4) + c4 * 16;
            __builtin_amdgcn_global_load_lds(c4 < thin_f4 ? reinterpret_cast<const uint4*>(x_b + ox) : zero16,
                                             (f32_lds_void*)(sb + F::OFF_X + pc * 1024), 16, 0, 0);
        } else {
            // lanes [0, 32): the stage's g rows (16 B each; zeros past the last row); lanes [32, 64): its mask rows
            const int rg_ = (lane & 31) < SR ? (lane & 31) : SR - 1;
            const uint32_t og = (uint32_t)(rg_ < lim ? rg_ : lim) * 16;
            const char* g_b = reinterpret_cast<const char*>(job.p + rbase * 4);
            const char* m_b = reinterpret_cast<const char*>(job.mask + rbase * 4);
            const uint4* src = lane < 32 ? (rg_ < n_valid ? reinterpret_cast<const uint4*>(g_b + og) : zero16)
                                         : reinterpret_cast<const uint4*>(m_b + og);
            __builtin_amdgcn_global_load_lds(src, (f32_lds_void*)(sb + F::OFF_G), 16, 0, 0);
        }
    };
    f32x16 acc[2] = {f32x16{}, f32x16{}}, acc0 = f32x16{};
    float w0acc[kVecRider ? IN_PAD : 1] = {};                          // kVecRider: dW_0[f][0..IN_PAD) of this thread's row group
    float bsum = 0.f, b0sum = 0.f, hacc[4] = {0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;                                                   // component (f & 3) of the row group's sum of g rows
    const bool gc1 = (f & 1) != 0, gc2 = (f & 2) != 0;
    const int tile0 = wave % MT, ks0 = wave / MT;                       // first-layer rider: this wave's tile and k-step phase
    // ---- phase 1 of a stage: rebuild into the panels Pw | Qw (k ascending, as the chain kernel forms the values), and the riders'
    // vector work ----
    auto rebuild_q = [&](const char* sb, float* __restrict__ Qw) {
        // a0[row][f] = relu(b0[f] + sum_k W0[f][k] x[row][k])
        const float* X = reinterpret_cast<const float*>(sb + F::OFF_X);
        const float* Z0 = reinterpret_cast<const float*>(sb + F::OFF_Z0);
        float z[RPT], o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { z[r] = lds_f(Z0 + (rb + r) * H + f); o[r] = b0v; }
        if constexpr (kLean) {
            const float4 w = lds_f4(w0_s + f * wstride);
            const float w4 = lds_f(w0_s + f * wstride + 4);
            float4 xv[RPT];
            float x4[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) { xv[r] = lds_f4(X + (rb + r) * 32); x4[r] = lds_f(X + (rb + r) * 32 + 4); }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                o[r] = fmaf(w.x, xv[r].x, o[r]); o[r] = fmaf(w.y, xv[r].y, o[r]); o[r] = fmaf(w.z, xv[r].z, o[r]); o[r] = fmaf(w.w, xv[r].w, o[r]);
                o[r] = fmaf(w4, x4[r], o[r]);
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {                         // rows ascending
                w0acc[0] = fmaf(z[r], xv[r].x, w0acc[0]); w0acc[1] = fmaf(z[r], xv[r].y, w0acc[1]);
                w0acc[2] = fmaf(z[r], xv[r].z, w0acc[2]); w0acc[3] = fmaf(z[r], xv[r].w, w0acc[3]);
                w0acc[4] = fmaf(z[r], x4[r], w0acc[4]);
            }
        } else
#pragma unroll
        for (int k4 = 0; k4 < thin_f4; ++k4) {
            const float4 w = lds_f4(w0_s + f * wstride + 4 * k4);
            float4 xv[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) xv[r] = lds_f4(X + (rb + r) * 32 + 4 * k4);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                o[r] = fmaf(w.x, xv[r].x, o[r]); o[r] = fmaf(w.y, xv[r].y, o[r]); o[r] = fmaf(w.z, xv[r].z, o[r]); o[r] = fmaf(w.w, xv[r].w, o[r]);
            }
            if constexpr (kVecRider && TG_F32DW_ABLATE != 6) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {                     // rows ascending
                    w0acc[4 * k4 + 0] = fmaf(z[r], xv[r].x, w0acc[4 * k4 + 0]); w0acc[4 * k4 + 1] = fmaf(z[r], xv[r].y, w0acc[4 * k4 + 1]);
                    w0acc[4 * k4 + 2] = fmaf(z[r], xv[r].z, w0acc[4 * k4 + 2]); w0acc[4 * k4 + 3] = fmaf(z[r], xv[r].w, w0acc[4 * k4 + 3]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) lds_st(Qw + (rb + r) * H + f, fmaxf(o[r], 0.f));
#pragma unroll
        for (int r = 0; r < RPT; ++r) b0sum += z[r];
    };
    auto rebuild_p = [&](const char* sb, float* __restrict__ Pw) {
        // dZ_top[row][f] = (sum_a g[row][a] W_head[a][f]) * bit(row, f)
        const float* Gm = reinterpret_cast<const float*>(sb + F::OFF_G);
        const uint32_t* Mm = reinterpret_cast<const uint32_t*>(sb + F::OFF_G + 512);
        const float* AT = reinterpret_cast<const float*>(sb + F::OFF_AT);
        float at[RPT];
        uint32_t mw[RPT];
        if constexpr (kLean) {
            float g1[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                at[r] = lds_f(AT + (rb + r) * H + f);
                g1[r] = lds_f(Gm + 4 * (rb + r));
                mw[r] = lds_u(Mm + 4 * (rb + r) + m_word);
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float v = whv[0] * g1[r];
                const float pv = ((mw[r] >> m_shift) & 1u) != 0u ? v : 0.f;
                lds_st(Pw + (rb + r) * H + f, pv);
                bsum += pv;
                hacc[0] = fmaf(g1[r], at[r], hacc[0]);
                gsum += g1[r];                                      // (threads f = 1..3 of a row group carry a copy nobody reads)
            }
            return;
        }
        float4 g4[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            at[r] = lds_f(AT + (rb + r) * H + f);
            g4[r] = lds_f4(Gm + 4 * (rb + r));
            mw[r] = lds_u(Mm + 4 * (rb + r) + m_word);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float v = whv[0] * g4[r].x;
            v = fmaf(whv[1], g4[r].y, v); v = fmaf(whv[2], g4[r].z, v); v = fmaf(whv[3], g4[r].w, v);
            const float pv = ((mw[r] >> m_shift) & 1u) != 0u ? v : 0.f;
            lds_st(Pw + (rb + r) * H + f, pv);
            bsum += pv;
#if TG_F32DW_ABLATE != 6                                    /* probe build 6: 56 of the rebuild's ~165 vector instructions gone */
            hacc[0] = fmaf(g4[r].x, at[r], hacc[0]); hacc[1] = fmaf(g4[r].y, at[r], hacc[1]);
            hacc[2] = fmaf(g4[r].z, at[r], hacc[2]); hacc[3] = fmaf(g4[r].w, at[r], hacc[3]);
            const float g_lo = gc1 ? g4[r].y : g4[r].x, g_hi = gc1 ? g4[r].w : g4[r].z;
            gsum += gc2 ? g_hi : g_lo;
#endif
        }
    };
    // ---- the first layer's rider of a stage: this wave's tile, its k-steps KS s + ks0, operands straight from the ring slot ----
    auto rider_half = [&](const char* sb, int h) {
        if constexpr (kVecRider) return;
        const float* Za = reinterpret_cast<const float*>(sb + F::OFF_Z0) + (2 * ks0 + kk) * H + 32 * tile0 + i;
        const float* Xa = reinterpret_cast<const float*>(sb + F::OFF_X) + (2 * ks0 + kk) * 32 + i;
        float za[2], xb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            za[s] = lds_f(Za + 2 * KS * (2 * h + s) * H);
            xb[s] = lds_f(Xa + 2 * KS * (2 * h + s) * 32);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(za[s], xb[s], acc0, 0, 0, 0);
    };
    int64_t sg_issue = my;
    if constexpr (kPipe) {
        const int n_my = my < n_st ? (int)((n_st - 1 - my) / nb) + 1 : 0;
        float* pan = Pp;                                                // set u: P at pan + 2 u SR H, Q behind it
        issue(sg_issue, 0);                                             // the first stage is on its way while the weights are fetched
        sg_issue += nb;
        load_weights();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                // the table is in place and the first stage has landed
        asm volatile("" ::: "memory");
        issue(sg_issue, 1);
        sg_issue += nb;
        rebuild_q(lds_c, pan + SR * H);
        rebuild_p(lds_c, pan);
        rider_half(lds_c, 0);
        rider_half(lds_c, 1);
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
            const int cur = k & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage k + 1 has landed; stage k's panels are complete; slot and panel set of stage k - 1 are free
            asm volatile("" ::: "memory");
#if TG_F32DW_ABLATE != 5
            issue(sg_issue, cur);                                       // stage k + 2 into the slot stage k was rebuilt from
#endif
            sg_issue += nb;
            const char* sn = lds_c + (cur ^ 1) * F::SLOT;               // stage k + 1's slot
            float* Pn = pan + (cur ^ 1) * 2 * SR * H;
            const float* Pa = pan + cur * 2 * SR * H + kk * H + 32 * wm + i;
            const float* Qa = Pa + SR * H - 32 * wm + 64 * wn;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float av[4], bv[4][2];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    av[s] = lds_f(Pa + 2 * (4 * h + s) * H);
                    bv[s][0] = lds_f(Qa + 2 * (4 * h + s) * H);
                    bv[s][1] = lds_f(Qa + 2 * (4 * h + s) * H + 32);
                }
#if TG_F32DW_ABLATE != 3
                if (h == 0) rebuild_q(sn, Pn + SR * H);
                else rebuild_p(sn, Pn);
#endif
#if TG_F32DW_ABLATE == 4
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(av[s]), "v"(bv[s][0]), "v"(bv[s][1]));
#else
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][1], acc[1], 0, 0, 0);
                }
                rider_half(sn, h);
#endif
            }
        }
    } else {
    int slot_issue = 0, slot = 0;
    __syncthreads();                                                    // the table is in place (ordinary stores: before any DMA)
#pragma unroll 1
    for (int t = 0; t < D - 1; ++t) {
        issue(sg_issue, slot_issue);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
    }
#if TG_F32DW_STAMPS
    fs_loop0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int64_t sg = my; sg < n_st; sg += nb) {
#if TG_F32DW_STAMPS
        unsigned long long st_t = __builtin_amdgcn_s_memtime();
        st[6] += 1;
#endif
        if (D == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // the stage has landed for every wave; every wave is done with the previous stage's slot and panels
        asm volatile("" ::: "memory");
        TG_FSTAMP(0)
        issue(sg_issue, slot_issue);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        char* sb = lds_c + slot * F::SLOT;
        slot = slot + 1 == D ? 0 : slot + 1;
        const float* X = reinterpret_cast<const float*>(sb + F::OFF_X);
        const float* Z0 = reinterpret_cast<const float*>(sb + F::OFF_Z0);
        TG_FSTAMP(1)
        rebuild_q(sb, Qp);
        rebuild_p(sb, Pp);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this thread's panel writes have landed ...
        TG_FSTAMP(2)
        __builtin_amdgcn_s_barrier();                               // ... and everyone's
        asm volatile("" ::: "memory");
        TG_FSTAMP(3)
        // ---- phase 2: products, in two halves of 4 k-steps (a half's operands in registers first: 16 registers, and the
        // second half's reads go out while the first half's products run) ----
        const float* Pa = Pp + kk * H + 32 * wm + i;
        const float* Qa = Qp + kk * H + 64 * wn + i;
        const float* Za = Z0 + (2 * ks0 + kk) * H + 32 * tile0 + i;
        const float* Xa = X + (2 * ks0 + kk) * 32 + i;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[4], bv[4][2], za[2], xb[2];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                av[s] = lds_f(Pa + 2 * (4 * h + s) * H);
                bv[s][0] = lds_f(Qa + 2 * (4 * h + s) * H);
                bv[s][1] = lds_f(Qa + 2 * (4 * h + s) * H + 32);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {                           // the rider's k-steps of this half: KS (2 h + s) + ks0
                za[s] = lds_f(Za + 2 * KS * (2 * h + s) * H);
                xb[s] = lds_f(Xa + 2 * KS * (2 * h + s) * 32);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(av[s]), "+v"(bv[s][0]), "+v"(bv[s][1]));     // (the products stay behind the wait)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][1], acc[1], 0, 0, 0);
                if (s % KS == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(za[s / KS], xb[s / KS], acc0, 0, 0, 0);
            }
        }
        TG_FSTAMP(5)
    }
#if TG_F32DW_STAMPS
    fs_loop1 = __builtin_amdgcn_s_memtime();
#endif
    }
#undef TG_FSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // no LDS-DMA may outlive the workgroup's LDS allocation
    __syncthreads();
    // ---- the row groups' partial sums and the k-step phases' partial tiles meet in LDS, added in a fixed order ----
    float* red = reinterpret_cast<float*>(lds_c);                       // [7][512] floats, then [MT][16][64]
    red[tid] = bsum;
    red[512 + tid] = b0sum;
#pragma unroll
    for (int a = 0; a < 4; ++a) red[(2 + a) * 512 + tid] = hacc[a];
    red[6 * 512 + tid] = gsum;
    float* tiles = red + 7 * 512;                                       // (kVecRider: the row groups' dW_0 rows, [RG][H][IN_PAD])
    if constexpr (kVecRider) {
#pragma unroll
        for (int k = 0; k < IN_PAD; ++k) tiles[tid * IN_PAD + k] = w0acc[k];
    } else if (ks0 > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tiles[(tile0 * 16 + r) * 64 + lane] = acc0[r];
    }
    __syncthreads();
    float* slab = ws + job.slab_off + (int64_t)my * job.slab_len;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int m0 = 32 * wm, n0 = 64 * wn + 32 * y;
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * H + n0 + col] = acc[y][r];
    }
    constexpr int W0C = kVecRider ? IN_PAD : 32;                        // columns of the first-layer rider's part of the slab (the host lays
    float* s1 = slab + H * H + H;                                       // the reduction out to match): [H][W0C] + [H]
    float* s2 = s1 + H * W0C + H;                                       // head rider: [4][H] + [4]
    if (tid < H) {
        float t[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            t[q] = red[q * 512 + tid];
#pragma unroll
            for (int g = 1; g < RG; ++g) t[q] += red[q * 512 + g * H + tid];
        }
        slab[H * H + tid] = t[0];
        s1[H * W0C + tid] = t[1];
        if constexpr (kVecRider) {
#pragma unroll
            for (int k = 0; k < IN_PAD; ++k) {                          // (the reduction reads in_dim <= IN_PAD columns of each 32-float row)
                float v = tiles[tid * IN_PAD + k];
#pragma unroll
                for (int g = 1; g < RG; ++g) v += tiles[(g * H + tid) * IN_PAD + k];
                s1[tid * W0C + k] = v;
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) s2[a * H + tid] = t[2 + a];
        if (tid < 4) {
            float gs = 0.f;                                             // (thread f < 4 of a row group holds component f), groups in order
#pragma unroll
            for (int g = 0; g < RG; ++g) gs += red[6 * 512 + g * H + tid];
            s2[4 * H + tid] = gs;
        }
    }
    if (!kVecRider && ks0 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc0[r] + tiles[(tile0 * 16 + r) * 64 + lane];
            s1[(32 * tile0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = v;
        }
    }
    TG_CLOCK_PROBE_END(g_probe_f32_dw)
#if TG_F32DW_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = g_f32_stamps3 + ((size_t)blockIdx.x * 8 + wave) * 12;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = st[k];
        o[7] = fs_entry; o[8] = fs_loop0; o[9] = fs_loop1; o[10] = __builtin_amdgcn_s_memtime(); o[11] = __builtin_amdgcn_s_memrealtime() - fs_rt0;
    }
#endif
}

// grad[m][n] += sum over the job's slabs, in a fixed order.
struct F32FinishDesc {
    const float* slab; float* grad; int64_t grad_ld;
    int32_t slab_len, n_slabs, N, m_out, n_out, first_elem;     // first_elem: in UNITS (below) once the launcher has laid them out
    int32_t vec;                                                // a unit = 4 consecutive floats of a row (else 1 float)
    // the optimizer step riding on the reduction (tg_mlp_f32_weight_grad_adam): the window IS a whole parameter tensor, p / m / v its
    // master, exp_avg and exp_avg_sq, adam_first the tensor's first element in the optimizer's index space; p == nullptr: none
    float* p; float* m; float* v; int64_t adam_first;
};
struct F32FinishArgs {
    F32FinishDesc d[2 * kF32DwMaxJobs]; int32_t n; int32_t total;
    // optional rider of the launch: sums[k] += sum over rows [0, n_loss_rows) of loss_work[row][k], in a fixed order (the chain
    // kernel's per-workgroup loss sums: this launch follows it in stream order, so no device-scope hand-off is needed)
    const double* loss_work; double* loss_sums; int32_t n_loss_rows; int32_t n_blocks;
    AdamScalars adam; int32_t adam_zero_grads;
    const GatherSegment* seg; const int32_t* inv_start; const int32_t* inv_dst;      // (seg == nullptr: no derived layouts pushed)
};

// A workgroup = 32 consecutive output units x 8 slab chunks: thread (el, c) adds slabs c, c + 8, ... of its unit (8 loads in
// flight), the 8 chunk sums meet in LDS and are added in chunk order: a fixed order whatever the launch, and ~500 slabs of 64-85 KB
// are read at memory speed (one thread per element walking all of them was a 100-us chain of dependent loads).  A unit is a float4
// where the window allows it (the H x H gradients: 512 contiguous bytes per workgroup and slab instead of 128).
__global__ __launch_bounds__(256) void mlp_f32_dw_finish_kernel(F32FinishArgs fa) {
    __shared__ float4 part[8][32];
    if ((int)blockIdx.x == fa.n_blocks) {                   // the extra workgroup: the loss sums (one wave)
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x, k = lane & 3, p16 = lane >> 2;     // 16 lanes per quantity: rows p16, p16 + 16, ...
            double t = 0.0;
            for (int r = p16; r < fa.n_loss_rows; r += 16) t += fa.loss_work[(int64_t)r * 4 + k];
#pragma unroll
            for (int off = 32; off >= 4; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane < 4) fa.loss_sums[lane] += t;
        }
        return;
    }
    const int el = threadIdx.x & 31, c = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float4 sum = float4{0.f, 0.f, 0.f, 0.f};
    float* dst = nullptr;
    float *ap = nullptr, *am = nullptr, *av = nullptr;
    int64_t ae = 0;
    bool vec = false;
    if (e < fa.total) {
        int k = 0;
#pragma unroll
        for (int t = 1; t < 2 * kF32DwMaxJobs; ++t)
            if (t < fa.n && e >= fa.d[t].first_elem) k = t;
        const F32FinishDesc d = fa.d[k];
        vec = d.vec != 0;
        const int le = (e - d.first_elem) * (vec ? 4 : 1);
        const int m = le / d.n_out, n = le - m * d.n_out;
        const float* src = d.slab + (int64_t)m * d.N + n;
        dst = d.grad + (int64_t)m * d.grad_ld + n;
        if (d.p) { ap = d.p + le; am = d.m + le; av = d.v + le; ae = d.adam_first + le; }
        int b = c;
        if (vec) {
            float4 s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = float4{0.f, 0.f, 0.f, 0.f};
            for (; b + 56 < d.n_slabs; b += 64) {           // slabs b, b + 8, ..., b + 56
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(b + 8 * u) * d.slab_len);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
            }
            for (; b < d.n_slabs; b += 8) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)b * d.slab_len);
                s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
            }
            sum.x = ((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x));
            sum.y = ((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y));
            sum.z = ((s[0].z + s[1].z) + (s[2].z + s[3].z)) + ((s[4].z + s[5].z) + (s[6].z + s[7].z));
            sum.w = ((s[0].w + s[1].w) + (s[2].w + s[3].w)) + ((s[4].w + s[5].w) + (s[6].w + s[7].w));
        } else {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (; b + 56 < d.n_slabs; b += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 8 * u) * d.slab_len];
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += v[u];
            }
            for (; b < d.n_slabs; b += 8) s[0] += src[(int64_t)b * d.slab_len];
            sum.x = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
    }
    part[c][el] = sum;
    __syncthreads();
    if (c == 0 && dst) {
        float4 t = part[0][el];
#pragma unroll
        for (int u = 1; u < 8; ++u) { t.x += part[u][el].x; t.y += part[u][el].y; t.z += part[u][el].z; t.w += part[u][el].w; }
        // With the optimizer step riding (ap): the element's gradient is complete here, so the thread that holds it applies Adam
        // to its parameter (adam_update: the operation sequence of tg_adam_step), writes the new value into the derived weight
        // layouts (adam_push) and leaves the gradient as the optimizer launch would have (zeroed, or the accumulated value).
        if (vec) {
            float4* d4 = reinterpret_cast<float4*>(dst);
            float4 g = *d4;
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
            if (ap) {
                float4 P = *reinterpret_cast<const float4*>(ap), Mv = *reinterpret_cast<const float4*>(am), Vv = *reinterpret_cast<const float4*>(av);
                P.x = adam_update(g.x, Mv.x, Vv.x, P.x, fa.adam);
                P.y = adam_update(g.y, Mv.y, Vv.y, P.y, fa.adam);
                P.z = adam_update(g.z, Mv.z, Vv.z, P.z, fa.adam);
                P.w = adam_update(g.w, Mv.w, Vv.w, P.w, fa.adam);
                *reinterpret_cast<float4*>(ap) = P; *reinterpret_cast<float4*>(am) = Mv; *reinterpret_cast<float4*>(av) = Vv;
                if (fa.seg) {
                    adam_push(ae, P.x, fa.seg, fa.inv_start, fa.inv_dst);
                    adam_push(ae + 1, P.y, fa.seg, fa.inv_start, fa.inv_dst);
                    adam_push(ae + 2, P.z, fa.seg, fa.inv_start, fa.inv_dst);
                    adam_push(ae + 3, P.w, fa.seg, fa.inv_start, fa.inv_dst);
                }
                if (fa.adam_zero_grads) g = float4{0.f, 0.f, 0.f, 0.f};
            }
            *d4 = g;
        } else {
            float g = *dst + t.x;
            if (ap) {
                float Mv = *am, Vv = *av;
                const float P = adam_update(g, Mv, Vv, *ap, fa.adam);
                *ap = P; *am = Mv; *av = Vv;
                if (fa.seg) adam_push(ae, P, fa.seg, fa.inv_start, fa.inv_dst);
                if (fa.adam_zero_grads) g = 0.f;
            }
            *dst = g;
        }
    }
}

static int f32_dw_max_blocks() { return 2 * device_cus(); }
static int f32_dw_slab_len(int H, int kind, int n) { return kind == F32DW_HEAD ? 4 * H + 4 : (n <= 32 ? H * 32 + H : H * H + H); }

}  // namespace tg

using namespace tg;

static int fill_f32_net(F32Net& net, const float* d_stream, int hidden, int n_hidden_layers, int in_pad, const char* what) {
    if (!(hidden == 64 || hidden == 128)) return set_error(TG_ERR_ARG, "%s: hidden width %d unsupported (64, 128)", what, hidden);
    if (!(n_hidden_layers >= 1 && n_hidden_layers <= kF32MaxHidden))
        return set_error(TG_ERR_ARG, "%s: %d hidden layers outside 1..%d", what, n_hidden_layers, kF32MaxHidden);
    if (!(in_pad >= 8 && in_pad <= 32 && in_pad % 8 == 0)) return set_error(TG_ERR_ARG, "%s: padded input width %d must be 8, 16, 24 or 32", what, in_pad);
    const int H = hidden, MT = H / 32, n_hh = n_hidden_layers - 1;
    const float* p = d_stream;
    net.w0 = reinterpret_cast<const uint4*>(p); p += (size_t)H * in_pad;
    net.bias = p; p += (size_t)n_hidden_layers * H;
    net.wh = p; p += (size_t)4 * H;
    net.bh = p; p += 4;
    net.blocks = reinterpret_cast<const uint4*>(p);
    net.n_hh = n_hh;
    net.k2 = in_pad / 2;
    (void)MT;
    return TG_OK;
}

extern "C" {

int64_t tg_mlp_f32_stream_floats(int32_t hidden, int32_t n_hidden_layers, int32_t in_pad) {
    if (!(hidden == 64 || hidden == 128) || n_hidden_layers < 1 || n_hidden_layers > kF32MaxHidden || in_pad < 8 || in_pad > 32 || in_pad % 8) return 0;
    const int64_t H = hidden, n_hh = n_hidden_layers - 1;
    return H * in_pad + n_hidden_layers * H + 4 * H + 4 + 2 * n_hh * H * H;
}

int tg_mlp_f32_blocks(void) { return device_cus(); }

int tg_mlp_f32_forward(const float* d_x, int32_t in_pad, const float* d_stream, int32_t hidden, int32_t n_hidden_layers, int64_t rows,
                       float* d_out, void* stream) {
    TG_REQUIRE(d_x && d_stream && d_out, "tg_mlp_f32_forward: null pointer");
    TG_REQUIRE(rows >= 0, "tg_mlp_f32_forward: negative row count");
    F32ChainArgs a{};
    if (int rc = fill_f32_net(a.net, d_stream, hidden, n_hidden_layers, in_pad, "tg_mlp_f32_forward")) return rc;
    if (rows == 0) return TG_OK;
    a.x = d_x; a.rows = rows; a.out = d_out;
    hipStream_t st = (hipStream_t)stream;
    return hidden == 128 ? launch_f32_chain<128, false>(a, st) : launch_f32_chain<64, false>(a, st);
}

int tg_mlp_f32_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, int32_t hidden, int32_t n_hidden_layers,
                                int64_t rows, void* const* d_acts, void* const* d_dz, void* d_top_maskbits, const tg_chain_loss* loss,
                                void* stream) {
    TG_REQUIRE(d_x && d_stream && d_acts && d_dz && loss, "tg_mlp_f32_forward_backward: null pointer");
    TG_REQUIRE(rows > 0, "tg_mlp_f32_forward_backward: no rows");
    TG_REQUIRE(loss->kind == 0 || loss->kind == 1, "tg_mlp_f32_forward_backward: kind %d", loss->kind);
    TG_REQUIRE(loss->act_dim >= 1 && loss->act_dim <= 4, "tg_mlp_f32_forward_backward: %d outputs unsupported (1..4)", loss->act_dim);
    TG_REQUIRE(loss->d_dout8 && loss->d_work, "tg_mlp_f32_forward_backward: null output");
    TG_REQUIRE(loss->kind == 1 ? loss->d_ret != nullptr : (loss->d_act && (loss->d_logp_old || loss->d_logp_old_out) && loss->d_adv),
               "tg_mlp_f32_forward_backward: missing per-row input");
    TG_REQUIRE(loss->kind == 1 || (loss->act_col_stride == 1 && loss->act_row_stride == loss->act_dim),
               "tg_mlp_f32_forward_backward: the actions must be contiguous [rows][act_dim]");
    F32ChainArgs a{};
    if (int rc = fill_f32_net(a.net, d_stream, hidden, n_hidden_layers, in_pad, "tg_mlp_f32_forward_backward")) return rc;
    for (int l = 0; l < n_hidden_layers; ++l) {
        // with >= 2 hidden layers the first activation and the top layer's dZ may be left out: the weight-gradient job of the layer
        // that would read them rebuilds them (tg_f32_dw_job.recompute) -- the latter needs the top layer's mask bits
        const bool a_opt = l == 0 && n_hidden_layers >= 2, z_opt = l == n_hidden_layers - 1 && n_hidden_layers >= 2 && d_top_maskbits;
        TG_REQUIRE((d_acts[l] || a_opt) && (d_dz[l] || z_opt), "tg_mlp_f32_forward_backward: buffer %d is null", l);
        a.acts[l] = (float*)d_acts[l];
        a.dz[l] = (float*)d_dz[l];
    }
    a.top_mask = (uint32_t*)d_top_maskbits;
    a.x = d_x; a.rows = rows;
    fill_f32_loss(a.loss, loss);
    hipStream_t st = (hipStream_t)stream;
    return hidden == 128 ? launch_f32_chain<128, true>(a, st) : launch_f32_chain<64, true>(a, st);
}

#if TG_F32DW_STAMPS
int tg_debug_f32_stamps(unsigned long long* host_out) {    /* diagnostic builds only: not part of the ABI */
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_stamps), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
int tg_debug_f32_stamps3(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_stamps3), sizeof(unsigned long long) * 4096 * 12) == hipSuccess ? 0 : -1;
}
int tg_debug_f32_stamps2(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f32_stamps2), sizeof(unsigned long long) * 4096 * 6) == hipSuccess ? 0 : -1;
}
#endif

int64_t tg_mlp_f32_weight_grad_workspace(int32_t hidden) {
    if (hidden != 64 && hidden != 128 && hidden != 256) return 0;
    // (a fused job's slab: the layer's gradient + both riders'; H = 256 runs one workgroup per CU: half the slots)
    return (int64_t)(hidden == 256 ? device_cus() : f32_dw_max_blocks()) * (hidden * hidden + hidden + hidden * 32 + hidden + 4 * hidden + 4) * (int64_t)sizeof(float);
}

int tg_mlp_f32_weight_grad(int32_t hidden, const tg_f32_dw_job* jobs, int32_t n_jobs, int64_t rows, void* d_workspace,
                           int64_t workspace_bytes, const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums, void* stream) {
    return tg_mlp_f32_weight_grad_adam(hidden, jobs, n_jobs, rows, d_workspace, workspace_bytes, d_loss_work, n_loss_rows, d_loss_sums,
                                       nullptr, stream);
}

int tg_mlp_f32_weight_grad_adam(int32_t hidden, const tg_f32_dw_job* jobs, int32_t n_jobs, int64_t rows, void* d_workspace,
                                int64_t workspace_bytes, const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums,
                                const tg_adam_rider* adam, void* stream) {
    TG_REQUIRE(jobs && d_workspace, "tg_mlp_f32_weight_grad: null pointer");
    if (adam) {
        TG_REQUIRE(adam->h_table && adam->n_tensors >= 1 && adam->n_tensors <= kAdamMaxTensors && adam->total >= 0 && adam->step >= 1,
                   "tg_mlp_f32_weight_grad_adam: bad optimizer table (%d tensors, %lld elements, step %lld)", adam->n_tensors,
                   (long long)adam->total, (long long)adam->step);
        TG_REQUIRE(1.0 - adam->beta1 < 0.5, "tg_mlp_f32_weight_grad_adam: beta1 = %g: lerp's other branch (weight >= 0.5) is not implemented", adam->beta1);
        TG_REQUIRE((adam->d_segments == nullptr) == (adam->d_inv_start == nullptr) && (adam->d_segments == nullptr) == (adam->d_inv_dst == nullptr) &&
                   (adam->d_segments == nullptr || (adam->n_segments >= 1 && adam->n_segments <= kGatherMaxSegments)),
                   "tg_mlp_f32_weight_grad_adam: the push tables come as all three or none, with 1..%d segments", kGatherMaxSegments);
        // (a launch of zero rows forms no gradient and must not step the optimizer either: the caller's optimizer.step() would
        // still have run -- refuse, so that the caller takes the separate launch)
        TG_REQUIRE(rows > 0, "tg_mlp_f32_weight_grad_adam: no rows: run the optimizer step as its own launch");
    }
    TG_REQUIRE(hidden == 64 || hidden == 128 || hidden == 256, "tg_mlp_f32_weight_grad: hidden width %d unsupported (64, 128, 256)", hidden);
    TG_REQUIRE(n_jobs >= 1 && n_jobs <= kF32DwMaxJobs, "tg_mlp_f32_weight_grad: %d jobs outside 1..%d", n_jobs, kF32DwMaxJobs);
    TG_REQUIRE(rows >= 0, "tg_mlp_f32_weight_grad: negative row count");
    TG_REQUIRE(workspace_bytes >= tg_mlp_f32_weight_grad_workspace(hidden), "tg_mlp_f32_weight_grad: workspace of %lld B is smaller than %lld B",
               (long long)workspace_bytes, (long long)tg_mlp_f32_weight_grad_workspace(hidden));
    if (rows == 0) return TG_OK;
    const int H = hidden;
    // Workgroup slots: two per CU.  The wide jobs (one H x H layer each) are bound by the fp32 matrix pipe, the light ones (first
    // layer: 16 products per 64 rows; head: vector arithmetic) by the bytes they stream: the wide jobs share HALF the slots -- one
    // wide workgroup per CU keeps every SIMD's matrix pipe busy -- and the light jobs share the rest in proportion to their bytes
    // per row, so that a CU streams a light job beside a wide job's products.
    // (time per row in units of one plain wide job; 0 = a light job, which shares what the wide jobs leave in proportion to its bytes)
    int64_t bytes[kF32DwMaxJobs], light_sum = 0;
    double weight[kF32DwMaxJobs], wide_sum = 0.0;
    int n_wide = 0;
    size_t shmem = H == 128 ? (size_t)F32DwGeom<128>::LDS_PLAIN : (size_t)F32DwGeom<64>::LDS_PLAIN;
    int ring_slots[kF32DwMaxJobs], fused_slab[kF32DwMaxJobs];
    // a net whose only H x H layer carries everything (two hidden layers, H = 128): the 8-wave form of that job
    // (padded input width 8: one barrier per stage; wider inputs: two)
    const bool use8 = hidden == 128 && n_jobs == 1 && jobs[0].kind == F32DW_MM && jobs[0].recompute == 3 &&
                      jobs[0].in_pad % 8 == 0 && jobs[0].in_pad >= 8 && jobs[0].in_pad <= 32;
    const bool use8_pipe = use8 && jobs[0].in_pad == 8;         // (its first-layer rider runs on the vector pipe: 8 slab columns, not 32)
    const int w0cols = use8_pipe ? 8 : 32;
    for (int j = 0; j < n_jobs; ++j) {
        const tg_f32_dw_job& jb = jobs[j];
        TG_REQUIRE(jb.kind == F32DW_MM || jb.kind == F32DW_HEAD, "tg_mlp_f32_weight_grad: job %d has kind %d", j, jb.kind);
        TG_REQUIRE(jb.d_p && jb.d_q && jb.d_wgrad, "tg_mlp_f32_weight_grad: job %d has a null pointer", j);
        weight[j] = 0.0;
        ring_slots[j] = fused_slab[j] = 0;
        if (jb.kind == F32DW_HEAD) {
            TG_REQUIRE(jb.n_cols == H && jb.m_out >= 1 && jb.m_out <= 4 && jb.n_out == H && jb.wgrad_ld >= H, "tg_mlp_f32_weight_grad: job %d: bad head window", j);
            bytes[j] = 4 * H + 16;
        } else {
            TG_REQUIRE(jb.n_cols == H || (jb.n_cols >= 8 && jb.n_cols <= 32 && jb.n_cols % 8 == 0), "tg_mlp_f32_weight_grad: job %d: %d columns", j, jb.n_cols);
            TG_REQUIRE(jb.m_out == H && jb.n_out >= 1 && jb.n_out <= jb.n_cols && jb.wgrad_ld >= jb.n_out, "tg_mlp_f32_weight_grad: job %d: bad window", j);
            TG_REQUIRE(jb.recompute >= 0 && jb.recompute <= 3 && (jb.recompute == 0 || (jb.n_cols == H && H != 256)), "tg_mlp_f32_weight_grad: job %d: recompute %d", j, jb.recompute);
            TG_REQUIRE(!(jb.recompute & 1) || (jb.d_w0 && jb.d_b0 && jb.in_pad >= 8 && jb.in_pad <= 32 && jb.in_pad % 8 == 0 && jb.in_dim >= 1 && jb.in_dim <= jb.in_pad &&
                                               jb.d_dz0 && jb.d_w0grad && jb.d_b0grad && jb.w0grad_ld >= jb.in_dim),
                       "tg_mlp_f32_weight_grad: job %d rebuilds the first activation: first-layer weights / input width / rider missing", j);
            TG_REQUIRE(!(jb.recompute & 2) || (jb.d_wh && jb.d_maskbits && jb.act_dim >= 1 && jb.act_dim <= 4 && jb.d_a_top && jb.d_whgrad && jb.d_bhgrad && jb.whgrad_ld >= H),
                       "tg_mlp_f32_weight_grad: job %d rebuilds the top dZ: head weights / mask bits / rider missing", j);
            bytes[j] = jb.n_cols == H ? 0 : 4 * H + 4 * jb.n_cols;
            if (jb.n_cols == H) {
                ++n_wide;
                // products per stage and wave: 32 (H = 128), + 8 for the first layer's tile; the head rider is vector work
                weight[j] = 1.0 + ((jb.recompute & 1) ? 0.25 : 0.0) + ((jb.recompute & 2) ? 0.05 : 0.0);
                wide_sum += weight[j];
            }
            if (jb.recompute) {
                auto fit = [&](auto geom) {
                    using F = decltype(geom);
                    ring_slots[j] = F::lds_bytes(3, jb.in_pad) <= F32DwGeom<128>::LDS_MAX ? 3 : 2;
                    fused_slab[j] = F::SLAB - ((jb.recompute & 1) ? H * (32 - w0cols) : 0);
                    return (size_t)F::lds_bytes(ring_slots[j], jb.in_pad);
                };
                size_t need = 0;
                if (H == 128) need = jb.recompute == 3 ? fit(F32FusedGeom<128, true, true>{}) : (jb.recompute == 2 ? fit(F32FusedGeom<128, true, false>{}) : fit(F32FusedGeom<128, false, true>{}));
                else need = jb.recompute == 3 ? fit(F32FusedGeom<64, true, true>{}) : (jb.recompute == 2 ? fit(F32FusedGeom<64, true, false>{}) : fit(F32FusedGeom<64, false, true>{}));
                TG_REQUIRE(need <= (size_t)F32DwGeom<128>::LDS_MAX, "tg_mlp_f32_weight_grad: job %d needs %zu B of LDS", j, need);
                shmem = need > shmem ? need : shmem;
            }
        }
        light_sum += bytes[j];
    }
    const int max_blocks = H == 256 ? device_cus() : f32_dw_max_blocks();      // (H = 256: one 8-wave workgroup per CU)
    // share of the slots that goes to the wide jobs: in proportion to estimated time per row -- a wide job's products at ~70 % of
    // the fp32 matrix rate of one workgroup per CU against a light job's bytes at the ~12 GB/s one workgroup streams
    // (H = 256, one workgroup per CU, measured with tools/f32_dw_jobs_probe.py: 0.26 us per row and wide workgroup; a light workgroup
    // streams ~17 GB/s)
    const double t_wide = wide_sum * (H == 256 ? 0.26 : (H == 128 ? 0.085 : 0.085 / 4)), t_light = (double)light_sum / (H == 256 ? 17000.0 : 12000.0);
    int wide_slots = 0;
    if (n_wide) {
        const double share = t_wide / (t_wide + t_light);
        wide_slots = n_wide < n_jobs ? (int)(share * max_blocks + 0.5) : max_blocks;
        if (wide_slots < n_wide) wide_slots = n_wide;
        if (wide_slots > max_blocks - (n_jobs - n_wide)) wide_slots = max_blocks - (n_jobs - n_wide);
    }
    const int light_slots = max_blocks - wide_slots;
    int alloc[kF32DwMaxJobs], used = 0;
    for (int j = 0; j < n_jobs; ++j) {
        alloc[j] = bytes[j] == 0 ? (int)(wide_slots * weight[j] / wide_sum) : (int)(bytes[j] * light_slots / light_sum);
        if (alloc[j] < 1) alloc[j] = 1;
        used += alloc[j];
    }
    while (used > max_blocks) {                              // (the minimum of one per job may overshoot)
        int big = 0;
        for (int j = 1; j < n_jobs; ++j)
            if (alloc[j] > alloc[big]) big = j;
        --alloc[big];
        --used;
    }
    {
        int n_desc = 0;
        for (int j = 0; j < n_jobs; ++j) n_desc += 2 + ((jobs[j].recompute & 1) ? 2 : 0) + ((jobs[j].recompute & 2) ? 2 : 0);
        TG_REQUIRE(n_desc <= 2 * kF32DwMaxJobs, "tg_mlp_f32_weight_grad: %d gradient windows exceed %d", n_desc, 2 * kF32DwMaxJobs);
    }
    F32DwArgs args{};
    args.n_jobs = n_jobs;
    F32FinishArgs fa{};
    int grid = 0, elems = 0;
    int64_t off = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const tg_f32_dw_job& jb = jobs[j];
        F32DwJob& dj = args.job[j];
        dj.p = jb.d_p; dj.q = jb.d_q; dj.kind = jb.kind; dj.n = jb.n_cols;
        dj.recompute = jb.recompute; dj.in_pad = jb.in_pad; dj.in_dim = jb.in_dim; dj.act_dim = jb.act_dim;
        dj.w0 = jb.d_w0; dj.b0 = jb.d_b0; dj.wh = jb.d_wh; dj.mask = jb.d_maskbits;
        dj.a_top = jb.d_a_top; dj.dz0 = jb.d_dz0; dj.ring_slots = ring_slots[j];
        dj.first_block = grid;
        // at least 4 stages per workgroup
        const int64_t n_st = ceil_div(rows, (int64_t)(bytes[j] == 0 ? (H >= 128 ? 16 : 32) : (H >= 128 ? 32 : 64)));
        const int cap = (int)(n_st < 4 ? 1 : (n_st / 4 > 1 << 20 ? 1 << 20 : n_st / 4));
        dj.n_blocks = alloc[j] < cap ? alloc[j] : cap;
        dj.slab_len = jb.recompute ? fused_slab[j] : f32_dw_slab_len(H, jb.kind, jb.n_cols);
        dj.slab_off = off;
        grid += dj.n_blocks;
        const int Nslab = jb.kind == F32DW_HEAD ? H : (jb.n_cols <= 32 ? 32 : H);
        F32FinishDesc& fd = fa.d[fa.n++];
        fd = F32FinishDesc{(const float*)d_workspace + off, jb.d_wgrad, jb.wgrad_ld, dj.slab_len, dj.n_blocks, Nslab, jb.m_out, jb.n_out, elems};
        elems += jb.m_out * jb.n_out;
        if (jb.d_bgrad) {
            const int boff = jb.kind == F32DW_HEAD ? 4 * H : (jb.n_cols <= 32 ? H * 32 : H * H);
            F32FinishDesc& fb = fa.d[fa.n++];
            fb = F32FinishDesc{(const float*)d_workspace + off + boff, jb.d_bgrad, (int64_t)H, dj.slab_len, dj.n_blocks, H, 1, jb.m_out, elems};
            elems += jb.m_out;
        }
        // the riders' parts of a fused job's slab: [H x H][H] | first layer [H x 32][H] | head [4 x H][4]
        int64_t roff = off + H * H + H;
        if (jb.recompute & 1) {
            fa.d[fa.n++] = F32FinishDesc{(const float*)d_workspace + roff, jb.d_w0grad, jb.w0grad_ld, dj.slab_len, dj.n_blocks, w0cols, H, jb.in_dim, elems};
            elems += H * jb.in_dim;
            fa.d[fa.n++] = F32FinishDesc{(const float*)d_workspace + roff + H * w0cols, jb.d_b0grad, (int64_t)H, dj.slab_len, dj.n_blocks, H, 1, H, elems};
            elems += H;
            roff += H * w0cols + H;
        }
        if (jb.recompute & 2) {
            fa.d[fa.n++] = F32FinishDesc{(const float*)d_workspace + roff, jb.d_whgrad, jb.whgrad_ld, dj.slab_len, dj.n_blocks, H, jb.act_dim, H, elems};
            elems += jb.act_dim * H;
            fa.d[fa.n++] = F32FinishDesc{(const float*)d_workspace + roff + 4 * H, jb.d_bhgrad, (int64_t)H, dj.slab_len, dj.n_blocks, H, 1, jb.act_dim, elems};
            elems += jb.act_dim;
        }
        off += (int64_t)dj.n_blocks * dj.slab_len;
    }
    // lay the windows out in units: float4 where a window's rows are contiguous and 16-B aligned in the slab and in the gradient
    int units = 0;
    for (int k = 0; k < fa.n; ++k) {
        F32FinishDesc& d = fa.d[k];
        const int cnt = d.m_out * d.n_out;
        d.vec = (d.n_out % 4 == 0 && (d.m_out == 1 || (d.N == d.n_out && d.grad_ld % 4 == 0)) && d.slab_len % 4 == 0 &&
                 ((uintptr_t)d.slab & 15) == 0 && ((uintptr_t)d.grad & 15) == 0) ? 1 : 0;
        d.first_elem = units;
        units += d.vec ? cnt / 4 : cnt;
    }
    fa.total = units;
    if (adam) {
        // every tensor of the optimizer's table must be exactly one of this launch's gradient windows, whole and contiguous
        // (a parameter this launch does not produce a gradient for would silently miss its step)
        uint64_t seen = 0;
        for (int k = 0; k < fa.n; ++k) {
            F32FinishDesc& d = fa.d[k];
            int t = -1;
            for (int q = 0; q < adam->n_tensors; ++q)
                if (adam->h_table[q].g == d.grad) t = q;
            TG_REQUIRE(t >= 0, "tg_mlp_f32_weight_grad_adam: gradient window %d is not a tensor of the optimizer's table", k);
            TG_REQUIRE(!(seen >> t & 1), "tg_mlp_f32_weight_grad_adam: tensor %d is the window of two jobs", t);
            seen |= 1ull << t;
            const int64_t numel = (t + 1 < adam->n_tensors ? adam->h_table[t + 1].first : adam->total) - adam->h_table[t].first;
            TG_REQUIRE((int64_t)d.m_out * d.n_out == numel && (d.m_out == 1 || d.grad_ld == d.n_out),
                       "tg_mlp_f32_weight_grad_adam: window %d (%d x %d, ld %lld) is not the whole contiguous tensor %d (%lld elements)", k,
                       d.m_out, d.n_out, (long long)d.grad_ld, t, (long long)numel);
            d.p = adam->h_table[t].p; d.m = adam->h_table[t].m; d.v = adam->h_table[t].v; d.adam_first = adam->h_table[t].first;
            TG_REQUIRE(d.p && d.m && d.v, "tg_mlp_f32_weight_grad_adam: tensor %d has a null pointer", t);
            if (d.vec && ((((uintptr_t)d.p | (uintptr_t)d.m | (uintptr_t)d.v) & 15) != 0)) d.vec = 0;
        }
        TG_REQUIRE(fa.n == adam->n_tensors, "tg_mlp_f32_weight_grad_adam: the launch produces %d gradient windows, the optimizer holds %d tensors",
                   fa.n, adam->n_tensors);
        // (units were laid out with the windows' own vec flags: lay them out again with the final ones)
        units = 0;
        for (int k = 0; k < fa.n; ++k) {
            F32FinishDesc& d = fa.d[k];
            d.first_elem = units;
            units += d.vec ? d.m_out * d.n_out / 4 : d.m_out * d.n_out;
        }
        fa.total = units;
        fa.adam = adam_scalars(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step);
        fa.adam_zero_grads = adam->zero_grads;
        fa.seg = reinterpret_cast<const GatherSegment*>(adam->d_segments); fa.inv_start = adam->d_inv_start; fa.inv_dst = adam->d_inv_dst;
    }
    TG_REQUIRE((d_loss_work == nullptr) == (d_loss_sums == nullptr) && n_loss_rows >= 0 && n_loss_rows <= 65536,
               "tg_mlp_f32_weight_grad: loss-sum rider needs both pointers and 0..65536 rows");
    fa.n_blocks = (int32_t)ceil_div(units, 32);
    fa.loss_work = d_loss_work; fa.loss_sums = d_loss_sums; fa.n_loss_rows = n_loss_rows;
    hipStream_t st = (hipStream_t)stream;
    if (use8) {
        auto launch8 = [&](auto kern) -> int {
            static LdsOptIn opt_in;
            if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32_weight_grad")) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), shmem, st, args.job[0], rows, (float*)d_workspace);
            return TG_OK;
        };
        // (in_pad 8: the one-barrier form with two panel sets -- 76 KiB)
        if (use8_pipe)
            shmem = 2 * (size_t)F32FusedGeom<128, true, true>::SLOT + 4 * 16 * 128 * 4 + 128 * (8 + 4) * 4;
        const bool lean = jobs[0].in_dim <= 5 && jobs[0].act_dim == 1;
        const int rc = jobs[0].in_pad == 8 ? (lean ? launch8(mlp_f32_dw_fused8_kernel<128, 8, true, true>) : launch8(mlp_f32_dw_fused8_kernel<128, 8, true>))
                     : jobs[0].in_pad == 16 ? launch8(mlp_f32_dw_fused8_kernel<128, 16, false>)
                     : jobs[0].in_pad == 24 ? launch8(mlp_f32_dw_fused8_kernel<128, 24, false>) : launch8(mlp_f32_dw_fused8_kernel<128, 32, false>);
        if (rc) return rc;
    } else if (hidden == 256) {
        if (int rc = launch_f32_wide_dw(args, rows, (float*)d_workspace, grid, st)) return rc;
    } else if (hidden == 128) {
        auto kern = mlp_f32_dw_kernel<128>;
        static LdsOptIn opt_in;
        if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32_weight_grad")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), shmem, st, args, rows, (float*)d_workspace);
    } else {
        auto kern = mlp_f32_dw_kernel<64>;
        static LdsOptIn opt_in;
        if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32_weight_grad")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), shmem, st, args, rows, (float*)d_workspace);
    }
    TG_LAUNCH_CHECK("tg_mlp_f32_weight_grad");
    hipLaunchKernelGGL(mlp_f32_dw_finish_kernel, dim3((unsigned)(fa.n_blocks + (d_loss_sums ? 1 : 0))), dim3(256), 0, st, fa);
    TG_LAUNCH_CHECK("tg_mlp_f32_weight_grad (finish)");
    return TG_OK;
}

}  // extern "C"
