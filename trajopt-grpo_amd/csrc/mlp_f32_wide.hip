// The fp32 chain learner at H = 256: the reference's own QuadPole factory at the reference's own precision
// (pipelines/quadpole_pipeline_ppo.py:54-58: GaussianActorCritic_NeuralNetwork 20-256x5-{4,1}, fp32) -- forward pass, loss head
// (algorithms/ppo.py:159-179) and backward-DATA pass of a row in ONE launch, as mlp_f32_chain.hip does for H = 64 / 128.
//
// Why a different machine than mlp_f32_chain.hip.  There a wave owns 32 rows on v_mfma_f32_32x32x2_f32 and keeps a layer's input
// AND output in registers: 2 x (H / 32) x 16 = 256 registers at H = 256 -- they do not exist at two waves per SIMD.  Here a wave
// owns 16 rows on v_mfma_f32_16x16x4_f32 (same 64 flop / clk / SIMD): a 16-feature tile is 4 accumulator registers, a layer's
// input + output 2 x 16 x 4 = 128.  Transposed form Y^T = W . X^T as everywhere in this tree: the 16 columns of a tile are 16 rows
// of the batch; lane (j = lane & 15, g = lane >> 4) of an accumulator tile t holds features 16 t + 4 g + r (r = 0..3) of row j, and
// register r of the four lane groups IS the B operand of MFMA step (t, r) of the next layer (contraction indices 16 t + 4 g + r,
// g = 0..3).  The A operand of that step, lane (i, g), is W[out i][16 t + 4 g + r]: a lane's four steps of one t are FOUR
// CONSECUTIVE floats of a weight row -- one ds_read_b128 per four products, and the weight stream is the matrix itself in 16-B
// pieces (mlp.F32WideStream).  The backward pass is the same machine on W^T with a mask multiply instead of bias + ReLU.
//
// Two 4-wave workgroups per CU instead of one 8-wave workgroup: the two waves of a SIMD then belong to DIFFERENT workgroups and
// drift out of phase, so one wave's epilogue / barrier wait hides behind the other's products (mlp_f32_chain.hip's eight waves
// meet at a barrier per block, their epilogues coincide, and its matrix pipe is 64 % busy).  Weights: one 16-feature output block
// (16 KiB = 16 pieces of 1 KiB) at a time through a ring of 3 slots, LDS-DMA, counted vmcnt + one raw s_barrier per block
// (mfma_ring.hpp's discipline).  Biases and the head stay in LDS; the ReLU mask bits of a wave's rows live in LDS between the passes.
#include <stdlib.h>
#include <type_traits>

#include "tg_common.hpp"
#include "mfma_ring.hpp"
#include "f32_loss.hpp"
#include "f32_dw.hpp"

#ifndef TG_F32W_ABLATE
#define TG_F32W_ABLATE 0           /* timing-only probe builds of the chain kernel (results meaningless): bit 0 = no block barrier, bit 1 = no
                                      weight DMA inside the rounds, bit 2 = no activation / dZ stores (tools/f32_wide_ablation.sh); of the wide
                                      weight-gradient job: bit 3 = no stage barrier, bit 4 = no DMA inside the stage loop */
#endif
#ifndef TG_F32R_TRAIN_WAVES
#define TG_F32R_TRAIN_WAVES 12     /* waves per CU of the resident kernel's training launches (probe builds: 16) */
#endif
#ifndef TG_F32W_NT_STORE
#define TG_F32W_NT_STORE 0         /* 1: the same for the H = 256 chain kernel */
#endif
#ifndef TG_F32R_NT_STORE
#define TG_F32R_NT_STORE 1         /* the resident kernel's activation / dZ tiles leave as non-temporal stores (A/B: profiles/r05_f32_res_kernel.md; 0 = plain) */
#endif
#ifndef TG_F32R_ABLATE
#define TG_F32R_ABLATE 0           /* ... of the resident H = 128 kernel: bit 0 = no activation / dZ / mask stores, bit 1 = no matrix products in
                                      the H x H tiles, bit 2 = no LDS reads of their weights, bit 3 = no head / loss arithmetic (tools/f32_res_ablation.sh) */
#endif

namespace tg {

constexpr int kWideH = 256;
constexpr int kWideMaxHidden = 5;       // hidden layers (the first + up to 4 H x H)

struct F32WideArgs {
    const float* x;             // [rows][in_pad] f32, zero padded (in_pad a multiple of 8, <= 32)
    int32_t in_pad;
    int32_t n_hh;               // H x H layers (hidden layers - 1)
    int64_t rows;
    const uint4* stream;        // blocks of 16 KiB: [first layer: 2][forward: n_hh x 16][backward: n_hh x 16, top layer first]
    const float* table;         // [kWideMaxHidden][H] hidden biases | [4][H] head weights (rows >= A zero) | [4] head bias (+ 12 pad)
    float* acts[kWideMaxHidden];    // training: post-ReLU outputs of the hidden layers f32 [rows][H]; [0] may be null (rebuilt by the
                                    // weight-gradient job), the others are written
    float* dz[kWideMaxHidden];      // training: d loss / d pre-activation f32 [rows][H]; [n_hh] (the top layer's) may be null
    float* out;                 // no-grad pass: head output f32 [rows][4]
    F32Loss loss;
};

__device__ static inline uint4 wide_lds_u4(const uint4* __restrict__ p) { return *p; }
__device__ static inline float4 wide_lds_f4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }
__device__ static inline float wide_lds_f(const float* __restrict__ p) { return *p; }
__device__ static inline uint2 wide_lds_u2(const uint32_t* __restrict__ p) { return *reinterpret_cast<const uint2*>(p); }
__device__ static inline void wide_lds_st2(uint32_t* __restrict__ p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }

constexpr int kWidePieces = 16;                         // 1-KiB pieces per block
constexpr int kWideSlot = kWidePieces * 64;             // uint4 per ring slot
constexpr int kWideD = 3, kWideP = kWideD - 1;          // ring slots, blocks in flight
constexpr int kWideWaves = 4;
constexpr int kWideTable = (kWideMaxHidden + 4) * kWideH + 16;      // floats

static size_t f32_wide_lds() {
    return (size_t)kWideD * kWideSlot * 16 + (size_t)kWideTable * 4 + (size_t)kWideMaxHidden * kWideWaves * 64 * 8 + 4 * 4 * 8;
}

TG_CLOCK_PROBE_VAR(g_probe_f32_wide, attach_probe_f32_wide)

template <bool kTrain>
__global__ __launch_bounds__(256, 2) void mlp_f32_wide_kernel(F32WideArgs a) {
    constexpr int H = kWideH, NT = H / 16, D = kWideD, P = kWideP, WPW = kWideWaves, KS = kWidePieces;
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    uint4* ring = lds;
    float* table = reinterpret_cast<float*>(ring + D * kWideSlot);
    float* wh_s = table + kWideMaxHidden * H;
    float* bh_s = wh_s + 4 * H;
    uint32_t* bits_s = reinterpret_cast<uint32_t*>(table + kWideTable);       // [layer][wave][64 lanes][2 words]
    double* red_s = reinterpret_cast<double*>(bits_s + kWideMaxHidden * WPW * 64 * 2);
    const int n_hh = a.n_hh;
    const int n_blocks = 2 + n_hh * NT * (kTrain ? 2 : 1);                    // blocks per round (the same sequence every round)
    const uint4* wfrag = a.stream;
    const int64_t rows = a.rows;
    const int64_t n_rounds = (rows + 63) / 64;
    F32Loss L = a.loss;                                                       // (a local copy: scalars in registers, not a re-read kernarg)
    if constexpr (kTrain) {
        f32_loss_from_device(L);
        TG_CLOCK_PROBE_BEGIN(g_probe_f32_wide)
    }
    // a.acts[l] / a.dz[l] with a run-time l would put the whole argument struct into scratch: uniform selects instead
    auto pick = [](float* const (&p)[kWideMaxHidden], int l) {
        float* r = p[0];
#pragma unroll
        for (int q = 1; q < kWideMaxHidden; ++q) r = l == q ? p[q] : r;
        return r;
    };
    for (int q = tid; q < kWideTable; q += 256) table[q] = a.table[q];

    // ---- ring prologue: blocks 0 .. P-1 in flight ----
    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
#pragma unroll
    for (int b = 0; b < P; ++b) {
        ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);
        pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;
        pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;
    }
    __syncthreads();                                                          // (the tables; drains the prologue's DMA too: once)
    // counted wait for the block about to be consumed (vector-memory operations retire in issue order, stores included): behind its
    // DMA there are always the (P - 1) x KS / WPW pieces of the later blocks, and -- in the training pass, where every H x H block
    // ends with exactly ONE store instruction (a lane past the last row re-stores the last row's identical bytes: the instruction is
    // issued whatever the row count) -- the stores of the two blocks before it.  A site whose two predecessors may not have stored
    // (the first layer's blocks, the first two blocks of a layer) counts only the DMA: stricter, never weaker.  Without the store
    // count every block also waited for the previous block's store and for a DMA piece issued one block ago: ~one L2 round trip.
    constexpr int kWait = (P - 1) * (KS / WPW);
    constexpr int kWaitFull = kWait + (kTrain ? P : 0);

    // A block's head: counted wait, barrier, the block's slot.  Its DMA (block + P into the slot just freed) is issued by dma_next()
    // AFTER the block's first operand reads are on their way: the LDS round trip then passes under the DMA's address arithmetic
    // instead of following it (the shared TG_RING_NEXT issues the DMA first).
#define TG_WIDE_HEAD(WAITN)                                                                                \
    TG_RING_WAIT(WAITN)                                                                                    \
    if (!(TG_F32W_ABLATE & 1)) __builtin_amdgcn_s_barrier();                                               \
    asm volatile("" ::: "memory");                                                                         \
    const uint4* cur = ring + cur_slot * KS * 64;                                                          \
    cur_slot = (cur_slot + 1 == D) ? 0 : cur_slot + 1;
    auto dma_next = [&]() {
        if (!(TG_F32W_ABLATE & 2)) ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);
        pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;
        pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;
    };
    auto relu_bits4 = [&](f32x4& v) {
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m |= (v[r] > 0.0f ? 1u : 0u) << r;
            v[r] = fmaxf(v[r], 0.0f);
        }
        return m;
    };
    // one 16-feature output tile against a whole H-wide operand: 64 products, the lane's A operands 16 B at a time
    // (the A operands of pieces t + 2, t + 3 are requested before the products of pieces t, t + 1 are issued: an LDS round trip is
    // ~100 cycles, eight products 256)
    auto tile_products = [&](const uint4* __restrict__ cur, const f32x4 (&xin)[NT], f32x4 acc) {
        const uint4* __restrict__ p = cur + lane;
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
        uint4 wa = wide_lds_u4(p), wb = wide_lds_u4(p + 64);
        dma_next();
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            uint4 na = wa, nb = wb;
            if (t + 2 < NT) { na = wide_lds_u4(p + (t + 2) * 64); nb = wide_lds_u4(p + (t + 3) * 64); }
            // two accumulator chains (even / odd pieces), alternating: a dependent v_mfma_f32_16x16x4_f32 can issue 40 cycles after its
            // predecessor, an independent one after 32 -- one chain alone caps a wave at 80 % of the pipe whenever its SIMD-mate stalls
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), xin[t][0], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.x), xin[t + 1][0], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), xin[t][1], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.y), xin[t + 1][1], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), xin[t][2], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.z), xin[t + 1][2], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), xin[t][3], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.w), xin[t + 1][3], acc1, 0, 0, 0);
            wa = na; wb = nb;
        }
        acc = acc + acc1;                                            // (a fixed order: deterministic)
        // pin that order (hipcc otherwise sinks every read to just in front of its products): 4 reads, then 8 products + the 2 reads
        // of the group after next, ...
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (i + 2 < NT / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        return acc;
    };
    auto bits_slot = [&](int layer) { return bits_s + ((layer * WPW + wave) * 64 + lane) * 2; };
    auto store_tile = [&](float* gptr, int64_t row, int mo, const f32x4& v) {
#if TG_F32W_ABLATE & 4
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
#if TG_F32W_NT_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gptr + row * H + 16 * mo + 4 * g));
#else
        *reinterpret_cast<float4*>(gptr + row * H + 16 * mo + 4 * g) = float4{v[0], v[1], v[2], v[3]};
#endif
#endif
    };

    double s_surr = 0.0, s_crit = 0.0, s_kl = 0.0, s_cnt = 0.0;

    for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t row = round * 64 + wave * 16 + j;
        const bool valid = row < rows;
        const int64_t rowc = valid ? row : rows - 1;
        f32x4 xin[NT], xout[NT];
        uint32_t mb0 = 0, mb1 = 0;                       // this layer's ReLU mask bits: tile t -> bits 4 (t & 7) .. + 3 of word t >> 3

        // ---- layer 0: K padded to 32 in the stream (8 steps per tile); lane group g holds x[8 g .. 8 g + 8) ----
        {
            float xr[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {                // (no load under a branch: clamped address, select)
                const int c0 = 8 * g + 4 * q;
                const bool in = c0 < a.in_pad;
                const float4 v = *reinterpret_cast<const float4*>(a.x + rowc * a.in_pad + (in ? c0 : 0));
                xr[4 * q] = in ? v.x : 0.f; xr[4 * q + 1] = in ? v.y : 0.f; xr[4 * q + 2] = in ? v.z : 0.f; xr[4 * q + 3] = in ? v.w : 0.f;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                TG_WIDE_HEAD(kWait)
                dma_next();
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    const int mo = 8 * b + tt;
                    const float4 b4 = wide_lds_f4(table + 16 * mo + 4 * g);
                    f32x4 acc = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint4 w4 = wide_lds_u4(cur + (2 * tt + q) * 64 + lane);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w4.x), xr[4 * q + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w4.y), xr[4 * q + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w4.z), xr[4 * q + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w4.w), xr[4 * q + 3], acc, 0, 0, 0);
                    }
                    const uint32_t m = relu_bits4(acc);
                    if (mo < 8) mb0 |= m << (4 * mo); else mb1 |= m << (4 * (mo - 8));
                    xin[mo] = acc;
                    if constexpr (kTrain) {
                        if (a.acts[0] != nullptr) store_tile(a.acts[0], rowc, mo, acc);
                    }
                }
            }
            if constexpr (kTrain) wide_lds_st2(bits_slot(0), uint2{mb0, mb1});
        }
        // ---- hidden H x H layers: one streamed block per 16 output features ----
        for (int l = 1; l <= n_hh; ++l) {
            mb0 = mb1 = 0;
            const float* bias_l = table + l * H + 4 * g;
            float* act_l = kTrain ? pick(a.acts, l) : nullptr;
#pragma unroll
            for (int mo = 0; mo < NT; ++mo) {
                if (mo >= 2) { TG_RING_WAIT(kWaitFull) } else { TG_RING_WAIT(kWait) }
                TG_WIDE_HEAD(0x3f)                               // (the counted wait was the line above: 0x3f never waits)
                const float4 b4 = wide_lds_f4(bias_l + 16 * mo);
                f32x4 acc = tile_products(cur, xin, f32x4{b4.x, b4.y, b4.z, b4.w});
                const uint32_t m = relu_bits4(acc);
                if (mo < 8) mb0 |= m << (4 * mo); else mb1 |= m << (4 * (mo - 8));
                xout[mo] = acc;
                if constexpr (kTrain) store_tile(act_l, rowc, mo, acc);
            }
            if constexpr (kTrain) wide_lds_st2(bits_slot(l), uint2{mb0, mb1});
#pragma unroll
            for (int t = 0; t < NT; ++t) xin[t] = xout[t];
        }
        // ---- head: <= 4 outputs as fp32 dot products over the lane's 64 features, the four lane groups added in a fixed order ----
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        const int A = kTrain ? L.A : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < A) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 w = wide_lds_f4(wh_s + k * H + 16 * t + 4 * g);
                    s = fmaf(xin[t][0], w.x, s);
                    s = fmaf(xin[t][1], w.y, s);
                    s = fmaf(xin[t][2], w.z, s);
                    s = fmaf(xin[t][3], w.w, s);
                }
                // (g0 + g1) + (g2 + g3): every lane group ends up with the same bits
                const float p1 = __shfl_xor(s, 16, 64);
                const float pair = (g & 1) ? p1 + s : s + p1;
                const float p2 = __shfl_xor(pair, 32, 64);
                o[k] = ((g & 2) ? p2 + pair : pair + p2) + wide_lds_f(bh_s + k);
            }
        if constexpr (!kTrain) {
            if (valid && g == 0) *reinterpret_cast<float4*>(a.out + row * 4) = float4{o[0], o[1], o[2], o[3]};
        } else {
            float gr[4], c_surr, c_crit, c_kl;
            f32_loss_row<false>(L, o, row, rowc, valid, g == 0, gr, c_surr, c_crit, c_kl);
            if (valid && g == 0) {
                s_surr += (double)c_surr; s_crit += (double)c_crit; s_kl += (double)c_kl; s_cnt += 1.0;
                *reinterpret_cast<float4*>(L.dout4 + row * 4) = float4{gr[0], gr[1], gr[2], gr[3]};
            }
            // ---- backward: dZ_top = (g . W_head) * (a_top > 0), then dZ_below = (W^T . dZ) * mask per layer, top down ----
            // (mb0 / mb1 still hold the top layer's mask bits)
            float* dz_top = pick(a.dz, n_hh);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t mw = (t < 8 ? mb0 >> (4 * t) : mb1 >> (4 * (t - 8)));
                float4 s = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < L.A) {
                        const float4 w = wide_lds_f4(wh_s + k * H + 16 * t + 4 * g);
                        s.x = fmaf(gr[k], w.x, s.x); s.y = fmaf(gr[k], w.y, s.y); s.z = fmaf(gr[k], w.z, s.z); s.w = fmaf(gr[k], w.w, s.w);
                    }
                f32x4 d;
                d[0] = (mw & 1u) ? s.x : 0.f;
                d[1] = (mw & 2u) ? s.y : 0.f;
                d[2] = (mw & 4u) ? s.z : 0.f;
                d[3] = (mw & 8u) ? s.w : 0.f;
                xin[t] = d;
                if (dz_top != nullptr) store_tile(dz_top, rowc, t, d);
            }
            for (int l = n_hh; l >= 1; --l) {
                const uint2 mk = wide_lds_u2(bits_slot(l - 1));
                float* dz_l = pick(a.dz, l - 1);
#pragma unroll
                for (int ko = 0; ko < NT; ++ko) {
                    // (the two blocks before any backward block stored: the top layer's last forward blocks, or this pass's own)
                    TG_WIDE_HEAD(kWaitFull)
                    f32x4 acc = tile_products(cur, xin, f32x4{0.f, 0.f, 0.f, 0.f});
                    const uint32_t mw = (ko < 8 ? mk.x >> (4 * ko) : mk.y >> (4 * (ko - 8)));
                    acc[0] = (mw & 1u) ? acc[0] : 0.f;
                    acc[1] = (mw & 2u) ? acc[1] : 0.f;
                    acc[2] = (mw & 4u) ? acc[2] : 0.f;
                    acc[3] = (mw & 8u) ? acc[3] : 0.f;
                    xout[ko] = acc;
                    store_tile(dz_l, rowc, ko, acc);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) xin[t] = xout[t];
            }
        }
    }
    // the ring's prefetches of a round that never came: let them land before the workgroup's LDS goes away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef TG_WIDE_HEAD
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
            for (int w = 0; w < WPW; ++w) t += red_s[w * 4 + tid];
            L.work[(int64_t)blockIdx.x * 4 + tid] = t;
        }
        TG_CLOCK_PROBE_END(g_probe_f32_wide)
    }
}

static int f32_wide_grid(int64_t rows) {
    const int64_t n_rounds = ceil_div(rows, 64);
    const int slots = 2 * device_cus();
    return (int)(n_rounds < slots ? n_rounds : slots);
}

template <bool kTrain>
static int launch_f32_wide(const F32WideArgs& args, hipStream_t st) {
    auto kern = mlp_f32_wide_kernel<kTrain>;
    const size_t shmem = f32_wide_lds();
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32w_forward")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)f32_wide_grid(args.rows)), dim3(256), shmem, st, args);
    TG_LAUNCH_CHECK("tg_mlp_f32w_forward");
    return TG_OK;
}

static int fill_f32_wide(F32WideArgs& a, const float* d_x, int32_t in_pad, const float* d_stream, const float* d_table, int32_t n_hidden_layers,
                         int64_t rows, const char* what) {
    if (!(n_hidden_layers >= 1 && n_hidden_layers <= kWideMaxHidden))
        return set_error(TG_ERR_ARG, "%s: %d hidden layers outside 1..%d", what, n_hidden_layers, kWideMaxHidden);
    if (!(in_pad >= 8 && in_pad <= 32 && in_pad % 8 == 0)) return set_error(TG_ERR_ARG, "%s: padded input width %d (a multiple of 8, <= 32)", what, in_pad);
    if (!d_x || !d_stream || !d_table) return set_error(TG_ERR_ARG, "%s: null pointer", what);
    if (rows < 0) return set_error(TG_ERR_ARG, "%s: negative row count", what);
    if (((uintptr_t)d_x | (uintptr_t)d_stream) & 15) return set_error(TG_ERR_ARG, "%s: input / stream not 16-B aligned", what);
    a.x = d_x; a.in_pad = in_pad; a.n_hh = n_hidden_layers - 1; a.rows = rows;
    a.stream = reinterpret_cast<const uint4*>(d_stream); a.table = d_table;
    return TG_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// The same 16-row machine with the whole weight stream RESIDENT in LDS: H = 128 with at most one H x H layer -- BASELINE configs[1]
// (C2: CartPole GRPO, fp32 5-128-128-1, pipelines/cartpole_pipeline_grpo.py:54-76).  mlp_f32_chain.hip runs that shape with 32 rows
// per wave on v_mfma_f32_32x32x2_f32: 128 registers of activations, two waves per SIMD, and a granularity of 32 rows per wave --
// at C2's ~176,000 rows per update that is 5.39 wave-rounds per SIMD, i.e. 6 (10 % of the launch idle), with the matrix pipe 64 %
// busy inside them.  Here a wave owns 16 rows (input + output = 64 registers): TWELVE waves per CU (sixteen in no-grad launches), no barrier
// anywhere in the row loop (a block is an offset into the resident stream), and the rows are dealt to the waves 16 at a time,
// wave-major across the CUs, so that every SIMD gets 10 or 11 wave-rounds of C2's 10.78.
//   stream  fwd blocks [n_hh][NT] then (training) bwd blocks [n_hh][NT], a block = NT pieces x 64 lanes x 16 B:
//           fwd piece t of block mo: W_l[16 mo + i][16 t + 4 g .. + 3];  bwd piece t of block ko: {W_l[16 t + 4 g + r][16 ko + i]}
//   w0      [NT tiles][K4 = in_pad / 4 steps][64 lanes] floats: lane (i, g) of step s holds W0[16 mo + i][4 s + g]
//   table   [2][H] hidden biases | [4][H] head weights | [4] head bias (+ 12 pad)
// Outputs as tg_mlp_f32_forward_backward's, including the top layer's mask bits in ITS format (the weight-gradient job of
// mlp_f32_chain.hip rebuilds the top dZ from them): the two kernels are interchangeable in front of tg_mlp_f32_weight_grad.
// The matrix pipe is 80 % busy (no-grad launches 89 %); what is left is vector instructions, which on this chip take the pipe's time
// whichever wave issues them (profiles/r05_f32_res_kernel.md) -- hence: tile PAIRS on one accumulator chain each, ReLU + mask bit
// in three instructions per element, the head accumulated tile by tile (the top activation is never whole in registers), the next
// round's input row and this round's loss inputs loaded ahead, a round's addresses as scalar base + small lane offset.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kResWaves = 16;                  // no-grad launches: 4 waves per SIMD (~95 registers)
constexpr int kResWavesTrain = TG_F32R_TRAIN_WAVES;            // training launches: 3 per SIMD (143 registers; 16 waves = 128 registers, 22 spilled: measured, no faster)
constexpr int res_waves(bool train) { return train ? kResWavesTrain : kResWaves; }
constexpr int kResMaxHidden = 2;

struct F32ResArgs {
    const float* x; int32_t in_pad; int32_t n_hh; int64_t rows;
    const uint4* stream; const float* w0; const float* table;
    float* acts[kResMaxHidden]; float* dz[kResMaxHidden];
    float* out; uint32_t* top_mask;
    int32_t out_dim;            // (no-grad launches: head outputs to compute, the others are written as 0)
    F32Loss loss;
};

template <int H>
static size_t f32_res_lds(int n_hh, int in_pad, bool train) {
    constexpr int NT = H / 16;
    return (size_t)n_hh * NT * (train ? 2 : 1) * NT * 1024 + (size_t)NT * (in_pad / 4) * 256 + (size_t)((kResMaxHidden + 4) * H + 16) * 4 +
           kResWaves * 4 * 8;
}

TG_CLOCK_PROBE_VAR(g_probe_f32_res, attach_probe_f32_res)

template <int H, int K4, bool kTrain>                                    // K4 = padded input width / 4: the first layer's products per tile
__global__ __launch_bounds__(64 * res_waves(kTrain)) void mlp_f32_res_kernel(F32ResArgs a) {
    constexpr int NT = H / 16, WPW = res_waves(kTrain), BLK = NT * 64;           // uint4 per block
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n_hh = a.n_hh;
    const int n_stream = n_hh * NT * (kTrain ? 2 : 1);
    uint4* strm = lds;
    float* w0_s = reinterpret_cast<float*>(strm + n_stream * BLK);      // [NT][K4][64]
    float* table = w0_s + NT * K4 * 64;
    float* wh_s = table + kResMaxHidden * H;
    float* bh_s = wh_s + 4 * H;
    double* red_s = reinterpret_cast<double*>(bh_s + 16);
    const int64_t rows = a.rows;
    const int64_t n_wr = (rows + 15) / 16;                               // wave-rounds of 16 rows
    F32Loss L = a.loss;
    if constexpr (kTrain) {
        f32_loss_from_device(L);
        TG_CLOCK_PROBE_BEGIN(g_probe_f32_res)
    }
    for (int q = tid; q < n_stream * BLK; q += 64 * WPW) strm[q] = a.stream[q];
    for (int q = tid; q < NT * K4 * 64; q += 64 * WPW) w0_s[q] = a.w0[q];
    for (int q = tid; q < (kResMaxHidden + 4) * H + 16; q += 64 * WPW) table[q] = a.table[q];
    __syncthreads();

    // ReLU in place + the four mask bits of the tile (bit r: register r stayed positive).  Three instructions per element, two of them
    // written out: every vector instruction of this kernel is time the matrix pipe does not get (profiles/r05_f32_res_kernel.md), and
    // hipcc makes five of `fmaxf` + a comparison (a canonicalising max in front of the max; class test + select + or for the bit)
    auto relu_bits4 = [&](f32x4& v, uint32_t& m, int sh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p; uint32_t b;
            asm("v_max_f32 %0, 0, %1" : "=v"(p) : "v"(v[r]));                       // (as fmaxf(x, 0))
            asm("v_min_u32 %0, 1, %1" : "=v"(b) : "v"(p));                          // +0 -> 0, any positive float -> 1
            m |= b << (sh + r);
            v[r] = p;
        }
    };
    // v[r] kept where bit sh + r of m is set, else +0: sign-extended one-bit field, and
    auto mask4 = [&](f32x4& v, uint32_t m, int sh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe((int)m, sh + r, 1);
            v[r] = __uint_as_float(__float_as_uint(v[r]) & keep);
        }
    };
    // TWO 16-feature output tiles against a whole H-wide operand: one accumulator chain each, alternating (a dependent product
    // issues every other slot), the A operands of the next step requested before this step's eight products
    auto tile_pair = [&](const uint4* __restrict__ blk_a, const uint4* __restrict__ blk_b, const f32x4 (&xin)[NT], f32x4& acc_a, f32x4& acc_b) {
        const uint4* __restrict__ pa = blk_a + lane;
        const uint4* __restrict__ pb = blk_b + lane;
#if TG_F32R_ABLATE & 4
        uint4 wa = uint4{(unsigned)lane, 1u, 2u, 3u}, wb = wa;
        asm volatile("" : "+v"(wa.x), "+v"(wa.y), "+v"(wa.z), "+v"(wa.w));
        asm volatile("" : "+v"(wb.x), "+v"(wb.y), "+v"(wb.z), "+v"(wb.w));
#else
        uint4 wa = wide_lds_u4(pa), wb = wide_lds_u4(pb);
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4 na = wa, nb = wb;
#if !(TG_F32R_ABLATE & 4)
            if (t + 1 < NT) { na = wide_lds_u4(pa + (t + 1) * 64); nb = wide_lds_u4(pb + (t + 1) * 64); }
#endif
#if TG_F32R_ABLATE & 2
            asm volatile("" ::"v"(wa.x), "v"(wa.y), "v"(wa.z), "v"(wa.w), "v"(wb.x), "v"(wb.y), "v"(wb.z), "v"(wb.w));
            acc_a[t & 3] += xin[t][0]; acc_b[t & 3] += xin[t][1];
#else
            acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), xin[t][0], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.x), xin[t][0], acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), xin[t][1], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.y), xin[t][1], acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), xin[t][2], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.z), xin[t][2], acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), xin[t][3], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb.w), xin[t][3], acc_b, 0, 0, 0);
#endif
            wa = na; wb = nb;
        }
#if !(TG_F32R_ABLATE & 6)
        // (pin the order -- and with it the number of pieces in registers at a time: hipcc otherwise hoists a whole block's reads)
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (i + 2 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#endif
    };
    // Addresses: a round's 16 rows start at a wave-uniform base (scalar registers), the lane adds a small offset -- `jr` = the lane's
    // row within the round, clamped into range (a lane past the last row re-does the last row: identical bytes).
    auto store_tile = [&](float* round_base, int jr, int mo, const f32x4& v) {
#if TG_F32R_ABLATE & 1
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
#if TG_F32R_NT_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(round_base + (jr * H + 4 * g) + 16 * mo));
#else
        *reinterpret_cast<float4*>(round_base + (jr * H + 4 * g) + 16 * mo) = float4{v[0], v[1], v[2], v[3]};
#endif
#endif
    };
    auto lane_row = [&](int64_t q) { const int64_t last = rows - 1 - q * 16; return last < j ? (int)last : j; };

    double s_surr = 0.0, s_crit = 0.0, s_kl = 0.0, s_cnt = 0.0;
    // wave-rounds dealt wave-major across the workgroups: q = k * (WPW * grid) + wave * grid + block
    const int64_t q_step = (int64_t)WPW * gridDim.x;
    int64_t q = (int64_t)wave * gridDim.x + blockIdx.x;
    float xr[K4];                                        // the round's input row: step s contracts inputs 4 s + g (g = the lane group)
    if (q < n_wr) {
        const float* xq = a.x + q * (16 * 4 * K4);
        const int jr = lane_row(q);
#pragma unroll
        for (int s = 0; s < K4; ++s) xr[s] = xq[jr * (4 * K4) + g + 4 * s];
    }
    for (; q < n_wr; q += q_step) {
        const int64_t row = q * 16 + j;
        const bool valid = row < rows;
        const int jr = lane_row(q);
        F32LossIn lin;
        if constexpr (kTrain) lin = f32_loss_load(L, q * 16, jr);        // (used after both layers: in flight behind them)
        f32x4 xin[NT];
        uint32_t mb0 = 0, mb1 = 0;                       // ReLU mask bits of layer 0 / layer 1: tile t -> bits 4 t .. 4 t + 3
        // ---- layer 0 ----
#pragma unroll
        for (int mo = 0; mo < NT; ++mo) {
            const float4 b4 = wide_lds_f4(table + 16 * mo + 4 * g);
            f32x4 acc = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int s = 0; s < K4; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wide_lds_f(w0_s + (mo * K4 + s) * 64 + lane), xr[s], acc, 0, 0, 0);
            relu_bits4(acc, mb0, 4 * mo);
            xin[mo] = acc;
            if constexpr (kTrain) {
                if (a.acts[0] != nullptr) store_tile(a.acts[0] + q * (16 * H), jr, mo, acc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (phase boundary: nothing hoisted across, registers stay bounded)
        {   // the next round's input row, into the registers the products above have just read (the last round re-reads its own)
            const int64_t qn = q + q_step < n_wr ? q + q_step : q;
            const float* xq = a.x + qn * (16 * 4 * K4);
            const int jn = lane_row(qn);
#pragma unroll
            for (int s = 0; s < K4; ++s) xr[s] = xq[jn * (4 * K4) + g + 4 * s];
        }
        // ---- head: <= 4 outputs as fp32 dot products over the lane's features (tiles in order, registers in order), accumulated tile
        // by tile as the top layer's tiles come out of the matrix pipe: the top activation is never whole in registers ----
        float sacc[4] = {0.f, 0.f, 0.f, 0.f};
        const int A = kTrain ? L.A : a.out_dim;
        auto head_tile = [&](int k, const f32x4& v, int t) {
#if !(TG_F32R_ABLATE & 8)
            const float4 w = wide_lds_f4(wh_s + k * H + 16 * t + 4 * g);
            sacc[k] = fmaf(v[0], w.x, sacc[k]);
            sacc[k] = fmaf(v[1], w.y, sacc[k]);
            sacc[k] = fmaf(v[2], w.z, sacc[k]);
            sacc[k] = fmaf(v[3], w.w, sacc[k]);
#endif
        };
        if (n_hh == 1) {
            // (one copy of the layer per output count: a run-time `k < A` inside would cut the schedule of every tile pair)
            auto layer1 = [&](auto a_tag) {
                constexpr int AA = decltype(a_tag)::value;
#pragma unroll
                for (int mo = 0; mo < NT; mo += 2) {
                    const float4 ba = wide_lds_f4(table + H + 16 * mo + 4 * g), bb = wide_lds_f4(table + H + 16 * (mo + 1) + 4 * g);
                    f32x4 acc_a = {ba.x, ba.y, ba.z, ba.w}, acc_b = {bb.x, bb.y, bb.z, bb.w};
                    tile_pair(strm + mo * BLK, strm + (mo + 1) * BLK, xin, acc_a, acc_b);
                    relu_bits4(acc_a, mb1, 4 * mo);
                    relu_bits4(acc_b, mb1, 4 * (mo + 1));
                    if constexpr (kTrain) { store_tile(a.acts[1] + q * (16 * H), jr, mo, acc_a); store_tile(a.acts[1] + q * (16 * H), jr, mo + 1, acc_b); }
#pragma unroll
                    for (int k = 0; k < AA; ++k) head_tile(k, acc_a, mo);
#pragma unroll
                    for (int k = 0; k < AA; ++k) head_tile(k, acc_b, mo + 1);
                    __builtin_amdgcn_sched_barrier(0);       // (the next pair's reads stay behind this pair's epilogue: registers)
                }
            };
            switch (A) {
                case 1: layer1(std::integral_constant<int, 1>{}); break;
                case 2: layer1(std::integral_constant<int, 2>{}); break;
                case 3: layer1(std::integral_constant<int, 3>{}); break;
                default: layer1(std::integral_constant<int, 4>{}); break;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < A) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) head_tile(k, xin[t], t);
                }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (phase boundary: nothing hoisted across, registers stay bounded)
        const uint32_t mtop = n_hh == 1 ? mb1 : mb0;
        float o[4] = {0.f, 0.f, 0.f, 0.f};                   // the four lane groups added in a fixed order
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < A) {
                const float p1 = __shfl_xor(sacc[k], 16, 64);
                const float pair = (g & 1) ? p1 + sacc[k] : sacc[k] + p1;
                const float p2 = __shfl_xor(pair, 32, 64);
                o[k] = ((g & 2) ? p2 + pair : pair + p2) + wide_lds_f(bh_s + k);
            }
        if constexpr (!kTrain) {
            if (valid && g == 0) *reinterpret_cast<float4*>(a.out + q * 64 + jr * 4) = float4{o[0], o[1], o[2], o[3]};
        } else {
            float gr[4], c_surr, c_crit, c_kl;
#if TG_F32R_ABLATE & 8
            gr[0] = o[0] + lin.adv; gr[1] = gr[2] = gr[3] = 0.f; c_surr = c_crit = c_kl = o[0];
#else
            f32_loss_compute<false>(L, lin, o, row, valid, g == 0, gr, c_surr, c_crit, c_kl);
#endif
            if (valid && g == 0) {
                s_surr += (double)c_surr; s_crit += (double)c_crit; s_kl += (double)c_kl; s_cnt += 1.0;
                *reinterpret_cast<float4*>(L.dout4 + q * 64 + jr * 4) = float4{gr[0], gr[1], gr[2], gr[3]};
            }
            // the top layer's mask bits in tg_mlp_f32_forward_backward's row format: feature f = 32 mt + 8 q + 4 hh + low is bit
            // low + 4 q + 16 (mt & 1) of word hh * (MT / 2) + (mt >> 1); this lane holds f = 16 t + 4 g + r, i.e. hh = g & 1,
            // q = 2 (t & 1) + (g >> 1), mt = t >> 1: its nibble of tile t goes to bit 8 (t & 1) + 4 (g >> 1) + 16 ((t >> 1) & 1) of
            // word hh * (MT / 2) + (t >> 2); the lane two groups on holds the other half of the same words
            if (a.top_mask != nullptr) {
                constexpr int NW = NT / 4;               // words per lane half (MT / 2)
                uint32_t wds[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    // nibbles 4 w .. 4 w + 3 of mtop -> bytes: (n0 | n1 << 8 | n2 << 16 | n3 << 24), then up by 4 in the upper groups
                    const uint32_t h = (mtop >> (16 * w)) & 0xffffu;
                    uint32_t v = (h | (h << 8)) & 0x00ff00ffu;                        // bytes 0 / 2 hold the nibble pairs (n1 n0) / (n3 n2)
                    v = (v | (v << 4)) & 0x0f0f0f0fu;                                  // one nibble per byte
                    v <<= 4 * (g >> 1);
                    wds[w] = v | (uint32_t)__shfl_xor((int)v, 32, 64);
                }
#if TG_F32R_ABLATE & 1
                asm volatile("" ::"v"(wds[0]), "v"(wds[1]));
#else
                if (g < 2) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) (a.top_mask + q * (16 * 2 * NW))[jr * (2 * NW) + g * NW + w] = wds[w];
                }
#endif
            }
            __builtin_amdgcn_sched_barrier(0);                   // (phase boundary: nothing hoisted across, registers stay bounded)
            // ---- backward: dZ_top = (g . W_head) * (a_top > 0), then dZ_0 = (W_1^T . dZ_1) * mask ----
            float* dz_top = n_hh == 1 ? a.dz[1] : a.dz[0];
            float* dz_top_q = dz_top + q * (16 * H);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < L.A) {
                        const float4 w = wide_lds_f4(wh_s + k * H + 16 * t + 4 * g);
                        d[0] = fmaf(gr[k], w.x, d[0]); d[1] = fmaf(gr[k], w.y, d[1]); d[2] = fmaf(gr[k], w.z, d[2]); d[3] = fmaf(gr[k], w.w, d[3]);
                    }
                mask4(d, mtop, 4 * t);
                xin[t] = d;
                if (dz_top != nullptr) store_tile(dz_top_q, jr, t, d);
            }
            if (n_hh == 1) {
#pragma unroll
                for (int ko = 0; ko < NT; ko += 2) {
                    f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
                    tile_pair(strm + (NT + ko) * BLK, strm + (NT + ko + 1) * BLK, xin, acc_a, acc_b);
                    mask4(acc_a, mb0, 4 * ko);
                    mask4(acc_b, mb0, 4 * (ko + 1));
                    store_tile(a.dz[0] + q * (16 * H), jr, ko, acc_a);
                    store_tile(a.dz[0] + q * (16 * H), jr, ko + 1, acc_b);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if constexpr (kTrain) {
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
            for (int w = 0; w < WPW; ++w) t += red_s[w * 4 + tid];
            L.work[(int64_t)blockIdx.x * 4 + tid] = t;
        }
        TG_CLOCK_PROBE_END(g_probe_f32_res)
    }
}

template <int K4, bool kTrain>
static int launch_f32_res_k(const F32ResArgs& args, hipStream_t st) {
    auto kern = mlp_f32_res_kernel<128, K4, kTrain>;
    const size_t shmem = f32_res_lds<128>(args.n_hh, args.in_pad, kTrain);
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_f32r_forward")) return rc;
    const int64_t wgs = ceil_div(ceil_div(args.rows, (int64_t)16), (int64_t)res_waves(kTrain));
    const int grid = (int)(wgs < device_cus() ? wgs : device_cus());
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * res_waves(kTrain)), shmem, st, args);
    TG_LAUNCH_CHECK("tg_mlp_f32r_forward");
    return TG_OK;
}
template <bool kTrain>
static int launch_f32_res(const F32ResArgs& args, hipStream_t st) {
    switch (args.in_pad) {
        case 8: return launch_f32_res_k<2, kTrain>(args, st);
        case 16: return launch_f32_res_k<4, kTrain>(args, st);
        case 24: return launch_f32_res_k<6, kTrain>(args, st);
        default: return launch_f32_res_k<8, kTrain>(args, st);
    }
}

static int fill_f32_res(F32ResArgs& a, const float* d_x, int32_t in_pad, const float* d_stream, const float* d_w0, const float* d_table,
                        int32_t hidden, int32_t n_hidden_layers, int64_t rows, const char* what) {
    if (hidden != 128) return set_error(TG_ERR_ARG, "%s: hidden width %d (128)", what, hidden);
    if (!(n_hidden_layers >= 1 && n_hidden_layers <= kResMaxHidden))
        return set_error(TG_ERR_ARG, "%s: %d hidden layers outside 1..%d (the whole stream must fit the LDS)", what, n_hidden_layers, kResMaxHidden);
    if (!(in_pad >= 8 && in_pad <= 32 && in_pad % 8 == 0)) return set_error(TG_ERR_ARG, "%s: padded input width %d (a multiple of 8, <= 32)", what, in_pad);
    if (!d_x || !d_w0 || !d_table || (n_hidden_layers > 1 && !d_stream)) return set_error(TG_ERR_ARG, "%s: null pointer", what);
    if (rows < 0) return set_error(TG_ERR_ARG, "%s: negative row count", what);
    if (((uintptr_t)d_stream) & 15) return set_error(TG_ERR_ARG, "%s: stream not 16-B aligned", what);
    a.x = d_x; a.in_pad = in_pad; a.n_hh = n_hidden_layers - 1; a.rows = rows;
    a.stream = reinterpret_cast<const uint4*>(d_stream); a.w0 = d_w0; a.table = d_table;
    return TG_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradients at H = 256 (what `loss.backward()` leaves in .grad, algorithms/ppo.py:181-183): dW_l = dZ_l^T . A_{l-1}, db_l =
// column sums of dZ_l, head dW_h = g^T . A_top -- the jobs of tg_mlp_f32_weight_grad, one 8-wave workgroup per CU.
//   wide job (a 256 x 256 layer): the whole gradient lives in the workgroup's accumulators -- wave (wm, wn) of a 2 x 4 grid owns a
//     128 x 64 block = 4 x 2 tiles of v_mfma_f32_32x32x2_f32 (128 registers).  The contraction runs over ROWS: A operand lane
//     (i, kk) = P[row 2 s + kk][m0 + i], B operand lane (j, kk) = Q[row 2 s + kk][n0 + j], straight out of row-major LDS panels
//     (one conflict-free ds_read_b32 per operand: 6 reads per 8 products).  A stage = 16 rows of P and of Q (32 KiB), LDS-DMA
//     through 3 slots, counted vmcnt + one raw s_barrier per stage; no per-stage epilogue, so the eight waves in lock step lose
//     nothing but the barrier's skew.  Bias sums: one thread per column and half stage, beside the matrix work.
//   light jobs (first layer: P = dZ_0, Q = the padded input rows; head: Q = the top activation, P = d loss / d output): bound by
//     the bytes of their one wide operand; a stage = 32 rows of it (32 KiB) + the narrow operand as a [32][32]-float image.
// Slabs: as mlp_f32_chain.hip's jobs write them -- [256][N] row-major then the 256 bias sums (head: [4][256] then 4) -- for the
// same fixed-order reduction launch (mlp_f32_dw_finish_kernel, which may carry the optimizer step).
// ------------------------------------------------------------------------------------------------------------------------
static __device__ uint4 g_f32w_zero16;
typedef __attribute__((address_space(3))) void f32w_lds_void;

constexpr int kWdSRW = 16, kWdSRL = 32;                                 // rows per stage: wide / light
constexpr int kWdSlotW = 2 * kWdSRW * kWideH * 4;                       // 32 KiB
constexpr int kWdSlotL = kWdSRL * kWideH * 4 + kWdSRL * 128;            // 36 KiB
constexpr int kWdDW = 4, kWdDL = 4;                                     // ring slots: wide jobs (bound by the matrix pipe: 9 GB/s per CU) /
                                                                        // light jobs (bound by the bytes in flight: 3 stages = 108 KiB)
constexpr int kWdNGW = 4, kWdNGL = 5;                                   // DMA instructions per wave and stage
constexpr int kWdLds = kWdDL * kWdSlotL > kWdDW * kWdSlotW ? kWdDL * kWdSlotL : kWdDW * kWdSlotW;

// `nrow` rows x 256 floats of `gsrc` (row-major) from row r0 on into a linear LDS panel: 1 KiB pieces = one row each, waves take
// pieces wave, wave + 8, ...  kZero: rows past the end arrive as zeros instead of as re-reads of the last row
// (`zero` = &g_f32w_zero16 held in registers by the caller: written in place, hipcc re-loads the symbol's address through the GOT in
// front of every DMA -- a scalar memory round trip per piece, ~15 % of a wide stage)
template <bool kZero, int NROW>
__device__ static inline void f32w_dma_rows(const float* __restrict__ gsrc, int64_t r0, int64_t rows, char* panel, int wave, int lane,
                                            const uint4* zero) {
#pragma unroll
    for (int t = 0; t < NROW / 8; ++t) {
        const int q = wave + 8 * t;
        const int64_t r = r0 + q;
        const uint4* src = reinterpret_cast<const uint4*>(gsrc + (r < rows ? r : rows - 1) * kWideH) + lane;
        if constexpr (kZero) src = r < rows ? src : zero;
        __builtin_amdgcn_global_load_lds(src, (f32w_lds_void*)(panel + q * 1024), 16, 0, 0);
    }
}

TG_CLOCK_PROBE_VAR(g_probe_f32_wide_dw, attach_probe_f32_wide_dw)

__global__ __launch_bounds__(512, 2) void mlp_f32_wide_dw_kernel(F32DwArgs args, int64_t rows, float* __restrict__ ws) {
    constexpr int H = kWideH;
    extern __shared__ uint4 lds[];
    char* lds_c = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const uint4* zero16 = &g_f32w_zero16;
    asm volatile("" : "+s"(zero16));                        // (opaque: the address stays in a register pair)
    TG_CLOCK_PROBE_BEGIN(g_probe_f32_wide_dw)
    // this workgroup's job (uniform)
    int ji = 0;
#pragma unroll
    for (int t = 1; t < kF32DwMaxJobs; ++t)
        if (t < args.n_jobs && (int)blockIdx.x >= args.job[t].first_block) ji = t;
    const float* jp = args.job[0].p; const float* jq = args.job[0].q;
    int jkind = args.job[0].kind, jn = args.job[0].n, jfirst = args.job[0].first_block, jnb = args.job[0].n_blocks, jslab = args.job[0].slab_len;
    int64_t joff = args.job[0].slab_off;
#pragma unroll
    for (int t = 1; t < kF32DwMaxJobs; ++t)
        if (ji == t) {
            jp = args.job[t].p; jq = args.job[t].q; jkind = args.job[t].kind; jn = args.job[t].n; jfirst = args.job[t].first_block;
            jnb = args.job[t].n_blocks; jslab = args.job[t].slab_len; joff = args.job[t].slab_off;
        }
    const int my = (int)blockIdx.x - jfirst, nb = jnb;
    const bool head = jkind == F32DW_HEAD, narrow = !head && jn <= 32;
    float* slab = ws + joff + (int64_t)my * jslab;

    if (!head && !narrow) {
        // ================= wide job =================
        constexpr int SR = kWdSRW, D = kWdDW, P = D - 1, NG = kWdNGW;
        const int64_t n_st = (rows + SR - 1) / SR;
        const int wm = wave >> 2, wn = wave & 3;
        f32x16 acc[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
        float bsum = 0.f;
        const int bcol = tid & 255, brow0 = (tid >> 8) * (SR / 2);       // bias: column, first row of this thread's half stage
        auto issue = [&](int64_t sg, int slot) {
            char* sb = lds_c + slot * kWdSlotW;
            f32w_dma_rows<true, SR>(jp, sg * SR, rows, sb, wave, lane, zero16);
            f32w_dma_rows<false, SR>(jq, sg * SR, rows, sb + SR * H * 4, wave, lane, zero16);
        };
        int64_t sg_issue = my;
        int slot_issue = 0, slot = 0;
#pragma unroll 1
        for (int t = 0; t < P; ++t) {
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        }
#pragma unroll 1
        for (int64_t sg = my; sg < n_st; sg += nb) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NG) : "memory");
#if !(TG_F32W_ABLATE & 8)                              /* probe builds (timing only): bit 3 = no stage barrier, bit 4 = no DMA inside the stage loop */
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
#if !(TG_F32W_ABLATE & 16)
            issue(sg_issue, slot_issue);
#endif
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
            const float* Pp = reinterpret_cast<const float*>(lds_c + slot * kWdSlotW);
            const float* Qp = Pp + SR * H;
            slot = slot + 1 == D ? 0 : slot + 1;
            const float* pa = Pp + kk * H + 128 * wm + i;
            const float* qa = Qp + kk * H + 64 * wn + i;
            float av[4], bv[2], an[4], bn[2];
#pragma unroll
            for (int x = 0; x < 4; ++x) av[x] = wide_lds_f(pa + 32 * x);
#pragma unroll
            for (int y = 0; y < 2; ++y) bv[y] = wide_lds_f(qa + 32 * y);
            float bcur = wide_lds_f(Pp + brow0 * H + bcol), bnext = 0.f;     // (bias column: consumed one step after it is read, like the operands)
#pragma unroll
            for (int s = 0; s < SR / 2; ++s) {
                if (s + 1 < SR / 2) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) an[x] = wide_lds_f(pa + (2 * s + 2) * H + 32 * x);
#pragma unroll
                    for (int y = 0; y < 2; ++y) bn[y] = wide_lds_f(qa + (2 * s + 2) * H + 32 * y);
                    bnext = wide_lds_f(Pp + (brow0 + s + 1) * H + bcol);
                }
                bsum += bcur;                                             // (rows past the end are zeros)
                bcur = bnext;
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < 4; ++x) av[x] = an[x];
#pragma unroll
                for (int y = 0; y < 2; ++y) bv[y] = bn[y];
            }
            // pin the order: a step's LDS reads (6 operands + 1 bias element, merged pairwise by hipcc: ~4 instructions) ahead of the
            // previous step's 8 products
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int s = 0; s < SR / 2; ++s) {
                if (s + 1 < SR / 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
        // the two half stages' bias sums meet in LDS and are added in a fixed order
        float* red = reinterpret_cast<float*>(lds_c);
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < H) slab[H * H + tid] = red[tid] + red[H + tid];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int m0 = 128 * wm + 32 * x, n0 = 64 * wn + 32 * y;
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(m0 + (r & 3) + 8 * (r >> 2) + 4 * kk) * H + n0 + i] = acc[x][y][r];
            }
    } else {
        // ================= light job: one wide operand, 32 rows per stage =================
        constexpr int SR = kWdSRL, D = kWdDL, P = D - 1, NG = kWdNGL;
        const int64_t n_st = (rows + SR - 1) / SR;
        const float* wide = head ? jq : jp;             // head: the top activation; first layer: the bottom dZ
        const float* thin = head ? jp : jq;             // head: g [rows][4]; first layer: x [rows][N]
        const int thin_f4 = head ? 1 : jn / 4;          // float4 per row of the narrow operand
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float hacc[4] = {0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
        const int bcol = tid & 255, brow0 = (tid >> 8) * (SR / 2);
        auto issue = [&](int64_t sg, int slot) {
            char* sb = lds_c + slot * kWdSlotL;
            const int64_t r0 = sg * SR;
            // (the first layer's dZ arrives as zeros past the end: its products and column sums need no masking; the head's
            // activation is clamped and its g rows arrive as zeros)
            if (head) f32w_dma_rows<false, SR>(wide, r0, rows, sb, wave, lane, zero16);
            else f32w_dma_rows<true, SR>(wide, r0, rows, sb, wave, lane, zero16);
            // the narrow operand as a zero-padded [SR][32 floats] image: 8 lanes per row, 8 rows per piece, 4 pieces (waves 4..7 repeat)
            const int piece = wave & 3;
            const int64_t r = r0 + piece * 8 + (lane >> 3);
            const int c4 = lane & 7;
            const uint4* src = (c4 < thin_f4 && r < rows) ? reinterpret_cast<const uint4*>(thin + r * (4 * thin_f4)) + c4 : zero16;
            __builtin_amdgcn_global_load_lds(src, (f32w_lds_void*)(sb + SR * H * 4 + piece * 1024), 16, 0, 0);
        };
        int64_t sg_issue = my;
        int slot_issue = 0, slot = 0;
#pragma unroll 1
        for (int t = 0; t < P; ++t) {
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
        }
#pragma unroll 1
        for (int64_t sg = my; sg < n_st; sg += nb) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NG) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(sg_issue, slot_issue);
            sg_issue += nb;
            slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
            const float* W = reinterpret_cast<const float*>(lds_c + slot * kWdSlotL);
            const float* T = W + SR * H;                // the narrow image [SR][32]
            slot = slot + 1 == D ? 0 : slot + 1;
            if (head) {
                // one column and half a stage per thread; the activation rows past the end are re-reads, their g rows zeros
#pragma unroll 4
                for (int r = 0; r < SR / 2; ++r) {
                    const float4 g4 = wide_lds_f4(T + 32 * (brow0 + r));
                    const float qv = wide_lds_f(W + (brow0 + r) * H + bcol);
                    hacc[0] = fmaf(g4.x, qv, hacc[0]); hacc[1] = fmaf(g4.y, qv, hacc[1]);
                    hacc[2] = fmaf(g4.z, qv, hacc[2]); hacc[3] = fmaf(g4.w, qv, hacc[3]);
                }
                if (tid < 4) {
                    for (int r = 0; r < SR; ++r) bsum += wide_lds_f(T + 32 * r + tid);
                }
            } else {
#pragma unroll 8
                for (int s = 0; s < SR / 2; ++s) {
                    const float av = wide_lds_f(W + (2 * s + kk) * H + 32 * wave + i);
                    const float bv = wide_lds_f(T + (2 * s + kk) * 32 + i);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
#pragma unroll 4
                for (int r = 0; r < SR / 2; ++r) bsum += wide_lds_f(W + (brow0 + r) * H + bcol);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* red = reinterpret_cast<float*>(lds_c);
        __syncthreads();
        if (head) {
            // the two half stages' sums, in a fixed order
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k * 512 + tid] = hacc[k];
            __syncthreads();
            if (tid < H) {
#pragma unroll
                for (int k = 0; k < 4; ++k) slab[k * H + tid] = red[k * 512 + tid] + red[k * 512 + H + tid];
            }
            if (tid < 4) slab[4 * H + tid] = bsum;
        } else {
            red[tid] = bsum;
            __syncthreads();
            if (tid < H) slab[H * 32 + tid] = red[tid] + red[H + tid];
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + i] = acc[r];
        }
    }
    TG_CLOCK_PROBE_END(g_probe_f32_wide_dw)
}

int launch_f32_wide_dw(const F32DwArgs& args, int64_t rows, float* d_workspace, int grid, hipStream_t st) {
    auto kern = mlp_f32_wide_dw_kernel;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, (size_t)kWdLds, opt_in, "tg_mlp_f32_weight_grad")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), (size_t)kWdLds, st, args, rows, d_workspace);
    return TG_OK;
}

int attach_probe_f32w(int which, void* d_probe) {
    if (which != 0) return attach_probe_f32_wide_dw(d_probe);
    const int rc = attach_probe_f32_wide(d_probe);
    return rc ? rc : attach_probe_f32_res(d_probe);
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_mlp_f32r_supported(int32_t hidden, int32_t n_hidden_layers, int32_t in_pad) {
    return hidden == 128 && n_hidden_layers >= 1 && n_hidden_layers <= kResMaxHidden && in_pad >= 8 && in_pad <= 32 && in_pad % 8 == 0 &&
           f32_res_lds<128>(n_hidden_layers - 1, in_pad, true) <= 160 * 1024;
}
int64_t tg_mlp_f32r_stream_floats(int32_t hidden, int32_t n_hidden_layers) { return (int64_t)2 * (n_hidden_layers - 1) * (hidden / 16) * (hidden / 16) * 256; }
int64_t tg_mlp_f32r_w0_floats(int32_t hidden, int32_t in_pad) { return (int64_t)(hidden / 16) * (in_pad / 4) * 64; }
int tg_mlp_f32r_grid(int64_t rows) {          // workgroups of a training launch = rows of partial loss sums it writes to tg_chain_loss.d_work
    const int64_t wgs = ceil_div(ceil_div(rows, (int64_t)16), (int64_t)kResWavesTrain);
    return (int)(wgs < device_cus() ? wgs : device_cus());
}
int64_t tg_mlp_f32r_table_floats(int32_t hidden) { return (int64_t)(kResMaxHidden + 4) * hidden + 16; }

int tg_mlp_f32r_forward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_w0, const float* d_table, int32_t hidden,
                        int32_t n_hidden_layers, int32_t out_dim, int64_t rows, float* d_out, void* stream) {
    F32ResArgs a{};
    if (int rc = fill_f32_res(a, d_x, in_pad, d_stream, d_w0, d_table, hidden, n_hidden_layers, rows, "tg_mlp_f32r_forward")) return rc;
    TG_REQUIRE(d_out, "tg_mlp_f32r_forward: null output");
    TG_REQUIRE(out_dim >= 1 && out_dim <= 4, "tg_mlp_f32r_forward: %d outputs (1..4)", out_dim);
    if (rows == 0) return TG_OK;
    a.out = d_out; a.out_dim = out_dim;
    return launch_f32_res<false>(a, (hipStream_t)stream);
}

int tg_mlp_f32r_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_w0, const float* d_table, int32_t hidden,
                                 int32_t n_hidden_layers, int64_t rows, void* const* d_acts, void* const* d_dz, void* d_top_maskbits,
                                 const tg_chain_loss* loss, void* stream) {
    F32ResArgs a{};
    if (int rc = fill_f32_res(a, d_x, in_pad, d_stream, d_w0, d_table, hidden, n_hidden_layers, rows, "tg_mlp_f32r_forward_backward")) return rc;
    TG_REQUIRE(loss && d_acts && d_dz, "tg_mlp_f32r_forward_backward: null pointer");
    TG_REQUIRE(loss->d_dout8 && loss->d_work, "tg_mlp_f32r_forward_backward: loss outputs missing");
    TG_REQUIRE(loss->act_dim >= 1 && loss->act_dim <= 4, "tg_mlp_f32r_forward_backward: %d outputs (1..4)", loss->act_dim);
    TG_REQUIRE(loss->kind == 1 ? (loss->d_ret != nullptr && loss->act_dim == 1)
                               : (loss->d_act && loss->d_adv && (loss->d_logp_old || loss->d_logp_old_out) && loss->act_row_stride == loss->act_dim &&
                                  loss->act_col_stride == 1),
               "tg_mlp_f32r_forward_backward: loss inputs missing (the actions must be contiguous [rows][A])");
    if (rows == 0) return TG_OK;
    for (int l = 0; l < n_hidden_layers; ++l) {
        // as tg_mlp_f32_forward_backward: the first activation / the top layer's dZ may be left out (rebuilt by the weight-gradient job)
        const bool a_opt = l == 0 && n_hidden_layers >= 2, z_opt = l == n_hidden_layers - 1 && n_hidden_layers >= 2 && d_top_maskbits;
        TG_REQUIRE((d_acts[l] || a_opt) && (d_dz[l] || z_opt), "tg_mlp_f32r_forward_backward: buffer %d is null", l);
        a.acts[l] = (float*)d_acts[l];
        a.dz[l] = (float*)d_dz[l];
    }
    a.top_mask = (uint32_t*)d_top_maskbits;
    fill_f32_loss(a.loss, loss);
    return launch_f32_res<true>(a, (hipStream_t)stream);
}

int tg_mlp_f32w_blocks(void) { return 2 * device_cus(); }

int64_t tg_mlp_f32w_stream_floats(int32_t n_hidden_layers) { return (int64_t)(2 + 2 * (n_hidden_layers - 1) * (kWideH / 16)) * kWidePieces * 256; }

int64_t tg_mlp_f32w_table_floats(void) { return kWideTable; }

int tg_mlp_f32w_forward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_table, int32_t n_hidden_layers, int64_t rows,
                        float* d_out, void* stream) {
    F32WideArgs a{};
    if (int rc = fill_f32_wide(a, d_x, in_pad, d_stream, d_table, n_hidden_layers, rows, "tg_mlp_f32w_forward")) return rc;
    TG_REQUIRE(d_out, "tg_mlp_f32w_forward: null output");
    if (rows == 0) return TG_OK;
    a.out = d_out;
    return launch_f32_wide<false>(a, (hipStream_t)stream);
}

int tg_mlp_f32w_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_table, int32_t n_hidden_layers,
                                 int64_t rows, void* const* d_acts, void* const* d_dz, const tg_chain_loss* loss, void* stream) {
    F32WideArgs a{};
    if (int rc = fill_f32_wide(a, d_x, in_pad, d_stream, d_table, n_hidden_layers, rows, "tg_mlp_f32w_forward_backward")) return rc;
    TG_REQUIRE(loss && d_acts && d_dz, "tg_mlp_f32w_forward_backward: null pointer");
    TG_REQUIRE(loss->d_dout8 && loss->d_work, "tg_mlp_f32w_forward_backward: loss outputs missing");
    TG_REQUIRE(loss->act_dim >= 1 && loss->act_dim <= 4, "tg_mlp_f32w_forward_backward: %d outputs (1..4)", loss->act_dim);
    TG_REQUIRE(loss->kind == 1 ? (loss->d_ret != nullptr && loss->act_dim == 1)
                               : (loss->d_act && loss->d_adv && (loss->d_logp_old || loss->d_logp_old_out) && loss->act_row_stride == loss->act_dim &&
                                  loss->act_col_stride == 1),
               "tg_mlp_f32w_forward_backward: loss inputs missing (the actions must be contiguous [rows][A])");
    if (rows == 0) return TG_OK;
    for (int l = 0; l < n_hidden_layers; ++l) {
        TG_REQUIRE((d_acts[l] || l == 0) && (d_dz[l] || l == n_hidden_layers - 1), "tg_mlp_f32w_forward_backward: buffer %d is null", l);
        a.acts[l] = (float*)d_acts[l];
        a.dz[l] = (float*)d_dz[l];
    }
    fill_f32_loss(a.loss, loss);
    return launch_f32_wide<true>(a, (hipStream_t)stream);
}

}  // extern "C"
