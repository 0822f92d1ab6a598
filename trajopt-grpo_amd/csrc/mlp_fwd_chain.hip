// Forward pass of the reference's ReLU MLP (models/neural_network.py:48-77: Linear(S,H) ReLU [Linear(H,H) ReLU]*
// Linear(H,A)) for 10^6..10^7 rows in ONE persistent launch that keeps a row's activations on chip from the input to
// the head and only WRITES them (the backward pass needs them): 64 B read + (L-1)*2H + 4*out_cols B written per row
// (2.6 KB at 20-256x5-4) against 5.2 KB for a chain of per-layer GEMMs, each of which re-reads what the previous
// one wrote.  With the HBM traffic halved the pass is bound by the matrix cores (557 kflop/row).
//
// Same machinery as fused_rollout.hip: transposed product Y^T = W . X^T with v_mfma_f32_32x32x16_bf16, 32 rows per
// wave (the MFMA columns), a layer's accumulator tile IS the next layer's B operand (bias-init, ReLU, bf16 pack), the
// weights stream L2 -> LDS by LDS-DMA through a ring shared by the 8 waves of the workgroup.  What differs:
//   * the fragment rows are permuted so that the 16 accumulator registers of lane (n, h) in output tile mt are the 16
//     CONSECUTIVE features 32 mt + 16 h + 0..15 of row n (32 B of its row), and the k-order of the next layer's A
//     fragments is 8 contiguous weights per lane (mlp.FragmentStream(layout="chain") packs them); every second tile
//     the wave transposes its 32 rows x 128 B through LDS and stores whole 128-B lines (store_pair);
//   * the input tile (32 rows x 32 padded features, bf16) also arrives by LDS-DMA, one round ahead, so that no
//     ordinary global load sits in the loop (hipcc would drain the ring with vmcnt(0) at its first use); for the
//     same reason every LDS read in the loop carries alias-scope metadata or is opaque to the compiler (see
//     bias_tile / lds_read_b128_opaque): a plain LDS read makes hipcc wait for ALL outstanding LDS-DMA;
//   * vector-memory operations retire in issue order, stores included, so the counted wait of the ring must allow
//     for the stores issued since the block it waits for.  The stores are unconditional (rows past the end are
//     clamped to the last row and rewrite it with identical bytes) and their number per block is a compile-time
//     pattern (4 after every odd tile), so every wait site has its own exact count.
#include "mfma_ring.hpp"

namespace tg {

constexpr int kChainMaxHidden = 8;
struct ChainActs { uint16_t* p[kChainMaxHidden]; uint32_t* m[kChainMaxHidden]; };   // activations, ReLU mask bits (or null)

// Accumulator start values = the tile's 32 biases (LDS table), 16 per lane half.  `__restrict__` on an inlined
// function's pointer parameters is what gives its LDS reads alias-scope metadata; hipcc makes an LDS read WITHOUT it
// wait for every outstanding LDS-DMA (vmcnt(0)), which would drain the weight ring at each block.
__device__ static inline f32x16 bias_tile(const float* __restrict__ b16) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(b16 + 4 * q);
        acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
    }
    return acc;
}

// Activation stores.  After two output tiles a lane (n, h) holds 4 x 16 B of row n's 128-B line (tile t, half h, 16-B
// piece sh at byte 64 t + 32 h + 16 sh).  Stored as they stand an instruction would write 32 rows x 32 B, and that
// pattern alone caps at 4.1 TB/s on this chip; so the wave transposes the 32 x 128 B through its LDS staging area
// (XOR-swizzled chunks: conflict-free both ways) and each of its 4 store instructions writes 8 WHOLE 128-B lines
// (5.1 TB/s for the same bytes).  `__restrict__`: see bias_tile.  Rows past the end are clamped: the lanes that
// computed them worked on the last row's input, so they rewrite the last row with identical bytes.
__device__ static inline void store_pair(uint4* __restrict__ st, uint16_t* __restrict__ g, int64_t row0, int64_t rows, int ld,
                                         int lane, bf16x8 a_lo, bf16x8 a_hi, bf16x8 b_lo, bf16x8 b_hi) {
    // staging image: 32 rows x 8 chunks of 16 B, chunk c of row n at n*8 + (c ^ ((n>>1)&7)): conflict-free b128 writes
    // (16 lanes = 16 rows, same chunk: row parity x swizzled chunk are 16 different bank groups) and reads (8 lanes per
    // row, two rows of opposite parity per 16-lane pass)
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
        // non-temporal: written once, read by a later kernel (A/B: 2.96 -> 2.91 ms per 2^22 rows)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(g + row * ld + 8 * c));
    }
}

// ReLU masks for the backward pass, 1 bit per activation (the backward-data kernels only need `activation > 0`: 32 B per
// row instead of re-reading the 512-B activation row).  `lo`/`hi` are a lane's 16 post-ReLU features of one tile
// (feature r in dword r>>1, half r&1); the result has bit k (even features 2k) and bit 16+k (odd features 2k+1), k < 8.
__device__ static inline uint32_t tile_mask_bits(bf16x8 lo, bf16x8 hi) {
    const uint4 a = __builtin_bit_cast(uint4, lo), b = __builtin_bit_cast(uint4, hi);
    // post-ReLU bf16 is +0 or positive: nonzero bits <=> positive; min(x, 1) per 16-bit half is that bit, shifted into place
    // and merged by one v_lshl_or_b32: 16 instructions per tile.  ONE asm statement: hipcc pads every statement boundary with
    // an s_nop (8 statements per tile cost as many issue slots as the arithmetic: the mask bits were 16 % of this kernel),
    // and from the C form it makes two compares + selects per dword.  Plain VALU -> VALU dependences need no wait states.
    uint32_t m, t0, t1;
    asm("v_pk_min_u16 %0, %3, %11\n\t"
        "v_pk_min_u16 %1, %4, %11\n\t"
        "v_pk_min_u16 %2, %5, %11\n\t"
        "v_lshl_or_b32 %0, %1, 1, %0\n\t"
        "v_pk_min_u16 %1, %6, %11\n\t"
        "v_lshl_or_b32 %0, %2, 2, %0\n\t"
        "v_pk_min_u16 %2, %7, %11\n\t"
        "v_lshl_or_b32 %0, %1, 3, %0\n\t"
        "v_pk_min_u16 %1, %8, %11\n\t"
        "v_lshl_or_b32 %0, %2, 4, %0\n\t"
        "v_pk_min_u16 %2, %9, %11\n\t"
        "v_lshl_or_b32 %0, %1, 5, %0\n\t"
        "v_pk_min_u16 %1, %10, %11\n\t"
        "v_lshl_or_b32 %0, %2, 6, %0\n\t"
        "v_lshl_or_b32 %0, %1, 7, %0"
        : "=&v"(m), "=&v"(t0), "=&v"(t1)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(0x00010001u));
    return m;
}

// Mask word i of lane (n, h) covers tiles 2i (bits 0-7 / 16-23) and 2i+1 (bits 8-15 / 24-31) of a layer's output `x`; a
// row's mask is [h = 0: MT/2 words][h = 1: MT/2 words] = H bits.  The words of layer l are formed while layer l+1 (or
// the head) runs its MFMAs -- `x` is that layer's input, the arithmetic hides in the matrix-core issue shadow -- and
// leave as one store per lane, 1 KiB contiguous per wave at H = 256.
template <int KS>
__device__ static inline uint32_t pair_mask_word(const bf16x8 (&x)[KS], int i) {
    return tile_mask_bits(x[4 * i], x[4 * i + 1]) | (tile_mask_bits(x[4 * i + 2], x[4 * i + 3]) << 8);
}
template <int MT>
__device__ static inline void store_mask_words(uint32_t* __restrict__ g, int64_t row, int h, const uint32_t (&w)[MT / 2]) {
    uint32_t* p = g + row * MT + h * (MT / 2);
    if constexpr (MT == 8) *reinterpret_cast<uint4*>(p) = uint4{w[0], w[1], w[2], w[3]};
    else *reinterpret_cast<uint2*>(p) = uint2{w[0], w[1]};
}

// One 32-feature output tile of a hidden layer: bias-init, K/16 MFMAs against the block's fragments, ReLU, bf16 pack.
template <int KS>
__device__ static inline void chain_tile(const uint4* __restrict__ cur, const float* __restrict__ b16, const bf16x8 (&xin)[KS],
                                         bf16x8& lo, bf16x8& hi, int lane) {
    f32x16 acc = bias_tile(b16);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[ks], acc, 0, 0, 0);
    }
    lo = relu_pack_bf16(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
    hi = relu_pack_bf16(acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]);
}

// 16 B from LDS without telling the compiler it is an LDS read (same reason); waits for it itself.
__device__ static inline uint4 lds_read_b128_opaque(const uint4* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint4*)p;
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

__device__ static inline int wave_of(unsigned tid) { return (int)(tid >> 6); }

#define TG_CHAIN_ADVANCE(WAITN) TG_RING_ADVANCE(WAITN)

// x [rows][32] bf16 (features >= in_dim zero); acts.p[l] [rows][H] bf16 for hidden layer l (kStore); out f32
// [rows][out_cols], out_cols in {8, 16}; bias f32 [(n_hh + 2)][H] (layer-major, natural feature order, head padded).
template <int H, int WPW, bool kStore, int D, bool kA0>
__global__ __launch_bounds__(64 * WPW, 2) void mlp_fwd_chain_kernel(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                                    const float* __restrict__ bias, int32_t n_hh, int64_t rows,
                                                                    ChainActs acts, float* __restrict__ out, int32_t out_cols) {
    constexpr int MT = H / 32, KS = H / 16;
    constexpr int P = D - 1;
    // counted wait for block c: all but the youngest N vector-memory operations have retired.  Behind DMA(c) there are
    // always the DMAs of the P-1 later blocks, plus the activation stores of the last P blocks: 4 per ODD output tile
    // (store_pair), none per even one -- two odd tiles lie behind an even site, one behind an odd site.  (At the layer
    // boundaries, the first-layer block with its 2 MT stores and the head with its 2-4 only ever add to that.)
    static_assert(P % 2 == 1 && P >= 3, "the store counts below are for a window of an odd number of blocks");
    constexpr int kWaitEven = (P - 1) * (KS / WPW) + (kStore ? 4 * ((P + 1) / 2) : 0);
    constexpr int kWaitOdd = (P - 1) * (KS / WPW) + (kStore ? 4 * ((P - 1) / 2) : 0);
    // kA0 == false: the first hidden activation is not stored (tg_mlp_weight_grad recomputes it, kind HR), so the
    // first-layer block has no stores behind it; the three sites whose window would count them wait for the DMAs alone
    constexpr int kWaitMin = (P - 1) * (KS / WPW);
    static_assert(KS % WPW == 0, "every wave moves the same number of 1-KiB pieces per block");
    extern __shared__ uint4 lds[];
    uint4* ring = lds;                                                  // D * KS * 64 uint4
    float* bias_s = reinterpret_cast<float*>(lds + D * KS * 64);        // (n_hh + 2) * H floats
    uint4* xs = reinterpret_cast<uint4*>(bias_s + (n_hh + 2) * H);      // WPW waves * 2 pieces * 64 uint4
    uint4* stage = xs + WPW * 128 + wave_of(threadIdx.x) * (32 * 8);    // per wave: 32 rows x 128 B (store_pair)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int64_t n_rounds = (rows + 32 * WPW - 1) / (32 * WPW);
    const int n_blocks = n_hh * MT + 2;

    for (int q = threadIdx.x; q < (n_hh + 2) * H; q += 64 * WPW) bias_s[q] = bias[q];
    __syncthreads();                                  // bias table in place; no DMA outstanding yet

    uint4* my_xs = xs + wave * 128;
    auto dma_x = [&](int64_t round) {
        // 32 rows x 64 B = two 1-KiB pieces; lane -> row 16 p + (lane >> 2), 16-B chunk lane & 3 (lands row-major)
        const int64_t base = round * (32 * WPW) + wave * 32;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int64_t r = base + 16 * p + (lane >> 2);
            r = r < rows ? r : rows - 1;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(x) + r * 4 + (lane & 3), (lds_void*)(my_xs + 64 * p), 16, 0,
                                             0);
        }
    };

    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
    dma_x(blockIdx.x);
    for (int b0 = 0; b0 < P; ++b0) {                  // blocks 0..P-1 in flight before the first round
        ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);
        pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;
        pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;
    }
    // the counted wait assumes the stores of three earlier blocks behind the block it waits for; before the first
    // block there are none, so the prologue is drained once
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t row0 = round * (32 * WPW) + wave * 32;
        int64_t row = row0 + col;
        row = row < rows ? row : rows - 1;            // clamped rows recompute and rewrite the last row (identical bytes)
        bf16x8 xin[KS], xout[KS];

        // ---- layer 0: [H x 32] . [32 x 32 rows]; one block holds all MT output tiles (2 k-steps each) ----
        {
            TG_CHAIN_ADVANCE(kWaitOdd)
            // the x tile was issued a full round ago (or in the prologue): it is older than everything the wait let pass
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xin[ks] = __builtin_bit_cast(bf16x8, lds_read_b128_opaque(my_xs + col * 4 + 2 * ks + h));
            dma_x(round + gridDim.x);                 // next round's tile (clamped past the end)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 acc = bias_tile(bias_s + 32 * mt + 16 * h);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(mt * 2 + ks) * 64 + lane]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int sh = 0; sh < 2; ++sh)
                    xout[2 * mt + sh] = relu_pack_bf16(acc[8 * sh], acc[8 * sh + 1], acc[8 * sh + 2], acc[8 * sh + 3], acc[8 * sh + 4],
                                                       acc[8 * sh + 5], acc[8 * sh + 6], acc[8 * sh + 7]);
                if (kStore && kA0 && (mt & 1))
                    store_pair(stage, acts.p[0] + 32 * (mt - 1), row0, rows, H, lane, xout[2 * mt - 2], xout[2 * mt - 1], xout[2 * mt],
                               xout[2 * mt + 1]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xin[ks] = xout[ks];
        }
        // ---- hidden H x H layers: one block per 32-feature output tile ----
        for (int l = 0; l < n_hh; ++l) {
            const float* bl = bias_s + (l + 1) * H + 16 * h;
            uint32_t mw[MT / 2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // (the mask bits of this layer's INPUT, activation l: one word per tile for the first MT/2 tiles, after the
                // barrier so that the arithmetic sits beside the tile's MFMAs)
                if (mt & 1) {
                    if (kStore && !kA0 && l == 0 && mt < 3) { TG_RING_WAIT(kWaitMin) } else { TG_RING_WAIT(kWaitOdd) }
                    TG_RING_NEXT
                    if (kStore && mt < MT / 2) mw[mt] = pair_mask_word<KS>(xin, mt);
                    chain_tile<KS>(cur, bl + 32 * mt, xin, xout[2 * mt], xout[2 * mt + 1], lane);
                    // (with the mask store one more store sits behind this tile than the wait sites count: stricter, never weaker)
                    if (kStore && mt == MT / 2 - 1 && acts.m[l]) store_mask_words<MT>(acts.m[l], row, h, mw);
                    if (kStore)
                        store_pair(stage, acts.p[l + 1] + 32 * (mt - 1), row0, rows, H, lane, xout[2 * mt - 2], xout[2 * mt - 1],
                                   xout[2 * mt], xout[2 * mt + 1]);
                } else {
                    if (kStore && !kA0 && l == 0 && mt < 3) { TG_RING_WAIT(kWaitMin) } else { TG_RING_WAIT(kWaitEven) }
                    TG_RING_NEXT
                    if (kStore && mt < MT / 2) mw[mt] = pair_mask_word<KS>(xin, mt);
                    chain_tile<KS>(cur, bl + 32 * mt, xin, xout[2 * mt], xout[2 * mt + 1], lane);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xin[ks] = xout[ks];
        }
        // ---- head: 32 padded output rows; features 0..15 are registers 0..15 of the h == 0 lanes ----
        {
            TG_CHAIN_ADVANCE(kWaitEven)
            if (kStore && acts.m[n_hh]) {                       // the last hidden activation's mask bits
                uint32_t mw[MT / 2];
#pragma unroll
                for (int i = 0; i < MT / 2; ++i) mw[i] = pair_mask_word<KS>(xin, i);
                store_mask_words<MT>(acts.m[n_hh], row, h, mw);
            }
            f32x16 acc = bias_tile(bias_s + (n_hh + 1) * H + 16 * h);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[ks], acc, 0, 0, 0);
            }
            // the h == 1 lanes (features 16..31: padding) take their partner's values and write the same bytes again
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __shfl(acc[r], col, 64);
            float* op = out + row * out_cols;
            *reinterpret_cast<float4*>(op) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(op + 4) = float4{v[4], v[5], v[6], v[7]};
            if (out_cols == 16) {
                *reinterpret_cast<float4*>(op + 8) = float4{v[8], v[9], v[10], v[11]};
                *reinterpret_cast<float4*>(op + 12) = float4{v[12], v[13], v[14], v[15]};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int H, bool kStore, int D, bool kA0>
static int chain_launch(const void* x, const void* wfrag, const float* bias, int n_hh, int64_t rows, const ChainActs& acts, float* out,
                        int out_cols, hipStream_t st) {
    constexpr int WPW = 8, KS = H / 16;
    const size_t shmem = (size_t)D * KS * 1024 + (size_t)(n_hh + 2) * H * sizeof(float) + (size_t)WPW * 2048 +
                         (size_t)WPW * 32 * 128;
    auto kern = mlp_fwd_chain_kernel<H, WPW, kStore, D, kA0>;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_forward_chain")) return rc;
    const int cus = device_cus();
    const int64_t n_rounds = ceil_div(rows, (int64_t)32 * WPW);
    const unsigned grid = (unsigned)(n_rounds < cus ? n_rounds : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPW), shmem, st, (const uint16_t*)x, (const uint4*)wfrag, bias, n_hh, rows, acts, out,
                       out_cols);
    TG_LAUNCH_CHECK("tg_mlp_forward_chain");
    return TG_OK;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_mlp_forward_chain(const void* d_x, const void* d_wfrag, const float* d_bias, int32_t hidden, int32_t n_hidden_layers,
                         int64_t rows, void* const* d_acts, void* const* d_masks, float* d_out, int32_t out_cols, void* stream) {
    TG_REQUIRE(d_x && d_wfrag && d_bias && d_out, "tg_mlp_forward_chain: null pointer");
    TG_REQUIRE(hidden == 128 || hidden == 256, "tg_mlp_forward_chain: hidden width %d unsupported (128, 256)", hidden);
    TG_REQUIRE(n_hidden_layers >= 1 && n_hidden_layers <= kChainMaxHidden, "tg_mlp_forward_chain: %d hidden layers outside 1..%d",
               n_hidden_layers, kChainMaxHidden);
    TG_REQUIRE(out_cols == 8 || out_cols == 16, "tg_mlp_forward_chain: out_cols %d must be 8 or 16", out_cols);
    TG_REQUIRE(rows >= 0, "tg_mlp_forward_chain: negative row count");
    if (rows == 0) return TG_OK;
    ChainActs acts{};
    if (d_acts)
        for (int l = 0; l < n_hidden_layers; ++l) {
            TG_REQUIRE(d_acts[l] || l == 0, "tg_mlp_forward_chain: activation buffer %d is null", l);
            acts.p[l] = (uint16_t*)d_acts[l];
            acts.m[l] = d_masks ? (uint32_t*)d_masks[l] : nullptr;
        }
    TG_REQUIRE(d_acts || !d_masks, "tg_mlp_forward_chain: mask bits are only produced together with the activations");
    hipStream_t st = (hipStream_t)stream;
    const int n_hh = n_hidden_layers - 1;
#define TG_CHAIN_ARGS d_x, d_wfrag, d_bias, n_hh, rows, acts, d_out, out_cols, st
    // ring of 4 slots, 3 blocks in flight (6 slots / 5 in flight measured the same: 2.94 vs 2.96 ms)
    const bool a0 = d_acts && d_acts[0];             // the first activation may be left out (recomputed by tg_mlp_weight_grad)
    if (hidden == 256)
        return !d_acts ? chain_launch<256, false, 4, true>(TG_CHAIN_ARGS)
                       : (a0 ? chain_launch<256, true, 4, true>(TG_CHAIN_ARGS) : chain_launch<256, true, 4, false>(TG_CHAIN_ARGS));
    return !d_acts ? chain_launch<128, false, 4, true>(TG_CHAIN_ARGS)
                   : (a0 ? chain_launch<128, true, 4, true>(TG_CHAIN_ARGS) : chain_launch<128, true, 4, false>(TG_CHAIN_ARGS));
#undef TG_CHAIN_ARGS
}

}  // extern "C"
