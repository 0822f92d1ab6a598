// Forward pass of the reference's ReLU MLP (models/neural_network.py:48-77: Linear(S,H) ReLU [Linear(H,H) ReLU]*
// Linear(H,A)) for 10^6..10^7 rows in ONE persistent launch that keeps a row's activations on chip from the input to
// the head and only WRITES them (the backward pass needs them): 64 B read + (L-1)*2H + 4*out_cols B written per row
// (2.6 KB at 20-256x5-4) against 5.2 KB for a chain of per-layer GEMMs, each of which re-reads what the previous
// one wrote.  With the HBM traffic halved the pass is bound by the matrix cores (557 kflop/row).
//
// Same machinery as fused_rollout.hip: transposed product Y^T = W . X^T on the matrix cores, 32 rows per wave (the MFMA
// columns), a layer's packed accumulators ARE the next layer's B operand (bias-init, ReLU, bf16 pack), the weights stream
// L2 -> LDS by LDS-DMA through a ring shared by the 8 waves of the workgroup.  What differs:
//   * the products are v_mfma_f32_16x16x32_bf16 (2 feature halves x 2 row tiles per 32-feature block): under the package's
//     power limit the chip holds a higher clock on this shape than on 32x32x16 (tools/mfma_shape_probe.hip: 1.53 vs 1.40
//     PFLOP/s in this loop), and this kernel is priced in joules (profiles/r02_fwd_chain_store_ablation.md);
//   * the fragment rows are permuted so that the 2 x 4 accumulator registers of lane (col, g) in output block mt are the 8
//     CONSECUTIVE features 32 mt + 8 g + 0..7 of its row (16 B of the row) -- then the next layer's k order is natural
//     (mlp.FragmentStream(layout="chain") builds the stream); every second block the wave transposes its 32 rows x 128 B
//     through LDS and stores whole 128-B lines (store_pair);
//   * the input tile (32 rows x 32 padded features, bf16) also arrives by LDS-DMA, one round ahead, so that no
//     ordinary global load sits in the loop (hipcc would drain the ring with vmcnt(0) at its first use); for the
//     same reason every LDS read in the loop carries alias-scope metadata or is opaque to the compiler (see
//     lds_float4 / lds_read_b128_opaque): a plain LDS read makes hipcc wait for ALL outstanding LDS-DMA;
//   * vector-memory operations retire in issue order, stores included, so the counted wait of the ring must allow
//     for the stores issued since the block it waits for.  The stores are unconditional (rows past the end are
//     clamped to the last row and rewrite it with identical bytes) and their number per block is a compile-time
//     pattern (4 after every odd block), so every wait site has its own exact count;
//   * the learner's training pass (tg_mlp_forward_chain_loss, template flag kHead) also carries the loss head: d loss / d output
//     per row in place of the head output, the loss sums, and the head's weight / bias gradient contracted on chip with the
//     top activation, which is then never written (see ChainLoss below).
#include "mfma_ring.hpp"

namespace tg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChainMaxHidden = 8;
TG_CLOCK_PROBE_VAR(g_probe_fwd_chain, attach_probe_fwd_chain)              // the learner's training pass (kHead)
TG_CLOCK_PROBE_VAR(g_probe_fwd_chain_plain, attach_probe_fwd_chain_plain)  // every other instantiation (no-grad passes, plain stores)
struct ChainActs { uint16_t* p[kChainMaxHidden]; uint32_t* m[kChainMaxHidden]; };   // activations, ReLU mask bits (or null)

// kHead: the loss head and the head's weight gradient inside the forward pass (tg_mlp_forward_chain_loss).  The clipped-surrogate
// / squared-error gradient of a row is a function of the head output the lane holds and of per-row inputs (loss_kernels.hip: the
// same arithmetic), so d loss / d output is known one block after the top activation: the eight waves pass that activation
// block by block through shared LDS tiles and split the product dOut^T . a_top over the OUTPUT (as mlp_bwd_chain.hip does for the
// first layer): the top activation is never written (-512 B per row) and tg_mlp_weight_grad has no DH job (-528 B per row).
struct ChainLoss {                          // (kept small: every field is a scalar register for the whole kernel)
    int32_t kind, A;                        // 0: actor (clipped surrogate + KL-ish penalty), 1: critic (squared error); A <= 4 outputs
                                            // 2: actor whose old policy is the current one: `logp_old` is WRITTEN (the row's own
                                            //    log-probability, which is then also its old one: ratio exactly 1), not read
    const float* act;                       // actor: [rows][A] contiguous;  critic: the returns [rows]
    const float* logp_old; const float* adv;
    float n_m, n_i;                         // normalisation of the advantage (actor) / return (critic): (x - n_m) * n_i
    float inv_var[4]; float logp_const, epsilon, surr_coef, critic_coef, kl_coef;
    const float* norm8;                     // non-null: n_m / n_i and the three coefficients come from this device f32 [8] (tg_ppo_norm's
                                            // output: PPO's normalisation constants and 1 / n never visit the host), read once at entry
    uint16_t* dout8;                        // out: d loss / d head output, bf16 [rows][8], zero padded (tg_mlp_backward_chain's input)
    float* head_slabs;                      // out: f32 [grid][4][16][H] partial head weight gradients
    double* work;                           // out: f64 [grid][4] partial loss sums (surrogate, squared error, KL, count)
    float* bias_partial;                    // out: f32 [grid][4] partial head bias gradients
};
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
__device__ static inline bf16x8 tr_frag16(const char* __restrict__ lo_p, const char* __restrict__ hi_p) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)lo_p);
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)hi_p);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ static inline void lds_store16(char* __restrict__ p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ static inline float lds_loadf(const float* __restrict__ p) { return *p; }
__device__ static inline double2 lds_loadd2(const double* __restrict__ p) { return *reinterpret_cast<const double2*>(p); }
__device__ static inline void lds_stored2(double* __restrict__ p, double2 v) { *reinterpret_cast<double2*>(p) = v; }

// 16 B from LDS without telling the compiler it is an LDS read: hipcc makes an LDS read WITHOUT alias-scope metadata wait for
// every outstanding LDS-DMA (vmcnt(0)), which would drain the weight ring at each block; waits for the read itself.
__device__ static inline uint4 lds_read_b128_opaque(const uint4* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint4*)p;
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// (`__restrict__` on an inlined function's pointer parameters is the other way to give its LDS reads that metadata)
__device__ static inline float4 lds_float4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// Activation stores.  After two 32-feature blocks a lane (col, g) holds, for each of its two rows (16 c + col), 2 x 16 B of the
// row's 128-B line (block b of the pair, bytes 64 b + 16 g).  Stored as they stand an instruction would write 16-B pieces;
// the wave transposes the 32 rows x 128 B through its LDS staging area (chunks XOR-swizzled by the row: conflict-free
// both ways) and each of its 4 store instructions writes 8 WHOLE 128-B lines (5.1 TB/s against 4.1 for 32-B pieces).
// Rows past the end are clamped: the lanes that computed them worked on the last row's input, so they rewrite the last row
// with identical bytes.
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
        // non-temporal: written once, read by a later kernel (A/B: 2.96 -> 2.91 ms per 2^22 rows)
        act_store16(act_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<act_u32x4*>(g + row * ld + 8 * ch));
    }
}

// ReLU masks for the backward pass, 1 bit per activation (the backward-data kernels only need `activation > 0`: 32 B per
// row instead of re-reading the 512-B activation row).  Memory layout per row: [half h = 0, 1][H / 64 words]; feature
// 32 mt + 16 h + r is bit (mt & 1) * 8 + (r >> 1) + 16 * (r & 1) of word mt >> 1 of half h.  Lane (col, g) holds the features
// 32 mt + 8 g + e (e = 0..7) of its two rows, i.e. h = g >> 1, r = 8 (g & 1) + e: dword d of a packed block holds e = 2 d (low
// half) and 2 d + 1 (high half), so min(x, 1) per 16-bit half (post-ReLU bf16 is +0 or positive: nonzero bits <=> positive)
// shifted left by (mt & 1) * 8 + 4 (g & 1) + d lands both bits.  `block_bits` leaves out the lane's 4 (g & 1).
// ONE asm statement per block: hipcc pads every statement boundary with an s_nop, and from the C form it makes two compares
// + selects per dword.  Plain VALU -> VALU dependences need no wait states.
__device__ static inline uint32_t block_bits(bf16x8 x) {
    const uint4 a = __builtin_bit_cast(uint4, x);
#if TG_ABLATE_CHAIN_VALU & 1
    return a.x;                                              // (probe build: the words are garbage, no instruction spent)
#endif
    uint32_t m, t0;
    asm("v_pk_min_u16 %0, %2, %6\n\t"
        "v_pk_min_u16 %1, %3, %6\n\t"
        "v_lshl_or_b32 %0, %1, 1, %0\n\t"
        "v_pk_min_u16 %1, %4, %6\n\t"
        "v_lshl_or_b32 %0, %1, 2, %0\n\t"
        "v_pk_min_u16 %1, %5, %6\n\t"
        "v_lshl_or_b32 %0, %1, 3, %0"
        : "=&v"(m), "=&v"(t0)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(0x00010001u));
    return m;
}
// word w of a lane's half-row: blocks 2 w (bits 0-3 / 16-19, + the lane's nibble) and 2 w + 1 (bits 8-11 / 24-27)
__device__ static inline uint32_t pair_mask_word(const bf16x8* x, int w, int nibble) {
    return (block_bits(x[2 * w]) | (block_bits(x[2 * w + 1]) << 8)) << nibble;
}
// The two lanes (g, g ^ 1) of a row own interleaved nibbles of the same words: one exchange (lane ^ 16, no LDS access), one OR,
// and the even lane stores the half-row's MT / 2 words.
template <int MT>
__device__ static inline void store_mask_words(uint32_t* __restrict__ g, int64_t row, int grp, const uint32_t (&w)[MT / 2]) {
#if TG_ABLATE_FUSED_CHAIN
    asm volatile("" ::"v"(w[0]), "v"(w[MT / 2 - 1]));          // (the words are still formed: a fused kernel needs them too)
    return;
#endif
    uint32_t full[MT / 2];
#pragma unroll
    for (int i = 0; i < MT / 2; ++i) full[i] = w[i] | (uint32_t)__builtin_amdgcn_ds_swizzle((int)w[i], 0x401F);   // xor 0x10
    if ((grp & 1) == 0) {
        int half_off = (grp >> 1) * (MT / 2);
        asm volatile("" : "+v"(half_off));        // (rebuilt at every use: hoisted, the 64-bit lane address is a register pair the
                                                  // fused-head variant does not have -- it spilled, and a spill reload drains the ring)
        uint32_t* p = g + row * MT + half_off;
        if constexpr (MT == 8) *reinterpret_cast<uint4*>(p) = uint4{full[0], full[1], full[2], full[3]};
        else *reinterpret_cast<uint2*>(p) = uint2{full[0], full[1]};
    }
}

// One 32-feature output block of a hidden layer for the wave's 2 x 16 rows: bias-init, H / 32 k-steps of 2 x 2
// v_mfma_f32_16x16x32_bf16 against the block's fragments, ReLU, bf16 pack.  (This shape, not 32x32x16: under the package's
// power limit the chip holds a higher clock on it -- 1.53 vs 1.40 PFLOP/s in this very loop, tools/mfma_shape_probe.hip.)
template <int K8>
__device__ static inline void chain_block(const uint4* __restrict__ cur, const float* __restrict__ b8, const bf16x8 (&xin)[2][K8],
                                          bf16x8 (&out)[2], int lane) {
    f32x4 acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float4 b4 = lds_float4(b8 + 4 * f);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[f][c] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    // s_setprio doubles as a scheduling fence around the block's products (hipcc otherwise threads the neighbouring blocks' pack
    // arithmetic through the MFMA sequence): -6 % on the pass without stores, neutral with them (same-box A/B)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < K8; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(ks * 2 + f) * 64 + lane]);
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xin[c][ks], acc[f][c], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
        out[c] = relu_pack_bf16(acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3], acc[1][c][0], acc[1][c][1], acc[1][c][2],
                                acc[1][c][3]);
}

__device__ static inline int wave_of(unsigned tid) { return (int)(tid >> 6); }

#define TG_CHAIN_ADVANCE(WAITN) TG_RING_ADVANCE(WAITN)

// x [rows][32] bf16 (features >= in_dim zero); acts.p[l] [rows][H] bf16 for hidden layer l (kStore); out f32
// [rows][out_cols], out_cols in {4, 8, 16}; bias f32 [(n_hh + 2)][H] (layer-major, natural feature order, head padded).
// A wave owns 32 rows as two 16-row MFMA column tiles (c = 0, 1): lane (col = lane & 15, g = lane >> 4) holds, per row
// 16 c + col and 32-feature block, the 8 consecutive features 8 g .. 8 g + 7 -- as accumulators (4 of half f = 0, 4 of f = 1:
// mlp.FragmentStream arranges the weight rows so), then packed: 16 B that are both the next layer's B operand for k-step =
// block (natural k order) and a contiguous piece of the activation row.
template <int H, int WPW, bool kStore, int D, bool kA0, bool kHead = false>
__global__ __launch_bounds__(64 * WPW, 2) void mlp_fwd_chain_kernel(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                                    const float* __restrict__ bias, int32_t n_hh, int64_t rows,
                                                                    ChainActs acts, float* __restrict__ out, int32_t out_cols,
                                                                    ChainLoss L) {
    static_assert(!kHead || (kStore && !kA0), "the fused head belongs to the learner's training pass");
    if constexpr (kHead) {
        if (L.norm8 != nullptr) {                                       // (uniform scalar loads, before anything is in flight)
            const int q = L.kind == 1 ? 2 : 0;
            L.n_m = L.norm8[q]; L.n_i = L.norm8[q + 1];
            L.surr_coef = L.norm8[4]; L.critic_coef = L.norm8[5]; L.kl_coef = L.norm8[6];
        }
    }
    constexpr int MT = H / 32, KS = H / 16, K8 = H / 32;               // blocks per layer, 1-KiB pieces per block, k-steps per block
    constexpr int P = D - 1;
    // counted wait for block c: all but the youngest N vector-memory operations have retired.  Behind DMA(c) there are
    // always the DMAs of the P-1 later blocks, plus the activation stores of the last P blocks: 4 per ODD output block
    // (store_pair), none per even one -- two odd blocks lie behind an even site, one behind an odd site.  (At the layer
    // boundaries, the first-layer block with its 2 MT stores and the head with its 1-2 only ever add to that.)
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
    // kHead: behind the staging area (which the head phase reuses as the shared tiles T[2][WPW][2 KiB]): per wave 1 KiB of d loss /
    // d output rows ([32][16] bf16), 1 KiB of per-row loss inputs ([8 fields][32 rows] f32), 16 lanes x (4 f64 + 4 f32) of sums
    // (addresses are rebuilt where they are used: the kernel has no registers to keep them in)
#define TG_HEAD_LDS                                                                                                              \
    char* tiles = reinterpret_cast<char*>(xs + WPW * 128);                                                                       \
    char* dtiles = tiles + WPW * 32 * 128;                                                                                       \
    [[maybe_unused]] float* lin = reinterpret_cast<float*>(dtiles + WPW * 1024) + wave_of(threadIdx.x) * 256;                    \
    [[maybe_unused]] double* sums_d = reinterpret_cast<double*>(dtiles + 2 * WPW * 1024) + (wave_of(threadIdx.x) * 16 + (threadIdx.x & 15)) * 4; \
    [[maybe_unused]] float* sums_f = reinterpret_cast<float*>(dtiles + 2 * WPW * 1024 + WPW * 16 * 32) + (wave_of(threadIdx.x) * 16 + (threadIdx.x & 15)) * 4; \
    [[maybe_unused]] float* hacc = reinterpret_cast<float*>(dtiles + 2 * WPW * 1024 + WPW * 16 * 48) + (wave_of(threadIdx.x) * MT * 16 + (threadIdx.x & 15)) * 4;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, col = lane & 15;
    const int64_t n_rounds = (rows + 32 * WPW - 1) / (32 * WPW);
    const int n_blocks = n_hh * MT + 2;
    if constexpr (kHead) { TG_CLOCK_PROBE_BEGIN(g_probe_fwd_chain) } else { TG_CLOCK_PROBE_BEGIN(g_probe_fwd_chain_plain) }

    for (int q = threadIdx.x; q < (n_hh + 2) * H; q += 64 * WPW) bias_s[q] = bias[q];
    if constexpr (kHead) {
        TG_HEAD_LDS
        if (grp == 0) {
#pragma unroll
            for (int b = 0; b < MT; ++b) *reinterpret_cast<float4*>(hacc + b * 64) = float4{0.f, 0.f, 0.f, 0.f};
            lds_stored2(sums_d, double2{0.0, 0.0});
            lds_stored2(sums_d + 2, double2{0.0, 0.0});
            *reinterpret_cast<float4*>(sums_f) = float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();                                  // bias table in place; no DMA outstanding yet

    uint4* my_xs = xs + wave * 128;
    auto dma_x = [&](int64_t round) {
        // 32 rows x 64 B = two 1-KiB pieces; lane -> row 16 p + (lane >> 2), 16-B chunk lane & 3 (lands row-major)
        const int64_t base = round * (32 * WPW) + wave * 32;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int64_t r = base + 16 * p + (lane >> 2);
            r = r < rows ? r : rows - 1; r = mem_row(r);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(x) + r * 4 + (lane & 3), (lds_void*)(my_xs + 64 * p), 16, 0,
                                             0);
        }
    };

    // kHead: the round's per-row loss inputs, one float per lane and field (lanes 0..31 = the wave's rows), to lin[field][row]
    [[maybe_unused]] auto dma_loss_inputs = [&](int64_t round) {
        TG_HEAD_LDS
        if (lane < 32) {
            int64_t r = round * (32 * WPW) + wave * 32 + lane;
            r = r < rows ? r : rows - 1; r = mem_row(r);
            if (L.kind != 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < L.A) __builtin_amdgcn_global_load_lds(L.act + r * L.A + k, (lds_void*)(lin + 32 * k), 4, 0, 0);
                // (kind 2 loads the slot it is about to write: never used, but every round issues the same number of DMAs)
                __builtin_amdgcn_global_load_lds(L.logp_old + r, (lds_void*)(lin + 32 * 4), 4, 0, 0);
                __builtin_amdgcn_global_load_lds(L.adv + r, (lds_void*)(lin + 32 * 5), 4, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds(L.act + r, (lds_void*)(lin + 32 * 0), 4, 0, 0);
            }
        }
    };
    // kHead: this wave's tiles of the head's weight gradient live in LDS, not in registers (32 more registers spill, and a spill
    // reload is a vector-memory operation: hipcc then drains the weight ring around it): only outputs 0..3 can be non-zero, i.e.
    // the 16 lanes g == 0 of every 16 x 16 tile: hacc[wave][block][lane] float4, read-modify-written by its one owner lane
    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
    dma_x(blockIdx.x);
    if constexpr (kHead) dma_loss_inputs(blockIdx.x);
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
        int64_t rowc[2];                              // clamped rows recompute and rewrite the last row (identical bytes)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            rowc[c] = row0 + 16 * c + col;
            rowc[c] = rowc[c] < rows ? rowc[c] : rows - 1; rowc[c] = mem_row(rowc[c]);
        }
        bf16x8 xin[2][K8], xout[2][K8];

        // ---- layer 0: [H x 32] . [32 x 32 rows]; one block holds all MT output blocks (one k-step, 2 halves each) ----
        {
            // (kHead: the top layer of the previous round stored nothing: count the DMAs alone)
            if constexpr (kHead) { TG_RING_WAIT(kWaitMin) } else { TG_RING_WAIT(kWaitOdd) }
            TG_RING_NEXT
            // the x tile was issued a full round ago (or in the prologue): it is older than everything the wait let pass
            bf16x8 x0[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) x0[c] = __builtin_bit_cast(bf16x8, lds_read_b128_opaque(my_xs + (16 * c + col) * 4 + grp));
            dma_x(round + gridDim.x);                 // next round's tile (clamped past the end)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float4 b4 = lds_float4(bias_s + 32 * mt + 8 * grp + 4 * f);
                    const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(mt * 2 + f) * 64 + lane]);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x0[c], f32x4{b4.x, b4.y, b4.z, b4.w}, 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    xout[c][mt] = relu_pack_bf16(acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3], acc[1][c][0], acc[1][c][1],
                                                 acc[1][c][2], acc[1][c][3]);
                if (kStore && kA0 && (mt & 1)) {
                    const bf16x8 pa[2] = {xout[0][mt - 1], xout[1][mt - 1]}, pb[2] = {xout[0][mt], xout[1][mt]};
                    store_pair(stage, acts.p[0] + 32 * (mt - 1), row0, rows, H, lane, pa, pb);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < K8; ++ks) xin[c][ks] = xout[c][ks];
        }
        // ---- hidden H x H layers: one block per 32 output features ----
        for (int l = 0; l < n_hh; ++l) {
            const float* bl = bias_s + (l + 1) * H + 8 * grp;
            uint32_t mw[2][MT / 2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // (the mask bits of this layer's INPUT, activation l: one word per row tile and block for the first MT blocks'
                // worth of words, after the barrier so that the arithmetic sits beside the block's MFMAs)
                // (kHead: the top layer stores nothing: from its fourth block on only DMAs lie behind a block, behind its third
                // one stored pair of the layer below)
                const bool top_layer = kHead && l == n_hh - 1;
                if ((kStore && !kA0 && l == 0 && mt < 3) || (top_layer && mt >= 3)) {
                    TG_RING_WAIT(kWaitMin)
                } else if ((mt & 1) || (top_layer && mt == 2)) {
                    TG_RING_WAIT(kWaitOdd)
                } else {
                    TG_RING_WAIT(kWaitEven)
                }
                TG_RING_NEXT
                if (kStore) {                                          // MT words per lane and layer, one per block
                    const int q = mt;                                  // (c, word) = (q / (MT/2), q % (MT/2))
                    mw[q / (MT / 2)][q % (MT / 2)] = pair_mask_word(xin[q / (MT / 2)], q % (MT / 2), 4 * (grp & 1));
                }
                bf16x8 o[2];
                chain_block<K8>(cur, bl + 32 * mt, xin, o, lane);
                xout[0][mt] = o[0];
                xout[1][mt] = o[1];
                if (mt & 1) {
                    // (with the mask stores more stores sit behind this block than the wait sites count: stricter, never weaker)
                    if (kStore && mt == MT - 1 && acts.m[l]) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) store_mask_words<MT>(acts.m[l], rowc[c], grp, mw[c]);
                    }
                    if (kStore && !top_layer) {
                        const bf16x8 pa[2] = {xout[0][mt - 1], xout[1][mt - 1]}, pb[2] = {xout[0][mt], xout[1][mt]};
                        store_pair(stage, acts.p[l + 1] + 32 * (mt - 1), row0, rows, H, lane, pa, pb);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < K8; ++ks) xin[c][ks] = xout[c][ks];
        }
        // ---- head: <= 16 outputs in half 0 of the block, natural order: register r of lane group g is output 4 g + r ----
        {
            if constexpr (kHead) { TG_RING_WAIT(kWaitMin) } else { TG_RING_WAIT(kWaitEven) }
            TG_RING_NEXT
            if (kStore && acts.m[n_hh]) {                       // the last hidden activation's mask bits
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint32_t mw[MT / 2];
#pragma unroll
                    for (int i = 0; i < MT / 2; ++i) mw[i] = pair_mask_word(xin[c], i, 4 * (grp & 1));
                    store_mask_words<MT>(acts.m[n_hh], rowc[c], grp, mw);
                }
            }
            const float4 b4 = lds_float4(bias_s + (n_hh + 1) * H + 4 * grp);
            f32x4 acc[2] = {f32x4{b4.x, b4.y, b4.z, b4.w}, f32x4{b4.x, b4.y, b4.z, b4.w}};
#pragma unroll
            for (int ks = 0; ks < K8; ++ks) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(ks * 2) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xin[c][ks], acc[c], 0, 0, 0);
            }
            if (out != nullptr && 4 * grp < out_cols) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    *reinterpret_cast<float4*>(out + rowc[c] * out_cols + 4 * grp) = float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
            }
            if constexpr (kHead) {
                TG_HEAD_LDS
                // ---- the loss head (loss_kernels.hip::surrogate_loss_kernel, same arithmetic): the g == 0 lanes hold outputs 0..3 of
                // their two rows; d loss / d output goes to dout8 (for the backward chain) and, padded to 16 columns, to this
                // wave's LDS tile (the A operand of the head's weight gradient) ----
                char* dt = dtiles + wave * 1024;
                if (grp == 0) {
                    // (few values live at a time: the kernel has no registers to spare, and a spill here costs the whole ring)
#pragma unroll 1
                    for (int c = 0; c < 2; ++c) {
                        float c_surr = 0.f, c_crit = 0.f, c_kl = 0.f, c_cnt = 0.f;
                        const int rl = 16 * c + col;
                        const bool valid = row0 + rl < rows;              // clamped duplicates of the last row contribute nothing
                        float g[4] = {0.f, 0.f, 0.f, 0.f};
                        if (valid) {
                            if (L.kind != 1) {
                                float quad = 0.f, dmu[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {              // (inv_var is 0 beyond the net's outputs; the tile slots hold zeros)
                                    const float d = (k < L.A ? lds_loadf(lin + 32 * k + rl) : 0.f) - acc[c][k];
                                    dmu[k] = d;
                                    quad += d * d * L.inv_var[k];
                                }
                                const float lp = -0.5f * quad + L.logp_const;
                                float lpo = lds_loadf(lin + 32 * 4 + rl);
                                if (L.kind == 2) {
                                    lpo = lp;
                                    const_cast<float*>(L.logp_old)[rowc[c]] = lp;      // (valid rows only: inside `if (valid)`)
                                }
                                const float adv = (lds_loadf(lin + 32 * 5 + rl) - L.n_m) * L.n_i;
                                const float rho = expf(lp - lpo);
                                const float lo = 1.0f - L.epsilon, hi = 1.0f + L.epsilon;
                                const float surr1 = rho * adv, surr2 = fminf(fmaxf(rho, lo), hi) * adv;
                                const bool inside = (rho >= lo) && (rho <= hi);
                                const float w = inside ? 1.0f : (surr1 < surr2 ? 1.0f : 0.0f);
                                c_surr = fminf(surr1, surr2);
                                float dlp = L.surr_coef * adv * rho * w;
                                if (L.kl_coef != 0.0f) {
                                    const float eo = expf(lpo);
                                    c_kl = eo * (lpo - lp);
                                    dlp -= L.kl_coef * eo;
                                }
#pragma unroll
                                for (int k = 0; k < 4; ++k) g[k] = dlp * dmu[k] * L.inv_var[k];
                            } else {
                                const float d = acc[c][0] - (lds_loadf(lin + rl) - L.n_m) * L.n_i;
                                c_crit = d * d;
                                g[0] = L.critic_coef * 2.0f * d;
                            }
                            c_cnt = 1.0f;
                        }
                        {
                            double2 t01 = lds_loadd2(sums_d);
                            t01.x += (double)c_surr; t01.y += (double)c_crit;
                            lds_stored2(sums_d, t01);
                            double2 t23 = lds_loadd2(sums_d + 2);
                            t23.x += (double)c_kl; t23.y += (double)c_cnt;
                            lds_stored2(sums_d + 2, t23);
                            float4 gs = lds_float4(sums_f);
                            gs.x += g[0]; gs.y += g[1]; gs.z += g[2]; gs.w += g[3];
                            lds_store16(reinterpret_cast<char*>(sums_f), __builtin_bit_cast(uint4, gs));
                        }
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const uint4 o = {__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{g[0], g[1]}, bf16x2)),
                                         __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{g[2], g[3]}, bf16x2)), 0u, 0u};
#if !TG_ABLATE_FUSED_CHAIN
                        if (valid) *reinterpret_cast<uint4*>(L.dout8 + rowc[c] * 8) = o;     // (a clamped duplicate must not zero the last row)
#endif
                        lds_store16(dt + rl * 32, o);
                        lds_store16(dt + rl * 32 + 16, uint4{0u, 0u, 0u, 0u});
                    }
                }
                // ---- the head's weight gradient: the top activation passes through the shared tiles block by block; wave i takes
                // the 16 features (i & 1) of the block over the rows of waves 2 (i >> 1), 2 (i >> 1) + 1 ----
                const int q4 = (lane >> 2) & 3, p4 = lane & 3;
                const int d_lo = (2 * grp) * 128 + q4 * 32 + p4 * 8, d_hi = (2 * grp + 1) * 128 + q4 * 32 + p4 * 8;
                const int nt = wave & 1;
                const int t_lo = (2 * grp) * 256 + q4 * 64 + (((2 * nt + (p4 >> 1)) ^ ((2 * grp) & 3)) * 16) + (p4 & 1) * 8;
                const int t_hi = (2 * grp + 1) * 256 + q4 * 64 + (((2 * nt + (p4 >> 1)) ^ ((2 * grp + 1) & 3)) * 16) + (p4 & 1) * 8;
                // tile b + 1 is written right behind barrier b (its buffer was last read for block b - 1, and every wave finished
                // those reads before it arrived at barrier b): the write's latency passes under block b's products
                auto write_tile = [&](int b) {
                    char* tw = tiles + ((b & 1) * WPW + wave) * 2048;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int rl = 16 * c + col;
                        lds_store16(tw + (rl >> 2) * 256 + (rl & 3) * 64 + ((grp ^ ((rl >> 2) & 3)) * 16), __builtin_bit_cast(uint4, xin[c][b]));
                    }
                };
#if TG_ABLATE_HEAD_RELAY != 1                                  /* probe build 1: no relay at all (timing only: the head's weight gradient is not formed) */
                write_tile(0);
#pragma unroll
                for (int b = 0; b < MT; ++b) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if TG_ABLATE_HEAD_RELAY != 2                                  /* probe build 2: the relay without its eight workgroup barriers (racy: timing only) */
                    __builtin_amdgcn_s_barrier();
#endif
                    asm volatile("" ::: "memory");
                    if (b + 1 < MT) write_tile(b + 1);
                    f32x4 t = {};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int v = 2 * (wave >> 1) + ks;
                        const char* db = dtiles + v * 1024;
                        const char* tb = tiles + ((b & 1) * WPW + v) * 2048;
                        const bf16x8 fa = tr_frag16(db + d_lo, db + d_hi), fb = tr_frag16(tb + t_lo, tb + t_hi);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, t, 0, 0, 0);
                    }
                    if (grp == 0) {
                        const float4 h = lds_float4(hacc + b * 64);
                        lds_store16(reinterpret_cast<char*>(hacc + b * 64),
                                    __builtin_bit_cast(uint4, float4{h.x + t[0], h.y + t[1], h.z + t[2], h.w + t[3]}));
                    }
                }
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next round's loss inputs overwrite what was read above)
                dma_loss_inputs(round + gridDim.x);
            }
        }
    }
    if constexpr (kHead) {
        TG_HEAD_LDS
        // partial head weight gradient: slab [workgroup][K quarter][16 outputs][H]; outputs 4 g + r (only < 8 are ever non-zero)
        float* slab = L.head_slabs + ((int64_t)blockIdx.x * 4 + (wave >> 1)) * 16 * H;
        if (grp == 0) {
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const float4 h = lds_float4(hacc + b * 64);
                const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[r * H + 32 * b + 16 * (wave & 1) + col] = hv[r];
            }
        }
        // loss sums and head bias sums: the 16 g == 0 lanes of every wave, added in a fixed order
        __syncthreads();
        if (threadIdx.x < 8) {
            const int j = threadIdx.x & 3;
            if (threadIdx.x < 4) {
                double t = 0.0;
                const double* sd = reinterpret_cast<const double*>(dtiles + 2 * WPW * 1024);
                for (int q = 0; q < WPW * 16; ++q) t += sd[q * 4 + j];
                L.work[(int64_t)blockIdx.x * 4 + j] = t;
            } else {
                float t = 0.f;
                const float* sf = reinterpret_cast<const float*>(dtiles + 2 * WPW * 1024 + WPW * 16 * 32);
                for (int q = 0; q < WPW * 16; ++q) t += sf[q * 4 + j];
                L.bias_partial[(int64_t)blockIdx.x * 4 + j] = t;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
    if constexpr (kHead) { TG_CLOCK_PROBE_END(g_probe_fwd_chain) } else { TG_CLOCK_PROBE_END(g_probe_fwd_chain_plain) }
}

template <int H, bool kStore, int D, bool kA0, bool kHead = false>
static int chain_launch(const void* x, const void* wfrag, const float* bias, int n_hh, int64_t rows, const ChainActs& acts, float* out,
                        int out_cols, hipStream_t st, const ChainLoss& loss = ChainLoss{}) {
    constexpr int WPW = 8, KS = H / 16;
    const size_t shmem = (size_t)D * KS * 1024 + (size_t)(n_hh + 2) * H * sizeof(float) + (size_t)WPW * 2048 +
                         (size_t)WPW * 32 * 128 + (kHead ? (size_t)2 * WPW * 1024 + (size_t)WPW * 16 * 48 + (size_t)WPW * (H / 32) * 16 * 16 : 0);
    auto kern = mlp_fwd_chain_kernel<H, WPW, kStore, D, kA0, kHead>;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_forward_chain")) return rc;
    const int cus = device_cus();
    const int64_t n_rounds = ceil_div(rows, (int64_t)32 * WPW);
    const unsigned grid = (unsigned)(n_rounds < cus ? n_rounds : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPW), shmem, st, (const uint16_t*)x, (const uint4*)wfrag, bias, n_hh, rows, acts, out,
                       out_cols, loss);
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
    TG_REQUIRE(out_cols == 4 || out_cols == 8 || out_cols == 16, "tg_mlp_forward_chain: out_cols %d must be 4, 8 or 16", out_cols);
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

int tg_mlp_forward_chain_blocks(void) { return device_cus(); }

int tg_mlp_forward_chain_loss(const void* d_x, const void* d_wfrag, const float* d_bias, int32_t hidden, int32_t n_hidden_layers,
                              int64_t rows, void* const* d_acts, void* const* d_masks, const tg_chain_loss* loss, void* stream) {
    TG_REQUIRE(d_x && d_wfrag && d_bias && d_acts && d_masks && loss, "tg_mlp_forward_chain_loss: null pointer");
    TG_REQUIRE(hidden == 128 || hidden == 256, "tg_mlp_forward_chain_loss: hidden width %d unsupported (128, 256)", hidden);
    TG_REQUIRE(n_hidden_layers >= 3 && n_hidden_layers <= kChainMaxHidden, "tg_mlp_forward_chain_loss: %d hidden layers outside 3..%d",
               n_hidden_layers, kChainMaxHidden);
    TG_REQUIRE(loss->kind == 0 || loss->kind == 1, "tg_mlp_forward_chain_loss: kind %d", loss->kind);
    TG_REQUIRE(loss->act_dim >= 1 && loss->act_dim <= 4, "tg_mlp_forward_chain_loss: %d outputs unsupported (1..4)", loss->act_dim);
    TG_REQUIRE(loss->d_dout8 && loss->d_head_slabs && loss->d_work && loss->d_bias_partial, "tg_mlp_forward_chain_loss: null output");
    TG_REQUIRE(loss->kind == 1 ? loss->d_ret != nullptr : (loss->d_act && (loss->d_logp_old || loss->d_logp_old_out) && loss->d_adv),
               "tg_mlp_forward_chain_loss: missing per-row input");
    TG_REQUIRE(loss->kind == 1 || (loss->act_col_stride == 1 && loss->act_row_stride == loss->act_dim),
               "tg_mlp_forward_chain_loss: the actions must be contiguous [rows][act_dim]");

    TG_REQUIRE(rows > 0, "tg_mlp_forward_chain_loss: no rows");
    ChainActs acts{};
    for (int l = 0; l < n_hidden_layers; ++l) {
        TG_REQUIRE(d_masks[l] && (d_acts[l] || l == 0 || l == n_hidden_layers - 1), "tg_mlp_forward_chain_loss: buffer %d is null", l);
        acts.p[l] = (uint16_t*)d_acts[l];
        acts.m[l] = (uint32_t*)d_masks[l];
    }
    TG_REQUIRE(!d_acts[0], "tg_mlp_forward_chain_loss: the first activation is recomputed by tg_mlp_weight_grad, not stored");
    ChainLoss L{};
    L.kind = loss->kind; L.A = loss->act_dim;
    L.act = loss->kind == 0 ? loss->d_act : loss->d_ret;
    L.logp_old = loss->d_logp_old; L.adv = loss->d_adv;
    if (loss->kind == 0 && loss->d_logp_old_out != nullptr) {      // the old policy is the current one: this pass writes the old log-probabilities
        L.kind = 2;
        L.logp_old = loss->d_logp_old_out;
    }
    L.n_m = loss->norm_mean; L.n_i = loss->norm_inv; L.norm8 = loss->d_norm8;
    float logdet = 0.f;
    for (int k = 0; k < 4; ++k) {
        L.inv_var[k] = k < loss->act_dim ? 1.0f / loss->var[k] : 0.f;
        if (k < loss->act_dim) logdet += logf(loss->var[k]);
    }
    L.logp_const = -0.5f * (float)loss->act_dim * 1.8378770664093453f - 0.5f * logdet;
    L.epsilon = loss->epsilon; L.surr_coef = loss->surr_coef; L.critic_coef = loss->critic_coef; L.kl_coef = loss->kl_coef;
    L.dout8 = (uint16_t*)loss->d_dout8; L.head_slabs = loss->d_head_slabs; L.work = loss->d_work; L.bias_partial = loss->d_bias_partial;
    hipStream_t st = (hipStream_t)stream;
    const int n_hh = n_hidden_layers - 1;
    return hidden == 256 ? chain_launch<256, true, 4, false, true>(d_x, d_wfrag, d_bias, n_hh, rows, acts, nullptr, 8, st, L)
                         : chain_launch<128, true, 4, false, true>(d_x, d_wfrag, d_bias, n_hh, rows, acts, nullptr, 8, st, L);
}

}  // extern "C"
