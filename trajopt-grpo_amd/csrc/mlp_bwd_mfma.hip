// Backward-data product of a hidden Linear layer fused with the ReLU backward and the bias gradient of
// the layer below it (bf16 operands, fp32 accumulate):
//
//     dZ_below[r][m] = ( sum_k dZ[r][k] * W[k][m] ) * (Act[r][m] > 0)          W = Linear.weight, [K = out][M = in]
//     partial[wg][m] = sum over the workgroup's rows of dZ_below[r][m]          (bias gradient, reduced on the host side)
//
// Under torch autograd (the reference, algorithms/*.py `loss.backward()` through models/neural_network.py:48-66)
// this is a GEMM that writes dA (512 B/row at M = 256) followed by a pass that re-reads dA and Act and writes dZ
// (1536 B/row): 2.5 KB/row of HBM traffic for a layer.  Here W (<= 128 KiB) sits in LDS for the whole launch and a
// wave streams 32-row tiles (NT = 2 MFMA column tiles of 16 rows) past it: read dZ (2K B/row) and Act (2M B/row), write dZ_below (2M B/row) = 1.5 KB/row,
// the floor for this step.  HBM-bound: 137 GFLOP per 2^20 rows against 1.5 GB.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, TRANSPOSED product  D^T[m][n] = sum_k W^T[m][k] * dZ^T[k][n]):
//   * B operand = the tile's rows straight from global memory: lane (n = lane & 15, q = lane >> 4) holds
//     dZ[row n][32 ks + 8 q + j], j < 8 -- 16 B per lane, 64 contiguous bytes per row and instruction, no LDS
//     staging, every byte read once;
//   * A operand = W^T from LDS in fragment order (one conflict-free ds_read_b128 per MFMA, shared by the NT row
//     tiles of the wave).  The order of the output features inside a PAIR of 16-feature m-blocks is free, so
//     fragment row m of block 2p+e carries feature 32p + 8(m>>2) + 4e + (m&3): the 2 x 4 accumulator registers
//     of lane (n, q) are then the 8 CONSECUTIVE features 32p + 8q + 0..7 of row n, and the Act load and the
//     dZ_below store of a pair are one 16 B access per lane (64 contiguous bytes per row).
//   The 16x16 shape (4 accumulators per block) is what keeps the per-lane bias-gradient sums at M/4 registers;
//   the 32x32 shape needs M/2 and spills at M = 256.
// `tg_dx_pack_weights` builds the fragment array from the row-major bf16 weight.
#include "tg_common.hpp"


namespace tg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ static inline float bf16_bits_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ static inline uint32_t f32_to_bf16_bits(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)x); }

// frag[((mb*KS + ks)*64 + lane)*8 + j] = W[32 ks + 8 (lane>>4) + j][feature(mb, lane & 15)],   KS = K/32, mb < M/16
__global__ __launch_bounds__(256) void dx_pack_weights_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ frag, int K,
                                                              int M) {
    const int total = K * M;
    const int KS = K >> 5;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
        const int ks = blk % KS, mb = blk / KS;
        const int m = lane & 15, q = lane >> 4;
        const int f = 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3);
        frag[i] = W[(int64_t)(32 * ks + 8 * q + j) * M + f];
    }
}

template <int M, int K, int NT, int kDxWaves, bool kBits>
__global__ __launch_bounds__(64 * kDxWaves, 1) void dx_relu_bias_kernel(const uint16_t* __restrict__ dz_in,
                                                                        const uint4* __restrict__ wfrag,
                                                                        const uint16_t* __restrict__ act,
                                                                        const uint32_t* __restrict__ maskbits,
                                                                        uint16_t* __restrict__ dz_out, int64_t rows,
                                                                        float* __restrict__ partial) {
    extern __shared__ uint4 lds[];
    constexpr int MP = M / 32, KS = K / 32;           // feature pairs-of-blocks, k-steps
    constexpr bool kHoistMask = (NT == 1);
    for (int i = threadIdx.x; i < M * K / 8; i += blockDim.x) lds[i] = wfrag[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t ntiles = (rows + 16 * NT - 1) / (16 * NT);
    const int64_t tstride = (int64_t)gridDim.x * kDxWaves;

    float bsum[MP][8];
#pragma unroll
    for (int p = 0; p < MP; ++p)
#pragma unroll
        for (int r = 0; r < 8; ++r) bsum[p][r] = 0.f;

    // loads of one tile: the B operand (dZ rows) and, for NT == 1, all mask words (requested up front: loaded inside
    // the feature loop, which the scheduling barriers keep in order, each would be needed ~0.1 us after its issue --
    // an HBM latency stall per feature pair)
    auto issue_loads = [&](int64_t tile, bf16x8 (&bb)[NT][KS], uint4 (&mm)[kHoistMask ? NT : 1][kHoistMask ? MP : 1],
                           uint32_t (&mb)[NT][MP / 2]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int64_t rr = tile * (16 * NT) + 16 * t + n;
            rr = rr < rows ? rr : rows - 1;
            const uint4* bp = reinterpret_cast<const uint4*>(dz_in + rr * K + 8 * q);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bb[t][ks] = __builtin_bit_cast(bf16x8, bp[4 * ks]);
            if constexpr (kBits) {
                // H mask bits per row as tg_mlp_forward_chain writes them: [half h = q>>1][MP/2 words]
                const uint32_t* wp = maskbits + rr * MP + (q >> 1) * (MP / 2);
                if constexpr (MP == 8) {
                    const uint4 v = *reinterpret_cast<const uint4*>(wp);
                    mb[t][0] = v.x; mb[t][1] = v.y; mb[t][2] = v.z; mb[t][3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < MP / 2; ++i) mb[t][i] = wp[i];
                }
            } else if constexpr (kHoistMask) {
                const uint4* mp = reinterpret_cast<const uint4*>(act + rr * M + 8 * q);
#pragma unroll
                for (int p = 0; p < MP; ++p) mm[t][p] = mp[4 * p];
            }
        }
    };
    // (also requesting the NEXT tile's operands before this tile's arithmetic was measured: slower, 1.31 vs 1.23 ms)
    for (int64_t tile = (int64_t)blockIdx.x * kDxWaves + wave; tile < ntiles; tile += tstride) {
        bf16x8 b[NT][KS];
        uint4 mka[kHoistMask ? NT : 1][kHoistMask ? MP : 1];
        uint32_t mbits[NT][MP / 2];
        issue_loads(tile, b, mka, mbits);
        bool ok[NT];
        const uint4* ap[NT];
        uint4* op[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = tile * (16 * NT) + 16 * t + n;
            ok[t] = row < rows;
            const int64_t rr = ok[t] ? row : rows - 1;
            ap[t] = kBits ? nullptr : reinterpret_cast<const uint4*>(act + rr * M + 8 * q);
            op[t] = reinterpret_cast<uint4*>(dz_out + rr * M + 8 * q);
        }
#pragma unroll
        for (int p = 0; p < MP; ++p) {
            uint4 mk[NT];
            if constexpr (!kBits) {
#pragma unroll
                for (int t = 0; t < NT; ++t) mk[t] = kHoistMask ? mka[t][p] : ap[t][4 * p];
            }
            f32x4 acc[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, lds[((2 * p + e) * KS + ks) * 64 + lane]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[t][ks], acc[t][e], 0, 0, 0);
                }
                if (M * K >= 256 * 256) __builtin_amdgcn_sched_barrier(0);   // at most K/32 fragments in flight
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t mw[4] = {mk[t].x, mk[t].y, mk[t].z, mk[t].w};
                // bit form: features 2w / 2w+1 of this lane are bits (p&1)*8 + 4(q&1) + w (+16) of word p>>1
                const uint32_t bw = kBits ? mbits[t][p >> 1] >> ((p & 1) * 8 + 4 * (q & 1)) : 0u;
                uint32_t ow[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // features 2w, 2w+1 of the lane's 8: accumulator (e = w>>1, r = 2(w&1) + {0,1}), rounded as a pair
                    // (one v_cvt_pk_bf16_f32) and ANDed with the pair's keep-mask
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const uint32_t pk = __builtin_bit_cast(
                        uint32_t, __builtin_convertvector(f32x2{acc[t][w >> 1][2 * (w & 1)], acc[t][w >> 1][2 * (w & 1) + 1]}, bf16x2));
                    uint32_t keep;
                    if constexpr (kBits) {
                        keep = ((bw >> w) & 0x00010001u) * 0xFFFFu;            // bit w -> low half, bit w+16 -> high half
                    } else {
                        // post-ReLU activations are >= 0: positive  <=>  nonzero magnitude bits and sign clear
                        const uint32_t a_lo = mw[w] & 0xFFFFu, a_hi = mw[w] >> 16;
                        keep = (((a_lo & 0x7FFFu) != 0 && !(a_lo & 0x8000u)) ? 0x0000FFFFu : 0u) |
                               (((a_hi & 0x7FFFu) != 0 && !(a_hi & 0x8000u)) ? 0xFFFF0000u : 0u);
                    }
                    ow[w] = ok[t] ? (pk & keep) : 0u;
                    bsum[p][2 * w] += bf16_bits_to_f32(ow[w] & 0xFFFFu);
                    bsum[p][2 * w + 1] += __uint_as_float(ow[w] & 0xFFFF0000u);
                }
                if (ok[t]) op[t][4 * p] = uint4{ow[0], ow[1], ow[2], ow[3]};
            }
            // keep the fragment reads of later feature blocks from being hoisted over this one (the fully unrolled
            // body would otherwise want all M*K/512 fragments live at once and spill)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // bias-gradient partial of this workgroup: sum the 16 row lanes of each quarter-wave, then the waves (fixed order)
    __syncthreads();                                  // every wave is done with the weight fragments
    float* red = reinterpret_cast<float*>(lds);       // [kDxWaves][M]
#pragma unroll
    for (int p = 0; p < MP; ++p)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float v = bsum[p][r];
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) v += __shfl_xor(v, s, 64);
            if (n == 0) red[wave * M + 32 * p + 8 * q + r] = v;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < M; c += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kDxWaves; ++w) s += red[w * M + c];
        partial[(int64_t)blockIdx.x * M + c] = s;
    }
}

static int dx_blocks() { return device_cus(); }

template <int M, int K, int NT, int kDxWaves, bool kBits>
static int dx_launch(const void* dz_in, const void* wfrag, const void* act, const void* maskbits, void* dz_out, int64_t rows,
                     float* partial, hipStream_t st) {
    const size_t shmem = (size_t)M * K * 2;
    auto kern = dx_relu_bias_kernel<M, K, NT, kDxWaves, kBits>;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_dx_relu_bias")) return rc;
    hipLaunchKernelGGL(kern, dim3(dx_blocks()), dim3(64 * kDxWaves), shmem, st, (const uint16_t*)dz_in, (const uint4*)wfrag,
                       (const uint16_t*)act, (const uint32_t*)maskbits, (uint16_t*)dz_out, rows, partial);
    TG_LAUNCH_CHECK("tg_dx_relu_bias");
    return TG_OK;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_dx_relu_bias_supported(int32_t k_dim, int32_t m_dim) {
    return (k_dim == 256 && m_dim == 256) || (k_dim == 128 && m_dim == 128) || (k_dim == 64 && m_dim == 64);
}

int tg_dx_relu_bias_blocks(void) { return dx_blocks(); }

int tg_dx_pack_weights(const void* d_w, void* d_wfrag, int32_t k_dim, int32_t m_dim, void* stream) {
    TG_REQUIRE(d_w && d_wfrag, "tg_dx_pack_weights: null pointer");
    TG_REQUIRE(k_dim > 0 && m_dim > 0 && k_dim % 32 == 0 && m_dim % 32 == 0,
               "tg_dx_pack_weights: K=%d and M=%d must be multiples of 32", k_dim, m_dim);
    const int total = k_dim * m_dim;
    hipLaunchKernelGGL(dx_pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_w,
                       (uint16_t*)d_wfrag, k_dim, m_dim);
    TG_LAUNCH_CHECK("tg_dx_pack_weights");
    return TG_OK;
}

int tg_dx_relu_bias(const void* d_dz_in, const void* d_wfrag, const void* d_act, const void* d_maskbits, void* d_dz_out,
                    int64_t rows, int32_t k_dim, int32_t m_dim, float* d_partial, void* stream) {
    TG_REQUIRE(d_dz_in && d_wfrag && (d_act || d_maskbits) && d_dz_out && d_partial, "tg_dx_relu_bias: null pointer");
    TG_REQUIRE(rows >= 0, "tg_dx_relu_bias: rows=%lld is negative", (long long)rows);
    TG_REQUIRE(tg_dx_relu_bias_supported(k_dim, m_dim), "tg_dx_relu_bias: K=%d, M=%d has no kernel (256x256, 128x128, 64x64)", k_dim,
               m_dim);
    hipStream_t st = (hipStream_t)stream;
#define TG_DX_ARGS d_dz_in, d_wfrag, d_act, d_maskbits, d_dz_out, rows, d_partial, st
#define TG_DX_CALL(M_, NT_) (d_maskbits ? dx_launch<M_, M_, NT_, 8, true>(TG_DX_ARGS) : dx_launch<M_, M_, NT_, 8, false>(TG_DX_ARGS))
    // (with 1-bit masks the 32-row tile no longer spills but is no faster: 1.04 vs 1.01 ms)
    // 16-row tiles and 8 waves measured fastest at 256 x 256 (1.34 ms per 2^22 rows; 32-row tiles spill: 1.71 ms;
    // 12 / 16 waves per workgroup spill harder: 1.46 / 2.31 ms)
    if (k_dim == 256) return TG_DX_CALL(256, 1);
    if (k_dim == 128) return TG_DX_CALL(128, 2);
    TG_REQUIRE(!d_maskbits, "tg_dx_relu_bias: mask bits need a width of 128 or 256 (tg_mlp_forward_chain writes them)");
    return dx_launch<64, 64, 2, 8, false>(TG_DX_ARGS);
#undef TG_DX_CALL
#undef TG_DX_ARGS
}

}  // extern "C"
