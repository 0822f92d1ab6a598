// Elementwise pieces of the MLP backward that PyTorch would run as separate passes:
// ReLU backward fused with the bias-gradient column sums (one read of dA and A, one write of dZ).
#include "tg_common.hpp"

namespace tg {

constexpr int kReluBlocks = 2048;

__device__ static inline float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// dZ = dA * (A > 0) in place;  partial[block][c] = sum over the block's rows of dZ[:, c].
// bf16, row-major [rows][cols], cols % 8 == 0 and cols <= 2048.  Each thread owns 8 consecutive columns.
__global__ __launch_bounds__(256) void relu_bwd_bias_bf16_kernel(uint16_t* __restrict__ dA, const uint16_t* __restrict__ A,
                                                                 int64_t rows, int cols, float* __restrict__ partial) {
    extern __shared__ float sh[];                   // [rows_per_pass][cols]
    const int tpr = cols >> 3;                      // threads per row
    const int rpp = blockDim.x / tpr;               // rows per pass
    const int rl = threadIdx.x / tpr, cl = (threadIdx.x % tpr) << 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (rl < rpp) {
        for (int64_t r = (int64_t)blockIdx.x * rpp + rl; r < rows; r += (int64_t)gridDim.x * rpp) {
            const int64_t off = r * cols + cl;
            uint4 d = *reinterpret_cast<const uint4*>(dA + off);
            const uint4 a = *reinterpret_cast<const uint4*>(A + off);
            uint32_t dv[4] = {d.x, d.y, d.z, d.w};
            const uint32_t av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // post-ReLU activations are >= 0: positive  <=>  nonzero magnitude bits and sign clear
                const uint16_t a_lo = (uint16_t)(av[j] & 0xFFFFu), a_hi = (uint16_t)(av[j] >> 16);
                const bool p_lo = (a_lo & 0x7FFFu) != 0 && !(a_lo & 0x8000u);
                const bool p_hi = (a_hi & 0x7FFFu) != 0 && !(a_hi & 0x8000u);
                uint32_t v = dv[j];
                if (!p_lo) v &= 0xFFFF0000u;
                if (!p_hi) v &= 0x0000FFFFu;
                dv[j] = v;
                acc[2 * j] += bf16_to_f32((uint16_t)(v & 0xFFFFu));
                acc[2 * j + 1] += bf16_to_f32((uint16_t)(v >> 16));
            }
            d.x = dv[0]; d.y = dv[1]; d.z = dv[2]; d.w = dv[3];
            *reinterpret_cast<uint4*>(dA + off) = d;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[rl * cols + cl + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += sh[r * cols + c];
        partial[(int64_t)blockIdx.x * cols + c] = s;
    }
}

__global__ __launch_bounds__(256) void relu_bwd_bias_f32_kernel(float* __restrict__ dA, const float* __restrict__ A,
                                                                int64_t rows, int cols, float* __restrict__ partial) {
    extern __shared__ float sh[];
    const int tpr = cols >> 2;                      // 4 floats (16 B) per thread
    const int rpp = blockDim.x / tpr;
    const int rl = threadIdx.x / tpr, cl = (threadIdx.x % tpr) << 2;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < rpp) {
        for (int64_t r = (int64_t)blockIdx.x * rpp + rl; r < rows; r += (int64_t)gridDim.x * rpp) {
            const int64_t off = r * cols + cl;
            float4 d = *reinterpret_cast<const float4*>(dA + off);
            const float4 a = *reinterpret_cast<const float4*>(A + off);
            d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f;
            d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
            acc[0] += d.x; acc[1] += d.y; acc[2] += d.z; acc[3] += d.w;
            *reinterpret_cast<float4*>(dA + off) = d;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[rl * cols + cl + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += sh[r * cols + c];
        partial[(int64_t)blockIdx.x * cols + c] = s;
    }
}

// Head backward fused with the top hidden layer's ReLU backward and bias gradient:
//   dA[r][c] = sum_k dout[r][k] * Wh[k][c]   (k < A <= 8: a rank-A product, computed on the fly in fp32)
//   dZ[r][c] = dA[r][c] * (Act[r][c] > 0)    written once (bf16 or f32);  partial[block][c] = column sums of dZ
// replaces the [rows x 8] x [8 x cols] GEMM (which only writes 512 B/row) plus the read of dA in relu_bwd_bias.
__device__ static inline uint16_t f32_to_bf16_rne(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

// `maskbits` (bf16 only): 1 bit per activation as tg_mlp_forward_chain writes them (cols bits per row: [half h][cols/64
// words], feature 32 mt + 16 h + r -> bit (mt&1)*8 + (r>>1) + 16 (r&1) of word mt>>1) instead of the activation row.
template <bool kBf16, int KA>        // KA = the head's output count rounded up to 1, 2, 4 or 8
__global__ __launch_bounds__(256) void head_bwd_relu_bias_kernel(const float* __restrict__ dout, int a_dim,
                                                                 const float* __restrict__ Wh, const void* __restrict__ act,
                                                                 const uint32_t* __restrict__ maskbits,
                                                                 void* __restrict__ dz, int64_t rows, int cols,
                                                                 float* __restrict__ partial) {
    extern __shared__ float sh[];
    constexpr int PER = 8;
    const int tpr = cols / PER;
    const int rpp = blockDim.x / tpr;
    const int rl = threadIdx.x / tpr, cl = (threadIdx.x % tpr) * PER;
    float acc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = 0.f;
    if (rl < rpp) {
        float w[KA][PER];                                     // this thread's columns of the head weights
#pragma unroll
        for (int k = 0; k < KA; ++k)
#pragma unroll
            for (int j = 0; j < PER; ++j) w[k][j] = (k < a_dim) ? Wh[(int64_t)k * cols + cl + j] : 0.f;
        // (issuing the loads of 2 or 4 rows per trip ahead of the arithmetic was measured: slower, the kernel is bound by
        // its vector ALU work and occupancy, not by load latency)
        const int mt = cl >> 5, hh = (cl >> 4) & 1, wpr = cols >> 5;          // mask bits: tile, lane half, words per row
        for (int64_t r = (int64_t)blockIdx.x * rpp + rl; r < rows; r += (int64_t)gridDim.x * rpp) {
            float d[KA];
#pragma unroll
            for (int k = 0; k < KA; ++k) d[k] = (k < a_dim) ? dout[r * a_dim + k] : 0.f;
            float v[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KA; ++k) s += d[k] * w[k][j];
                v[j] = s;
            }
            const int64_t off = r * cols + cl;
            if constexpr (kBf16) {
                uint32_t av[4] = {0u, 0u, 0u, 0u}, bw = 0u;
                if (maskbits) {
                    // this thread's 8 features cl .. cl+7: tile mt = cl>>5, half h = (cl>>4)&1, r16 = (cl&15) + j
                    bw = maskbits[r * wpr + hh * (wpr >> 1) + (mt >> 1)] >> ((mt & 1) * 8 + ((cl & 15) >> 1));
                } else {
                    const uint4 a = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(act) + off);
                    av[0] = a.x; av[1] = a.y; av[2] = a.z; av[3] = a.w;
                }
                uint32_t ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint16_t a_lo = (uint16_t)(av[j] & 0xFFFFu), a_hi = (uint16_t)(av[j] >> 16);
                    const bool p_lo = maskbits ? ((bw >> j) & 1u) != 0 : ((a_lo & 0x7FFFu) != 0 && !(a_lo & 0x8000u));
                    const bool p_hi = maskbits ? ((bw >> (j + 16)) & 1u) != 0 : ((a_hi & 0x7FFFu) != 0 && !(a_hi & 0x8000u));
                    const uint16_t o_lo = p_lo ? f32_to_bf16_rne(v[2 * j]) : (uint16_t)0;
                    const uint16_t o_hi = p_hi ? f32_to_bf16_rne(v[2 * j + 1]) : (uint16_t)0;
                    acc[2 * j] += bf16_to_f32(o_lo);
                    acc[2 * j + 1] += bf16_to_f32(o_hi);
                    ov[j] = (uint32_t)o_lo | ((uint32_t)o_hi << 16);
                }
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dz) + off) = uint4{ov[0], ov[1], ov[2], ov[3]};
            } else {
                const float* ap = reinterpret_cast<const float*>(act) + off;
                float* zp = reinterpret_cast<float*>(dz) + off;
                const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                const float am[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int j = 0; j < PER; ++j) { v[j] = am[j] > 0.f ? v[j] : 0.f; acc[j] += v[j]; }
                *reinterpret_cast<float4*>(zp) = float4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4*>(zp + 4) = float4{v[4], v[5], v[6], v[7]};
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) sh[rl * cols + cl + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += sh[r * cols + c];
        partial[(int64_t)blockIdx.x * cols + c] = s;
    }
}

// Weight-gradient epilogue.  The split-K batched GEMM of mlp.py leaves n_batches partial products [M][K] in fp32 and a
// row tail shorter than one batch row count; PyTorch finished that with a reduction, a tail GEMM (50-95 us for < 128
// rows on hipBLASLt), and two additions.  One launch instead:
//   grad[m][k] += sum_b partial[b][m][k] + sum_{r < tail} dz_tail[r][m] * a_tail[r][k],   m < m_out, k < k_out.
// One thread per output element; the partials are read coalesced along k, summed in a fixed order (eight interleaved
// chains, then the tail rows in order): deterministic.
template <bool kBf16>
__global__ __launch_bounds__(256) void dw_finish_kernel(const float* __restrict__ partial, int n_batches, int M, int K,
                                                        const void* __restrict__ dz_tail, const void* __restrict__ a_tail, int tail,
                                                        float* __restrict__ grad, int64_t grad_ld, int m_out, int k_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m_out * k_out) return;
    const int m = idx / k_out, k = idx - m * k_out;
    const int64_t mk = (int64_t)M * K;
    const float* p = partial + (int64_t)m * K + k;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= n_batches; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += p[(int64_t)(b + q) * mk];      // 8 loads in flight per thread
    }
    for (; b < n_batches; ++b) acc[0] += p[(int64_t)b * mk];
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (kBf16) {
        const uint16_t* dz = static_cast<const uint16_t*>(dz_tail);
        const uint16_t* a = static_cast<const uint16_t*>(a_tail);
#pragma unroll 8
        for (int r = 0; r < tail; ++r) s = __builtin_fmaf(bf16_to_f32(dz[(int64_t)r * M + m]), bf16_to_f32(a[(int64_t)r * K + k]), s);
    } else {
        const float* dz = static_cast<const float*>(dz_tail);
        const float* a = static_cast<const float*>(a_tail);
#pragma unroll 8
        for (int r = 0; r < tail; ++r) s = __builtin_fmaf(dz[(int64_t)r * M + m], a[(int64_t)r * K + k], s);
    }
    grad[(int64_t)m * grad_ld + k] += s;
}

struct ColsumOut { float* p[8]; };

// out[v][c] += sum_b partial[b][v][c]: the per-workgroup column sums the backward kernels leave (bias gradients), added
// into up to 8 separate gradient vectors in one launch.  Block = 64 columns x 16 block-groups; every thread sums its
// group's blocks in order, the 16 group sums are added in order: deterministic.
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, int n_blocks, int n_vec, int width,
                                                             ColsumOut out) {
    __shared__ float sh[16][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;                       // flat (v, c)
    const int total = n_vec * width;
    float s = 0.f;
    if (col < total) {
        const int per = (n_blocks + 15) / 16;
        const int b0 = grp * per, b1 = min(b0 + per, n_blocks);
        for (int b = b0; b < b1; ++b) s += partial[(int64_t)b * total + col];
    }
    sh[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && col < total) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sh[g][cl];
        const int v = col / width, c = col - v * width;
        out.p[v][c] += t;
    }
}

// Head of the backward pass: dz[r][0..A) = dout[r][:] in the compute dtype, columns A..out_pad-1 zero, and the column
// sums of dout per workgroup (the head's bias gradient; tg_colsum_finish adds them up).  One thread per row.
constexpr int kHeadPrepBlocks = 1024;
template <bool kBf16>
__global__ __launch_bounds__(256) void head_prep_kernel(const float* __restrict__ dout, int64_t rows, int A, int out_pad,
                                                        void* __restrict__ dz, float* __restrict__ partial) {
    __shared__ float sh[4][8];                                  // one row of column sums per wave
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (k < A) ? dout[r * A + k] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
        if (kBf16) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[2 * k], v[2 * k + 1]}, bf16x2));
            uint16_t* d = static_cast<uint16_t*>(dz) + r * out_pad;
            *reinterpret_cast<uint4*>(d) = uint4{w[0], w[1], w[2], w[3]};
            if (out_pad > 8) *reinterpret_cast<uint4*>(d + 8) = uint4{0u, 0u, 0u, 0u};
        } else {
            float* d = static_cast<float*>(dz) + r * out_pad;
            for (int k = 0; k < out_pad; ++k) d[k] = k < 8 ? v[k] : 0.f;
        }
    }
    // wave reduction in a fixed order (xor butterflies), then the 4 waves in order
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float x = acc[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
        acc[k] = x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[wave][k] = acc[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < A) partial[blockIdx.x * A + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

}  // namespace tg



using namespace tg;

extern "C" {

int tg_relu_bwd_bias_blocks(void) { return kReluBlocks; }

int tg_relu_bwd_bias(void* d_dA, const void* d_A, int64_t rows, int32_t cols, int32_t is_bf16, float* d_partial,
                     void* stream) {
    TG_REQUIRE(d_dA && d_A && d_partial, "tg_relu_bwd_bias: null pointer");
    const int per = is_bf16 ? 8 : 4;
    TG_REQUIRE(rows >= 0 && cols > 0 && cols % per == 0 && cols / per <= 256,
               "tg_relu_bwd_bias: cols=%d must be a multiple of %d and <= %d", cols, per, 256 * per);
    const int tpr = cols / per, rpp = 256 / tpr;
    const size_t shmem = (size_t)rpp * cols * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16) {
        hipLaunchKernelGGL(relu_bwd_bias_bf16_kernel, dim3(kReluBlocks), dim3(256), shmem, st, (uint16_t*)d_dA,
                           (const uint16_t*)d_A, rows, cols, d_partial);
    } else {
        hipLaunchKernelGGL(relu_bwd_bias_f32_kernel, dim3(kReluBlocks), dim3(256), shmem, st, (float*)d_dA, (const float*)d_A,
                           rows, cols, d_partial);
    }
    TG_LAUNCH_CHECK("tg_relu_bwd_bias");
    return TG_OK;
}

int tg_head_bwd_relu_bias(const float* d_dout, int32_t act_dim, const float* d_whead, const void* d_act, const void* d_maskbits,
                          void* d_dz, int64_t rows, int32_t cols, int32_t is_bf16, float* d_partial, void* stream) {
    TG_REQUIRE(d_dout && d_whead && (d_act || d_maskbits) && d_dz && d_partial, "tg_head_bwd_relu_bias: null pointer");
    TG_REQUIRE(!d_maskbits || (is_bf16 && (cols == 128 || cols == 256)),
               "tg_head_bwd_relu_bias: mask bits need bf16 and 128 or 256 columns (tg_mlp_forward_chain writes them)");
    TG_REQUIRE(act_dim >= 1 && act_dim <= 8, "tg_head_bwd_relu_bias: act_dim %d outside 1..8", act_dim);
    TG_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols / 8 <= 256, "tg_head_bwd_relu_bias: cols=%d must be a multiple of 8 and <= 2048",
               cols);
    const int tpr = cols / 8, rpp = 256 / tpr;
    const size_t shmem = (size_t)rpp * cols * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define TG_HEAD_LAUNCH(BF, KA_)                                                                                          \
    hipLaunchKernelGGL((head_bwd_relu_bias_kernel<BF, KA_>), dim3(kReluBlocks), dim3(256), shmem, st, d_dout, act_dim, d_whead, \
                       d_act, (const uint32_t*)(BF ? d_maskbits : nullptr), d_dz, rows, cols, d_partial)
    const int ka = act_dim <= 1 ? 1 : act_dim <= 2 ? 2 : act_dim <= 4 ? 4 : 8;
    if (is_bf16) {
        switch (ka) {
            case 1: TG_HEAD_LAUNCH(true, 1); break;
            case 2: TG_HEAD_LAUNCH(true, 2); break;
            case 4: TG_HEAD_LAUNCH(true, 4); break;
            default: TG_HEAD_LAUNCH(true, 8); break;
        }
    } else {
        switch (ka) {
            case 1: TG_HEAD_LAUNCH(false, 1); break;
            case 2: TG_HEAD_LAUNCH(false, 2); break;
            case 4: TG_HEAD_LAUNCH(false, 4); break;
            default: TG_HEAD_LAUNCH(false, 8); break;
        }
    }
#undef TG_HEAD_LAUNCH
    TG_LAUNCH_CHECK("tg_head_bwd_relu_bias");
    return TG_OK;
}

int tg_dw_finish(const float* d_partial, int32_t n_batches, int32_t m_dim, int32_t k_dim, const void* d_dz_tail, const void* d_a_tail,
                 int32_t tail, int32_t is_bf16, float* d_grad, int64_t grad_ld, int32_t m_out, int32_t k_out, void* stream) {
    TG_REQUIRE(d_grad && (d_partial || n_batches == 0), "tg_dw_finish: null pointer");
    TG_REQUIRE(n_batches >= 0 && m_dim > 0 && k_dim > 0, "tg_dw_finish: bad partial shape [%d][%d][%d]", n_batches, m_dim, k_dim);
    TG_REQUIRE(m_out >= 0 && m_out <= m_dim && k_out >= 0 && k_out <= k_dim && grad_ld >= k_out,
               "tg_dw_finish: window %d x %d (ld %lld) outside %d x %d", m_out, k_out, (long long)grad_ld, m_dim, k_dim);
    TG_REQUIRE(tail >= 0 && tail <= 4096 && (tail == 0 || (d_dz_tail && d_a_tail)), "tg_dw_finish: bad tail (%d rows)", tail);
    if (m_out == 0 || k_out == 0) return TG_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div((int64_t)m_out * k_out, 256));
    if (is_bf16)
        hipLaunchKernelGGL(dw_finish_kernel<true>, grid, dim3(256), 0, st, d_partial, n_batches, m_dim, k_dim, d_dz_tail, d_a_tail, tail,
                           d_grad, grad_ld, m_out, k_out);
    else
        hipLaunchKernelGGL(dw_finish_kernel<false>, grid, dim3(256), 0, st, d_partial, n_batches, m_dim, k_dim, d_dz_tail, d_a_tail, tail,
                           d_grad, grad_ld, m_out, k_out);
    TG_LAUNCH_CHECK("tg_dw_finish");
    return TG_OK;
}

int tg_colsum_finish(const float* d_partial, int32_t n_blocks, int32_t n_vec, int32_t width, float* const* d_out, void* stream) {
    TG_REQUIRE(d_partial && d_out, "tg_colsum_finish: null pointer");
    TG_REQUIRE(n_blocks >= 0 && n_vec >= 1 && n_vec <= 8 && width >= 1, "tg_colsum_finish: bad shape [%d][%d][%d] (at most 8 vectors)",
               n_blocks, n_vec, width);
    ColsumOut out{};
    for (int v = 0; v < n_vec; ++v) {
        TG_REQUIRE(d_out[v], "tg_colsum_finish: output %d is null", v);
        out.p[v] = d_out[v];
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)ceil_div((int64_t)n_vec * width, 64)), dim3(1024), 0, (hipStream_t)stream,
                       d_partial, n_blocks, n_vec, width, out);
    TG_LAUNCH_CHECK("tg_colsum_finish");
    return TG_OK;
}

int tg_head_prep_blocks(void) { return kHeadPrepBlocks; }

int tg_head_prep(const float* d_dout, int64_t rows, int32_t act_dim, int32_t out_pad, int32_t is_bf16, void* d_dz, float* d_partial,
                 void* stream) {
    TG_REQUIRE(d_partial && ((d_dout && d_dz) || rows == 0), "tg_head_prep: null pointer");
    TG_REQUIRE(rows >= 0 && act_dim >= 1 && act_dim <= 8 && out_pad >= 8 && out_pad % 8 == 0 && out_pad <= 16,
               "tg_head_prep: act_dim %d / out_pad %d unsupported (1..8 outputs, padded to 8 or 16)", act_dim, out_pad);
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16)
        hipLaunchKernelGGL(head_prep_kernel<true>, dim3(kHeadPrepBlocks), dim3(256), 0, st, d_dout, rows, act_dim, out_pad, d_dz, d_partial);
    else
        hipLaunchKernelGGL(head_prep_kernel<false>, dim3(kHeadPrepBlocks), dim3(256), 0, st, d_dout, rows, act_dim, out_pad, d_dz, d_partial);
    TG_LAUNCH_CHECK("tg_head_prep");
    return TG_OK;
}

}  // extern "C"
