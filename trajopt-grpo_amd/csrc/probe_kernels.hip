// Measurement instrument, not a product kernel: the bare inner loop of the chain kernels (mlp_fwd_chain.hip, mlp_bwd_chain.hip,
// fused_rollout.hip; mlp_f32_chain.hip for the fp32 form) with nothing around it -- A fragments from LDS, B fragments in
// registers, dependent accumulator chains, two waves per SIMD, one workgroup per CU, random operands, no global traffic after the
// prologue.  Launched back to back until the clock has settled it runs at the matrix rate the PACKAGE sustains under its power
// limit with this operand pattern: the update kernels of /root/reference's `loss.backward()` + forward passes
// (algorithms/ppo.py:147-183) cannot run faster than this loop whatever their schedule, and bench.py quotes it as
// `roofline.sustained_peak` beside the datasheet peak (round 2 measured 1.53 PFLOP/s for the bf16 form with a stand-alone program,
// tools/mfma_shape_probe.hip: 0.61 of the 2.5 PFLOP/s dense peak; the fp32 form 106 of 157 TFLOP/s).
#include "mfma_ring.hpp"

namespace tg {

TG_CLOCK_PROBE_VAR(g_probe_mfma_loop, attach_probe_mfma_loop)

// dtype 0: per iteration a wave multiplies one 16-KiB weight block (32 output features x 256 inputs, the chain kernels' fragment
// image) into its 32 rows: 8 k-steps x 2 feature halves x 2 row tiles of v_mfma_f32_16x16x32_bf16 = 32 MFMAs, 16 ds_read_b128.
__global__ __launch_bounds__(512, 2) void mfma_loop_bf16_kernel(const uint4* __restrict__ w, const uint4* __restrict__ xg,
                                                                float* __restrict__ out, int iters) {
    extern __shared__ uint4 lds[];                       // 4 blocks of 16 KiB
    TG_CLOCK_PROBE_BEGIN(g_probe_mfma_loop)
    for (int i = threadIdx.x; i < 4 * 1024; i += 512) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 x[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) x[ks] = __builtin_bit_cast(bf16x8, xg[((size_t)blockIdx.x * 512 + threadIdx.x) * 16 + ks]);
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint4* cur = lds + (it & 3) * 1024;
        f32x4 acc[2][2] = {};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(ks * 2 + f) * 64 + lane]);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x[ks * 2 + r], acc[f][r], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) sink += acc[f][r][q];
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
    TG_CLOCK_PROBE_END(g_probe_mfma_loop)
}

// dtype 1: the fp32 chain learner's loop: per iteration a wave runs two dependent chains of 64 v_mfma_f32_32x32x2_f32 (two
// 32-feature output tiles of a 128-wide layer for its 32 rows): A = 4 k-steps per ds_read_b128, B = the activations in registers.
__global__ __launch_bounds__(512, 2) void mfma_loop_f32_kernel(const uint4* __restrict__ w, const uint4* __restrict__ xg,
                                                               float* __restrict__ out, int iters) {
    extern __shared__ uint4 lds[];                       // 4 blocks of 16 KiB = [16 groups of 4 steps][64 lanes] float4
    TG_CLOCK_PROBE_BEGIN(g_probe_mfma_loop)
    for (int i = threadIdx.x; i < 4 * 1024; i += 512) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float x[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 v = __builtin_bit_cast(float4, xg[((size_t)blockIdx.x * 512 + threadIdx.x) * 16 + q]);
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        f32x16 acc[2] = {};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a = __builtin_bit_cast(float4, lds[((it + t) & 3) * 1024 + g * 64 + lane]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x[4 * g], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x[4 * g + 1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x[4 * g + 2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x[4 * g + 3], acc[t], 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[0][r] + acc[1][r];
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
    TG_CLOCK_PROBE_END(g_probe_mfma_loop)
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_mfma_sustained_probe_blocks(void) { return device_cus(); }

double tg_mfma_sustained_probe_flops(int32_t dtype, int32_t iters) {
    // per wave and iteration: bf16 32 MFMAs x 2 x 16 x 16 x 32 flop; fp32 128 MFMAs x 2 x 32 x 32 x 2 flop -- 524,288 either way
    (void)dtype;
    return (double)device_cus() * 8.0 * (double)iters * 524288.0;
}

int tg_mfma_sustained_probe(int32_t dtype, int32_t iters, const void* d_w, const void* d_x, float* d_out, void* stream) {
    TG_REQUIRE(dtype == 0 || dtype == 1, "tg_mfma_sustained_probe: dtype %d (0 bf16, 1 f32)", dtype);
    TG_REQUIRE(iters > 0 && d_w && d_x && d_out, "tg_mfma_sustained_probe: bad argument");
    hipStream_t st = (hipStream_t)stream;
    static LdsOptIn opt_in[2];
    const void* kern = dtype == 0 ? (const void*)mfma_loop_bf16_kernel : (const void*)mfma_loop_f32_kernel;
    if (int rc = reserve_dynamic_lds(kern, 65536, opt_in[dtype], "tg_mfma_sustained_probe")) return rc;
    if (dtype == 0)
        hipLaunchKernelGGL(mfma_loop_bf16_kernel, dim3((unsigned)device_cus()), dim3(512), 65536, st, (const uint4*)d_w, (const uint4*)d_x, d_out, iters);
    else
        hipLaunchKernelGGL(mfma_loop_f32_kernel, dim3((unsigned)device_cus()), dim3(512), 65536, st, (const uint4*)d_w, (const uint4*)d_x, d_out, iters);
    TG_LAUNCH_CHECK("tg_mfma_sustained_probe");
    return TG_OK;
}

}  // extern "C"
