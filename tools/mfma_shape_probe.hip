// Microbenchmark: the chain kernels' inner loop (A fragments from LDS by ds_read_b128, B fragments in registers, one accumulator
// chain per output tile, 2 waves per SIMD, 256 workgroups) with v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 at equal flops,
// LDS bytes and registers, on random data, after ~1 s of back-to-back launches (steady clocks).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape_probe tools/mfma_shape_probe.hip && ./mfma_shape_probe 40000
// Round 2, one MI355X: 32x32x16 1.40-1.46 PFLOP/s, 16x16x32 1.53-1.54, 64 rows per wave (A fragment reused) 1.37 -- the power-limited
// ceiling of this operand pattern (the
// forward chain without stores runs at 1.36).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ w, const uint4* __restrict__ xg, float* __restrict__ out, int iters) {
    extern __shared__ uint4 lds[];                       // 4 blocks of 16 KiB
    for (int i = threadIdx.x; i < 4 * 1024; i += 512) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 x[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) x[ks] = __builtin_bit_cast(bf16x8, xg[(blockIdx.x * 512 + threadIdx.x) * 16 + ks]);
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint4* cur = lds + (it & 3) * 1024;
        if (SHAPE == 32) {
            f32x16 acc = {};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sink += acc[r];
        } else {
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
        // keep the operands changing a little (not constant-foldable), like a chain does
        x[it & 15] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(uint4, x[it & 15]));
    }
    out[blockIdx.x * 512 + threadIdx.x] = sink;
}

// 64 rows per wave: every A fragment read from LDS feeds TWO MFMAs (two 32-row blocks), 4 waves per workgroup = one per SIMD
// (the operands of 64 rows need the whole 512-register file).  Same flops per workgroup and launch as k<32>.
__global__ __launch_bounds__(256, 1) void k64(const uint4* __restrict__ w, const uint4* __restrict__ xg, float* __restrict__ out, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 4 * 1024; i += 256) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 x0[16], x1[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        x0[ks] = __builtin_bit_cast(bf16x8, xg[(blockIdx.x * 512 + threadIdx.x) * 16 + ks]);
        x1[ks] = __builtin_bit_cast(bf16x8, xg[(blockIdx.x * 512 + 256 + threadIdx.x) * 16 + ks]);
    }
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint4* cur = lds + (it & 3) * 1024;
        f32x16 a0 = {}, a1 = {};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x0[ks], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x1[ks], a1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += a0[r] + a1[r];
        x0[it & 15] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(uint4, x0[it & 15]));
    }
    out[blockIdx.x * 512 + threadIdx.x] = sink;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    uint4 *w, *x; float* out;
    hipMalloc(&w, 64 * 1024); hipMalloc(&x, 256 * 512 * 16 * 16); hipMalloc(&out, 256 * 512 * 4);
    // random bf16 in [-1, 1)
    uint16_t* hw = (uint16_t*)malloc(64 * 1024); uint16_t* hx = (uint16_t*)malloc(256 * 512 * 16 * 16);
    srand(1);
    auto rb = []() { union { float f; uint32_t u; } c; c.f = (rand() / (float)RAND_MAX) * 2.f - 1.f; return (uint16_t)(c.u >> 16); };
    for (int i = 0; i < 32 * 1024; ++i) hw[i] = rb();
    for (int i = 0; i < 256 * 512 * 16 * 8; ++i) hx[i] = rb();
    hipMemcpy(w, hw, 64 * 1024, hipMemcpyHostToDevice); hipMemcpy(x, hx, 256 * 512 * 16 * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k64, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
        for (int shape : {32, 16, 64}) {
            // ~2 s of back-to-back launches before timing: steady-state clocks
            for (int rep = 0; rep < 12; ++rep) {
                if (rep == 8) hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 65536, 0, w, x, out, iters);
                else if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 65536, 0, w, x, out, iters);
                else hipLaunchKernelGGL(k64, dim3(256), dim3(256), 65536, 0, w, x, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = 4.0 * 256 * 8 * (double)iters * 16 * 32768.0;
            printf("%s: %.2f ms per launch, %.1f TFLOP/s\n", shape == 32 ? "32x32x16, 8 waves x 32 rows" : shape == 16 ? "16x16x32, 8 waves x 32 rows" : "32x32x16, 4 waves x 64 rows (A fragment reused)", ms / 4, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
