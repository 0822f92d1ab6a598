// Shared host/device helpers for the gfx950 kernels (error reporting, launch math, Philox).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/trajopt_grpo_hip.h"

namespace tg {

int set_error(int code, const char* fmt, ...);

#define TG_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ::tg::set_error(TG_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define TG_LAUNCH_CHECK(what)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess)                                                       \
            return ::tg::set_error(TG_ERR_HIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// Individually rounded fp32 operations.  HIP's __fmul_rn/__fadd_rn are plain `*`/`+` and get
// contracted into FMAs under the default -ffp-contract=fast-honor-pragmas; the reference (NumPy /
// torch CPU) rounds every operation, so the float32 control path and the reward-to-go recurrence
// use these to stay bit-compatible.
__device__ static inline float rn_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ static inline float rn_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ static inline float rn_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ static inline float rn_div(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Per-DEVICE launch state (one process may drive several GPUs): the device ordinal of the calling thread, its CU count,
// and the opt-in to more than 64 KiB of dynamic LDS per kernel (gfx950 has 160 KiB per CU), cached per device.
constexpr int kMaxDevices = 16;
static inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        d = 0;
    }
    return (d < 0 || d >= kMaxDevices) ? 0 : d;
}
static inline int device_cus() {
    static int n[kMaxDevices] = {};
    const int d = current_device();
    if (n[d] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        n[d] = cus;
    }
    return n[d];
}
struct LdsOptIn { size_t bytes[kMaxDevices] = {}; };
static inline int reserve_dynamic_lds(const void* kernel, size_t bytes, LdsOptIn& cache, const char* what) {
    if (bytes > 160 * 1024) return set_error(TG_ERR_ARG, "%s: %zu B of LDS needed (> 160 KiB)", what, bytes);
    const int d = current_device();
    if (bytes > 64 * 1024 && bytes > cache.bytes[d]) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return set_error(TG_ERR_HIP, "%s: cannot reserve %zu B of LDS (%s)", what, bytes, hipGetErrorString(e));
        }
        cache.bytes[d] = bytes;
    }
    return TG_OK;
}

// Swarm termination: lanes [k*agents, (k+1)*agents) of a wavefront are the bodies of one environment; the env
// truncates when any of them does.  One 64-bit ballot, then each lane tests its own segment of the mask.
__device__ static inline bool any_in_segment(bool flag, int agents) {
    if (agents <= 1) return flag;
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63;
    const unsigned long long seg = (agents >= 64 ? ~0ull : ((1ull << agents) - 1ull)) << (lane & ~(agents - 1));
    return (b & seg) != 0ull;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): counter-based, so the draw for (env, t) does
// not depend on launch geometry or on how envs are sharded across GPUs.
// ---------------------------------------------------------------------------
struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    static constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;

    __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t hi0 = __umulhi(M0, c[0]), hi1 = __umulhi(M1, c[2]);
#else
        uint32_t hi0 = (uint32_t)(((uint64_t)M0 * c[0]) >> 32), hi1 = (uint32_t)(((uint64_t)M1 * c[2]) >> 32);
#endif
        uint32_t lo0 = M0 * c[0], lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }

    // key = seed (64 bit); counter = (idx lo, idx hi, sub, stream)
    __host__ __device__ static inline void draw(uint64_t seed, uint64_t idx, uint32_t sub, uint32_t stream,
                                                uint32_t (&out)[4]) {
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        out[0] = (uint32_t)idx; out[1] = (uint32_t)(idx >> 32); out[2] = sub; out[3] = stream;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(out, k0, k1);
            k0 += W0; k1 += W1;
        }
    }

    // uniform in (0,1]: never 0, so log() is finite
    __host__ __device__ static inline float u01(uint32_t x) { return ((x >> 8) + 1u) * (1.0f / 16777216.0f); }
    // uniform in [0,1) with 53 bits from two words
    __host__ __device__ static inline double u01d(uint32_t a, uint32_t b) {
        return (double)((((uint64_t)a << 32) | b) >> 11) * (1.0 / 9007199254740992.0);
    }
};

}  // namespace tg
