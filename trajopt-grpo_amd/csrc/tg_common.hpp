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

// ---- in-kernel clock probe (tg_clock_probe_attach; bench.py's roofline.clock_GHz) ----
// A translation unit that owns a probed kernel declares TG_CLOCK_PROBE_VAR(var, attach_fn) at namespace scope: a per-TU device
// pointer (no -fgpu-rdc: every TU has its own) and the host function that sets it.  The kernel brackets its body with
// TG_CLOCK_PROBE_BEGIN(var) / TG_CLOCK_PROBE_END(var).  Nothing is held in registers in between (the chain kernels have none to
// spare): thread 0 parks the entry stamps in the probe buffer itself -- a store issued BEFORE any LDS-DMA of the kernel, so the
// rings' counted vmcnt waits, which count the operations issued BEHIND a block, are not affected -- and reads them back (volatile:
// no forwarding through a register) at the exit, behind the kernel's final vmcnt(0).
#define TG_CLOCK_PROBE_VAR(var, attach_fn)                                                                                 \
    static __device__ unsigned long long* var = nullptr;                                                                   \
    int attach_fn(void* d_probe) {                                                                                         \
        unsigned long long* p = (unsigned long long*)d_probe;                                                              \
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(var), &p, sizeof(p));                                                  \
        return e == hipSuccess ? TG_OK : set_error(TG_ERR_HIP, "tg_clock_probe_attach: %s", hipGetErrorString(e));        \
    }
#define TG_CLOCK_PROBE_BEGIN(var)                                                                                          \
    if (threadIdx.x == 0) {                                                                                                \
        volatile unsigned long long* pr__ = var;                                                                           \
        if (pr__ != nullptr && blockIdx.x < 4096u) {                                                                       \
            pr__[4 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime();                                                       \
            pr__[5 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();                                                   \
        }                                                                                                                  \
    }
#define TG_CLOCK_PROBE_END(var)                                                                                            \
    if (threadIdx.x == 0) {                                                                                                \
        unsigned long long* pr__ = var;                                                                                    \
        if (pr__ != nullptr && blockIdx.x < 4096u) {                                                                       \
            const unsigned long long c1__ = __builtin_amdgcn_s_memtime(), r1__ = __builtin_amdgcn_s_memrealtime();         \
            const volatile unsigned long long* pk__ = pr__ + 4 + 2 * blockIdx.x;                                           \
            atomicAdd(pr__ + 0, c1__ - pk__[0]);                                                                           \
            atomicAdd(pr__ + 1, r1__ - pk__[1]);                                                                           \
            atomicAdd(pr__ + 2, 1ull);                                                                                     \
        }                                                                                                                  \
    }
int attach_probe_fwd_chain(void*);
int attach_probe_fwd_chain_plain(void*);
int attach_probe_bwd_chain(void*);
int attach_probe_weight_grad(void*);
int attach_probe_f32(int which, void*);
int attach_probe_f32w(int which, void*);
int attach_probe_mfma_loop(void*);

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
