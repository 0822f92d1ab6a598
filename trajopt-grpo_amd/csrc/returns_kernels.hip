// Returns and advantages on the time-major SoA trajectory ([T][n], env index fastest).
//
// Each lane owns one environment and walks its time axis backwards, so the reward-to-go
// recurrence is evaluated in exactly the reference's order (bit-for-bit fp32, no FMA
// contraction), while 32 time steps of loads are kept in flight per lane so the launch is
// HBM-bound rather than latency-bound.  The episode mask is what segments the scan: a
// zero mask at t+1 cuts the carry exactly as `gamma * R[t+1] * m[t+1]` does in the reference.
// Group statistics are reduced across lanes with wavefront shuffles, deterministically.
#include "tg_common.hpp"

namespace tg {

constexpr int kChunk = 32;  // time steps of independent loads in flight per lane

// R[T-1] = r m;  R[t] = r[t] m[t] + (gamma R[t+1]) m[t+1].   grpo.py:66-74 == ppo.py:100-111
__global__ __launch_bounds__(256) void rtg_scan_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ mask,
                                                       float gamma, float* __restrict__ rtg, int64_t n, int32_t T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float carry = 0.0f;   // gamma-discounted, masked R[t+1]: (gamma * R[t+1]) * m[t+1]
    for (int32_t t_hi = T; t_hi > 0; t_hi -= kChunk) {
        const int32_t cnt = t_hi < kChunk ? t_hi : kChunk;
        float r[kChunk];
        uint8_t m[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const int64_t idx = (int64_t)(t_hi - 1 - k) * n + i;
                r[k] = rew[idx];
                m[k] = mask[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const float mf = (float)m[k];
                const float R = rn_add(rn_mul(r[k], mf), carry);
                rtg[(int64_t)(t_hi - 1 - k) * n + i] = R;
                carry = rn_mul(rn_mul(gamma, R), mf);
            }
        }
    }
}

// GAE, ppo.py:112-124:  delta_t = r_t + gamma V_{t+1} m_{t+1} - V_t   (UNMASKED r_t),
// A_t = delta_t + gamma lam A_{t+1} m_{t+1};  ret = V + A.
__global__ __launch_bounds__(256) void gae_scan_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                       const uint8_t* __restrict__ mask, float gamma, float lam,
                                                       float* __restrict__ adv, float* __restrict__ ret, int64_t n,
                                                       int32_t T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float next_v_m = 0.0f;   // V[t+1] * m[t+1]
    float next_a_m = 0.0f;   // (gamma*lam*A[t+1]) * m[t+1]
    const float gl = rn_mul(gamma, lam);
    // same walk as rtg_scan_kernel: kChunk time steps of independent loads in flight per lane, then the serial recurrence
    // over them in the reference's order (one dependent HBM round trip per time step otherwise)
    for (int32_t t_hi = T; t_hi > 0; t_hi -= kChunk) {
        const int32_t cnt = t_hi < kChunk ? t_hi : kChunk;
        float r[kChunk], v[kChunk];
        uint8_t m[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const int64_t idx = (int64_t)(t_hi - 1 - k) * n + i;
                r[k] = rew[idx];
                v[k] = val[idx];
                m[k] = mask[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const int32_t t = t_hi - 1 - k;
                const int64_t idx = (int64_t)t * n + i;
                const float mf = (float)m[k];
                float a;
                if (t == T - 1) {
                    a = rn_sub(r[k], v[k]);
                } else {
                    const float delta = rn_sub(rn_add(r[k], rn_mul(gamma, next_v_m)), v[k]);
                    a = rn_add(delta, next_a_m);
                }
                adv[idx] = a;
                ret[idx] = rn_add(v[k], a);
                next_v_m = rn_mul(v[k], mf);
                next_a_m = rn_mul(rn_mul(gl, a), mf);
            }
        }
    }
}

// one env's masked moments (count, sum, sum of squares) in fp64, ascending time: the ONE body behind env_moments_kernel and
// ppo_returns_kernel (their sums must agree bit for bit)
__device__ static inline void lane_moments(const float* x, const uint8_t* __restrict__ mask, int64_t n, int32_t T, int64_t i,
                                           double& cnt, double& s1, double& s2) {
    cnt = 0, s1 = 0, s2 = 0;
    for (int32_t t0 = 0; t0 < T; t0 += kChunk) {
        const int32_t c = (T - t0) < kChunk ? (T - t0) : kChunk;
        float v[kChunk];
        uint8_t m[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < c) {
                const int64_t idx = (int64_t)(t0 + k) * n + i;
                v[k] = x[idx];
                m[k] = mask[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < c && m[k]) {
                const double d = (double)v[k];
                cnt += 1.0; s1 += d; s2 += d * d;
            }
        }
    }
}

// per-env masked partial moments (count, sum, sum of squares) in fp64
__global__ __launch_bounds__(256) void env_moments_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                          int64_t n, int32_t T, double* __restrict__ work) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double cnt, s1, s2;
    lane_moments(x, mask, n, T, i, cnt, s1, s2);
    work[i] = cnt;
    work[n + i] = s1;
    work[2 * n + i] = s2;
}

// PPO's advantages and returns with their per-env moments in ONE launch (algorithms/ppo.py:100-124, :138-139): what
// tg_rtg_scan + `rtg - V` (Monte Carlo) or tg_gae_scan, followed by the env stage of tg_masked_moments on each of the two, computes
// -- the same operations in the same order, so the same bits.  A lane owns an env: the backward recurrence writes adv / ret
// [T][n], then the lane reads its own column back in ascending time for the fp64 sums (same-thread store -> load: program order).
// work: f64 [3][2 n]: plane j = {count, sum, sum of squares}, advantages in columns [0, n), returns in [n, 2 n) -- the layout
// group_moments_kernel reduces as two groups of n.
template <bool kGae>
__global__ __launch_bounds__(256) void ppo_returns_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                          const uint8_t* __restrict__ mask, float gamma, float lam, float* adv,
                                                          float* ret, int64_t n, int32_t T, double* __restrict__ work) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float carry = 0.0f;      // Monte Carlo: (gamma * R[t+1]) * m[t+1]
    float next_v_m = 0.0f;   // GAE: V[t+1] * m[t+1]
    float next_a_m = 0.0f;   // GAE: (gamma*lam*A[t+1]) * m[t+1]
    const float gl = rn_mul(gamma, lam);
    for (int32_t t_hi = T; t_hi > 0; t_hi -= kChunk) {
        const int32_t cnt = t_hi < kChunk ? t_hi : kChunk;
        float r[kChunk], v[kChunk];
        uint8_t m[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const int64_t idx = (int64_t)(t_hi - 1 - k) * n + i;
                r[k] = rew[idx];
                v[k] = val[idx];
                m[k] = mask[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (k < cnt) {
                const int32_t t = t_hi - 1 - k;
                const int64_t idx = (int64_t)t * n + i;
                const float mf = (float)m[k];
                if constexpr (kGae) {                            // gae_scan_kernel's step
                    float a;
                    if (t == T - 1) {
                        a = rn_sub(r[k], v[k]);
                    } else {
                        const float delta = rn_sub(rn_add(r[k], rn_mul(gamma, next_v_m)), v[k]);
                        a = rn_add(delta, next_a_m);
                    }
                    adv[idx] = a;
                    ret[idx] = rn_add(v[k], a);
                    next_v_m = rn_mul(v[k], mf);
                    next_a_m = rn_mul(rn_mul(gl, a), mf);
                } else {                                         // rtg_scan_kernel's step, then ppo.py:111's A = R - V
                    const float R = rn_add(rn_mul(r[k], mf), carry);
                    ret[idx] = R;
                    adv[idx] = rn_sub(R, v[k]);
                    carry = rn_mul(rn_mul(gamma, R), mf);
                }
            }
        }
    }
    double c0, s1, s2;
    lane_moments(adv, mask, n, T, i, c0, s1, s2);
    work[i] = c0; work[2 * n + i] = s1; work[4 * n + i] = s2;
    lane_moments(ret, mask, n, T, i, c0, s1, s2);
    work[n + i] = c0; work[3 * n + i] = s1; work[5 * n + i] = s2;
}

// ppo.py:138-139 and the 1 / n of :165-179 on the device, from the (all-reduced) moments [2][3] = {count, sum, sum of squares} of
// the valid advantages and returns: out f32 [8] = {adv mean, 1 / (adv std + 1e-8), ret mean, 1 / (ret std + 1e-8), -1 / n, c1 / n,
// kl_coeff / n, n}.  torch's own sequence of fp64 / fp32 operations (unbiased std, clamped at 0; NaN for n < 2, like torch).
__global__ void ppo_norm_kernel(const double* __restrict__ moments, double c1, double kl_coeff, float* __restrict__ out) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int q = 0; q < 2; ++q) {
        const double cnt = moments[3 * q], s1 = moments[3 * q + 1], s2 = moments[3 * q + 2];
        const double mean = s1 / cnt;
        const double var = (s2 - s1 * mean) / (cnt - 1.0);
        const float stdf = (float)sqrt(var < 0.0 ? 0.0 : var);            // (NaN stays NaN)
        out[2 * q] = (float)mean;
        out[2 * q + 1] = rn_div(1.0f, rn_add(stdf, 1e-8f));
    }
    const double n = moments[0];
    out[4] = (float)(-1.0 / n);
    out[5] = (float)(c1 / n);
    out[6] = (float)(kl_coeff / n);
    out[7] = (float)n;
}

// one workgroup per group: fixed-order reduction of the group's per-env partials
__global__ __launch_bounds__(256) void group_moments_kernel(const double* __restrict__ work, int64_t n, int64_t group_size,
                                                            double* __restrict__ moments) {
    __shared__ double sh[3][4];
    const int64_t g = blockIdx.x;
    const int64_t base = g * group_size;
    double acc[3] = {0, 0, 0};
    for (int64_t e = threadIdx.x; e < group_size; e += blockDim.x) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += work[j * n + base + e];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) sh[j][w] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        moments[g * 3 + j] = ((sh[j][0] + sh[j][1]) + sh[j][2]) + sh[j][3];
    }
}

// out = (x - mean_g) / std_g  (mode 0, grpo.py:115)   or   / (std_g + 1e-8)  (mode 1, ppo.py:138-139)
__global__ __launch_bounds__(256) void group_normalize_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                              const double* __restrict__ moments, int mode,
                                                              float* __restrict__ out, int64_t n, int32_t T,
                                                              int64_t group_size) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t t0 = blockIdx.y * kChunk;
    if (i >= n) return;
    const double* mo = moments + (i / group_size) * 3;
    const double cnt = mo[0], s1 = mo[1], s2 = mo[2];
    const double mean = s1 / cnt;
    // unbiased variance (torch.std default); NaN for cnt < 2 exactly like torch
    const double var = (s2 - s1 * mean) / (cnt - 1.0);
    const float meanf = (float)mean;
    const float stdf = (float)sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
    const float denom = mode == 0 ? stdf : rn_add(stdf, 1e-8f);
    const int32_t c = (T - t0) < kChunk ? (T - t0) : kChunk;
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
        if (k < c) {
            const int64_t idx = (int64_t)(t0 + k) * n + i;
            const float v = x[idx];
            out[idx] = mask[idx] ? rn_div(rn_sub(v, meanf), denom) : 0.0f;
        }
    }
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_rtg_scan(const float* d_rew, const uint8_t* d_mask, float gamma, float* d_rtg, int64_t n, int32_t T, void* stream) {
    TG_REQUIRE(d_rew && d_mask && d_rtg, "tg_rtg_scan: null pointer");
    TG_REQUIRE(n >= 0 && T > 0, "tg_rtg_scan: bad sizes n=%lld T=%d", (long long)n, T);
    if (n == 0) return TG_OK;
    const int block = n <= ((int64_t)1 << 18) ? 64 : 256;
    hipLaunchKernelGGL(rtg_scan_kernel, dim3((unsigned)ceil_div(n, block)), dim3(block), 0, (hipStream_t)stream, d_rew, d_mask,
                       gamma, d_rtg, n, T);
    TG_LAUNCH_CHECK("tg_rtg_scan");
    return TG_OK;
}

int tg_gae_scan(const float* d_rew, const float* d_values, const uint8_t* d_mask, float gamma, float lam, float* d_adv,
                float* d_ret, int64_t n, int32_t T, void* stream) {
    TG_REQUIRE(d_rew && d_values && d_mask && d_adv && d_ret, "tg_gae_scan: null pointer");
    TG_REQUIRE(n >= 0 && T > 0, "tg_gae_scan: bad sizes");
    if (n == 0) return TG_OK;
    const int block = n <= ((int64_t)1 << 18) ? 64 : 256;
    hipLaunchKernelGGL(gae_scan_kernel, dim3((unsigned)ceil_div(n, block)), dim3(block), 0, (hipStream_t)stream, d_rew,
                       d_values, d_mask, gamma, lam, d_adv, d_ret, n, T);
    TG_LAUNCH_CHECK("tg_gae_scan");
    return TG_OK;
}

int tg_masked_moments(const float* d_x, const uint8_t* d_mask, int64_t n, int32_t T, int64_t group_size, double* d_moments,
                      double* d_work, void* stream) {
    TG_REQUIRE(d_x && d_mask && d_moments && d_work, "tg_masked_moments: null pointer");
    TG_REQUIRE(n > 0 && T > 0 && group_size > 0 && n % group_size == 0,
               "tg_masked_moments: n=%lld must be a positive multiple of group_size=%lld", (long long)n, (long long)group_size);
    const int block = n <= ((int64_t)1 << 18) ? 64 : 256;
    hipLaunchKernelGGL(env_moments_kernel, dim3((unsigned)ceil_div(n, block)), dim3(block), 0, (hipStream_t)stream, d_x, d_mask,
                       n, T, d_work);
    TG_LAUNCH_CHECK("tg_masked_moments(env)");
    hipLaunchKernelGGL(group_moments_kernel, dim3((unsigned)(n / group_size)), dim3(256), 0, (hipStream_t)stream, d_work, n,
                       group_size, d_moments);
    TG_LAUNCH_CHECK("tg_masked_moments(group)");
    return TG_OK;
}

int tg_ppo_returns(const float* d_rew, const float* d_values, const uint8_t* d_mask, float gamma, float lam, int monte_carlo,
                   float* d_adv, float* d_ret, int64_t n, int32_t T, double* d_moments, double* d_work, void* stream) {
    TG_REQUIRE(d_rew && d_values && d_mask && d_adv && d_ret && d_moments && d_work, "tg_ppo_returns: null pointer");
    TG_REQUIRE(n > 0 && T > 0, "tg_ppo_returns: bad sizes n=%lld T=%d", (long long)n, T);
    TG_REQUIRE(d_adv != d_ret, "tg_ppo_returns: adv and ret must be distinct buffers");
    hipStream_t st = (hipStream_t)stream;
    const int block = n <= ((int64_t)1 << 18) ? 64 : 256;
    if (monte_carlo)
        hipLaunchKernelGGL(ppo_returns_kernel<false>, dim3((unsigned)ceil_div(n, block)), dim3(block), 0, st, d_rew, d_values, d_mask, gamma,
                           lam, d_adv, d_ret, n, T, d_work);
    else
        hipLaunchKernelGGL(ppo_returns_kernel<true>, dim3((unsigned)ceil_div(n, block)), dim3(block), 0, st, d_rew, d_values, d_mask, gamma,
                           lam, d_adv, d_ret, n, T, d_work);
    TG_LAUNCH_CHECK("tg_ppo_returns");
    // the per-env partials as two groups of n (advantages | returns): tg_masked_moments' group stage, once
    hipLaunchKernelGGL(group_moments_kernel, dim3(2), dim3(256), 0, st, d_work, 2 * n, n, d_moments);
    TG_LAUNCH_CHECK("tg_ppo_returns(group)");
    return TG_OK;
}

int tg_ppo_norm(const double* d_moments, double c1, double kl_coeff, float* d_norm8, void* stream) {
    TG_REQUIRE(d_moments && d_norm8, "tg_ppo_norm: null pointer");
    hipLaunchKernelGGL(ppo_norm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_moments, c1, kl_coeff, d_norm8);
    TG_LAUNCH_CHECK("tg_ppo_norm");
    return TG_OK;
}

int tg_group_normalize(const float* d_x, const uint8_t* d_mask, const double* d_moments, int mode, float* d_out, int64_t n,
                       int32_t T, int64_t group_size, void* stream) {
    TG_REQUIRE(d_x && d_mask && d_moments && d_out, "tg_group_normalize: null pointer");
    TG_REQUIRE(mode == 0 || mode == 1, "tg_group_normalize: mode %d", mode);
    TG_REQUIRE(n > 0 && T > 0 && group_size > 0 && n % group_size == 0, "tg_group_normalize: bad sizes");
    const int block = 256;
    hipLaunchKernelGGL(group_normalize_kernel, dim3((unsigned)ceil_div(n, block), (unsigned)ceil_div(T, kChunk)), dim3(block), 0,
                       (hipStream_t)stream, d_x, d_mask, d_moments, mode, d_out, n, T, group_size);
    TG_LAUNCH_CHECK("tg_group_normalize");
    return TG_OK;
}

}  // extern "C"
