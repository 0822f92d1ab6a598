// The learner's prologue as a handful of launches: what /root/reference does between `buffer.sample()` and the first forward pass
// of `learn()` (algorithms/grpo.py:66-115 -- reward-to-go, group statistics, masking; algorithms/ppo.py:126-139 -- the boolean-mask
// gather `obs[mask]`, `act[mask]`, `adv[mask]`) and what rounds 1-3 ran as ~25 torch launches (nonzero, three index_selects, pads,
// fills).
//
//   tg_returns_moments   reward-to-go AND the per-env masked moments of it in ONE launch, for rollouts of a few thousand envs.  The
//                        stand-alone kernels (returns_kernels.hip) give every env one lane that walks its own time axis with 32 loads
//                        in flight: right at 65,536 envs (HBM-bound), but at 4,096 envs the launch is 64 waves, each waiting for 16
//                        rounds of dependent global-load latency (37 + 32 us for 18 MB).  Here a workgroup owns 32 envs: all four
//                        waves bring the block's whole [T][32] strip of rewards and masks into LDS with coalesced loads, ONE lane
//                        per env runs the recurrence backwards out of LDS in the reference's order -- bit-identical to tg_rtg_scan
//                        -- and then the masked moments forwards in ascending time (bit-identical to tg_masked_moments' first
//                        stage), and all waves store the returns.  Horizons up to 1,024 steps (160 KiB of LDS).
//   tg_learn_count       valid rows per 1,024-entry chunk of the flat [T*n] mask and their exclusive prefix: the row number of every
//                        valid (t, n) in time-major order (what `mask.nonzero()` enumerates), the total, and a flag when the total is
//                        not what the host was told (the rollout's own statistic).
//   tg_learn_compact     one pass over the mask that writes, for every valid row, its flat index, the observation as a padded
//                        compute-dtype input row (bf16 or f32, zero padding, the ones column the backward chain wants), the action
//                        row, and up to two per-row scalars gathered from [T][n] arrays -- the first one optionally normalised on
//                        the fly with tg_group_normalize's arithmetic (GRPO's advantage: no [T][n] advantage array at all).
#include "tg_common.hpp"

namespace tg {

// ---------------------------------------------------------------------------------------------------------------
// returns + per-env moments, small-n form
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRmEnvs = 32;          // envs per workgroup
constexpr int kRmChunk = 32;         // time steps a lane takes into registers at a time
constexpr int kRmMaxHorizon = (160 * 1024 - 16) / (kRmEnvs * 5);      // 1023

// LDS, all of it dynamic: R[T][32] f32 (rewards in, returns out, in place) + M[T][32] u8 + 16 B for the block's last live step.
// T * 160 + 16 B <= 160 KiB, i.e. T <= 1023 (kRmMaxHorizon; the host checks -- a static __shared__ beside a dynamic allocation of
// exactly 160 KiB does not launch).
__global__ __launch_bounds__(256) void returns_moments_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ mask, float gamma,
                                                              float* __restrict__ rtg, int64_t n, int32_t T, double* __restrict__ work) {
    extern __shared__ float s_dyn[];
    float* s_r = s_dyn;                                                 // [T][32]
    uint8_t* s_m = reinterpret_cast<uint8_t*>(s_dyn + (size_t)T * kRmEnvs);   // [T][32]
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * kRmEnvs;
    const int le = tid & (kRmEnvs - 1), lt0 = tid / kRmEnvs;            // env column; first row (rows 8 apart)
    const bool env_ok = e0 + le < n;
    // The scans below are serial in time, and most of the horizon is padding once the block's longest episode has ended (CartPole at
    // C2: ~100 of 500 steps): the loading threads find the block's last step that is not all zero bits -- beyond it reward * mask
    // + carry is +0 and the moments add nothing, exactly what the strips already hold -- and the scans stop there.
    int& s_last = *reinterpret_cast<int*>(s_m + (size_t)T * kRmEnvs);        // (T * 160 B in: 16-B aligned)
    if (tid == 0) s_last = -1;
    __syncthreads();
    int my_last = -1;
    // ---- all waves: the block's [T][32] strips of rewards and masks into LDS (coalesced 128-B / 32-B row segments, 8 rows per
    // pass and thread group, every load of a pass in flight together) ----
    // (unconditional loads from clamped addresses: a load under a branch is issued, waited for and only then followed by the next
    // one -- the first version of this kernel spent 60 of its 78 us that way; 2 x 16 loads per thread are in flight here)
    const int64_t e_c = env_ok ? e0 + le : n - 1;
    for (int t0 = 0; t0 < T; t0 += 128) {
        float vr[16]; uint8_t vm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = t0 + lt0 + 8 * k;
            const int64_t idx = (int64_t)(t < T ? t : T - 1) * n + e_c;
            vr[k] = rew[idx];
            vm[k] = mask[idx];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = t0 + lt0 + 8 * k;
            if (t < T) { s_r[t * kRmEnvs + le] = env_ok ? vr[k] : 0.f; s_m[t * kRmEnvs + le] = env_ok ? vm[k] : (uint8_t)0; }
            const bool live = env_ok && t < T && (vm[k] != 0 || __float_as_uint(vr[k]) != 0u);
            my_last = live && t > my_last ? t : my_last;
        }
    }
    if (my_last >= 0) atomicMax(&s_last, my_last);
    __syncthreads();
    // steps [0, t_stop) are scanned: t_stop = the first multiple of the chunk length past the last live step (or the horizon)
    const int t_live = s_last + 1;
    const int t_stop = (t_live + kRmChunk - 1) / kRmChunk * kRmChunk < T ? (t_live + kRmChunk - 1) / kRmChunk * kRmChunk : T;
    if (tid < kRmEnvs) {
        // ---- one lane per env.  Backward: R = r m + carry; carry = (gamma R) m -- the reference's order (grpo.py:66-74), individually
        // rounded.  A chunk's operands are read into registers first (independent LDS reads), so the dependent chain is three
        // vector operations per step and nothing else ----
        // (no load and no arithmetic under a branch: the first version guarded every step of the unrolled chunk with `k < cnt`, and
        // hipcc then waits for each LDS read where it stands -- 105 cycles per step instead of a dozen.  Full chunks run unguarded,
        // the < 32 steps that remain one by one.)
        float carry = 0.0f;
        int t_hi = t_stop;
        for (; t_hi >= kRmChunk; t_hi -= kRmChunk) {
            float r[kRmChunk], m[kRmChunk];
#pragma unroll
            for (int k = 0; k < kRmChunk; ++k) {
                r[k] = s_r[(t_hi - 1 - k) * kRmEnvs + tid];
                m[k] = (float)s_m[(t_hi - 1 - k) * kRmEnvs + tid];
            }
#pragma unroll
            for (int k = 0; k < kRmChunk; ++k) {
                const float R = rn_add(rn_mul(r[k], m[k]), carry);
                s_r[(t_hi - 1 - k) * kRmEnvs + tid] = R;
                carry = rn_mul(rn_mul(gamma, R), m[k]);
            }
        }
        for (int t = t_hi - 1; t >= 0; --t) {
            const float mf = (float)s_m[t * kRmEnvs + tid];
            const float R = rn_add(rn_mul(s_r[t * kRmEnvs + tid], mf), carry);
            s_r[t * kRmEnvs + tid] = R;
            carry = rn_mul(rn_mul(gamma, R), mf);
        }
        // ---- forward: masked moments over the valid steps in ascending time (tg_masked_moments' env stage: same order, same fp64
        // sums).  Selects instead of a branch per step: adding +0.0 to a sum that started at +0.0 changes no bit, and the fused
        // multiply-add is what `s2 += d * d` compiles to there ----
        double cnt = 0.0, s1 = 0.0, s2 = 0.0;
        auto add = [&](float v, uint8_t mk) {
            const double d = (double)v;
            const double q = __builtin_fma(d, d, s2);
            cnt += mk ? 1.0 : 0.0;
            s1 += mk ? d : 0.0;
            s2 = mk ? q : s2;
        };
        int t0 = 0;
        for (; t0 + kRmChunk <= t_stop; t0 += kRmChunk) {
            float v[kRmChunk]; uint8_t m[kRmChunk];
#pragma unroll
            for (int k = 0; k < kRmChunk; ++k) {
                v[k] = s_r[(t0 + k) * kRmEnvs + tid];
                m[k] = s_m[(t0 + k) * kRmEnvs + tid];
            }
#pragma unroll
            for (int k = 0; k < kRmChunk; ++k) add(v[k], m[k]);
        }
        for (; t0 < t_stop; ++t0) add(s_r[t0 * kRmEnvs + tid], s_m[t0 * kRmEnvs + tid]);
        if (e0 + tid < n) {
            work[e0 + tid] = cnt;
            work[n + e0 + tid] = s1;
            work[2 * n + e0 + tid] = s2;
        }
    }
    __syncthreads();
    // ---- all waves: the returns out ----
    for (int t0 = 0; t0 < T; t0 += 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + lt0 + 8 * k;
            if (env_ok && t < T) rtg[(int64_t)t * n + e0 + le] = s_r[t * kRmEnvs + le];
        }
    }
}

// one workgroup per group: fixed-order reduction of the group's per-env partials (returns_kernels.hip::group_moments_kernel's twin:
// that one is static to its translation unit)
__global__ __launch_bounds__(256) void group_moments2_kernel(const double* __restrict__ work, int64_t n, int64_t group_size,
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

// ---------------------------------------------------------------------------------------------------------------
// compaction of the valid rows
// ---------------------------------------------------------------------------------------------------------------
constexpr int kChunkElems = 1024;     // flat mask entries per workgroup (256 threads x 4)

__device__ static inline uint32_t load_mask4(const uint8_t* __restrict__ mask, int64_t f0, int64_t M) {
    // 4 consecutive mask bytes as flags in bits 0, 8, 16, 24 (entries past the end: 0)
    if (f0 + 3 < M && ((uintptr_t)(mask + f0) & 3) == 0) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(mask + f0);
        return ((w & 0x000000FFu) ? 1u : 0u) | ((w & 0x0000FF00u) ? 0x100u : 0u) | ((w & 0x00FF0000u) ? 0x10000u : 0u) |
               ((w & 0xFF000000u) ? 0x1000000u : 0u);
    }
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (f0 + k < M && mask[f0 + k]) v |= 1u << (8 * k);
    return v;
}

__global__ __launch_bounds__(256) void mask_count_kernel(const uint8_t* __restrict__ mask, int64_t M, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sh[4];
    const int64_t f0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t c = __popc(load_mask4(mask, f0, M));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// counts[c] -> exclusive prefix in place; total[0] = sum, total[1] = (sum != expected) when expected >= 0.  One workgroup.
__global__ __launch_bounds__(1024) void chunk_scan_kernel(uint32_t* __restrict__ counts, int64_t n_chunks, int64_t expected,
                                                          int64_t* __restrict__ total) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_chunks; base += 1024) {
        const int64_t i = base + tid;
        const uint64_t v = i < n_chunks ? counts[i] : 0u;
        uint64_t x = v;                                                 // inclusive scan within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n_chunks) counts[i] = (uint32_t)(before + x - v);
        __syncthreads();
        if (tid == 1023) carry_s = before + x;
        __syncthreads();
    }
    if (tid == 0) {
        total[0] = (int64_t)carry_s;
        total[1] = (expected >= 0 && (int64_t)carry_s != expected) ? 1 : 0;
    }
}

struct CompactArgs {
    const uint8_t* mask; const uint32_t* offsets; int64_t M, n, rows_cap;
    const void* obs; int64_t obs_feat_stride; int32_t S, obs_f64;
    const float* act; int64_t act_comp_stride; int32_t A;
    void* xin; int32_t in_pad, xin_bf16, ones_col;
    float* act_rows; int64_t* idx;
    const float* src0; float* dst0; const float* src1; float* dst1;
    const double* moments; int32_t norm_mode; int64_t group_size;
};

__device__ static inline uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));    // round to nearest even, as torch's copy_
}

template <typename OT>
__global__ __launch_bounds__(256) void compact_rows_kernel(CompactArgs a) {
    __shared__ uint32_t sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t f0 = ((int64_t)blockIdx.x * 256 + tid) * 4;
    const uint32_t flags = load_mask4(a.mask, f0, a.M);
    const uint32_t c = __popc(flags);
    uint32_t x = c;                                                     // inclusive scan of the per-thread counts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += sh[w];
    int64_t out = (int64_t)a.offsets[blockIdx.x] + before + x - c;     // row number of this thread's first valid entry
    if (c == 0) return;
    const OT* obs = reinterpret_cast<const OT*>(a.obs);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        if (!((flags >> (8 * k)) & 1u)) continue;
        const int64_t f = f0 + k;
        if (out >= a.rows_cap) return;                                  // more valid rows than the host was told: flagged by the scan
        a.idx[out] = f;
        // observation row: obs[s][t][e], element (t, e) = flat entry f of feature plane s (the plane holds T + 1 slots: f indexes the first T)
        if (a.xin_bf16) {
            uint32_t* row = reinterpret_cast<uint32_t*>(a.xin) + out * (a.in_pad / 2);
            for (int s0 = 0; s0 < a.in_pad; s0 += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s = s0 + j;
                    v[j] = s < a.S ? (float)obs[(int64_t)s * a.obs_feat_stride + f] : (s == a.ones_col ? 1.0f : 0.0f);
                }
                *reinterpret_cast<uint4*>(row + s0 / 2) = uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                                  pack_bf16x2(v[6], v[7])};
            }
        } else {
            float* row = reinterpret_cast<float*>(a.xin) + out * a.in_pad;
            for (int s0 = 0; s0 < a.in_pad; s0 += 4) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = s0 + j;
                    v[j] = s < a.S ? (float)obs[(int64_t)s * a.obs_feat_stride + f] : (s == a.ones_col ? 1.0f : 0.0f);
                }
                *reinterpret_cast<float4*>(row + s0) = float4{v[0], v[1], v[2], v[3]};
            }
        }
        if (a.act_rows) {
            for (int j = 0; j < a.A; ++j) a.act_rows[out * a.A + j] = a.act[(int64_t)j * a.act_comp_stride + f];
        }
        if (a.dst0) {
            float v = a.src0[f];
            if (a.moments) {
                // tg_group_normalize's arithmetic (returns_kernels.hip::group_normalize_kernel), for this one entry
                const double* mo = a.moments + ((f % a.n) / a.group_size) * 3;
                const double cnt = mo[0], s1 = mo[1], s2 = mo[2];
                const double mean = s1 / cnt;
                const double var = (s2 - s1 * mean) / (cnt - 1.0);
                const float meanf = (float)mean;
                const float stdf = (float)sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
                const float denom = a.norm_mode == 0 ? stdf : rn_add(stdf, 1e-8f);
                v = rn_div(rn_sub(v, meanf), denom);
            }
            a.dst0[out] = v;
        }
        if (a.dst1) a.dst1[out] = a.src1[f];
        ++out;
    }
}

// dst[idx[r]] = src[r * stride]: the critic's values of the valid rows back onto the [T][n] grid (what `V = zeros; V.index_copy_(0, idx,
// v)` did, algorithms/ppo.py:93 evaluated on valid rows only)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int64_t stride, const int64_t* __restrict__ idx,
                                                           int64_t rows, float* __restrict__ dst) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x)
        dst[idx[r]] = src[r * stride];
}

// dst0[r] = src0[idx[r]], dst1[r] = src1[idx[r]]: `adv[mask]`, `returns[mask]` (ppo.py:126-135) in one launch
__global__ __launch_bounds__(256) void gather_rows2_kernel(const int64_t* __restrict__ idx, int64_t rows, const float* __restrict__ src0,
                                                           float* __restrict__ dst0, const float* __restrict__ src1, float* __restrict__ dst1) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = idx[r];
        dst0[r] = src0[f];
        if (dst1) dst1[r] = src1[f];
    }
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_returns_moments_max_horizon(void) { return kRmMaxHorizon; }

int tg_returns_moments(const float* d_rew, const uint8_t* d_mask, float gamma, float* d_rtg, int64_t n, int32_t T, int64_t group_size,
                       double* d_moments, double* d_work, void* stream) {
    TG_REQUIRE(d_rew && d_mask && d_rtg && d_moments && d_work, "tg_returns_moments: null pointer");
    TG_REQUIRE(n > 0 && T > 0 && group_size > 0 && n % group_size == 0,
               "tg_returns_moments: n=%lld must be a positive multiple of group_size=%lld", (long long)n, (long long)group_size);
    hipStream_t st = (hipStream_t)stream;
    TG_REQUIRE(T <= tg_returns_moments_max_horizon(), "tg_returns_moments: horizon %d > %d (the block's [T][32] strips live in LDS: use tg_rtg_scan + tg_masked_moments)",
               T, tg_returns_moments_max_horizon());
    const size_t shmem = (size_t)T * kRmEnvs * 5 + 16;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)returns_moments_kernel, shmem, opt_in, "tg_returns_moments")) return rc;
    hipLaunchKernelGGL(returns_moments_kernel, dim3((unsigned)ceil_div(n, kRmEnvs)), dim3(256), shmem, st, d_rew, d_mask, gamma, d_rtg, n, T, d_work);
    TG_LAUNCH_CHECK("tg_returns_moments");
    hipLaunchKernelGGL(group_moments2_kernel, dim3((unsigned)(n / group_size)), dim3(256), 0, st, d_work, n, group_size, d_moments);
    TG_LAUNCH_CHECK("tg_returns_moments(group)");
    return TG_OK;
}

int64_t tg_learn_count_workspace(int64_t entries) { return (ceil_div(entries > 0 ? entries : 1, kChunkElems) + 4) * (int64_t)sizeof(uint32_t); }

int tg_learn_count(const uint8_t* d_mask, int64_t entries, int64_t expected_rows, void* d_work, int64_t work_bytes, int64_t* d_total,
                   void* stream) {
    TG_REQUIRE(d_mask && d_work && d_total, "tg_learn_count: null pointer");
    TG_REQUIRE(entries > 0 && entries < ((int64_t)1 << 40), "tg_learn_count: %lld mask entries", (long long)entries);
    TG_REQUIRE(work_bytes >= tg_learn_count_workspace(entries), "tg_learn_count: workspace of %lld B < %lld", (long long)work_bytes,
               (long long)tg_learn_count_workspace(entries));
    const int64_t n_chunks = ceil_div(entries, kChunkElems);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, d_mask, entries, (uint32_t*)d_work);
    TG_LAUNCH_CHECK("tg_learn_count");
    hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, st, (uint32_t*)d_work, n_chunks, expected_rows, d_total);
    TG_LAUNCH_CHECK("tg_learn_count(scan)");
    return TG_OK;
}

int tg_learn_compact(const tg_compact_args* p, void* stream) {
    TG_REQUIRE(p && p->d_mask && p->d_offsets && p->d_obs && p->d_xin && p->d_idx, "tg_learn_compact: null pointer");
    TG_REQUIRE(p->n > 0 && p->T > 0 && p->S > 0 && p->S <= 64, "tg_learn_compact: bad sizes (n %lld, T %d, S %d)", (long long)p->n, p->T, p->S);
    TG_REQUIRE(p->obs_dtype == TG_F32 || p->obs_dtype == TG_F64, "tg_learn_compact: obs dtype %d", p->obs_dtype);
    TG_REQUIRE(p->in_pad >= p->S && p->in_pad <= 64 && p->in_pad % (p->xin_bf16 ? 8 : 4) == 0,
               "tg_learn_compact: in_pad %d (a multiple of %d, >= S)", p->in_pad, p->xin_bf16 ? 8 : 4);
    TG_REQUIRE(p->ones_col < p->in_pad && (p->ones_col < 0 || p->ones_col >= p->S), "tg_learn_compact: ones column %d inside the observation", p->ones_col);
    TG_REQUIRE(!p->d_act_rows || (p->d_act && p->A > 0 && p->A <= 16), "tg_learn_compact: actions: null pointer or A = %d", p->A);
    TG_REQUIRE((!p->d_dst0 || p->d_src0) && (!p->d_dst1 || p->d_src1), "tg_learn_compact: a per-row output without its source");
    TG_REQUIRE(!p->d_moments || (p->d_dst0 && p->group_size > 0 && p->n % p->group_size == 0 && (p->norm_mode == 0 || p->norm_mode == 1)),
               "tg_learn_compact: normalisation needs dst0, a group size dividing n and mode 0 / 1");
    TG_REQUIRE(p->rows_cap >= 0, "tg_learn_compact: negative row capacity");
    CompactArgs a{};
    a.mask = p->d_mask; a.offsets = (const uint32_t*)p->d_offsets; a.M = (int64_t)p->T * p->n; a.n = p->n; a.rows_cap = p->rows_cap;
    a.obs = p->d_obs; a.obs_feat_stride = p->obs_feat_stride; a.S = p->S; a.obs_f64 = p->obs_dtype == TG_F64;
    a.act = p->d_act; a.act_comp_stride = (int64_t)p->T * p->n; a.A = p->A;
    a.xin = p->d_xin; a.in_pad = p->in_pad; a.xin_bf16 = p->xin_bf16; a.ones_col = p->ones_col;
    a.act_rows = p->d_act_rows; a.idx = p->d_idx;
    a.src0 = p->d_src0; a.dst0 = p->d_dst0; a.src1 = p->d_src1; a.dst1 = p->d_dst1;
    a.moments = p->d_moments; a.norm_mode = p->norm_mode; a.group_size = p->group_size;
    if (p->rows_cap == 0) return TG_OK;
    const int64_t n_chunks = ceil_div(a.M, kChunkElems);
    hipStream_t st = (hipStream_t)stream;
    if (a.obs_f64) hipLaunchKernelGGL(compact_rows_kernel<double>, dim3((unsigned)n_chunks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(compact_rows_kernel<float>, dim3((unsigned)n_chunks), dim3(256), 0, st, a);
    TG_LAUNCH_CHECK("tg_learn_compact");
    return TG_OK;
}

int tg_scatter_rows(const float* d_src, int64_t src_stride, const int64_t* d_idx, int64_t rows, float* d_dst, void* stream) {
    TG_REQUIRE(rows >= 0 && src_stride >= 1, "tg_scatter_rows: bad sizes");
    if (rows == 0) return TG_OK;
    TG_REQUIRE(d_src && d_idx && d_dst, "tg_scatter_rows: null pointer");
    const unsigned grid = (unsigned)(ceil_div(rows, 256) < 4096 ? ceil_div(rows, 256) : 4096);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, src_stride, d_idx, rows, d_dst);
    TG_LAUNCH_CHECK("tg_scatter_rows");
    return TG_OK;
}

int tg_gather_rows2(const int64_t* d_idx, int64_t rows, const float* d_src0, float* d_dst0, const float* d_src1, float* d_dst1, void* stream) {
    TG_REQUIRE(rows >= 0, "tg_gather_rows2: negative row count");
    if (rows == 0) return TG_OK;
    TG_REQUIRE(d_idx && d_src0 && d_dst0 && ((d_src1 == nullptr) == (d_dst1 == nullptr)), "tg_gather_rows2: null pointer");
    const unsigned grid = (unsigned)(ceil_div(rows, 256) < 4096 ? ceil_div(rows, 256) : 4096);
    hipLaunchKernelGGL(gather_rows2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_idx, rows, d_src0, d_dst0, d_src1, d_dst1);
    TG_LAUNCH_CHECK("tg_gather_rows2");
    return TG_OK;
}

}  // extern "C"
