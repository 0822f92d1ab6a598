// Environment kernels for gfx950: reset, standalone step, and the fused GPU-resident
// rollout step (sample action -> Env.step -> record -> terminate).  One lane per
// environment, struct-of-arrays state so every load/store is a coalesced 256-B (f32)
// or 512-B (f64) wavefront access.
#include "env_dynamics.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace tg {

// ---------------------------------------------------------------------------
// reset
// ---------------------------------------------------------------------------
template <typename Env, typename R>
__global__ __launch_bounds__(256) void reset_kernel(R* __restrict__ state, int64_t ld, int64_t n, uint64_t seed,
                                                    uint32_t stream_id, int64_t key_offset, int64_t key_div, int variant) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rnd[4];
    Philox::draw(seed, (uint64_t)((key_offset + i) / key_div), 0xFFFFFFFFu /* sub = reset */, stream_id, rnd);
    R o[Env::S];
    if constexpr (Env::kBalanceTerminates) Env::reset(rnd, o, variant);   // Pendulum: swingup or near-upright start
    else Env::reset(rnd, o);
#pragma unroll
    for (int k = 0; k < Env::S; ++k) state[k * ld + i] = o[k];
}

// ---------------------------------------------------------------------------
// standalone Env.step on SoA state
// ---------------------------------------------------------------------------
template <typename Env, typename R>
__global__ __launch_bounds__(256) void step_kernel(typename Env::C c, const R* __restrict__ state, int64_t ld,
                                                   const float* __restrict__ action, int64_t ld_a,
                                                   R* __restrict__ next, int64_t ld_next, int32_t* __restrict__ steps,
                                                   R* __restrict__ time_balanced, R* __restrict__ reward,
                                                   uint8_t* __restrict__ truncated, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    R s[Env::S], o[Env::S];
    float a[Env::A];
#pragma unroll
    for (int k = 0; k < Env::S; ++k) s[k] = state[k * ld + i];
#pragma unroll
    for (int k = 0; k < Env::A; ++k) a[k] = action[k * ld_a + i];
    const int steps_after = steps[i] + 1;
    R r;
    const StepOut out = Env::step(s, a, c, steps_after, o, r);
#pragma unroll
    for (int k = 0; k < Env::S; ++k) next[k * ld_next + i] = o[k];
    steps[i] = steps_after;
    reward[i] = r;
    truncated[i] = out.truncated ? 1 : 0;
    if (time_balanced != nullptr) time_balanced[i] = out.balanced ? time_balanced[i] + c.dt : (R)0;
}

// ---------------------------------------------------------------------------
// fused rollout step
// ---------------------------------------------------------------------------
struct Sigma { float v[8]; };

// kSpec: issue the state / mean loads before the alive test resolves (one memory round trip instead of
// two on the critical path).  Used for small launches, which are latency-bound; large launches are
// bandwidth-bound and skip the loads of ended envs instead.
template <typename Env, typename R, bool kSample, bool kSpec>
__global__ __launch_bounds__(256) void rollout_step_kernel(typename Env::C c, R* __restrict__ obs,
                                                           float* __restrict__ act, R* __restrict__ rew,
                                                           uint8_t* __restrict__ mask, int32_t* __restrict__ len,
                                                           int64_t n, int32_t T, int32_t t,
                                                           const float* __restrict__ mean, int64_t mean_rs, Sigma sigma,
                                                           const uint64_t* __restrict__ rng, int64_t env_offset,
                                                           int32_t agents) {
    constexpr int S = Env::S, A = Env::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < n;
    const int64_t ic = in_range ? i : n - 1;          // clamp so every lane's addresses are valid
    const int64_t T1 = (int64_t)T + 1;
    R s[S], o[S];
    float mu[A], a[A];
    const int32_t my_len = len[ic];
    if constexpr (kSpec) {
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = obs[(k * T1 + t) * n + ic];
        if constexpr (kSample) {
#pragma unroll
            for (int k = 0; k < A; ++k) mu[k] = mean[ic * mean_rs + k];
        } else {
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = act[((int64_t)k * T + t) * n + ic];
        }
    }
    // len == 0 while the episode runs (for Pendulum: -(consecutive balanced steps so far), see StepOut)
    const bool alive = in_range && (Env::kBalanceTerminates ? my_len <= 0 : my_len == 0);
    // wavefront-wide termination: if all 64 envs of this wave have ended, leave before touching the
    // trajectory again (wave-uniform branch, no divergence)
    if (__ballot(alive) == 0ull) return;
    if constexpr (!kSpec) {
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = alive ? obs[(k * T1 + t) * n + ic] : (R)0;
        if constexpr (kSample) {
#pragma unroll
            for (int k = 0; k < A; ++k) mu[k] = alive ? mean[ic * mean_rs + k] : 0.0f;
        } else {
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = alive ? act[((int64_t)k * T + t) * n + ic] : 0.0f;
        }
    }
    if constexpr (kSample) {
        // a = mean + sigma * eps, eps ~ N(0, I): Box-Muller on Philox words keyed by the GLOBAL env index
        // and t (independent of sharding / launch geometry).  Hardware log / sin / cos (revolutions).
        uint32_t rnd[4];
        Philox::draw(rng[0], (uint64_t)(env_offset + ic), (uint32_t)t, (uint32_t)rng[1], rnd);
        float eps[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (2 * h < A) {
                const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(Philox::u01(rnd[2 * h])));
                const float rev = Philox::u01(rnd[2 * h + 1]);
                eps[2 * h] = rad * __builtin_amdgcn_cosf(rev);
                eps[2 * h + 1] = rad * __builtin_amdgcn_sinf(rev);
            }
        }
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = rn_add(mu[k], rn_mul(sigma.v[k], eps[k]));
    }

    R r;
    const StepOut out = Env::step(s, a, c, t + 1, o, r);
    // the worker also stops at t == max_steps (rollout_worker.py:51); a swarm stops when any of its bodies does
    bool ended = out.truncated;
    int balanced_steps = 0;
    if constexpr (Env::kBalanceTerminates) {
        balanced_steps = out.balanced ? 1 - my_len : 0;
        ended = ended || (balanced_steps >= c.term_steps);                  // terminated, pendulum_env.py:151
    }
    const bool done = any_in_segment(alive && ended, agents) || (t + 1 >= T);
    const bool carry = alive && !done;
    if (in_range) {
        // every lane of a live wave stores (zeros for ended envs, which is what the padding must hold):
        // whole 256-B lines leave the wave, so no partial-line read-modify-write in L2 / HBM
        if constexpr (kSample) {
#pragma unroll
            for (int k = 0; k < A; ++k) act[((int64_t)k * T + t) * n + i] = alive ? a[k] : 0.0f;
        }
        rew[(int64_t)t * n + i] = alive ? r : (R)0;
        mask[(int64_t)t * n + i] = alive ? 1 : 0;
#pragma unroll
        for (int k = 0; k < S; ++k) obs[(k * T1 + t + 1) * n + i] = carry ? o[k] : (R)0;
        if (alive && done) len[i] = t + 1;
        else if (Env::kBalanceTerminates && alive) len[i] = -balanced_steps;
    }
}

// ---------------------------------------------------------------------------
// teacher-forced rollout, all time steps in ONE launch (tg_rollout_forced)
// ---------------------------------------------------------------------------
// `Env.step` over recorded actions for steps [t_begin, t_end): what `rollout_step_kernel<..., kSample = false>` does one launch per
// time step (replays of recorded trajectories, rollout/rollout_worker.py:51-68 with the action given), with the state kept in
// REGISTERS between steps.  One wavefront owns 64 envs for the whole range: per env-step it reads the action (A floats, prefetched
// kAhead steps ahead from clamped addresses -- no load sits under a branch) and writes the next observation, reward and mask byte;
// the state is never re-read.  At 65,536 envs the per-step launch moves 12 MB in ~5 us (one wave per SIMD and a kernel boundary per
// step: latency-bound, 0.3 of the HBM roofline); this form is a plain write stream of the trajectory.  Same step function, same
// contraction-off arithmetic, same "every lane of a live wave stores, zeros for ended envs" rule: bit-identical to the per-step path.
// (Measured and dropped: 32 envs per wave so that every SIMD holds two waves -- 0.66 against 0.51 ms at 65,536 envs: the step is
// bound by its ~500 vector instructions, not by their latency.)
constexpr int kForcedAhead = 8;
template <typename Env, typename R>
__global__ __launch_bounds__(64) void rollout_forced_kernel(typename Env::C c, R* __restrict__ obs, const float* __restrict__ act,
                                                            R* __restrict__ rew, uint8_t* __restrict__ mask, int32_t* __restrict__ len,
                                                            int64_t n, int32_t T, int32_t t_begin, int32_t t_end, int32_t agents) {
    constexpr int S = Env::S, A = Env::A;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < n;
    const int64_t ic = in_range ? i : n - 1;
    const int64_t T1 = (int64_t)T + 1;
    int32_t my_len = len[ic];
    bool alive = in_range && (Env::kBalanceTerminates ? my_len <= 0 : my_len == 0);
    R s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const R v = obs[(k * T1 + t_begin) * n + ic];
        s[k] = alive ? v : (R)0;
    }
    float ab[kForcedAhead][A];
    auto load_actions = [&](int t, float (&dst)[A]) {
        const int tc = t < T ? t : T - 1;
#pragma unroll
        for (int k = 0; k < A; ++k) dst[k] = act[((int64_t)k * T + tc) * n + ic];
    };
#pragma unroll
    for (int q = 0; q < kForcedAhead; ++q) load_actions(t_begin + q, ab[q]);
    for (int tb = t_begin; tb < t_end; tb += kForcedAhead) {
#pragma unroll
        for (int q = 0; q < kForcedAhead; ++q) {
            const int t = tb + q;
            if (t >= t_end || __ballot(alive) == 0ull) { tb = t_end; break; }      // (wave-uniform)
            float a[A];
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = ab[q][k];
            load_actions(t + kForcedAhead, ab[q]);
            R o[S], r;
            const StepOut out = Env::step(s, a, c, t + 1, o, r);
            bool ended = out.truncated;
            int balanced_steps = 0;
            if constexpr (Env::kBalanceTerminates) {
                balanced_steps = out.balanced ? 1 - my_len : 0;
                ended = ended || (balanced_steps >= c.term_steps);
            }
            const bool done = any_in_segment(alive && ended, agents) || (t + 1 >= T);
            const bool carry = alive && !done;
            if (in_range) {
                rew[(int64_t)t * n + i] = alive ? r : (R)0;
                mask[(int64_t)t * n + i] = alive ? 1 : 0;
#pragma unroll
                for (int k = 0; k < S; ++k) obs[(k * T1 + t + 1) * n + i] = carry ? o[k] : (R)0;
            }
            if (alive && done) my_len = t + 1;
            else if (Env::kBalanceTerminates && alive) my_len = -balanced_steps;
#pragma unroll
            for (int k = 0; k < S; ++k) s[k] = carry ? o[k] : (R)0;
            alive = carry;
        }
    }
    if (in_range) len[i] = my_len;
}

// sum of episode lengths (= env-steps executed = sum of mask) and #episodes ended
__global__ __launch_bounds__(256) void rollout_finish_kernel(const int32_t* __restrict__ len, int64_t n,
                                                             uint64_t* __restrict__ counters) {
    __shared__ unsigned long long s_sum[4], s_done[4];
    unsigned long long sum = 0, done = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t l = len[i];
        sum += (unsigned long long)l;
        done += (l > 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off, 64);
        done += __shfl_down(done, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = sum; s_done[w] = done; }
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[0] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        counters[1] = s_done[0] + s_done[1] + s_done[2] + s_done[3];
    }
}

__global__ void rng_advance_kernel(uint64_t* rng) { rng[1] += 1; }

// ---- the launches around a rollout, folded (a 3-ms step at 4,096 envs spent 60 us in ~15 of them) ----
// tg_rollout_begin: every buffer of the trajectory zeroed by ONE launch (six memsets before).
struct ZeroRegions { void* p[6]; unsigned long long units[6]; unsigned long long bytes[6]; };   // units = ceil(bytes / 16)
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroRegions z, unsigned long long total_units) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += stride) {
        unsigned long long v = u;
        int r = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (r == k && v >= z.units[k]) { v -= z.units[k]; r = k + 1; }
        char* base = reinterpret_cast<char*>(z.p[r]);
        if ((v + 1) * 16 <= z.bytes[r]) {
            *reinterpret_cast<uint4*>(base + v * 16) = uint4{0u, 0u, 0u, 0u};
        } else {
            for (unsigned long long b = v * 16; b < z.bytes[r]; ++b) base[b] = 0;
        }
    }
}

// tg_rollout_finish_stats: what followed a rollout as ~8 launches (tg_rollout_finish, tg_rng_advance, a reward sum, fills and copies
// in Rollout_Buffer.sample()) as two: per-workgroup partial reward sums in f64, then one workgroup that adds them in a fixed order
// (deterministic), sums the episode lengths and advances the RNG stream.  stats = {sum of rewards, n, sum of lengths} as f64.
template <typename R>
__global__ __launch_bounds__(256) void reward_partial_kernel(const R* __restrict__ rew, int64_t M, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) acc += (double)rew[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void finish_stats_kernel(const int32_t* __restrict__ len, int64_t n, uint64_t* __restrict__ counters,
                                                           const double* __restrict__ partial, int32_t n_partial, uint64_t* rng,
                                                           double* __restrict__ stats) {
    __shared__ unsigned long long s_sum[4], s_done[4];
    __shared__ double s_part[256];
    s_part[threadIdx.x] = (int)threadIdx.x < n_partial ? partial[threadIdx.x] : 0.0;      // (all partials in flight at once)
    unsigned long long sum = 0, done = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t l = len[i];
        sum += (unsigned long long)(l > 0 ? l : 0);
        done += (l > 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off, 64);
        done += __shfl_down(done, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = sum; s_done[w] = done; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long steps = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        counters[0] = steps;
        counters[1] = s_done[0] + s_done[1] + s_done[2] + s_done[3];
        double r = 0.0;
        for (int k = 0; k < n_partial; ++k) r += s_part[k];
        stats[0] = r;
        stats[1] = (double)n;
        stats[2] = (double)steps;
        if (rng != nullptr) rng[1] += 1;
    }
}

// Quadrotor._dynamics (12-state, explicit Euler).  quadrotor_env.py:128-169
template <typename R>
__global__ __launch_bounds__(256) void quadrotor12_kernel(const R* __restrict__ st, int64_t ld,
                                                          const R* __restrict__ ctl, int64_t ld_c, R* __restrict__ nx,
                                                          int64_t ld_n, int64_t n, R mass, R arm, R Ixx, R Iyy, R Izz,
                                                          R tc, R g, R dt) {
    using M = Math<R>;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    R s[12], u[4];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = st[k * ld + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = ctl[k * ld_c + i];
    const R phi = s[6], th = s[7], p = s[9], q = s[10], r = s[11];
    R sphi, cphi, sth, cth;
    M::sincos_(phi, &sphi, &cphi);
    M::sincos_(th, &sth, &cth);
    const R tth = M::tan_(th);
    const R ut = ((u[0] + u[1]) + u[2]) + u[3];
    const R im = (R)1 / mass;
    R rate[12];
    rate[0] = s[3]; rate[1] = s[4]; rate[2] = s[5];
    rate[3] = im * ((-sth) * ut);                       // third column of R (the [2][1] typo at :144 is unused)
    rate[4] = im * ((sphi * cth) * ut);
    rate[5] = im * ((cphi * cth) * ut + (-mass * g));
    rate[6] = (p + sphi * tth * q) + cphi * tth * r;
    rate[7] = cphi * q + (-sphi) * r;
    rate[8] = sphi / cth * q + cphi / cth * r;
    const R s22 = (R)(1.4142135623730951 / 2.0);
    rate[9] = (s22 * (u[0] + u[2] - u[1] - u[3]) * arm - (Izz - Iyy) * q * r) / Ixx;
    rate[10] = (s22 * (u[2] + u[3] - u[0] - u[1]) * arm - (Izz - Ixx) * p * r) / Iyy;
    rate[11] = (tc * (u[0] + u[3] - u[1] - u[2])) / Izz;
#pragma unroll
    for (int k = 0; k < 12; ++k) nx[k * ld_n + i] = s[k] + rate[k] * dt;
}

// ---------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------
static inline dim3 env_grid(int64_t n, int& block) {
    // small launches are latency-bound: one wave per workgroup spreads them over all CUs
    block = (n <= (int64_t)1 << 18) ? 64 : 256;
    return dim3((unsigned)ceil_div(n, block));
}

template <template <typename> class EnvT, typename R>
static int reset_dispatch(const tg_env_params* p, void* state, int64_t ld, int64_t n, uint64_t seed, uint64_t stream_id,
                          int64_t key_offset, int64_t key_div, hipStream_t st) {
    int block;
    dim3 grid = env_grid(n, block);
    hipLaunchKernelGGL((reset_kernel<EnvT<R>, R>), grid, dim3(block), 0, st, (R*)state, ld, n, seed, (uint32_t)stream_id,
                       key_offset, key_div, (int)(p->p[3] != 0.0));
    TG_LAUNCH_CHECK("tg_env_reset");
    return TG_OK;
}

template <template <typename> class EnvT, typename R>
static int step_dispatch(const tg_env_params* p, const void* state, int64_t ld, const float* action, int64_t ld_a,
                         void* next, int64_t ld_next, int32_t* steps, void* tb, void* reward, uint8_t* trunc, int64_t n,
                         hipStream_t st) {
    int block;
    dim3 grid = env_grid(n, block);
    auto c = EnvT<R>::C::make(*p);
    hipLaunchKernelGGL((step_kernel<EnvT<R>, R>), grid, dim3(block), 0, st, c, (const R*)state, ld, action, ld_a, (R*)next,
                       ld_next, steps, (R*)tb, (R*)reward, trunc, n);
    TG_LAUNCH_CHECK("tg_env_step");
    return TG_OK;
}

template <template <typename> class EnvT, typename R>
static int rollout_dispatch(const tg_env_params* p, const tg_traj* tr, int32_t t, const float* mean, int64_t mean_rs,
                            const float* sigma, const uint64_t* rng, int64_t env_offset, hipStream_t st) {
    int block;
    dim3 grid = env_grid(tr->n, block);
    auto c = EnvT<R>::C::make(*p);
    Sigma sg;
    memset(&sg, 0, sizeof(sg));
    const bool spec = tr->n <= ((int64_t)1 << 18);
#define TG_RS(SAMPLE, SPEC)                                                                                            \
    hipLaunchKernelGGL((rollout_step_kernel<EnvT<R>, R, SAMPLE, SPEC>), grid, dim3(block), 0, st, c, (R*)tr->d_obs,      \
                       tr->d_act, (R*)tr->d_rew, tr->d_mask, tr->d_len, tr->n, tr->horizon, t, mean, mean_rs, sg, rng, \
                       env_offset, p->agents)
    if (mean != nullptr) {
        for (int k = 0; k < EnvT<R>::A; ++k) sg.v[k] = sigma[k];
        if (spec) TG_RS(true, true); else TG_RS(true, false);
    } else {
        if (spec) TG_RS(false, true); else TG_RS(false, false);
    }
#undef TG_RS
    TG_LAUNCH_CHECK("tg_rollout_step");
    return TG_OK;
}

template <template <typename> class EnvT, typename R>
static int forced_dispatch(const tg_env_params* p, const tg_traj* tr, int32_t t_begin, int32_t t_end, hipStream_t st) {
    auto c = EnvT<R>::C::make(*p);
    hipLaunchKernelGGL((rollout_forced_kernel<EnvT<R>, R>), dim3((unsigned)ceil_div(tr->n, 64)), dim3(64), 0, st, c, (R*)tr->d_obs,
                       (const float*)tr->d_act, (R*)tr->d_rew, tr->d_mask, tr->d_len, tr->n, tr->horizon, t_begin, t_end, p->agents);
    TG_LAUNCH_CHECK("tg_rollout_forced");
    return TG_OK;
}

#define TG_ENV_SWITCH(env_id, dtype, CALL)                                                     \
    switch ((env_id) * 2 + (dtype)) {                                                          \
        case TG_ENV_CARTPOLE * 2 + TG_F32: return CALL(CartPoleEnv, float);                    \
        case TG_ENV_CARTPOLE * 2 + TG_F64: return CALL(CartPoleEnv, double);                   \
        case TG_ENV_QUADPOLE2D * 2 + TG_F32: return CALL(QuadPole2DEnv, float);                \
        case TG_ENV_QUADPOLE2D * 2 + TG_F64: return CALL(QuadPole2DEnv, double);               \
        case TG_ENV_QUADPOLE * 2 + TG_F32: return CALL(QuadPoleEnv, float);                    \
        case TG_ENV_QUADPOLE * 2 + TG_F64: return CALL(QuadPoleEnv, double);                   \
        case TG_ENV_PENDULUM * 2 + TG_F32: return CALL(PendulumEnv, float);                    \
        case TG_ENV_PENDULUM * 2 + TG_F64: return CALL(PendulumEnv, double);                   \
        default: return set_error(TG_ERR_UNSUPPORTED, "unsupported env_id %d / dtype %d", (int)(env_id), (int)(dtype)); \
    }

static int cartpole_time_trunc_step(int max_steps, double timestep) {
    // cartpole_env.py:27,151,168: `_time += timestep` accumulated in fp64 vs max_steps*timestep
    const double max_time = max_steps * timestep;
    double t = 0;
    int k = 0;
    for (;;) {
        ++k;
        t += timestep;
        if (t > max_time) return k;
        if (k > max_steps + 8) return k;  // unreachable for sane inputs; bounds the loop
    }
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_env_dims(int env_id, int* obs_dim, int* act_dim) {
    static const int dims[5][2] = {{5, 1}, {10, 2}, {20, 4}, {12, 4}, {3, 1}};
    TG_REQUIRE(env_id >= 0 && env_id < 5, "tg_env_dims: bad env_id %d", env_id);
    if (obs_dim) *obs_dim = dims[env_id][0];
    if (act_dim) *act_dim = dims[env_id][1];
    return TG_OK;
}

int tg_env_finalize_params(tg_env_params* p) {
    TG_REQUIRE(p != nullptr, "tg_env_finalize_params: null params");
    TG_REQUIRE(p->max_steps > 0 && p->timestep > 0, "tg_env_finalize_params: max_steps/timestep must be positive");
    const bool timed = p->env_id == TG_ENV_CARTPOLE || p->env_id == TG_ENV_PENDULUM;   // `_time > max_time` clauses
    p->time_trunc_step = timed ? cartpole_time_trunc_step(p->max_steps, p->timestep) : p->max_steps;
    if (p->env_id == TG_ENV_PENDULUM) {
        // consecutive balanced steps after which the fp64-accumulated `_time_balanced` first exceeds 5 s (pendulum_env.py:135,151)
        double tb = 0;
        int k = 0;
        do { tb = tb + p->timestep; ++k; } while (!(tb > 5.0) && k < (1 << 24));
        p->p[4] = (double)k;
    }
    return TG_OK;
}

int tg_env_default_params(int env_id, int max_steps, tg_env_params* out) {
    TG_REQUIRE(out != nullptr, "tg_env_default_params: null out");
    memset(out, 0, sizeof(*out));
    out->env_id = env_id;
    out->max_steps = max_steps > 0 ? max_steps : 500;
    out->timestep = 0.02;
    switch (env_id) {
        case TG_ENV_CARTPOLE: {  // cartpole_env.py:7-16
            const double d[] = {1.0, 1.0, 0.5, 9.80665};
            memcpy(out->p, d, sizeof(d));
            break;
        }
        case TG_ENV_QUADPOLE2D: {  // quadrotor_env.py:875-895
            const double d[] = {1.5, 0.5, 4e-1, 0.5, 0.75, 9.80665, 2.0, 0.25};
            memcpy(out->p, d, sizeof(d));
            break;
        }
        case TG_ENV_QUADPOLE: {  // quadrotor_env.py:362-382
            const double d[] = {1.5, 0.5, 9.80665, 0.5, 4e-1, 4e-1, 2.5e-1, 0.1, 0.5, 1.5};
            memcpy(out->p, d, sizeof(d));
            break;
        }
        case TG_ENV_PENDULUM: {  // pendulum_env.py:8-16: mass, length, gravity, swingup (0/1); p[4] = term steps (finalize)
            const double d[] = {1.0, 0.5, 9.80665, 0.0};
            memcpy(out->p, d, sizeof(d));
            out->timestep = 0.05;
            out->max_steps = max_steps > 0 ? max_steps : 200;
            break;
        }
        case TG_ENV_QUADROTOR12: {  // quadrotor_env.py:9-16
            const double d[] = {1.0, 0.2, 0.005, 0.005, 0.006, 0.017, 9.80665};
            memcpy(out->p, d, sizeof(d));
            out->timestep = 0.05;
            out->max_steps = max_steps > 0 ? max_steps : 200;
            break;
        }
        default:
            return set_error(TG_ERR_ARG, "tg_env_default_params: bad env_id %d", env_id);
    }
    return tg_env_finalize_params(out);
}

int tg_env_reset(const tg_env_params* p, int dtype, void* d_state, int64_t ld, int64_t n, uint64_t seed,
                 uint64_t stream_id, int64_t key_offset, int64_t key_div, void* stream) {
    TG_REQUIRE(p && d_state, "tg_env_reset: null pointer");
    TG_REQUIRE(n >= 0 && ld >= n && key_div >= 1 && key_offset >= 0, "tg_env_reset: bad sizes n=%lld ld=%lld key_div=%lld",
               (long long)n, (long long)ld, (long long)key_div);
    if (n == 0) return TG_OK;
#define CALL(E, R) reset_dispatch<E, R>(p, d_state, ld, n, seed, stream_id, key_offset, key_div, (hipStream_t)stream)
    TG_ENV_SWITCH(p->env_id, dtype, CALL)
#undef CALL
}

int tg_env_step(const tg_env_params* p, int dtype, const void* d_state, int64_t ld, const float* d_action, int64_t ld_a,
                void* d_next, int64_t ld_next, int32_t* d_steps, void* d_time_balanced, void* d_reward,
                uint8_t* d_truncated, int64_t n, void* stream) {
    TG_REQUIRE(p && d_state && d_action && d_next && d_steps && d_reward && d_truncated, "tg_env_step: null pointer");
    TG_REQUIRE(n >= 0 && ld >= n && ld_a >= n && ld_next >= n, "tg_env_step: leading dimensions smaller than n=%lld",
               (long long)n);
    if (n == 0) return TG_OK;
#define CALL(E, R) \
    step_dispatch<E, R>(p, d_state, ld, d_action, ld_a, d_next, ld_next, d_steps, d_time_balanced, d_reward, d_truncated, n, (hipStream_t)stream)
    TG_ENV_SWITCH(p->env_id, dtype, CALL)
#undef CALL
}

int tg_quadrotor12_dynamics(const tg_env_params* p, int dtype, const void* d_state, int64_t ld, const void* d_control,
                            int64_t ld_c, void* d_next, int64_t ld_next, int64_t n, void* stream) {
    TG_REQUIRE(p && d_state && d_control && d_next, "tg_quadrotor12_dynamics: null pointer");
    TG_REQUIRE(p->env_id == TG_ENV_QUADROTOR12, "tg_quadrotor12_dynamics: params are for env %d", p->env_id);
    TG_REQUIRE(n >= 0 && ld >= n && ld_c >= n && ld_next >= n, "tg_quadrotor12_dynamics: bad leading dimensions");
    if (n == 0) return TG_OK;
    const dim3 grid((unsigned)ceil_div(n, 256));
    const double* q = p->p;
    if (dtype == TG_F32) {
        hipLaunchKernelGGL(quadrotor12_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_state, ld,
                           (const float*)d_control, ld_c, (float*)d_next, ld_next, n, (float)q[0], (float)q[1], (float)q[2],
                           (float)q[3], (float)q[4], (float)q[5], (float)q[6], (float)p->timestep);
    } else if (dtype == TG_F64) {
        hipLaunchKernelGGL(quadrotor12_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)d_state, ld,
                           (const double*)d_control, ld_c, (double*)d_next, ld_next, n, q[0], q[1], q[2], q[3], q[4], q[5],
                           q[6], p->timestep);
    } else {
        return set_error(TG_ERR_UNSUPPORTED, "tg_quadrotor12_dynamics: dtype %d", dtype);
    }
    TG_LAUNCH_CHECK("tg_quadrotor12_dynamics");
    return TG_OK;
}

int tg_rollout_begin(const tg_traj* tr, int obs_dim, int act_dim, void* stream) {
    TG_REQUIRE(tr && tr->d_obs && tr->d_act && tr->d_rew && tr->d_mask && tr->d_len && tr->d_counters,
               "tg_rollout_begin: null trajectory pointer");
    TG_REQUIRE(tr->n > 0 && tr->horizon > 0 && obs_dim > 0 && act_dim > 0, "tg_rollout_begin: bad sizes");
    TG_REQUIRE(tr->dtype == TG_F32 || tr->dtype == TG_F64, "tg_rollout_begin: bad dtype %d", tr->dtype);
    const size_t rs = tr->dtype == TG_F64 ? 8 : 4;
    const size_t n = (size_t)tr->n, T = (size_t)tr->horizon;
    hipStream_t st = (hipStream_t)stream;
    // the whole obs buffer linearly (a strided 2-D fill that spares slot 0 runs at ~0.5 TB/s, 10x slower): the caller writes the
    // initial states into slot 0 AFTER this call (tg_env_reset or a copy).  All six buffers in ONE launch (six memsets were
    // six fill kernels: ~35 us of a 3-ms step at 4,096 envs).
    ZeroRegions z{};
    void* ptrs[6] = {tr->d_obs, tr->d_act, tr->d_rew, tr->d_mask, tr->d_len, tr->d_counters};
    const size_t bytes[6] = {(size_t)obs_dim * (T + 1) * n * rs, (size_t)act_dim * T * n * sizeof(float), T * n * rs, T * n,
                             n * sizeof(int32_t), 4 * sizeof(uint64_t)};
    unsigned long long total = 0;
    for (int k = 0; k < 6; ++k) {
        TG_REQUIRE(((uintptr_t)ptrs[k] & 15) == 0, "tg_rollout_begin: buffer %d is not 16-byte aligned", k);
        z.p[k] = ptrs[k]; z.bytes[k] = bytes[k]; z.units[k] = (bytes[k] + 15) / 16;
        total += z.units[k];
    }
    const unsigned long long want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < (unsigned long long)device_cus() * 16 ? want : (unsigned long long)device_cus() * 16);
    hipLaunchKernelGGL(zero_regions_kernel, dim3(grid), dim3(256), 0, st, z, total);
    TG_LAUNCH_CHECK("tg_rollout_begin");
    return TG_OK;
}

int tg_rollout_step(const tg_env_params* p, const tg_traj* tr, int32_t t, const float* d_mean, int64_t mean_row_stride,
                    const float* sigma, const uint64_t* d_rng, int64_t env_offset, void* stream) {
    TG_REQUIRE(p && tr && tr->d_obs && tr->d_act && tr->d_rew && tr->d_mask && tr->d_len, "tg_rollout_step: null pointer");
    TG_REQUIRE(tr->n > 0 && tr->horizon > 0 && t >= 0 && t < tr->horizon, "tg_rollout_step: t=%d outside horizon %d", t,
               tr->horizon);
    TG_REQUIRE(tr->horizon == p->max_steps, "tg_rollout_step: trajectory horizon %d != env.max_steps %d", tr->horizon,
               p->max_steps);
    TG_REQUIRE(p->agents <= 1 || (p->agents <= 64 && (p->agents & (p->agents - 1)) == 0 && tr->n % p->agents == 0),
               "tg_rollout_step: agents=%d must be a power of two <= 64 dividing n", p->agents);
    if (d_mean != nullptr) {
        int S, A;
        tg_env_dims(p->env_id, &S, &A);
        TG_REQUIRE(sigma != nullptr && d_rng != nullptr, "tg_rollout_step: sampling needs sigma and d_rng");
        TG_REQUIRE(mean_row_stride >= A, "tg_rollout_step: mean_row_stride %lld < act_dim %d", (long long)mean_row_stride, A);
    }
#define CALL(E, R) rollout_dispatch<E, R>(p, tr, t, d_mean, mean_row_stride, sigma, d_rng, env_offset, (hipStream_t)stream)
    TG_ENV_SWITCH(p->env_id, tr->dtype, CALL)
#undef CALL
}

int tg_rollout_forced(const tg_env_params* p, const tg_traj* tr, int32_t t_begin, int32_t t_end, void* stream) {
    TG_REQUIRE(p && tr && tr->d_obs && tr->d_act && tr->d_rew && tr->d_mask && tr->d_len, "tg_rollout_forced: null pointer");
    TG_REQUIRE(tr->n > 0 && tr->horizon > 0 && t_begin >= 0 && t_begin <= t_end && t_end <= tr->horizon,
               "tg_rollout_forced: steps [%d, %d) outside horizon %d", t_begin, t_end, tr->horizon);
    TG_REQUIRE(tr->horizon == p->max_steps, "tg_rollout_forced: trajectory horizon %d != env.max_steps %d", tr->horizon, p->max_steps);
    TG_REQUIRE(p->agents <= 1 || (p->agents <= 64 && (p->agents & (p->agents - 1)) == 0 && tr->n % p->agents == 0),
               "tg_rollout_forced: agents=%d must be a power of two <= 64 dividing n", p->agents);
    if (t_begin == t_end) return TG_OK;
#define CALL(E, R) forced_dispatch<E, R>(p, tr, t_begin, t_end, (hipStream_t)stream)
    TG_ENV_SWITCH(p->env_id, tr->dtype, CALL)
#undef CALL
}

int tg_rollout_finish(const tg_traj* tr, void* stream) {
    TG_REQUIRE(tr && tr->d_len && tr->d_counters && tr->n > 0, "tg_rollout_finish: bad trajectory");
    hipLaunchKernelGGL(rollout_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tr->d_len, tr->n, tr->d_counters);
    TG_LAUNCH_CHECK("tg_rollout_finish");
    return TG_OK;
}

int tg_rollout_finish_stats_workspace(void) { return 256 * (int)sizeof(double); }

int tg_rollout_finish_stats(const tg_traj* tr, uint64_t* d_rng, double* d_stats, double* d_work, void* stream) {
    TG_REQUIRE(tr && tr->d_len && tr->d_counters && tr->d_rew && tr->n > 0 && tr->horizon > 0, "tg_rollout_finish_stats: bad trajectory");
    TG_REQUIRE(d_stats && d_work, "tg_rollout_finish_stats: null pointer");
    TG_REQUIRE(tr->dtype == TG_F32 || tr->dtype == TG_F64, "tg_rollout_finish_stats: bad dtype %d", tr->dtype);
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)tr->horizon * tr->n;
    const int64_t want = ceil_div(M, 256 * 8);
    const int nb = (int)(want < 256 ? want : 256);
    if (tr->dtype == TG_F64) hipLaunchKernelGGL(reward_partial_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, (const double*)tr->d_rew, M, d_work);
    else hipLaunchKernelGGL(reward_partial_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)tr->d_rew, M, d_work);
    TG_LAUNCH_CHECK("tg_rollout_finish_stats(rewards)");
    hipLaunchKernelGGL(finish_stats_kernel, dim3(1), dim3(256), 0, st, tr->d_len, tr->n, tr->d_counters, (const double*)d_work, nb, d_rng, d_stats);
    TG_LAUNCH_CHECK("tg_rollout_finish_stats");
    return TG_OK;
}

int tg_rng_advance(uint64_t* d_rng, void* stream) {
    TG_REQUIRE(d_rng != nullptr, "tg_rng_advance: null pointer");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_rng);
    TG_LAUNCH_CHECK("tg_rng_advance");
    return TG_OK;
}

}  // extern "C"
