// Fused GPU-resident rollout for float32 policies: the whole T-step loop in ONE persistent launch, every product in
// fp32 (v_mfma_f32_32x32x2_f32) -- the reference's own precision (rollout/rollout_worker.py:19-84 with the fp32
// policies of policies/actor_critic.py:107-138).
//
// The bf16 kernel (fused_rollout.hip) is built for throughput at 10^4..10^5 envs; this one is built for LATENCY at the
// reference's net sizes (128 x 3, 128 x 4, 64-wide nets) and a few thousand envs, where the per-step launch path
// spends its time between kernels (~20 us per time step for 4,096 CartPole envs):
//   * a workgroup owns 32 envs for the whole rollout (many small workgroups: 4,096 envs = 128 CUs busy);
//   * wave w of its H/32 waves owns the 32-feature output tile w of EVERY layer, and its rows of every weight matrix
//     stay in REGISTERS for the whole rollout (H/2 registers per H x H layer): no weight traffic at all after the
//     prologue, neither HBM nor LDS;
//   * transposed form Y^T = W . X^T: the 32 columns of a tile are the 32 envs.  A layer's output tile goes to LDS in
//     groups of 4 consecutive features ([H/4][32 envs][4]), one raw s_barrier, and every wave reads the whole
//     activation vector back as its B operands (lane (env, kh) takes group 2q + kh: ds_read_b128, conflict-free).
//     The k order this implies (step 4q + j multiplies feature 8q + 4kh + j) is folded into the weight registers;
//   * the head (A <= 4 outputs) is a 16-term fp32 dot product per lane on the last accumulators, summed over the
//     2 H/32 partials through LDS in a fixed order;
//   * every wave steps the workgroup's 32 envs redundantly (same instructions, same inputs: same bits), so the env
//     state never has to be broadcast; wave 0 records the trajectory.  Sampling (Philox keyed by global env index
//     and t), dynamics, recording and termination are the code of rollout_step_kernel.
// HBM traffic per env-step is the trajectory record only.
#include "env_dynamics.hpp"

#include <string.h>

namespace tg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SigmaF32 { float v[8]; };

// accumulator start values: rows (r&3) + 8(r>>2) + 4h of a 32-row tile, `b` = tile base + 4h
__device__ static inline f32x16 bias_rows_f32(const float* __restrict__ b) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(b + 8 * q);
        acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
    }
    return acc;
}

__device__ static inline float4 f32r_lds_f4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

__device__ static inline void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // not __syncthreads(): that would also wait for the trajectory stores
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// H = hidden width (64 / 128), NHH = number of H x H layers (hidden layers - 1).
// Tables: `wstream` f32 [H/32 waves][K1/2 + NHH*H/2 registers][64 lanes]; `tab` f32 [(NHH+1)*H biases][4*H head
// weights, rows >= A zero][4 head biases].
template <typename Env, int H, int NHH>
__global__ __launch_bounds__(64 * (H / 32)) void fused_rollout_f32_kernel(
    typename Env::C c, float* __restrict__ obs, float* __restrict__ act, float* __restrict__ rew, uint8_t* __restrict__ mask,
    int32_t* __restrict__ len, int64_t n, int32_t T, int32_t t0, int32_t t1, const float* __restrict__ wstream,
    const float* __restrict__ tab, SigmaF32 sigma, const uint64_t* __restrict__ rng, int64_t env_offset, int32_t agents) {
    constexpr int S = Env::S, A = Env::A, WPW = H / 32;
    constexpr int K1 = (S + 7) / 8 * 8, R1 = K1 / 2, RH = H / 2, RW = R1 + NHH * RH;
    constexpr int NB = (NHH + 1) * H, NTAB = NB + 4 * H + 4;
    static_assert(S <= 32 && A <= 4, "state <= 32 features, <= 4 actions");
    extern __shared__ float lds_f[];
    float* tab_s = lds_f;                                   // NTAB floats
    float* actb = tab_s + NTAB;                             // 2 buffers x [H/4 groups][32 envs][4]
    float* red = actb + 2 * H * 32;                         // 2 buffers x [A][2 WPW partials][32 envs]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int64_t i = (int64_t)blockIdx.x * 32 + col;       // lanes 32..63 shadow lanes 0..31 (they carry the upper k-halves)
    const bool in_range = (i < n) && (h == 0) && (wave == 0);
    const int64_t ic = (i < n) ? i : n - 1;
    const int64_t T1 = (int64_t)T + 1;

    for (int q = threadIdx.x; q < NTAB; q += 64 * WPW) tab_s[q] = tab[q];
    float w1[R1], wh[NHH > 0 ? NHH : 1][RH];
    {
        const float* my = wstream + (int64_t)wave * RW * 64 + lane;
#pragma unroll
        for (int r = 0; r < R1; ++r) w1[r] = my[r * 64];
#pragma unroll
        for (int l = 0; l < NHH; ++l)
#pragma unroll
            for (int r = 0; r < RH; ++r) wh[l][r] = my[(R1 + l * RH + r) * 64];
    }
    float s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = obs[(k * T1 + t0) * n + ic];
    const int32_t len0 = len[ic];
    bool alive = (i < n) && (Env::kBalanceTerminates ? len0 <= 0 : len0 == 0);
    int balanced_steps = Env::kBalanceTerminates ? -len0 : 0;
    __syncthreads();
    int par = 0, rpar = 0;

    for (int32_t t = t0; t < t1; ++t) {
        if (__ballot(alive) == 0ull) break;                  // the same 32 envs in every wave: a uniform exit
        // ---- layer 1: B operands straight from the state registers (feature 8q + 4kh + j at step 4q + j) ----
        f32x16 acc = bias_rows_f32(tab_s + 32 * wave + 4 * h);
#pragma unroll
        for (int q = 0; q < K1 / 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = (8 * q + j < S) ? s[(8 * q + j < S) ? 8 * q + j : 0] : 0.0f;
                const float hi = (8 * q + 4 + j < S) ? s[(8 * q + 4 + j < S) ? 8 * q + 4 + j : 0] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[4 * q + j], h ? hi : lo, acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
        // ---- H x H layers: exchange the activation vector through LDS, multiply with the register-resident rows ----
#pragma unroll
        for (int l = 0; l < NHH; ++l) {
            float* buf = actb + par * (H * 32);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(buf + ((8 * wave + 2 * g + h) * 32 + col) * 4) =
                    float4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            lds_barrier();
            acc = bias_rows_f32(tab_s + (l + 1) * H + 32 * wave + 4 * h);
#pragma unroll
            for (int q = 0; q < H / 8; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(buf + ((2 * q + h) * 32 + col) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[l][4 * q], x.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[l][4 * q + 1], x.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[l][4 * q + 2], x.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[l][4 * q + 3], x.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
            par ^= 1;
        }
        // ---- head: per-lane partial dot products over this lane's 16 features, summed through LDS in a fixed order ----
        float mu[A];
        {
            float* rb = red + rpar * (A * 2 * WPW * 32);
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float* hw = tab_s + NB + k * H + 32 * wave + 4 * h;
                float p = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 w4 = *reinterpret_cast<const float4*>(hw + 8 * g);
                    p = __builtin_fmaf(acc[4 * g], w4.x, p);
                    p = __builtin_fmaf(acc[4 * g + 1], w4.y, p);
                    p = __builtin_fmaf(acc[4 * g + 2], w4.z, p);
                    p = __builtin_fmaf(acc[4 * g + 3], w4.w, p);
                }
                rb[(k * 2 * WPW + 2 * wave + h) * 32 + col] = p;
            }
            lds_barrier();
#pragma unroll
            for (int k = 0; k < A; ++k) {
                float m = tab_s[NB + 4 * H + k];
#pragma unroll
                for (int q = 0; q < 2 * WPW; ++q) m += rb[(k * 2 * WPW + q) * 32 + col];
                mu[k] = m;
            }
            rpar ^= 1;
        }
        // ---- sample, step, record (same arithmetic and RNG keys as rollout_step_kernel) ----
        float a[A];
        {
            uint32_t rnd[4];
            Philox::draw(rng[0], (uint64_t)(env_offset + ic), (uint32_t)t, (uint32_t)rng[1], rnd);
            float eps[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (2 * hh < A) {
                    const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(Philox::u01(rnd[2 * hh])));
                    const float rev = Philox::u01(rnd[2 * hh + 1]);
                    eps[2 * hh] = rad * __builtin_amdgcn_cosf(rev);
                    eps[2 * hh + 1] = rad * __builtin_amdgcn_sinf(rev);
                }
            }
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = rn_add(mu[k], rn_mul(sigma.v[k], eps[k]));
        }
        float o[S], r;
        const StepOut out = Env::step(s, a, c, t + 1, o, r);
        bool ended = out.truncated;
        if constexpr (Env::kBalanceTerminates) {
            balanced_steps = out.balanced ? balanced_steps + 1 : 0;
            ended = ended || (balanced_steps >= c.term_steps);              // terminated, pendulum_env.py:151
        }
        const bool done = any_in_segment(alive && ended, agents) || (t + 1 >= T);
        const bool carry = alive && !done;
        if (in_range) {
#pragma unroll
            for (int k = 0; k < A; ++k) act[((int64_t)k * T + t) * n + i] = alive ? a[k] : 0.0f;
            rew[(int64_t)t * n + i] = alive ? r : 0.0f;
            mask[(int64_t)t * n + i] = alive ? 1 : 0;
#pragma unroll
            for (int k = 0; k < S; ++k) obs[(k * T1 + t + 1) * n + i] = carry ? o[k] : 0.0f;
            if (alive && done) len[i] = t + 1;
        }
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = carry ? o[k] : 0.0f;
        alive = carry;
    }
    if (Env::kBalanceTerminates && in_range && alive) len[i] = -balanced_steps;   // a later segment [t1, ..) picks the count up
}

// The same rollout with SIXTEEN envs per workgroup on v_mfma_f32_16x16x4_f32, for env counts that leave CUs without a workgroup at 32
// envs each (BASELINE configs[1]: 4,096 CartPole envs = 128 workgroups on 256 CUs).  A time step is a chain of dependent products
// -- 2 H^2 x envs flop per H x H layer on the workgroup's one CU -- so halving the envs per workgroup halves the step's latency and
// fills the other half of the chip.  Wave w still owns features [32 w, 32 w + 32) of every layer, as TWO 16-feature tiles (two
// independent accumulator chains, alternating: a dependent 16x16x4 product issues every 40 cycles, an independent one every 32);
// lane (j, g) = (env j, k sub-index g); the activation vector crosses the waves as [H / 4 groups][16 envs][4] and a lane reads
// group 4 q + g for the four steps 4 q .. 4 q + 3 (features 16 q + 4 g + e at step 4 q + e: folded into the weight registers).
// Weight stream: f32 [H/32 waves][K1/2 + NHH*H/2 registers][64 lanes], register tt * (k / 4) + s of lane (i, g) of wave w =
// W[32 w + 16 tt + i][first layer: 4 s + g | H x H: 16 (s >> 2) + 4 g + (s & 3)].  Everything after the head (sampling, dynamics,
// recording, termination) is the code above; lanes 16..63 shadow lanes 0..15.
typedef float f32x4r __attribute__((ext_vector_type(4)));

template <typename Env, int H, int NHH>
__global__ __launch_bounds__(64 * (H / 32)) void fused_rollout_f32x16_kernel(
    typename Env::C c, float* __restrict__ obs, float* __restrict__ act, float* __restrict__ rew, uint8_t* __restrict__ mask,
    int32_t* __restrict__ len, int64_t n, int32_t T, int32_t t0, int32_t t1, const float* __restrict__ wstream,
    const float* __restrict__ tab, SigmaF32 sigma, const uint64_t* __restrict__ rng, int64_t env_offset, int32_t agents) {
    constexpr int S = Env::S, A = Env::A, WPW = H / 32, E = 16;
    constexpr int K1 = (S + 7) / 8 * 8, R1 = K1 / 2, RH = H / 2, RW = R1 + NHH * RH, S1 = K1 / 4, SH = H / 4;
    constexpr int NB = (NHH + 1) * H, NTAB = NB + 4 * H + 4;
    static_assert(S <= 32 && A <= 4, "state <= 32 features, <= 4 actions");
    extern __shared__ float lds_f[];
    float* tab_s = lds_f;                                   // NTAB floats
    float* actb = tab_s + NTAB;                             // 2 buffers x [H/4 groups][16 envs][4]
    float* red = actb + 2 * H * E;                          // 2 buffers x [A][4 WPW partials][16 envs]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int64_t i = (int64_t)blockIdx.x * E + col;
    const bool in_range = (i < n) && (g == 0) && (wave == 0);
    const int64_t ic = (i < n) ? i : n - 1;
    const int64_t T1 = (int64_t)T + 1;

    for (int q = threadIdx.x; q < NTAB; q += 64 * WPW) tab_s[q] = tab[q];
    float w1[2][S1], wh[NHH > 0 ? NHH : 1][2][SH];
    {
        const float* my = wstream + (int64_t)wave * RW * 64 + lane;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < S1; ++r) w1[tt][r] = my[(tt * S1 + r) * 64];
#pragma unroll
        for (int l = 0; l < NHH; ++l)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < SH; ++r) wh[l][tt][r] = my[(R1 + l * RH + tt * SH + r) * 64];
    }
    float s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = obs[(k * T1 + t0) * n + ic];
    const int32_t len0 = len[ic];
    bool alive = (i < n) && (Env::kBalanceTerminates ? len0 <= 0 : len0 == 0);
    int balanced_steps = Env::kBalanceTerminates ? -len0 : 0;
    __syncthreads();
    int par = 0, rpar = 0;
    auto bias4 = [&](const float* b) { const float4 v = *reinterpret_cast<const float4*>(b); return f32x4r{v.x, v.y, v.z, v.w}; };

    for (int32_t t = t0; t < t1; ++t) {
        if (__ballot(alive) == 0ull) break;                  // the same 16 envs in every wave: a uniform exit
        // ---- the step's noise first: it depends on (env, t) only, and its ~100 vector instructions fill the products' latencies ----
        float eps[4];
        {
            uint32_t rnd[4];
            Philox::draw(rng[0], (uint64_t)(env_offset + ic), (uint32_t)t, (uint32_t)rng[1], rnd);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (2 * hh < A) {
                    const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(Philox::u01(rnd[2 * hh])));
                    const float rev = Philox::u01(rnd[2 * hh + 1]);
                    eps[2 * hh] = rad * __builtin_amdgcn_cosf(rev);
                    eps[2 * hh + 1] = rad * __builtin_amdgcn_sinf(rev);
                }
            }
        }
        // ---- layer 1: B operands straight from the state registers (feature 4 s + g at step s) ----
        f32x4r acc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = bias4(tab_s + 32 * wave + 16 * tt + 4 * g);
#pragma unroll
        for (int r = 0; r < S1; ++r) {
            float x = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * r + e < S) x = (g == e) ? s[(4 * r + e < S) ? 4 * r + e : 0] : x;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[tt][r], x, acc[tt], 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = fmaxf(acc[tt][r], 0.0f);
        // ---- H x H layers: exchange the activation vector through LDS, multiply with the register-resident rows ----
#pragma unroll
        for (int l = 0; l < NHH; ++l) {
            float* buf = actb + par * (H * E);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                *reinterpret_cast<float4*>(buf + ((8 * wave + 4 * tt + g) * E + col) * 4) = float4{acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
            lds_barrier();
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[tt] = bias4(tab_s + (l + 1) * H + 32 * wave + 16 * tt + 4 * g);
            // (the next group's operands are requested before this group's eight products: hipcc otherwise reads each group into the
            //  same registers right in front of its products -- eight exposed LDS latencies per layer and step)
            float4 x = f32r_lds_f4(buf + (g * E + col) * 4);
#pragma unroll
            for (int q = 0; q < H / 16; ++q) {
                float4 xn = x;
                if (q + 1 < H / 16) xn = f32r_lds_f4(buf + ((4 * (q + 1) + g) * E + col) * 4);
                const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_sched_barrier(0);               // (the two chains stay interleaved, the read stays in front of the group)
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[l][0][4 * q + e], xe[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[l][1][4 * q + e], xe[e], acc[1], 0, 0, 0);
                }
                x = xn;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tt][r] = fmaxf(acc[tt][r], 0.0f);
            par ^= 1;
        }
        // ---- head: per-lane partial dot products over this lane's 8 features, summed through LDS in a fixed order ----
        float mu[A];
        {
            float* rb = red + rpar * (A * 4 * WPW * E);
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float* hw = tab_s + NB + k * H + 32 * wave + 4 * g;
                float p = 0.0f;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float4 w4 = *reinterpret_cast<const float4*>(hw + 16 * tt);
                    p = __builtin_fmaf(acc[tt][0], w4.x, p);
                    p = __builtin_fmaf(acc[tt][1], w4.y, p);
                    p = __builtin_fmaf(acc[tt][2], w4.z, p);
                    p = __builtin_fmaf(acc[tt][3], w4.w, p);
                }
                rb[(k * 4 * WPW + 4 * wave + g) * E + col] = p;
            }
            lds_barrier();
#pragma unroll
            for (int k = 0; k < A; ++k) {
                float m = tab_s[NB + 4 * H + k];
#pragma unroll
                for (int q = 0; q < 4 * WPW; ++q) m += rb[(k * 4 * WPW + q) * E + col];
                mu[k] = m;
            }
            rpar ^= 1;
        }
        // ---- sample, step, record (same arithmetic and RNG keys as rollout_step_kernel) ----
        float a[A];
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = rn_add(mu[k], rn_mul(sigma.v[k], eps[k]));
        float o[S], r;
        const StepOut out = Env::step(s, a, c, t + 1, o, r);
        bool ended = out.truncated;
        if constexpr (Env::kBalanceTerminates) {
            balanced_steps = out.balanced ? balanced_steps + 1 : 0;
            ended = ended || (balanced_steps >= c.term_steps);              // terminated, pendulum_env.py:151
        }
        const bool done = any_in_segment(alive && ended, agents) || (t + 1 >= T);
        const bool carry = alive && !done;
        if (in_range) {
#pragma unroll
            for (int k = 0; k < A; ++k) act[((int64_t)k * T + t) * n + i] = alive ? a[k] : 0.0f;
            rew[(int64_t)t * n + i] = alive ? r : 0.0f;
            mask[(int64_t)t * n + i] = alive ? 1 : 0;
#pragma unroll
            for (int k = 0; k < S; ++k) obs[(k * T1 + t + 1) * n + i] = carry ? o[k] : 0.0f;
            if (alive && done) len[i] = t + 1;
        }
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = carry ? o[k] : 0.0f;
        alive = carry;
    }
    if (Env::kBalanceTerminates && in_range && alive) len[i] = -balanced_steps;   // a later segment [t1, ..) picks the count up
}

template <template <typename> class EnvT, int H, int NHH>
static int fused_f32_launch(const tg_env_params* p, const tg_traj* tr, const float* wstream, const float* tab, const float* sigma,
                            const uint64_t* rng, int64_t env_offset, int t0, int t1, int block_envs, hipStream_t st) {
    using Env = EnvT<float>;
    constexpr int WPW = H / 32, A = Env::A;
    auto c = Env::C::make(*p);
    SigmaF32 sg;
    memset(&sg, 0, sizeof(sg));
    for (int k = 0; k < A; ++k) sg.v[k] = sigma[k];
    static_assert(sizeof(float) * ((NHH + 1) * H + 4 * H + 4 + 2 * H * 32 + 2 * 4 * 2 * WPW * 32) <= 64 * 1024, "LDS budget");
    if (block_envs == 16) {
        const size_t shmem = sizeof(float) * ((size_t)(NHH + 1) * H + 4 * H + 4 + 2 * H * 16 + 2 * A * 4 * WPW * 16);
        const dim3 grid((unsigned)ceil_div(tr->n, 16));
        hipLaunchKernelGGL((fused_rollout_f32x16_kernel<Env, H, NHH>), grid, dim3(64 * WPW), shmem, st, c, (float*)tr->d_obs, tr->d_act,
                           (float*)tr->d_rew, tr->d_mask, tr->d_len, tr->n, tr->horizon, t0, t1, wstream, tab, sg, rng, env_offset,
                           p->agents);
    } else {
        const size_t shmem = sizeof(float) * ((size_t)(NHH + 1) * H + 4 * H + 4 + 2 * H * 32 + 2 * A * 2 * WPW * 32);
        const dim3 grid((unsigned)ceil_div(tr->n, 32));
        hipLaunchKernelGGL((fused_rollout_f32_kernel<Env, H, NHH>), grid, dim3(64 * WPW), shmem, st, c, (float*)tr->d_obs, tr->d_act,
                           (float*)tr->d_rew, tr->d_mask, tr->d_len, tr->n, tr->horizon, t0, t1, wstream, tab, sg, rng, env_offset,
                           p->agents);
    }
    TG_LAUNCH_CHECK("tg_fused_rollout_f32");
    return TG_OK;
}

template <template <typename> class EnvT>
static int fused_f32_dispatch(int hidden, int n_hh, const tg_env_params* p, const tg_traj* tr, const float* wstream, const float* tab,
                              const float* sigma, const uint64_t* rng, int64_t env_offset, int t0, int t1, int block_envs, hipStream_t st) {
#define TG_F32_CASE(HH, NN) \
    case HH * 10 + NN: return fused_f32_launch<EnvT, HH, NN>(p, tr, wstream, tab, sigma, rng, env_offset, t0, t1, block_envs, st);
    switch (hidden * 10 + n_hh) {
        TG_F32_CASE(64, 0) TG_F32_CASE(64, 1) TG_F32_CASE(64, 2) TG_F32_CASE(64, 3)
        TG_F32_CASE(128, 0) TG_F32_CASE(128, 1) TG_F32_CASE(128, 2) TG_F32_CASE(128, 3)
        default: break;
    }
#undef TG_F32_CASE
    return set_error(TG_ERR_UNSUPPORTED, "tg_fused_rollout_f32: hidden width %d with %d hidden layers is not instantiated "
                     "(widths 64 / 128, 1..4 hidden layers)", hidden, n_hh + 1);
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_fused_rollout_f32_supported(int32_t hidden, int32_t n_hidden_layers) {
    return (hidden == 64 || hidden == 128) && n_hidden_layers >= 1 && n_hidden_layers <= 4;
}

int tg_fused_rollout_f32_block_envs(int64_t n, int32_t agents) {
    // 16 envs per workgroup while that leaves no CU with two workgroups (and a swarm env's bodies fit a 16-lane group), else 32
    return (n <= (int64_t)16 * device_cus() && agents <= 16) ? 16 : 32;
}

int tg_fused_rollout_f32(const tg_env_params* p, const tg_traj* tr, const float* d_wstream, const float* d_tab, int32_t hidden,
                         int32_t n_hidden_layers, int32_t block_envs, const float* sigma, const uint64_t* d_rng, int64_t env_offset,
                         int32_t t_begin, int32_t t_end, void* stream) {
    TG_REQUIRE(p && tr && d_wstream && d_tab && sigma && d_rng, "tg_fused_rollout_f32: null pointer");
    TG_REQUIRE(tr->d_obs && tr->d_act && tr->d_rew && tr->d_mask && tr->d_len, "tg_fused_rollout_f32: null trajectory pointer");
    TG_REQUIRE(tr->dtype == TG_F32, "tg_fused_rollout_f32: float32 trajectories only");
    TG_REQUIRE(tr->n > 0 && tr->horizon == p->max_steps, "tg_fused_rollout_f32: horizon %d != env.max_steps %d", tr->horizon,
               p->max_steps);
    TG_REQUIRE(0 <= t_begin && t_begin <= t_end && t_end <= tr->horizon, "tg_fused_rollout_f32: bad step range [%d, %d)", t_begin,
               t_end);
    TG_REQUIRE(block_envs == 16 || block_envs == 32, "tg_fused_rollout_f32: %d envs per workgroup (16 or 32: the layout of d_wstream)", block_envs);
    TG_REQUIRE(p->agents <= 1 || (p->agents <= block_envs && (p->agents & (p->agents - 1)) == 0 && tr->n % p->agents == 0),
               "tg_fused_rollout_f32: agents=%d must be a power of two <= %d dividing n", p->agents, block_envs);
    if (t_begin == t_end) return TG_OK;
    const int n_hh = n_hidden_layers - 1;
    hipStream_t st = (hipStream_t)stream;
    switch (p->env_id) {
        case TG_ENV_CARTPOLE: return fused_f32_dispatch<CartPoleEnv>(hidden, n_hh, p, tr, d_wstream, d_tab, sigma, d_rng, env_offset, t_begin, t_end, block_envs, st);
        case TG_ENV_QUADPOLE2D: return fused_f32_dispatch<QuadPole2DEnv>(hidden, n_hh, p, tr, d_wstream, d_tab, sigma, d_rng, env_offset, t_begin, t_end, block_envs, st);
        case TG_ENV_QUADPOLE: return fused_f32_dispatch<QuadPoleEnv>(hidden, n_hh, p, tr, d_wstream, d_tab, sigma, d_rng, env_offset, t_begin, t_end, block_envs, st);
        case TG_ENV_PENDULUM: return fused_f32_dispatch<PendulumEnv>(hidden, n_hh, p, tr, d_wstream, d_tab, sigma, d_rng, env_offset, t_begin, t_end, block_envs, st);
        default: return set_error(TG_ERR_UNSUPPORTED, "tg_fused_rollout_f32: env %d is not instantiated", p->env_id);
    }
}

}  // extern "C"
