// Fused GPU-resident rollout: the whole T-step loop of a rollout in ONE persistent launch.
//
// Reference path replaced: rollout/rollout_worker.py:19-84 -- per step one policy forward
// (policies/actor_critic.py:107-138 -> models/neural_network.py:67-77), one sample, one Env.step.
// The unfused path (rollout.py) launches ~9 kernels per time step and streams every activation through
// HBM; here a workgroup owns 256 environments for the whole rollout:
//   * the env state lives in registers (one lane per env, 64 envs per wavefront) and is never re-read;
//   * the actor MLP runs on the matrix cores in TRANSPOSED form, Y^T = W . X^T, with
//     v_mfma_f32_32x32x16_bf16: the 32 lanes of a tile are 32 envs, so an env's activations never leave
//     its wave, and a layer's accumulator tile IS the next layer's B operand (bias-init, ReLU, bf16
//     pack -- no LDS transpose; the k-order permutation this implies is folded into the weight stream);
//   * the weights (bf16, pre-arranged on the host in MFMA A-fragment order, 1 KiB per wave-instruction)
//     stream L2 -> LDS once per workgroup per step and are shared by its 4 waves (and 2 env tiles each);
//   * sampling (Philox keyed by global env index and t, identical to rollout_step_kernel), dynamics,
//     trajectory recording and episode termination follow in the same launch; a wave whose 64 envs have
//     all ended skips its MFMA work, a workgroup whose envs have all ended leaves the time loop.
// HBM traffic per env-step is the trajectory record only (S*4 + A*4 + 4 + 1 B written, nothing read).
#include "env_dynamics.hpp"
#include "mfma_ring.hpp"

#include <stdlib.h>
#include <string.h>

namespace tg {


struct SigmaF { float v[8]; };

constexpr int kCompactEvery = 8;   // time steps between compactions of a workgroup's running envs

// Accumulator start values: rows (r&3) + 8(r>>2) + 4h of a 32-row tile, `b` = tile base + 4h.  The `__restrict__`
// parameter of this inlined function gives its LDS reads alias-scope metadata; without it hipcc makes an LDS read wait
// for EVERY outstanding LDS-DMA (vmcnt(0)), draining the weight ring (it did, in the first-layer and head blocks).
__device__ static inline f32x16 bias_rows(const float* __restrict__ b) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(b + 8 * q);
        acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
    }
    return acc;
}

// First-layer B fragment from a bf16 feature row in LDS: element j of lane half h is feature 16 ks + 8 (j>>2) + 4 h + (j&3);
// `p` = row + 16 ks + 4 h.  (`__restrict__` for the same reason as bias_rows.)
__device__ static inline bf16x8 x_fragment(const unsigned short* __restrict__ p) {
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 8);
    const uint4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

// the trajectory stores of a step are issued behind the blocks in flight; not counting them only makes the wait stricter
#define TG_STAGE_ADVANCE TG_RING_ADVANCE((P - 1) * (KS / WPW))

// NT = env tiles (of 32) per wave, WPW = waves per workgroup.
//   NT = 2, WPW = 4: 64 envs per wave, one lane per env, one wave per SIMD (the X fragments of two tiles fill
//                    the register file), 256 envs per workgroup.
//   NT = 1, WPW = 8: 32 envs per wave (lanes 32..63 only carry the upper k-halves of the MFMA operands), half the
//                    registers, so two waves share a SIMD and fill each other's barrier / LDS / epilogue gaps;
//                    ONE weight ring feeds all 8 waves (256 envs per workgroup).  Measured fastest.
//   NT = 1, WPW = 4: 128 envs per workgroup, two workgroups per CU (twice the weight stream per CU); more
//                    workgroups for small env counts.
template <typename Env, int H, int NT, int WPW>
__global__ __launch_bounds__(64 * WPW, (NT == 2) ? 1 : 2) void fused_rollout_kernel(
    typename Env::C c, float* __restrict__ obs, float* __restrict__ act, float* __restrict__ rew,
    uint8_t* __restrict__ mask, int32_t* __restrict__ len, int64_t n, int32_t T, int32_t t0, int32_t t1,
    const uint4* __restrict__ wfrag, const float* __restrict__ bias, int32_t n_hh, SigmaF sigma,
    const uint64_t* __restrict__ rng, int64_t env_offset, int32_t agents) {
    constexpr int S = Env::S, A = Env::A, MT = H / 32, KS = H / 16;
    constexpr int D = (NT == 1 && WPW == 4) ? 3 : 4, P = D - 1;   // ring slots, blocks in flight (LDS is shared by 2 workgroups at 4 x NT=1)
    static_assert(KS % WPW == 0, "every wave moves the same number of 1-KiB pieces per block");
    static_assert(S <= 32 && A <= 4, "state must fit one padded 32-feature tile; actions the first 4 head rows");
    // the compaction staging ([32 * WPW][S + 1] floats) aliases xs (WPW * 64 * 32 bf16 = 4096 * WPW bytes);
    // per wave: 32 records of S + 1 floats in 64 * 32 bf16; compaction only runs with NT == 1
    static_assert(NT != 1 || 32 * (S + 1) * 4 <= 64 * 32 * 2, "compaction records must fit the input staging area they alias");
    extern __shared__ uint4 lds[];
    uint4* ring = lds;                                                  // D * KS * 64 uint4
    float* bias_s = reinterpret_cast<float*>(lds + D * KS * 64);        // (n_hh + 2) * H floats
    unsigned short* xs = reinterpret_cast<unsigned short*>(bias_s + (n_hh + 2) * H);   // WPW waves * 64 rows * 32 bf16
    int* flags = reinterpret_cast<int*>(xs + WPW * 64 * 32);            // WPW ints: per-wave "some env alive"

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    // NT = 2: lane == env.  NT = 1: lanes 0..31 own the wave's 32 envs, lanes 32..63 shadow them (never recorded).
    const int64_t wg_base = (int64_t)blockIdx.x * (32 * NT * WPW);
    int64_t i = (NT == 2) ? wg_base + threadIdx.x : wg_base + wave * 32 + col;
    bool in_range = (i < n) && (NT == 2 || h == 0);
    int64_t ic = (i < n) ? i : n - 1;
    const int64_t T1 = (int64_t)T + 1;

    for (int q = threadIdx.x; q < (n_hh + 2) * H; q += 64 * WPW) bias_s[q] = bias[q];

    float s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = obs[(k * T1 + t0) * n + ic];
    // len == 0 while the episode runs; Pendulum keeps -(consecutive balanced steps) there (env_dynamics.hpp, StepOut)
    const int32_t len0 = len[ic];
    bool alive = in_range && (Env::kBalanceTerminates ? len0 <= 0 : len0 == 0);
    int balanced_steps = Env::kBalanceTerminates ? -len0 : 0;
    unsigned short* my_x = xs + (wave * 64 + lane) * 32;
    // zero the padding features once (columns S..31 never change)
#pragma unroll
    for (int k = S; k < 32; ++k) my_x[k] = 0;
    const int n_blocks = n_hh * MT + 2;
    __syncthreads();                                  // bias table and padding are in place; no DMA outstanding yet
    int pre_pos = 0, pre_slot = 0, cur_slot = 0;
    for (int b0 = 0; b0 < P; ++b0) {                  // blocks 0..P-1 in flight before the first step
        ring_dma_block<KS, WPW>(wfrag + (int64_t)pre_pos * KS * 64, ring + pre_slot * KS * 64, wave, lane);
        pre_pos = (pre_pos + 1 == n_blocks) ? 0 : pre_pos + 1;
        pre_slot = (pre_slot + 1 == D) ? 0 : pre_slot + 1;
    }

    for (int32_t t = t0; t < t1; ++t) {
        bool wave_alive;
        if (NT == 1 && agents <= 1 && !Env::kBalanceTerminates && t > t0 && ((t - t0) % kCompactEvery) == 0) {
            // ---- compaction: the workgroup's running envs move to its lowest lanes, so the waves that hold only
            // ended envs (and, between compactions, fill up with them) stop doing MFMA work.  An env keeps its
            // identity `i`: recording and the Philox key follow the env, not the lane.
            const bool own = alive && (h == 0);
            const unsigned long long ball = __ballot(own);
            const int cnt = __popcll(ball);
            const int rank = __popcll(ball & ((1ull << lane) - 1ull));
            if (lane == 0) flags[wave] = cnt;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int offset = 0, total = 0;
#pragma unroll
            for (int w = 0; w < WPW; ++w) {
                const int cw = flags[w];
                offset += (w < wave) ? cw : 0;
                total += cw;
            }
            if (total == 0) break;                            // every env of this workgroup has ended
            float* stage = reinterpret_cast<float*>(xs);      // [32*WPW][S+1] floats; xs is rewritten every step anyway
            if (own) {
                float* d = stage + (offset + rank) * (S + 1);
#pragma unroll
                for (int k = 0; k < S; ++k) d[k] = s[k];
                d[S] = __int_as_float((int)(i - wg_base));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int slot = wave * 32 + col;
            if (slot < total) {
                const float* d = stage + slot * (S + 1);
#pragma unroll
                for (int k = 0; k < S; ++k) s[k] = d[k];
                i = wg_base + __float_as_int(d[S]);
                alive = true;
                in_range = (h == 0);
                ic = i;
            } else {
                alive = false;
                in_range = false;                             // an empty slot records nothing
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // staging reads are done before xs is rewritten
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = S; k < 32; ++k) my_x[k] = 0;         // the staging overwrote the zero padding of the feature rows
            wave_alive = __ballot(alive) != 0ull;
        } else {
            wave_alive = __ballot(alive) != 0ull;
            // leave the time loop when every env of this workgroup has ended (raw barrier: keeps the DMA ring in flight)
            if (lane == 0) flags[wave] = wave_alive ? 1 : 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int any_alive = 0;
#pragma unroll
            for (int w = 0; w < WPW; ++w) any_alive |= flags[w];
            if (any_alive == 0) break;
        }

        // ---- layer-1 input: this wave's 64 states as bf16 rows in LDS, read back in B-fragment order ----
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const __bf16 b = (__bf16)s[k];
            my_x[k] = __builtin_bit_cast(unsigned short, b);
        }
        bf16x8 xin[NT][KS], xout[NT][KS];
#pragma unroll
        for (int tile = 0; tile < NT; ++tile) {
            const unsigned short* row = xs + (wave * 64 + tile * 32 + col) * 32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xin[tile][ks] = x_fragment(row + 16 * ks + 4 * h);
        }

        // ---- layer 1: [H x 32] . [32 x 64 envs]; one block holds all MT output tiles (2 k-steps each) ----
        {
            TG_STAGE_ADVANCE
            if (wave_alive) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int tile = 0; tile < NT; ++tile) {
                        f32x16 acc = bias_rows(bias_s + 32 * mt + 4 * h);
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const bf16x8 a = __builtin_bit_cast(bf16x8, cur[(mt * 2 + ks) * 64 + lane]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[tile][ks], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int sh = 0; sh < 2; ++sh) {
                            xout[tile][2 * mt + sh] = relu_pack_bf16(acc[8 * sh], acc[8 * sh + 1], acc[8 * sh + 2], acc[8 * sh + 3],
                                                                     acc[8 * sh + 4], acc[8 * sh + 5], acc[8 * sh + 6], acc[8 * sh + 7]);
                        }
                    }
                }
            }
#pragma unroll
            for (int tile = 0; tile < NT; ++tile)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xin[tile][ks] = xout[tile][ks];
        }
        // ---- hidden H x H layers: blocks alternate odd (slot 1) / even (slot 0) ----
        for (int l = 0; l < n_hh; ++l) {
            const float* bl = bias_s + (l + 1) * H;
            auto tile_gemm = [&](const uint4* __restrict__ cur, const int mt) {
                if (wave_alive) {
#pragma unroll
                    for (int tile = 0; tile < NT; ++tile) {
                        f32x16 acc = bias_rows(bl + 32 * mt + 4 * h);
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[tile][ks], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int sh = 0; sh < 2; ++sh) {
                            xout[tile][2 * mt + sh] = relu_pack_bf16(acc[8 * sh], acc[8 * sh + 1], acc[8 * sh + 2], acc[8 * sh + 3],
                                                                     acc[8 * sh + 4], acc[8 * sh + 5], acc[8 * sh + 6], acc[8 * sh + 7]);
                        }
                    }
                }
            };
#pragma unroll
            for (int mp = 0; mp < MT / 2; ++mp) {
                {
                    TG_STAGE_ADVANCE
                    tile_gemm(cur, 2 * mp);
                }
                {
                    TG_STAGE_ADVANCE
                    tile_gemm(cur, 2 * mp + 1);
                }
            }
#pragma unroll
            for (int tile = 0; tile < NT; ++tile)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xin[tile][ks] = xout[tile][ks];
        }
        // ---- head: 32 padded output rows (the first A are the action means), no activation ----
        float mu[A];
        {
            TG_STAGE_ADVANCE
            const float* bl = bias_s + (n_hh + 1) * H;
            f32x16 acc2[NT];
#pragma unroll
            for (int tile = 0; tile < NT; ++tile) {
                f32x16 acc = bias_rows(bl + 4 * h);
                if (wave_alive) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, cur[ks * 64 + lane]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[tile][ks], acc, 0, 0, 0);
                    }
                }
                acc2[tile] = acc;
            }
            // rows 0..3 of the head tile sit in registers 0..3 of the h == 0 lanes; env (32*tile + col) is lane
            // 32*tile + col, so tile 1's means cross from lane col to lane col + 32
#pragma unroll
            for (int k = 0; k < A; ++k) {
                if constexpr (NT == 2) {
                    const float other = __shfl(acc2[1][k], col, 64);
                    mu[k] = (lane < 32) ? acc2[0][k] : other;
                } else {
                    mu[k] = acc2[0][k];            // lanes 0..31; the shadow lanes' values are never recorded
                }
            }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <template <typename> class EnvT, int H, int NT, int WPW>
static int fused_launch(const tg_env_params* p, const tg_traj* tr, const void* wfrag, const float* bias, int n_hh,
                        const float* sigma, const uint64_t* rng, int64_t env_offset, int t0, int t1, hipStream_t st) {
    using Env = EnvT<float>;
    constexpr int KS = H / 16;
    auto c = Env::C::make(*p);
    SigmaF sg;
    memset(&sg, 0, sizeof(sg));
    for (int k = 0; k < Env::A; ++k) sg.v[k] = sigma[k];
    const size_t shmem = (size_t)((NT == 1 && WPW == 4) ? 3 : 4) * KS * 1024 + (size_t)(n_hh + 2) * H * sizeof(float) +
                         (size_t)WPW * 64 * 32 * 2 + 4 * WPW;
    auto kern = fused_rollout_kernel<Env, H, NT, WPW>;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_fused_rollout")) return rc;
    const dim3 grid((unsigned)ceil_div(tr->n, 32 * NT * WPW));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WPW), shmem, st, c, (float*)tr->d_obs, tr->d_act, (float*)tr->d_rew, tr->d_mask,
                       tr->d_len, tr->n, tr->horizon, t0, t1, (const uint4*)wfrag, bias, n_hh, sg, rng, env_offset, p->agents);
    TG_LAUNCH_CHECK("tg_fused_rollout");
    return TG_OK;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_fused_rollout(const tg_env_params* p, const tg_traj* tr, const void* d_wfrag, const float* d_bias, int32_t hidden,
                     int32_t n_hidden_layers, const float* sigma, const uint64_t* d_rng, int64_t env_offset, int32_t t_begin,
                     int32_t t_end, void* stream) {
    TG_REQUIRE(p && tr && d_wfrag && d_bias && sigma && d_rng, "tg_fused_rollout: null pointer");
    TG_REQUIRE(tr->d_obs && tr->d_act && tr->d_rew && tr->d_mask && tr->d_len, "tg_fused_rollout: null trajectory pointer");
    TG_REQUIRE(tr->dtype == TG_F32, "tg_fused_rollout: float32 trajectories only");
    TG_REQUIRE(tr->n > 0 && tr->horizon == p->max_steps, "tg_fused_rollout: horizon %d != env.max_steps %d", tr->horizon,
               p->max_steps);
    TG_REQUIRE(0 <= t_begin && t_begin <= t_end && t_end <= tr->horizon, "tg_fused_rollout: bad step range [%d, %d)", t_begin,
               t_end);
    TG_REQUIRE(n_hidden_layers >= 1 && n_hidden_layers <= 16, "tg_fused_rollout: %d hidden layers unsupported", n_hidden_layers);
    TG_REQUIRE(p->agents <= 1 || (p->agents <= 32 && (p->agents & (p->agents - 1)) == 0 && tr->n % p->agents == 0),
               "tg_fused_rollout: agents=%d must be a power of two <= 32 dividing n", p->agents);
    if (t_begin == t_end) return TG_OK;
    const int n_hh = n_hidden_layers - 1;
    hipStream_t st = (hipStream_t)stream;
    // variant: 0 = NT 1 x 8 waves (default), 1 = NT 1 x 4 waves (two workgroups per CU; default below 32,768 envs, where
    // it gives twice as many workgroups).  (NT 2 x 4 waves was measured slower at every size, DESIGN_HISTORY: not instantiated.)
    const int variant = tr->n < 32768 ? 1 : 0;
#define CALL(E, HH)                                                                                                       \
    (variant == 0 ? fused_launch<E, HH, 1, 8>(p, tr, d_wfrag, d_bias, n_hh, sigma, d_rng, env_offset, t_begin, t_end, st) \
                  : fused_launch<E, HH, 1, 4>(p, tr, d_wfrag, d_bias, n_hh, sigma, d_rng, env_offset, t_begin, t_end, st))
    switch (p->env_id * 1000 + hidden) {
        case TG_ENV_CARTPOLE * 1000 + 128: return CALL(CartPoleEnv, 128);
        case TG_ENV_CARTPOLE * 1000 + 256: return CALL(CartPoleEnv, 256);
        case TG_ENV_QUADPOLE2D * 1000 + 128: return CALL(QuadPole2DEnv, 128);
        case TG_ENV_QUADPOLE2D * 1000 + 256: return CALL(QuadPole2DEnv, 256);
        case TG_ENV_QUADPOLE * 1000 + 128: return CALL(QuadPoleEnv, 128);
        case TG_ENV_QUADPOLE * 1000 + 256: return CALL(QuadPoleEnv, 256);
        case TG_ENV_PENDULUM * 1000 + 128: return CALL(PendulumEnv, 128);
        case TG_ENV_PENDULUM * 1000 + 256: return CALL(PendulumEnv, 256);
        default:
            return set_error(TG_ERR_UNSUPPORTED, "tg_fused_rollout: env %d with hidden width %d is not instantiated "
                             "(widths 128 and 256)", p->env_id, hidden);
    }
#undef CALL
}

}  // extern "C"
