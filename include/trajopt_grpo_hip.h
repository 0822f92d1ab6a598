/*
 * trajopt_grpo_hip.h -- C ABI of the MI355X (gfx950) rollout / returns / loss kernels.
 *
 * The reference (Dyllon-Preston/trajopt-grpo @ 2025-06-13) is pure Python and has no
 * FFI of its own; its drop-in seam is the duck-typed Python surface of SURVEY.md 8(b).
 * This library is what that surface's GPU implementation binds (ctypes, see
 * INTEGRATION.md): each entry point names the reference code whose arithmetic it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (hipMalloc / torch CUDA storage);
 *   - `stream` is a hipStream_t passed as void*; all calls only ENQUEUE work on it
 *     (no allocation, no synchronisation: they are hipGraph-capturable);
 *   - return value: 0 = ok, negative = error (TG_ERR_*); tg_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI;
 *   - struct-of-arrays layouts, env index fastest:
 *       state      [S][ld]            component-major, ld >= n
 *       trajectory obs [S][T+1][n], act [A][T][n], rew [T][n], mask [T][n] (u8),
 *       len [n] (i32).  Slot t of obs is the observation BEFORE action t
 *       (rollout/rollout_worker.py:53); slot t+1 is written only while the
 *       episode is still running, so padding stays zero (rollout_worker.py:37-41).
 *   - flat env index n = g*E + e  (buffers/rollout_buffer.py:85-89).
 */
#ifndef TRAJOPT_GRPO_HIP_H
#define TRAJOPT_GRPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 12

enum { TG_OK = 0, TG_ERR_ARG = -1, TG_ERR_HIP = -2, TG_ERR_UNSUPPORTED = -3 };

/* environments (environments/cartpole_env.py, environments/quadrotor_env.py) */
enum { TG_ENV_CARTPOLE = 0, TG_ENV_QUADPOLE2D = 1, TG_ENV_QUADPOLE = 2, TG_ENV_QUADROTOR12 = 3, TG_ENV_PENDULUM = 4 };
/* arithmetic type of the environment state / trajectory */
enum { TG_F32 = 0, TG_F64 = 1 };

/* Physical parameters.  p[] meaning per env (defaults = the reference constructors):
 *  CARTPOLE   (cartpole_env.py:7-16):  p0 masscart, p1 masspole, p2 length, p3 gravity
 *  QUADPOLE2D (quadrotor_env.py:875-895): p0 mq, p1 mp, p2 I, p3 Lq, p4 Lp, p5 gravity,
 *                                         p6 bound (|x|,|z| <= p6), p7 balance_radius
 *  QUADPOLE   (quadrotor_env.py:362-382): p0 mass, p1 load_mass, p2 gravity, p3 tether,
 *                                         p4 Ixx, p5 Iyy, p6 Izz, p7 torque_constant, p8 arm,
 *                                         p9 bound
 *  QUADROTOR12(quadrotor_env.py:9-16):   p0 mass, p1 arm_length, p2 Ixx, p3 Iyy, p4 Izz,
 *                                         p5 torque_constant, p6 gravity
 *  PENDULUM   (pendulum_env.py:8-16):    p0 mass, p1 length, p2 gravity, p3 swingup (0 / 1); p4 is filled by
 *                                         tg_env_finalize_params: the number of CONSECUTIVE balanced steps after which
 *                                         the float-accumulated `_time_balanced` first exceeds 5 s (:135, :151).
 *                                         Pendulum is the one env whose episodes TERMINATE (balanced for > 5 s); in a
 *                                         rollout the running count lives in d_len as a negative number.
 * agents: rollout kernels only.  With agents = k > 1, every k consecutive env slots form ONE environment of k
 *  bodies that share the policy and terminate together: when any body truncates, all k stop at that step
 *  (segmented wavefront ballot).  The reference's QuadrotorSwarm is an empty subclass (quadrotor_env.py:185-186),
 *  so this semantics is defined by this build and has no oracle beyond k = 1 == the plain env.
 * time_trunc_step: CartPole and Pendulum -- first step count at which the reference's
 *  float-accumulated `_time > max_time` fires (cartpole_env.py:168, pendulum_env.py:150); filled by
 *  tg_env_default_params / tg_env_finalize_params. */
typedef struct tg_env_params {
    int32_t env_id;
    int32_t max_steps;
    int32_t time_trunc_step;
    int32_t agents;          /* swarm: agents per env (power of two <= 64), 0/1 = single-body env */
    double  timestep;
    double  p[12];
} tg_env_params;

/* GPU-resident trajectory of one rollout (this rank's shard). */
typedef struct tg_traj {
    void*     d_obs;      /* real [S][T+1][n] */
    float*    d_act;      /* f32  [A][T][n]   (policies emit float32: actor_critic.py:138) */
    void*     d_rew;      /* real [T][n] */
    uint8_t*  d_mask;     /* u8   [T][n] */
    int32_t*  d_len;      /* i32  [n]; <= 0 while the episode is running (0; Pendulum: -balanced steps), else its length */
    uint64_t* d_counters; /* u64  [4]: [0] env-steps executed (= sum of mask), [1] episodes ended */
    int64_t   n;
    int32_t   horizon;    /* T = env.max_steps */
    int32_t   dtype;      /* TG_F32 | TG_F64 */
} tg_traj;

const char* tg_last_error(void);
int  tg_abi_version(void);

int  tg_env_dims(int env_id, int* obs_dim, int* act_dim);
int  tg_env_default_params(int env_id, int max_steps, tg_env_params* out);
/* recompute derived fields (time_trunc_step) after the caller edited max_steps/timestep */
int  tg_env_finalize_params(tg_env_params* p);

/* reset(): draw initial states with the reference's distributions
 * (cartpole_env.py:102-119, quadrotor_env.py:530-576, :930-961) from a counter-based
 * Philox4x32-10 stream keyed by (seed, stream_id, (key_offset + i) / key_div): with
 * key_div = E all E episodes of a group share one draw (the `restart=True` semantics of
 * rollout_worker.py:70-71); key_offset = this shard's first global env index, so results
 * do not depend on how envs are sharded over GPUs.  Writes d_state[S][ld], columns 0..n-1. */
int  tg_env_reset(const tg_env_params* p, int dtype, void* d_state, int64_t ld, int64_t n,
                  uint64_t seed, uint64_t stream_id, int64_t key_offset, int64_t key_div,
                  void* stream);

/* step(): one Env.step for n independent envs on SoA state (may be in place:
 * d_next == d_state).  cartpole_env.py:138-182, quadrotor_env.py:625-713, :1132-1223.
 * d_action f32 [A][ld_a]; d_steps i32 [n] in/out (count before -> after);
 * d_time_balanced real [n] in/out or NULL; d_reward real [n]; d_truncated u8 [n]. */
int  tg_env_step(const tg_env_params* p, int dtype, const void* d_state, int64_t ld,
                 const float* d_action, int64_t ld_a, void* d_next, int64_t ld_next,
                 int32_t* d_steps, void* d_time_balanced, void* d_reward,
                 uint8_t* d_truncated, int64_t n, void* stream);

/* Quadrotor._dynamics (quadrotor_env.py:113-169): pure explicit-Euler map, raw control
 * real [4][ld_c]; state real [12][ld]. */
int  tg_quadrotor12_dynamics(const tg_env_params* p, int dtype, const void* d_state, int64_t ld,
                             const void* d_control, int64_t ld_c, void* d_next, int64_t ld_next,
                             int64_t n, void* stream);

/* ---- GPU-resident rollout (replaces RolloutManager.rollout / RolloutWorker.run_episodes,
 *      rollout/rollout_manager.py:85-125, rollout/rollout_worker.py:19-84) ---- */

/* zero the trajectory + counters (padding must be zero: rollout_worker.py:37-41).  Clears ALL of obs, slot 0
 * included: write the initial states (tg_env_reset into slot 0, or a copy) after this call. */
int  tg_rollout_begin(const tg_traj* tr, int obs_dim, int act_dim, void* stream);

/* fused time step t for all n envs:
 *   action source: d_mean != NULL  -> a = mean[i][:] + sigma * eps, eps ~ N(0,I) from
 *                                     Philox keyed (seed, stream_id, env_offset + i, t)
 *                                     (MultivariateNormal(mean, diag(sigma^2)).sample(),
 *                                     actor_critic.py:131-133); action recorded in act[:,t,i];
 *                  d_mean == NULL  -> teacher forcing: act[:,t,i] is read (parity runs);
 *   then Env.step, reward/mask/len recording and episode termination.  A wavefront whose 64
 *   envs have all ended (ballot == 0) exits before touching memory; env-steps are counted
 *   with one atomic per wavefront (popcount of the ballot).
 * d_rng: device u64[2] = {seed, stream_id} (read by the kernel so a captured graph can be
 * replayed with a fresh stream id).  sigma: host float[A] = sqrt(diag(cov)). */
int  tg_rollout_step(const tg_env_params* p, const tg_traj* tr, int32_t t, const float* d_mean,
                     int64_t mean_row_stride, const float* sigma, const uint64_t* d_rng,
                     int64_t env_offset, void* stream);

/* Teacher-forced steps [t_begin, t_end) in ONE launch: tg_rollout_step(d_mean = NULL) for every t of the range, the state kept in
 * registers between steps (a wavefront owns 64 envs for the whole range) -- per env-step the recorded action is read and the next
 * observation, reward and mask byte are written, nothing else.  Bit-identical to the per-step launches. */
int  tg_rollout_forced(const tg_env_params* p, const tg_traj* tr, int32_t t_begin, int32_t t_end, void* stream);

/* counters[0] = sum of episode lengths (= env-steps executed = sum of mask),
 * counters[1] = episodes ended.  rollout/rollout_worker.py:67-68 */
int  tg_rollout_finish(const tg_traj* tr, void* stream);

/* tg_rollout_finish + tg_rng_advance (d_rng may be NULL) + the statistics Rollout_Buffer.sample() reads
 * (buffers/rollout_buffer.py:70: the average episode return), in two launches: d_stats f64 [3] = {sum of all rewards (fixed
 * summation order: deterministic), n, sum of episode lengths}.  d_work: tg_rollout_finish_stats_workspace() bytes. */
int  tg_rollout_finish_stats_workspace(void);
int  tg_rollout_finish_stats(const tg_traj* tr, uint64_t* d_rng, double* d_stats, double* d_work, void* stream);

/* The whole step range [t_begin, t_end) of a rollout in ONE persistent launch: actor MLP on the matrix cores
 * (bf16 weights, fp32 accumulate), sampling, Env.step, recording and termination, with the env state held in
 * registers.  Same results contract as tg_rollout_step in sampling mode (same Philox keys); the means differ
 * from the GEMM path only by bf16/fp32 summation order.  The actor must be Linear(S,H) ReLU [Linear(H,H) ReLU]*
 * Linear(H,A) with H in {128, 256}, S <= 32, A <= 4.
 *   d_wfrag: bf16 weights in MFMA A-fragment order (see mlp.fragment_stream): blocks of H/16 KiB --
 *            first layer (all output tiles), then one block per 32-row output tile of every H x H layer, then the
 *            head padded to 32 rows;  d_bias: f32 [(n_hidden_layers + 1)][H] (head row padded with zeros). */
int  tg_fused_rollout(const tg_env_params* p, const tg_traj* tr, const void* d_wfrag, const float* d_bias,
                      int32_t hidden, int32_t n_hidden_layers, const float* sigma, const uint64_t* d_rng,
                      int64_t env_offset, int32_t t_begin, int32_t t_end, void* stream);

/* The same for float32 policies (the reference's own precision; rollout/rollout_worker.py:19-84): every product in
 * fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), one workgroup per 32 envs, the weights register-resident for the
 * whole rollout -- built for latency at the reference's net sizes and a few thousand envs.  The means differ from the
 * GEMM path only by fp32 summation order.  Actor: Linear(S,H) ReLU [Linear(H,H) ReLU]^(n_hidden_layers-1) Linear(H,A),
 * H in {64, 128}, 1..4 hidden layers, S <= 32, A <= 4 (tg_fused_rollout_f32_supported()).
 *   block_envs: 32, or 16 -- the same rollout with sixteen envs per workgroup on v_mfma_f32_16x16x4_f32, for env counts that leave
 *              CUs without a workgroup at 32 (BASELINE configs[1]: 4,096 envs = 128 workgroups on 256 CUs): a time step is a chain
 *              of dependent products on the workgroup's one CU, so half the envs are half the step's latency.
 *              tg_fused_rollout_f32_block_envs(n, agents) says which one the launch should use; it decides the layout of d_wstream.
 *   d_wstream: f32 [H/32 waves][K1/2 + (n_hidden_layers-1)*H/2 registers][64 lanes], K1 = S rounded up to 8, first layer then
 *              the H x H layers (zero beyond the matrix; trajopt-grpo_amd/mlp.py `RegisterStreamF32`):
 *              block_envs 32: register 4q + j of lane (m, kh) of wave w is W[32w + m][8q + 4kh + j];
 *              block_envs 16: register tt * (k / 4) + s of lane (i, g) of wave w is W[32w + 16 tt + i][first layer: 4 s + g;
 *                             H x H: 16 (s >> 2) + 4 g + (s & 3)]   (k = the layer's padded input width, tt = 0, 1);
 *   d_tab:     f32 [(n_hidden_layers)*H hidden biases][4*H head weights, rows >= A zero][4 head biases]. */
int  tg_fused_rollout_f32_supported(int32_t hidden, int32_t n_hidden_layers);
int  tg_fused_rollout_f32_block_envs(int64_t n, int32_t agents);
int  tg_fused_rollout_f32(const tg_env_params* p, const tg_traj* tr, const float* d_wstream, const float* d_tab,
                          int32_t hidden, int32_t n_hidden_layers, int32_t block_envs, const float* sigma, const uint64_t* d_rng,
                          int64_t env_offset, int32_t t_begin, int32_t t_end, void* stream);

/* d_rng[1] += 1 (enqueued; one thread) */
int  tg_rng_advance(uint64_t* d_rng, void* stream);

/* ---- returns / advantages (algorithms/grpo.py:66-74,110-115; algorithms/ppo.py:93-139) ---- */

/* reward-to-go reverse scan, fp32, bit-for-bit the reference recurrence:
 *   R[T-1] = r[T-1] m[T-1];  R[t] = r[t] m[t] + (gamma R[t+1]) m[t+1]      [T][n] layout */
int  tg_rtg_scan(const float* d_rew, const uint8_t* d_mask, float gamma, float* d_rtg,
                 int64_t n, int32_t T, void* stream);

/* GAE branch (ppo.py:112-124): adv and ret = values + adv */
int  tg_gae_scan(const float* d_rew, const float* d_values, const uint8_t* d_mask, float gamma,
                 float lam, float* d_adv, float* d_ret, int64_t n, int32_t T, void* stream);

/* masked moments per group of `group_size` consecutive envs (all T steps):
 * d_moments f64 [n/group_size][3] = {count, sum, sum of squares}; deterministic
 * (no float atomics).  d_work: f64 [3*n] scratch. */
int  tg_masked_moments(const float* d_x, const uint8_t* d_mask, int64_t n, int32_t T,
                       int64_t group_size, double* d_moments, double* d_work, void* stream);

/* mode 0 (GRPO, grpo.py:115): out = (x - mean_g) / std_g          (unbiased std, eps inside std)
 * mode 1 (PPO,  ppo.py:138-139): out = (x - mean_g) / (std_g + 1e-8)
 * masked-out entries are written as 0. */
int  tg_group_normalize(const float* d_x, const uint8_t* d_mask, const double* d_moments,
                        int mode, float* d_out, int64_t n, int32_t T, int64_t group_size,
                        void* stream);

/* ---- policy log-prob + fused clipped-surrogate loss (forward + backward) ----
 * Gaussian policy with fixed diagonal covariance (actor_critic.py:100-103,159-160):
 *   logp = -1/2 sum_k (a_k - mu_k)^2 / var_k - k/2 ln 2pi - 1/2 sum_k ln var_k
 * rows are samples; mean is row-major [M][A] (row stride given); act element (i,k) is at
 * d_act[i*act_row_stride + k*act_col_stride]. */
int  tg_gaussian_logp(const float* d_mean, int64_t mean_row_stride, const float* d_act,
                      int64_t act_row_stride, int64_t act_col_stride, const float* var,
                      int act_dim, float* d_logp, int64_t M, void* stream);

typedef struct tg_loss_args {
    const float*   d_mean;  int64_t mean_row_stride;            /* [M][A] policy means */
    const float*   d_act;   int64_t act_row_stride, act_col_stride;
    const float*   d_logp_old;                                  /* [M] */
    const float*   d_adv;                                       /* [M] */
    const float*   d_value;                                     /* [M] or NULL (GRPO) */
    const float*   d_ret;                                       /* [M] or NULL (GRPO) */
    const uint8_t* d_mask;                                      /* [M] or NULL = all rows valid */
    const float*   d_norm;   /* NULL, or device f32[4] {adv_mean, adv_inv_std, ret_mean, ret_inv_std}
                                applied on the fly: x <- (x - mean) * inv_std   (ppo.py:138-139) */
    float   var[8];          /* diag(cov) */
    int32_t act_dim;
    float   epsilon;         /* clip range */
    float   surr_coef;       /* d(total)/d(sum_i min(rho A, clip(rho) A)):
                                GRPO +1/G (descent on J as written, grpo.py:137-145);
                                PPO  -1/n_valid (ppo.py:165) */
    float   critic_coef;     /* PPO: c1/n_valid  (ppo.py:169,179); 0 for GRPO */
    float   kl_coef;         /* PPO: kl_coeff/n_valid  (ppo.py:175-176); 0 for GRPO */
    float*  d_grad_mean;     /* [M][A] row-major (row stride A): d total / d mean */
    float*  d_grad_value;    /* [M] or NULL */
    double* d_sums;          /* f64[4]: sum surrogate, sum (V-R)^2, sum exp(lp_old)(lp_old-lp), #valid */
    double* d_work;          /* f64[4*tg_loss_work_blocks()] scratch */
    int64_t M;
    const float* d_coef;     /* NULL, or device f32[3] {surr_coef, critic_coef, kl_coef}: used INSTEAD of the three host fields
                                (tg_ppo_norm's output + 4: PPO's 1 / n_valid stays on the device) */
} tg_loss_args;

int  tg_loss_work_blocks(void);
int  tg_surrogate_loss(const tg_loss_args* a, void* stream);

/* ---- MLP backward glue (models/neural_network.py:67-77 under torch autograd in the reference) ----
 * dZ = dA * (A > 0) in place (A = post-ReLU activations) fused with the bias-gradient column sums:
 * d_partial f32 [tg_relu_bwd_bias_blocks()][cols]; the caller sums it over axis 0.
 * Row-major [rows][cols]; bf16 needs cols % 8 == 0, f32 cols % 4 == 0. */
int  tg_relu_bwd_bias_blocks(void);
int  tg_relu_bwd_bias(void* d_dA, const void* d_A, int64_t rows, int32_t cols, int32_t is_bf16,
                      float* d_partial, void* stream);

/* The head's backward-data product fused into the top hidden layer's ReLU backward:
 *   dZ[r][c] = (sum_{k<act_dim} dout[r][k] * Whead[k][c]) * (A[r][c] > 0),  d_partial as tg_relu_bwd_bias.
 * d_dout f32 [rows][act_dim] (contiguous), d_whead f32 [act_dim][cols] (the head's master weights),
 * d_act / d_dz [rows][cols] bf16 (is_bf16) or f32; cols % 8 == 0.
 * d_maskbits (optional, bf16, cols 128 / 256): the layer's ReLU mask bits from tg_mlp_forward_chain (cols / 8 bytes
 * per row) are read instead of d_act (which may then be NULL): 4 B instead of 16 B per thread. */
int  tg_head_bwd_relu_bias(const float* d_dout, int32_t act_dim, const float* d_whead, const void* d_act,
                           const void* d_maskbits, void* d_dz, int64_t rows, int32_t cols, int32_t is_bf16,
                           float* d_partial, void* stream);

/* Bias gradients: out[v][c] += sum_{b<n_blocks} partial[b][v][c] for n_vec <= 8 gradient vectors of `width` entries in
 * one launch (d_partial = the per-workgroup column sums tg_mlp_backward_chain / tg_dx_relu_bias / tg_relu_bwd_bias /
 * tg_head_prep leave; d_out = HOST array of n_vec device pointers).  Fixed summation order. */
int  tg_colsum_finish(const float* d_partial, int32_t n_blocks, int32_t n_vec, int32_t width, float* const* d_out,
                      void* stream);

/* Head of the backward pass: d_dz[r][k] = dout[r][k] (k < act_dim; compute dtype bf16 / f32), zero for the padding
 * columns up to out_pad (8 or 16), and d_partial f32 [tg_head_prep_blocks()][act_dim] = per-workgroup column sums of
 * dout (the head's bias gradient, finished by tg_colsum_finish with width act_dim). */
int  tg_head_prep_blocks(void);
int  tg_head_prep(const float* d_dout, int64_t rows, int32_t act_dim, int32_t out_pad, int32_t is_bf16, void* d_dz,
                  float* d_partial, void* stream);

/* Weight-gradient epilogue (replaces the reduction, tail GEMM and additions PyTorch would run after a split-K batched
 * GEMM dz^T a; torch autograd's accumulation into Linear.weight.grad):
 *   grad[m][k] += sum_{b<n_batches} partial[b][m][k] + sum_{r<tail} dz_tail[r][m] * a_tail[r][k],  m < m_out, k < k_out
 * d_partial f32 [n_batches][m_dim][k_dim]; d_dz_tail [tail][m_dim], d_a_tail [tail][k_dim] (bf16 if is_bf16 else f32;
 * may be NULL when tail == 0); d_grad f32 with row stride grad_ld.  Fixed summation order: deterministic. */
int  tg_dw_finish(const float* d_partial, int32_t n_batches, int32_t m_dim, int32_t k_dim, const void* d_dz_tail,
                  const void* d_a_tail, int32_t tail, int32_t is_bf16, float* d_grad, int64_t grad_ld, int32_t m_out,
                  int32_t k_out, void* stream);

/* A hidden layer's backward-data product on the matrix cores, fused with the ReLU backward and the bias gradient
 * of the layer below (bf16 operands, fp32 accumulate; replaces `dA = dZ @ W` + tg_relu_bwd_bias):
 *   dZ_below[r][m] = (sum_{k<k_dim} dZ[r][k] * W[k][m]) * (A[r][m] > 0)
 *   d_partial f32 [tg_dx_relu_bias_blocks()][m_dim]: per-workgroup column sums of dZ_below (caller sums axis 0).
 * d_dz_in bf16 [rows][k_dim], d_act / d_dz_out bf16 [rows][m_dim] (row-major, contiguous, distinct buffers),
 * W = Linear.weight bf16 [k_dim = out_features][m_dim = in_features].  d_wfrag is W re-ordered by
 * tg_dx_pack_weights (k_dim*m_dim bf16, MFMA A-fragment order; rebuild it whenever W changes).
 * tg_dx_relu_bias_supported(k_dim, m_dim) != 0 for the shapes that have a kernel (square 64 / 128 / 256).
 * d_maskbits (optional, widths 128 / 256): the ReLU mask bits of A written by tg_mlp_forward_chain (m_dim / 8 bytes per
 * row) are read instead of d_act (which may then be NULL): 1.03 instead of 1.5 KB of traffic per row at 256. */
int  tg_dx_relu_bias_supported(int32_t k_dim, int32_t m_dim);
int  tg_dx_relu_bias_blocks(void);
int  tg_dx_pack_weights(const void* d_w, void* d_wfrag, int32_t k_dim, int32_t m_dim, void* stream);
int  tg_dx_relu_bias(const void* d_dz_in, const void* d_wfrag, const void* d_act, const void* d_maskbits, void* d_dz_out,
                     int64_t rows, int32_t k_dim, int32_t m_dim, float* d_partial, void* stream);

/* ---- MLP forward, all layers in one persistent launch (models/neural_network.py:67-77) ----
 * Linear(in<=32, H) ReLU [Linear(H, H) ReLU]^(n_hidden_layers-1) Linear(H, out<=out_cols), H in {128, 256}, bf16
 * operands, fp32 accumulate / bias / head output.  A row's activations stay on chip between layers and are only
 * written (for the backward pass):
 *   d_x      bf16 [rows][32]   input, features >= in zero-padded
 *   d_wfrag  bf16 weight stream in MFMA fragment order (trajopt-grpo_amd/mlp.py `FragmentStream(layout="chain")`
 *            documents and builds it), d_bias f32 [n_hidden_layers + 1][H] (natural order, head row zero-padded)
 *   d_acts   HOST array of n_hidden_layers device pointers, bf16 [rows][H] each (post-ReLU outputs of the hidden
 *            layers, row-major), or NULL to skip storing them.  d_acts[0] alone may be NULL: the first activation is
 *            then left out (tg_mlp_weight_grad kind HR recomputes it; its mask bits are still written)
 *   d_masks  HOST array of n_hidden_layers device pointers, u32 [rows][H / 32] each, or NULL: the ReLU masks of the
 *            stored activations, 1 bit each (all the backward-data kernels need of them).  Per row: [lane half h = 0, 1]
 *            [H / 64 words]; feature 32 mt + 16 h + r is bit (mt & 1) * 8 + (r >> 1) + 16 * (r & 1) of word mt >> 1
 *   d_out    f32 [rows][out_cols], out_cols in {4, 8, 16} (columns >= out hold the padded head rows: zeros + bias 0; 4 for nets
 *            with <= 4 outputs: a 16-B row, so that tg_rollout_step reads no padding with the policy mean) */
int  tg_mlp_forward_chain(const void* d_x, const void* d_wfrag, const float* d_bias, int32_t hidden,
                          int32_t n_hidden_layers, int64_t rows, void* const* d_acts, void* const* d_masks,
                          float* d_out, int32_t out_cols, void* stream);

/* The training forward pass with the LOSS HEAD and the head's weight gradient inside it.  The clipped-surrogate / squared-error
 * gradient of a row (what tg_surrogate_loss + tg_head_prep produce: algorithms/ppo.py:159-179, grpo.py:122-140) is a function of the
 * head output the kernel holds and of per-row inputs, so d loss / d output is written directly (bf16 [rows][8], the backward
 * chain's input) and contracted on chip with the top hidden activation: that activation is NOT written (d_acts[n - 1] may be
 * NULL, as d_acts[0]) and tg_mlp_weight_grad needs no DH job.
 *   kind 0 (actor): d_act (contiguous [rows][act_dim]), d_logp_old, d_adv, var[act_dim], epsilon, surr_coef, kl_coef
 *   kind 1 (critic): d_ret, critic_coef              act_dim <= 4
 *   norm_mean / norm_inv: the advantage (actor) / return (critic) enters as (x - norm_mean) * norm_inv (0 and 1: as is)
 *   d_head_slabs  f32 [tg_mlp_forward_chain_blocks()][4][16][H] partial head weight gradients (rows 0..act_dim-1 of each [16][H]);
 *                 d_bias_partial f32 [blocks][4]; d_work f64 [blocks][4] partial sums (surrogate, squared error, KL, count):
 *                 the caller adds the first `grid` = min(blocks, ceil(rows / 256)) of each in order. */
typedef struct tg_chain_loss {
    int32_t      kind, act_dim;
    const float* d_act; int64_t act_row_stride, act_col_stride;
    const float* d_logp_old; const float* d_adv; const float* d_ret;
    float        norm_mean, norm_inv;
    float        var[4];
    float        epsilon, surr_coef, critic_coef, kl_coef;
    void*        d_dout8; float* d_head_slabs; double* d_work; float* d_bias_partial;
    /* kind 0 (tg_mlp_forward_chain_loss, tg_mlp_f32_forward_backward; NULL otherwise): the old policy IS the current one (ppo.py:142-143 always; GRPO when
     * nothing has touched either net since `old_policy.load_state_dict(policy.state_dict())`, grpo.py:148): the row's log-probability
     * is written here and used as its own old log-probability (ratio exactly 1, as in the reference, whose two forward passes are
     * the same arithmetic) -- d_logp_old is not read, and the caller needs no no-grad pass of the old policy. */
    float*       d_logp_old_out;
    /* NULL, or device f32[8] as tg_ppo_norm writes it {adv mean, adv 1/(std+eps), ret mean, ret 1/(std+eps), surr_coef, critic_coef,
     * kl_coef, n}: the head then takes its normalisation pair (kind 0: [0], [1]; kind 1: [2], [3]) and its coefficients ([4], [5], [6])
     * from there instead of from norm_mean / norm_inv / *_coef above -- PPO's batch statistics (ppo.py:138-139) and 1 / n_valid
     * (:165-179) never visit the host. */
    const float* d_norm8;
} tg_chain_loss;
int  tg_mlp_forward_chain_blocks(void);
int  tg_mlp_forward_chain_loss(const void* d_x, const void* d_wfrag, const float* d_bias, int32_t hidden, int32_t n_hidden_layers,
                               int64_t rows, void* const* d_acts, void* const* d_masks, const tg_chain_loss* loss, void* stream);

/* ---- MLP backward-data pass, all hidden layers in one persistent launch ----
 * For Linear(in, H) ReLU [Linear(H, H) ReLU]^(n_hidden_layers-1) Linear(H, out <= 8), H in {128, 256}, 3..6 hidden
 * layers, bf16:
 *   dZ_top   = (dOut . W_head)  * (a_top   > 0)
 *   dZ_below = (dZ   . W_layer) * (a_below > 0)           for every hidden-to-hidden layer, from the top down
 * a row's gradient stays on chip from the head to the first hidden layer; read 16 B + H/8 B of mask bits per layer,
 * write 2 H B per layer (H = 256, 5 layers: 2.7 instead of 4.8 KB per row).
 *   d_dout8   bf16 [rows][8]: d loss / d head output, columns >= out zero
 *   d_wfrag   bf16 stream of the TRANSPOSED weights, head first then the hidden-to-hidden layers top down
 *             (trajopt-grpo_amd/mlp.py `FragmentStream(layout="chain", transposed=True)`)
 *   d_dz      HOST array of n_hidden_layers device pointers, bf16 [rows][H]: outputs, TOP hidden layer first.  d_dz[0] may
 *             be NULL (only with d_partial == NULL): the top layer's dZ is then not written -- it is a function of d_dout8
 *             and the layer's mask bits, and tg_mlp_weight_grad (kind RH) rebuilds it on chip
 *   d_masks   HOST array of the same layers' ReLU mask bits (u32 [rows][H/32], as tg_mlp_forward_chain writes them)
 *   d_partial f32 [tg_mlp_backward_chain_blocks()][n_hidden_layers][H]: per-workgroup column sums of the dZ (bias
 *             gradients; the caller sums axis 0).  Deterministic.  NULL: no column sums (tg_mlp_weight_grad forms the
 *             bias gradients inside its contraction; the kernel is ~10 % shorter without them). */
int  tg_mlp_backward_chain_blocks(void);
int  tg_mlp_backward_chain(const void* d_dout8, const void* d_wfrag, int32_t hidden, int32_t n_hidden_layers, int64_t rows,
                           void* const* d_dz, const void* const* d_masks, float* d_partial, void* stream);
/* The same pass with the FIRST layer's weight gradient formed inside it: dW0 = dZ_bottom^T . x contracts over the rows the kernel
 * already holds, so the bottom layer's dZ is not written (d_dz[n_hidden_layers - 1] may be NULL) and tg_mlp_weight_grad needs no
 * HX job.  d_x = the net input, bf16 [rows][32], zero padded -- with column 31 set to ONE if the caller wants the first layer's
 * bias gradient as column 31 of the result.  d_w0_slabs: f32 [*n_slabs][H][32] partial gradients (2 per workgroup; buffer of at
 * least 2 * tg_mlp_backward_chain_blocks() * H * 32 floats = slab_floats); the caller adds the *n_slabs slabs in order.
 * Replaces: the bottom Linear's weight.grad / bias.grad of `loss.backward()` (algorithms/ppo.py:181-183). */
int  tg_mlp_backward_chain_w0(const void* d_dout8, const void* d_wfrag, int32_t hidden, int32_t n_hidden_layers, int64_t rows,
                              void* const* d_dz, const void* const* d_masks, const void* d_x, float* d_w0_slabs, int64_t slab_floats,
                              int32_t* n_slabs, void* stream);

/* ---- MLP weight gradients, every layer in one persistent launch (what `loss.backward()` leaves in Linear.weight.grad /
 *      .bias.grad: algorithms/ppo.py:181-183, algorithms/grpo.py:143-145) ----
 *   wgrad[m][n] += sum over rows r of P[r][m] * Q[r][n]   (m < m_out, n < n_out)      bgrad[m] += sum over rows of P[r][m]
 * one job per Linear; P = the layer's dZ (tg_mlp_backward_chain), Q = the layer's input.  bf16 operands, fp32 sums, every
 * operand byte read once.  A workgroup keeps one job's whole H x H fp32 gradient in registers for its share of the rows and
 * leaves it as a partial ("slab") in the workspace; a second kernel adds the slabs in a fixed order into the gradient
 * windows (e.g. the learner's flat all-reduce bucket): deterministic.
 *   kind HH: P bf16 [rows][H], Q bf16 [rows][H]          hidden-to-hidden layers
 *        HX: P bf16 [rows][H], Q bf16 [rows][32]         first layer (Q = the zero-padded net input)
 *        DH: P bf16 [rows][8], Q bf16 [rows][H]          head (P = d loss / d head output, zero padded; no bias sums:
 *                                                        tg_head_prep produces them)
 *        HR: P bf16 [rows][H], Q bf16 [rows][32]         second layer WITHOUT its stored input: the first hidden
 *                                                        activation relu(W0 x + b0) is recomputed on chip from the net
 *                                                        input row (d_w0frag = first block of the forward chain's weight
 *                                                        stream, d_b0 = first-layer bias f32 [H]); bit-identical to
 *                                                        what tg_mlp_forward_chain would have stored
 *        RH: P bf16 [rows][8],  Q bf16 [rows][H]         top hidden-to-hidden layer WITHOUT its stored dZ: dZ_top =
 *                                                        (dOut . W_head) * (a_top > 0) is recomputed on chip from P = d loss /
 *                                                        d head output (as DH's P), d_aux = the top hidden layer's ReLU mask
 *                                                        bits (u32 [rows][H/32], tg_mlp_forward_chain) and d_whfrag = the first
 *                                                        block of the backward chain's weight stream (W_head^T);
 *                                                        bit-identical to what tg_mlp_backward_chain would have stored
 * H in {128, 256}.  d_workspace: tg_mlp_weight_grad_workspace(H) bytes. */
enum { TG_DW_HH = 0, TG_DW_HX = 1, TG_DW_DH = 2, TG_DW_HR = 3, TG_DW_RH = 4 };
typedef struct tg_dw_job {
    const void* d_p;
    const void* d_q;
    float*      d_wgrad;      /* f32, row stride wgrad_ld */
    float*      d_bgrad;      /* f32 [m_out] or NULL */
    int64_t     wgrad_ld;
    int32_t     kind;
    int32_t     m_out, n_out;
    const void* d_aux;        /* RH: the top hidden layer's ReLU mask bits; otherwise unused */
} tg_dw_job;
int64_t tg_mlp_weight_grad_workspace(int32_t hidden);
/* tg_mlp_weight_grad_ex: the same launch pair, with up to 4 more fixed-order slab reductions riding on the second launch -- the
 * chain kernels' OWN partial gradients (tg_mlp_forward_chain_loss's head slabs and bias partials, tg_mlp_backward_chain_w0's
 * first-layer slabs), which the learner otherwise adds with three torch launches each:
 *   d_grad[m * grad_ld + n] += sum over s < n_slabs of d_slab[s * slab_stride + m * row_pitch + n],  m < m_out, n < n_out
 * -- and, optionally, the forward chain's per-workgroup f64 loss sums: d_loss_sums[k] += sum over r < n_loss_rows of
 * d_loss_work[r * 4 + k] (k < 4), rows in order.  Everything deterministic, no atomics. */
typedef struct tg_slab_sum {
    const float* d_slab; int64_t slab_stride; int32_t n_slabs; int32_t row_pitch;
    float* d_grad; int64_t grad_ld; int32_t m_out, n_out;
} tg_slab_sum;
int  tg_mlp_weight_grad_ex(int32_t hidden, const tg_dw_job* jobs, int32_t n_jobs, int64_t rows, const void* d_w0frag, const float* d_b0,
                           const void* d_whfrag, void* d_workspace, int64_t workspace_bytes, const tg_slab_sum* extra, int32_t n_extra,
                           const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums, void* stream);
int  tg_mlp_weight_grad(int32_t hidden, const tg_dw_job* jobs, int32_t n_jobs, int64_t rows, const void* d_w0frag,
                        const float* d_b0, const void* d_whfrag, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- The update in the reference's own precision (fp32) at the reference's own net sizes ----
 * Linear(S <= 32, H) ReLU [Linear(H, H) ReLU]{0..3} Linear(H, A <= 4), H in {64, 128}: pipelines/cartpole_pipeline_grpo.py:54-76,
 * cartpole_pipeline_ppo.py:54-79 (models/neural_network.py:67-77); every product is v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
 *   d_x       f32 [rows][in_pad], in_pad in {8, 16, 24, 32}, columns >= S zero
 *   d_stream  f32, tg_mlp_f32_stream_floats(H, n_hidden_layers, in_pad) floats (trajopt-grpo_amd/mlp.py `F32ChainStream` documents
 *             and builds it): [first layer in MFMA fragment order: H x in_pad][hidden biases: n x H][head weights: 4 x H, rows >= A
 *             zero][head bias: 4][forward blocks of the H x H layers][backward (transposed) blocks, top layer first]
 * tg_mlp_f32_forward: the no-grad pass (models/neural_network.py:67-77): d_out f32 [rows][4] (columns >= A: zeros + bias 0).
 * tg_mlp_f32_forward_backward: forward pass + loss head + backward-DATA pass of every row in ONE launch (algorithms/ppo.py:159-183,
 *   grpo.py:122-145 up to the weight gradients): `loss` as for tg_mlp_forward_chain_loss, except that d_dout8 receives d loss /
 *   d head output as F32 [rows][4], d_head_slabs / d_bias_partial are not used (the head's gradient is a job of
 *   tg_mlp_f32_weight_grad) and d_work is f64 [tg_mlp_f32_blocks()][4]; the caller adds the first min(blocks, ceil(rows / 256)).
 *   d_acts / d_dz: HOST arrays of n_hidden_layers device pointers, f32 [rows][H] each: the post-ReLU outputs of the hidden layers
 *   and d loss / d their pre-activations (index = layer, 0 = first hidden layer), written for tg_mlp_f32_weight_grad.  With >= 2
 *   hidden layers d_acts[0] and (given d_top_maskbits: u32 [rows][4], the top layer's ReLU mask, 1 bit per feature) d_dz[n - 1]
 *   may be NULL: the weight-gradient job that would read them rebuilds them on chip (tg_f32_dw_job.recompute).  Mask layout:
 *   feature 32 mt + 8 q + 4 hh + low (q < 4, hh < 2, low < 4) is bit low + 4 q + 16 (mt % 2) of word hh * (H / 64) + mt / 2.
 * tg_mlp_f32_weight_grad: every job's wgrad += P^T Q (and bgrad += column sums of P) in one launch + one fixed-order reduction
 *   (deterministic, no float atomics).  kind MM: P f32 [rows][H] (a layer's dZ), Q f32 [rows][n_cols] (its input: n_cols = H, or
 *   the padded net input, n_cols = in_pad), window m_out = H x n_out <= n_cols; kind HEAD: P = d loss / d head output f32
 *   [rows][4], Q = the top activation f32 [rows][H], window m_out = A x n_out = H, bgrad [A]. */
int64_t tg_mlp_f32_stream_floats(int32_t hidden, int32_t n_hidden_layers, int32_t in_pad);
int  tg_mlp_f32_blocks(void);
int  tg_mlp_f32_forward(const float* d_x, int32_t in_pad, const float* d_stream, int32_t hidden, int32_t n_hidden_layers,
                        int64_t rows, float* d_out, void* stream);
int  tg_mlp_f32_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, int32_t hidden, int32_t n_hidden_layers,
                                 int64_t rows, void* const* d_acts, void* const* d_dz, void* d_top_maskbits,
                                 const tg_chain_loss* loss, void* stream);
/* The same passes at H = 256 (the reference's QuadPole factory at its own precision: pipelines/quadpole_pipeline_ppo.py:54-58,
 * 20-256x5-{4,1} fp32), csrc/mlp_f32_wide.hip: a wave owns 16 rows on v_mfma_f32_16x16x4_f32, two 4-wave workgroups per CU.
 *   d_stream  f32, tg_mlp_f32w_stream_floats(n_hidden_layers) floats, in 16-KiB blocks of 16 pieces x 64 lanes x 16 B, lane = (i = lane & 15,
 *             g = lane >> 4) (trajopt-grpo_amd/mlp.py `F32WideStream`):
 *               first layer, 2 blocks: block b, piece 2 tt + q (tt < 8, q < 2): W0[16 (8 b + tt) + i][8 g + 4 q .. + 3] (zero beyond the inputs)
 *               forward, layer l = 1 .. n_hidden_layers - 1, 16 blocks mo, piece t:  W_l[16 mo + i][16 t + 4 g .. + 3]
 *               backward, layer l = n_hidden_layers - 1 .. 1, 16 blocks ko, piece t: {W_l[16 t + 4 g + r][16 ko + i], r = 0..3}
 *   d_table   f32 [tg_mlp_f32w_table_floats()]: [5][256] hidden biases | [4][256] head weights (rows >= outputs zero) | [4] head bias
 *   d_acts / d_dz as for tg_mlp_f32_forward_backward (d_acts[0] and d_dz[top] may be NULL). */
int64_t tg_mlp_f32w_stream_floats(int32_t n_hidden_layers);
int64_t tg_mlp_f32w_table_floats(void);
int  tg_mlp_f32w_blocks(void);
int  tg_mlp_f32w_forward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_table, int32_t n_hidden_layers, int64_t rows,
                         float* d_out, void* stream);
int  tg_mlp_f32w_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_table, int32_t n_hidden_layers,
                                  int64_t rows, void* const* d_acts, void* const* d_dz, const tg_chain_loss* loss, void* stream);
/* The H = 128 net with at most ONE H x H layer (BASELINE configs[1], C2: 5-128-128-1, pipelines/cartpole_pipeline_grpo.py:54-76) on the
 * same 16-row machine with its whole weight stream resident in LDS (csrc/mlp_f32_wide.hip, mlp_f32_res_kernel): 12 waves per CU (16 without gradients), no
 * barrier in the row loop, rows dealt 16 at a time wave-major across the CUs (C2's ~176,000 rows are 10.78 wave-rounds per SIMD: 11
 * here, 6 x 32-row rounds = 12 in tg_mlp_f32_forward_backward).  Same outputs as tg_mlp_f32_forward[_backward] (incl. the top layer's
 * mask bits): interchangeable in front of tg_mlp_f32_weight_grad.
 *   d_stream f32 [tg_mlp_f32r_stream_floats]: forward blocks [8 mo][8 pieces t][64 lanes] x 16 B: W_1[16 mo + i][16 t + 4 g .. + 3], then
 *            backward blocks [8 ko][8 t][64]: {W_1[16 t + 4 g + r][16 ko + i]}   (lane = (i = lane & 15, g = lane >> 4); empty with one hidden layer)
 *   d_w0     f32 [tg_mlp_f32r_w0_floats]: [8 tiles mo][in_pad / 4 steps s][64 lanes]: W0[16 mo + i][4 s + g] (zero beyond the inputs)
 *   d_table  f32 [tg_mlp_f32r_table_floats]: [2][128] hidden biases | [4][128] head weights | [4] head bias | 12 zeros */
int  tg_mlp_f32r_supported(int32_t hidden, int32_t n_hidden_layers, int32_t in_pad);
int64_t tg_mlp_f32r_stream_floats(int32_t hidden, int32_t n_hidden_layers);
int64_t tg_mlp_f32r_w0_floats(int32_t hidden, int32_t in_pad);
int64_t tg_mlp_f32r_table_floats(int32_t hidden);
int  tg_mlp_f32r_grid(int64_t rows);          /* workgroups of a training launch over `rows` rows = rows of f64 [4] partial sums it leaves in d_work */
int  tg_mlp_f32r_forward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_w0, const float* d_table, int32_t hidden,
                         int32_t n_hidden_layers, int32_t out_dim, int64_t rows, float* d_out /* [rows][4], columns >= out_dim zero */, void* stream);
int  tg_mlp_f32r_forward_backward(const float* d_x, int32_t in_pad, const float* d_stream, const float* d_w0, const float* d_table, int32_t hidden,
                                  int32_t n_hidden_layers, int64_t rows, void* const* d_acts, void* const* d_dz, void* d_top_maskbits,
                                  const tg_chain_loss* loss, void* stream);
enum { TG_F32DW_MM = 0, TG_F32DW_HEAD = 1 };
typedef struct tg_f32_dw_job {
    const float* d_p;
    const float* d_q;
    float*       d_wgrad;     /* f32, row stride wgrad_ld */
    float*       d_bgrad;     /* f32 [m_out] or NULL */
    int64_t      wgrad_ld;
    int32_t      kind, n_cols, m_out, n_out;
    /* wide MM job with operands REBUILT on chip instead of read, and the net's two light gradients riding on it (nets with >= 2
     * hidden layers; tg_mlp_f32_forward_backward then need not store the rebuilt matrices, and no MM job for the first layer /
     * HEAD job is given):
     *   bit 0 (the second layer's job): Q = relu(W0 x + b0), d_q = the net input f32 [rows][in_pad]; rider: the FIRST layer's
     *     gradient -- d_dz0 = the bottom dZ f32 [rows][H], d_w0grad f32 H x in_dim (row stride w0grad_ld) += dZ_0^T x, d_b0grad [H];
     *   bit 1 (the top layer's job): P = (g . W_head) * mask, d_p = d loss / d output f32 [rows][4], d_maskbits = the top layer's
     *     mask bits u32 [rows][4] written by tg_mlp_f32_forward_backward; rider: the HEAD's gradient -- d_a_top = the top
     *     activation f32 [rows][H], d_whgrad f32 act_dim x H (row stride whgrad_ld) += g^T A_top, d_bhgrad [act_dim]. */
    int32_t      recompute, in_pad, in_dim, act_dim;
    const float* d_w0;        /* Linear 0 weight f32 [H][in_dim], bias f32 [H] (the master tensors) */
    const float* d_b0;
    const float* d_wh;        /* head weight f32 [act_dim][H] */
    const uint32_t* d_maskbits;
    const float* d_a_top;
    float*       d_whgrad;
    float*       d_bhgrad;
    int64_t      whgrad_ld;
    const float* d_dz0;
    float*       d_w0grad;
    float*       d_b0grad;
    int64_t      w0grad_ld;
} tg_f32_dw_job;
int64_t tg_mlp_f32_weight_grad_workspace(int32_t hidden);
/* d_loss_work / n_loss_rows / d_loss_sums (optional, both pointers or neither): the reduction launch also adds rows [0, n_loss_rows)
 * of d_loss_work (f64 [.][4]: tg_mlp_f32_forward_backward's d_work, n_loss_rows = min(blocks, ceil(rows / 256))) into d_loss_sums
 * (f64 [4]) in a fixed order -- the loss statistics of an update without reduction / accumulation launches of their own. */
int  tg_mlp_f32_weight_grad(int32_t hidden, const tg_f32_dw_job* jobs, int32_t n_jobs, int64_t rows, void* d_workspace,
                            int64_t workspace_bytes, const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums, void* stream);

/* ---- Optimizer step and derived weight layouts (pipelines: torch.optim.Adam; algorithms/grpo.py:145, ppo.py:183) ----
 * tg_adam_step: torch.optim.Adam's DEFAULT update (foreach path: no amsgrad, weight decay, maximize, capturable) of n_tensors
 *   fp32 tensors in one launch, the same fp32 operation sequence per element as torch's kernels (bit-identical results).
 *   d_table: DEVICE array of n_tensors descriptors (param, grad, exp_avg, exp_avg_sq device pointers; `first` = the running
 *   element offset of the tensor in the launch's index space, ascending; total = the sum of the sizes).  step = the 1-based
 *   step number AFTER the increment (torch's state['step']).  zero_grads != 0: every gradient element is set to 0 once it has
 *   been read -- the NEXT update's optimizer.zero_grad(set_to_none=False) (algorithms/grpo.py:143, ppo.py:181) without a launch.
 * tg_gather_streams: dst[j] = master tensor (code[j] >> 24) element (code[j] & 0xFFFFFF), or 0 where code[j] < 0, converted
 *   to bf16 (is_bf16) or kept f32, for every segment in one launch (d_segments: DEVICE array, `first` as above; the master
 *   tensors are the `p` pointers of d_table).  Rebuilds every derived weight layout after a step. */
typedef struct tg_adam_tensor { float* p; float* g; float* m; float* v; int64_t first; } tg_adam_tensor;
typedef struct tg_gather_segment { void* dst; const int32_t* code; int64_t first; int32_t is_bf16; int32_t pad; } tg_gather_segment;
int  tg_adam_step(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, double lr, double beta1, double beta2, double eps,
                  int64_t step, int32_t zero_grads, void* stream);
int  tg_gather_streams(const tg_gather_segment* d_segments, int32_t n_segments, int64_t total, const tg_adam_tensor* d_table,
                       void* stream);
/* d_flag[0] |= 1 when, for any tensor of the table, an element of `p` differs bitwise from the same element of `g` (m, v unused). */
int  tg_params_differ(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, int32_t* d_flag, void* stream);
/* tg_adam_step + tg_gather_streams in ONE launch: the thread that updates element e of the launch's index space also writes the new
 * value to every layout position derived from it -- d_inv_start int32 [total + 1] / d_inv_dst: the segments' codes inverted (CSR);
 * a destination is (segment << 26 | element of that segment).  Positions whose code is negative (padding) are not touched: the
 * layouts must have been built once by tg_gather_streams.  Same update, bit for bit. */
int  tg_adam_step_push(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, double lr, double beta1, double beta2, double eps,
                       int64_t step, int32_t zero_grads, const tg_gather_segment* d_segments, int32_t n_segments,
                       const int32_t* d_inv_start, const int32_t* d_inv_dst, void* stream);
/* tg_mlp_f32_weight_grad with the optimizer step RIDING on its reduction launch: the thread that completes a gradient element applies
 * tg_adam_step's update to its parameter and (push tables given) tg_adam_step_push's layout writes, then leaves the gradient zeroed
 * (zero_grads) or holding the accumulated value -- what `loss.backward(); optimizer.step()` (algorithms/grpo.py:144-145) leaves, bit
 * for bit, in the launches of the backward pass alone.  h_table: a HOST copy of the optimizer's tensor table; every tensor of it must
 * be exactly one gradient window of `jobs`, whole and contiguous (else an error: a parameter without a window would miss its step).
 * Only where nothing stands between the gradients and the step: one rank (no all-reduce), one chunk of rows per update. */
typedef struct tg_adam_rider {
    const tg_adam_tensor*    h_table;
    int32_t                  n_tensors;
    int32_t                  zero_grads;
    int64_t                  total;
    double                   lr, beta1, beta2, eps;
    int64_t                  step;          /* 1-based, after the increment */
    const tg_gather_segment* d_segments;    /* optional (all three or none): tg_adam_step_push's tables */
    int32_t                  n_segments;
    int32_t                  pad;
    const int32_t*           d_inv_start;
    const int32_t*           d_inv_dst;
} tg_adam_rider;
int  tg_mlp_f32_weight_grad_adam(int32_t hidden, const tg_f32_dw_job* jobs, int32_t n_jobs, int64_t rows, void* d_workspace,
                                 int64_t workspace_bytes, const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums,
                                 const tg_adam_rider* adam, void* stream);

/* ---- The learner's prologue (algorithms/grpo.py:66-115, algorithms/ppo.py:126-139: returns, group statistics, `x[mask]`) ----
 * tg_returns_moments: tg_rtg_scan + tg_masked_moments of the returns in two launches instead of three, in the form for rollouts of
 *   a few thousand envs (a workgroup stages 64-step strips of 32 envs through LDS; one lane per env runs the recurrence in the
 *   reference's order): d_rtg and d_moments are BIT-identical to tg_rtg_scan / tg_masked_moments.  d_work: f64 [3*n] scratch.
 *   T <= tg_returns_moments_max_horizon() (the strips of a workgroup's 32 envs live in LDS).
 * tg_learn_count: valid entries per 1,024-entry chunk of the flat mask u8 [entries] (time-major [T][n]) and their exclusive prefix
 *   into d_work (tg_learn_count_workspace(entries) bytes); d_total[0] = the number of valid entries, d_total[1] = 1 when
 *   expected_rows >= 0 and differs from it (the host sized its buffers from the rollout's own statistic: a mask edited since then,
 *   or a hand-built trajectory, must not pass silently).
 * tg_learn_compact: for every valid (t, e), in time-major order (the order of `mask.nonzero()`), row r = its rank:
 *     d_idx[r] = t*n + e;   d_xin[r][0..in_pad) = obs[.][t][e] converted to bf16 (xin_bf16) or f32, zero padded, 1 in column
 *     `ones_col` (or -1);   d_act_rows[r][0..A) = act[.][t][e] (optional);   d_dst0[r] = d_src0[t*n + e],  d_dst1[r] = d_src1[..]
 *   (optional [T][n] f32 sources; with d_moments, d_dst0 is tg_group_normalize(mode = norm_mode) of d_src0 evaluated for the valid
 *   entries only -- same arithmetic, bit-identical).  d_offsets = tg_learn_count's d_work.  Rows >= rows_cap are not written.
 *   obs: [S] planes of obs_feat_stride elements (TG_F32 / TG_F64), entry (t, e) at t*n + e; act: f32 [A][T][n]. */
typedef struct tg_compact_args {
    const uint8_t* d_mask; const void* d_offsets;
    int64_t n; int32_t T; int32_t S; int32_t A; int32_t obs_dtype;
    const void* d_obs; int64_t obs_feat_stride; const float* d_act;
    void* d_xin; int32_t in_pad; int32_t xin_bf16; int32_t ones_col; int32_t norm_mode;
    float* d_act_rows; int64_t* d_idx;
    const float* d_src0; float* d_dst0; const float* d_src1; float* d_dst1;
    const double* d_moments; int64_t group_size;
    int64_t rows_cap;
} tg_compact_args;
int  tg_returns_moments_max_horizon(void);
int  tg_returns_moments(const float* d_rew, const uint8_t* d_mask, float gamma, float* d_rtg, int64_t n, int32_t T,
                        int64_t group_size, double* d_moments, double* d_work, void* stream);
int64_t tg_learn_count_workspace(int64_t entries);
int  tg_learn_count(const uint8_t* d_mask, int64_t entries, int64_t expected_rows, void* d_work, int64_t work_bytes, int64_t* d_total,
                    void* stream);
int  tg_learn_compact(const tg_compact_args* args, void* stream);

/* ---- PPO's prologue (algorithms/ppo.py:93-139) without a host round trip ----
 * tg_scatter_rows: d_dst[d_idx[r]] = d_src[r * src_stride] for r < rows -- the critic's values of the valid rows (the no-grad
 *   forward's padded output, column 0) back onto the zeroed [T][n] grid (ppo.py:93 evaluated on the valid rows only).
 * tg_ppo_returns: returns and advantages of every (t, e) and their masked fp64 moments in two launches.  monte_carlo != 0:
 *   R = tg_rtg_scan(rew, mask, gamma), A = R - V (ppo.py:100-111); else tg_gae_scan (ppo.py:112-124: A by GAE(gamma, lam), R = V + A).
 *   d_adv / d_ret f32 [T][n] (distinct); d_moments f64 [2][3] = {count, sum, sum of squares} of the valid advantages, then of the
 *   valid returns; d_work f64 [6 n] scratch.  Bit-identical to tg_rtg_scan / `rtg - V` / tg_gae_scan + tg_masked_moments(group = n).
 * tg_ppo_norm: from d_moments (after the ranks' all-reduce, if any) the f32 [8] the loss heads read through
 *   tg_chain_loss.d_norm8 / tg_loss_args.d_norm + d_coef: {adv mean, 1 / (adv std + 1e-8), ret mean, 1 / (ret std + 1e-8),
 *   -1 / n, c1 / n, kl_coeff / n, n} with the unbiased std clamped at 0, in torch's own order of fp64 / fp32 operations
 *   (ppo.py:138-139, :165-179).
 * tg_gather_rows2: d_dst0[r] = d_src0[d_idx[r]] (and d_dst1 / d_src1 when given): `advantages[mask]`, `returns[mask]`. */
int  tg_scatter_rows(const float* d_src, int64_t src_stride, const int64_t* d_idx, int64_t rows, float* d_dst, void* stream);
int  tg_ppo_returns(const float* d_rew, const float* d_values, const uint8_t* d_mask, float gamma, float lam, int monte_carlo,
                    float* d_adv, float* d_ret, int64_t n, int32_t T, double* d_moments, double* d_work, void* stream);
int  tg_ppo_norm(const double* d_moments, double c1, double kl_coeff, float* d_norm8, void* stream);
int  tg_gather_rows2(const int64_t* d_idx, int64_t rows, const float* d_src0, float* d_dst0, const float* d_src1, float* d_dst1,
                     void* stream);

/* ---- Measurement instruments (bench.py's roofline object; nothing on the product path calls them) ----
 * tg_clock_probe_attach: the update's persistent kernels are bound by the package power limit, i.e. by the shader clock the chip
 *   can hold while they run -- a clock neither rocm-smi's sclk nor a kernel duration shows.  With a probe attached, thread 0 of
 *   every workgroup of the named kernel family stamps s_memtime (shader-clock ticks) and s_memrealtime (100 MHz) at entry and
 *   exit and adds the differences to d_probe[0] / d_probe[1] (and 1 to d_probe[2]): clock = 0.1 GHz x d_probe[0] / d_probe[1].
 *   d_probe: DEVICE array of TG_CLOCK_PROBE_U64 zeroed uint64 (the entry stamps are parked behind the sums), NULL detaches.
 *   Synchronous (a symbol write); per device.  Detached (the default) a kernel pays one scalar load and a branch.
 * tg_mfma_sustained_probe: the chain kernels' bare inner loop -- A fragments from LDS (ds_read_b128), B fragments in registers,
 *   dependent accumulator chains, two waves per SIMD, one workgroup per CU, random data, no global traffic after the prologue --
 *   for `iters` iterations: dtype 0 = v_mfma_f32_16x16x32_bf16 (the bf16 chain kernels' shape), 1 = v_mfma_f32_32x32x2_f32 (the
 *   fp32 chain learner's).  Run back to back until the clock has settled it measures the matrix rate the package sustains under
 *   its power limit: the ceiling the roofline's `sustained_peak` quotes beside the datasheet peak.  d_w: 64 KiB of operands,
 *   d_x: blocks x 512 x 256 B, d_out: blocks x 512 floats; tg_mfma_sustained_probe_flops: flops of one launch. */
enum { TG_PROBE_FWD_CHAIN = 0, TG_PROBE_BWD_CHAIN = 1, TG_PROBE_WEIGHT_GRAD = 2, TG_PROBE_F32_CHAIN = 3, TG_PROBE_F32_WEIGHT_GRAD = 4,
       TG_PROBE_MFMA_LOOP = 5, TG_PROBE_FWD_CHAIN_PLAIN = 6 /* tg_mlp_forward_chain without the loss head: no-grad passes */ };
#define TG_CLOCK_PROBE_U64 (4 + 2 * 4096)
int  tg_clock_probe_attach(int32_t family, void* d_probe);
int  tg_mfma_sustained_probe_blocks(void);
double tg_mfma_sustained_probe_flops(int32_t dtype, int32_t iters);
int  tg_mfma_sustained_probe(int32_t dtype, int32_t iters, const void* d_w, const void* d_x, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRAJOPT_GRPO_HIP_H */
