// The loss head of the fp32 chain learners (mlp_f32_chain.hip: H = 64 / 128; mlp_f32_wide.hip: H = 256): the arithmetic of
// loss_kernels.hip::surrogate_loss_kernel (algorithms/ppo.py:159-179, grpo.py:122-140) for ONE row whose head outputs a lane holds.
#pragma once
#include "tg_common.hpp"

namespace tg {

struct F32Loss {            // (as ChainLoss of mlp_fwd_chain.hip)
    int32_t kind, A;        // 0: actor (clipped surrogate + KL-ish penalty), 1: critic (squared error)
    const float* act;       // actor: [rows][A] contiguous; critic: the returns [rows]
    const float* logp_old; const float* adv;
    float* logp_old_out;    // non-null: the old policy is the current one -- the row's log-probability is its old log-probability, written here
    float n_m, n_i;         // normalisation of the advantage (actor) / return (critic): (x - n_m) * n_i
    float inv_var[4]; float logp_const, epsilon, surr_coef, critic_coef, kl_coef;
    const float* norm8;     // non-null: n_m / n_i and the three coefficients come from this device f32 [8] (tg_ppo_norm's output)
    float* dout4;           // out: d loss / d head output, f32 [rows][4] (columns >= A zero)
    double* work;           // out: f64 [grid][4] partial loss sums (surrogate, squared error, KL, count)
};

// host: tg_chain_loss -> F32Loss
static inline void fill_f32_loss(F32Loss& L, const tg_chain_loss* loss) {
    L.kind = loss->kind; L.A = loss->act_dim;
    L.act = loss->kind == 0 ? loss->d_act : loss->d_ret;
    L.logp_old = loss->d_logp_old; L.adv = loss->d_adv; L.logp_old_out = loss->kind == 0 ? loss->d_logp_old_out : nullptr;
    L.n_m = loss->norm_mean; L.n_i = loss->norm_inv; L.norm8 = loss->d_norm8;
    float logdet = 0.f;
    for (int k = 0; k < 4; ++k) {
        L.inv_var[k] = k < loss->act_dim ? 1.0f / loss->var[k] : 0.f;
        if (k < loss->act_dim) logdet += logf(loss->var[k]);
    }
    L.logp_const = -0.5f * (float)loss->act_dim * 1.8378770664093453f - 0.5f * logdet;
    L.epsilon = loss->epsilon; L.surr_coef = loss->surr_coef; L.critic_coef = loss->critic_coef; L.kl_coef = loss->kl_coef;
    L.dout4 = (float*)loss->d_dout8; L.work = loss->d_work;
}

// device, at kernel entry: the normalisation pair and the coefficients from the device when PPO keeps them there
__device__ static inline void f32_loss_from_device(F32Loss& L) {
    if (L.norm8 != nullptr) {
        const int q = L.kind == 1 ? 2 : 0;
        L.n_m = L.norm8[q]; L.n_i = L.norm8[q + 1];
        L.surr_coef = L.norm8[4]; L.critic_coef = L.norm8[5]; L.kl_coef = L.norm8[6];
    }
}

// The per-row inputs of the loss head, loadable ahead of the row's products (f32_loss_load) ...
struct F32LossIn { float act[4]; float lpo, adv; };

// (every load unconditional within its uniform branch: columns >= A re-read the last action column)
// row = base + r: `base` wave-uniform (it goes into the scalar part of the address), `r` the lane's part
template <typename R>
__device__ static inline F32LossIn f32_loss_load(const F32Loss& L, int64_t base, R r) {
    F32LossIn in;
    in.act[1] = in.act[2] = in.act[3] = 0.f; in.lpo = 0.f; in.adv = 0.f;
    if (L.kind == 0) {
        const float* act = L.act + base * L.A;
#pragma unroll
        for (int k = 0; k < 4; ++k) in.act[k] = act[r * L.A + (k < L.A ? k : L.A - 1)];
        if (L.logp_old_out == nullptr) in.lpo = (L.logp_old + base)[r];
        in.adv = (L.adv + base)[r];
    } else {
        in.act[0] = (L.act + base)[r];
    }
    return in;
}
__device__ static inline F32LossIn f32_loss_load(const F32Loss& L, int64_t rowc) { return f32_loss_load(L, (int64_t)0, rowc); }

// ... and the arithmetic: head outputs o[4] -> d loss / d output g[4] and the row's contributions to the loss sums.  `row` / `valid` /
// `writer` decide what is written (one lane per row writes).  kZeroInvalid: a lane past the last row gets g = 0 (mlp_f32_chain.hip);
// false: it keeps the clamped row's own g -- the 16-row kernels let such lanes recompute and re-store the last row's values
// (identical bytes), so that every store instruction is issued whatever the row count.
template <bool kZeroInvalid = true>
__device__ static inline void f32_loss_compute(const F32Loss& L, const F32LossIn& in, const float (&o)[4], int64_t row, bool valid, bool writer,
                                               float (&g)[4], float& c_surr, float& c_crit, float& c_kl) {
    g[0] = g[1] = g[2] = g[3] = 0.f;
    c_surr = c_crit = c_kl = 0.f;
    if (L.kind == 0) {
        float quad = 0.f, dmu[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = (k < L.A ? in.act[k] : 0.f) - o[k];
            dmu[k] = d;
            quad += d * d * L.inv_var[k];
        }
        const float lp = -0.5f * quad + L.logp_const;
        float lpo;
        if (L.logp_old_out != nullptr) {
            lpo = lp;
            if (valid && writer) L.logp_old_out[row] = lp;
        } else {
            lpo = in.lpo;
        }
        const float adv = (in.adv - L.n_m) * L.n_i;
        const float rho = expf(lp - lpo);
        const float lo = 1.0f - L.epsilon, hi = 1.0f + L.epsilon;
        const float surr1 = rho * adv, surr2 = fminf(fmaxf(rho, lo), hi) * adv;
        const bool inside = (rho >= lo) && (rho <= hi);
        const float w = inside ? 1.0f : (surr1 < surr2 ? 1.0f : 0.0f);
        c_surr = fminf(surr1, surr2);
        float dlp = L.surr_coef * adv * rho * w;
        if (L.kl_coef != 0.0f) {
            const float eo = expf(lpo);
            c_kl = eo * (lpo - lp);
            dlp -= L.kl_coef * eo;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = dlp * dmu[k] * L.inv_var[k];
    } else {
        const float d = o[0] - (in.act[0] - L.n_m) * L.n_i;
        c_crit = d * d;
        g[0] = L.critic_coef * 2.0f * d;
    }
    if constexpr (kZeroInvalid) {
        if (!valid) { g[0] = g[1] = g[2] = g[3] = 0.f; }
    }
}

// One row, inputs loaded on the spot (`rowc` = the row clamped into range).
template <bool kZeroInvalid = true>
__device__ static inline void f32_loss_row(const F32Loss& L, const float (&o)[4], int64_t row, int64_t rowc, bool valid, bool writer,
                                           float (&g)[4], float& c_surr, float& c_crit, float& c_kl) {
    const F32LossIn in = f32_loss_load(L, rowc);
    f32_loss_compute<kZeroInvalid>(L, in, o, row, valid, writer, g, c_surr, c_crit, c_kl);
}

}  // namespace tg
