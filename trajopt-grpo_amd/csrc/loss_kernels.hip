// Gaussian log-prob and the fused clipped-surrogate loss (forward + backward in one pass).
//
// One lane per sample.  The kernel reads the policy mean (GEMM output), the action, the
// old log-prob and the advantage (and value / return for PPO), forms
//   logp, rho = exp(logp - logp_old), min(rho A, clip(rho) A), (V - R)^2, exp(lp_old)(lp_old - lp)
// and writes d(total)/d(mean) (and d/dV) directly, so the whole loss head costs one read of
// its inputs and one write of the gradient instead of ~15 elementwise launches.
// Loss scalars are reduced deterministically: wavefront shuffle -> per-block partial in
// fp64 -> fixed-order final pass (no float atomics).
#include "tg_common.hpp"

namespace tg {

constexpr int kLossBlocks = 1024;   // persistent grid-stride blocks
constexpr int kLossThreads = 256;

struct Var8 { float inv_var[8]; float logp_const; };

__device__ static inline float gaussian_logp(const float* mu, const float* a, const Var8& v, int A) {
    float quad = 0.0f;
#pragma unroll 4
    for (int k = 0; k < A; ++k) {
        const float d = a[k] - mu[k];
        quad += d * d * v.inv_var[k];
    }
    return -0.5f * quad + v.logp_const;
}

template <int A>
__global__ __launch_bounds__(256) void gaussian_logp_kernel(const float* __restrict__ mean, int64_t mean_rs,
                                                            const float* __restrict__ act, int64_t act_rs, int64_t act_cs,
                                                            Var8 v, float* __restrict__ logp, int64_t M) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        float mu[A], a[A];
#pragma unroll
        for (int k = 0; k < A; ++k) { mu[k] = mean[i * mean_rs + k]; a[k] = act[i * act_rs + k * act_cs]; }
        logp[i] = gaussian_logp(mu, a, v, A);
    }
}

struct LossK {
    const float* mean; int64_t mean_rs;
    const float* act; int64_t act_rs, act_cs;
    const float* logp_old; const float* adv; const float* value; const float* ret;
    const uint8_t* mask; const float* norm; const float* coef;
    Var8 v;
    float epsilon, surr_coef, critic_coef, kl_coef;
    float* grad_mean; float* grad_value; double* work; int64_t M;
};

template <int A>
__global__ __launch_bounds__(256) void surrogate_loss_kernel(LossK p) {
    __shared__ double sh[4][4];
    double s_surr = 0, s_crit = 0, s_kl = 0, s_cnt = 0;
    float n_am = 0.f, n_ai = 1.f, n_rm = 0.f, n_ri = 1.f;
    if (p.norm != nullptr) { n_am = p.norm[0]; n_ai = p.norm[1]; n_rm = p.norm[2]; n_ri = p.norm[3]; }
    // (locals: writing into the by-value argument struct would send it to scratch)
    float surr_coef = p.surr_coef, critic_coef = p.critic_coef, kl_coef = p.kl_coef;
    if (p.coef != nullptr) { surr_coef = p.coef[0]; critic_coef = p.coef[1]; kl_coef = p.coef[2]; }
    const float lo = 1.0f - p.epsilon, hi = 1.0f + p.epsilon;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.M; i += (int64_t)gridDim.x * blockDim.x) {
        const bool valid = p.mask == nullptr || p.mask[i] != 0;
        float g[A];
#pragma unroll
        for (int k = 0; k < A; ++k) g[k] = 0.0f;
        float gv = 0.0f;
        if (valid) {
            float mu[A], a[A];
#pragma unroll
            for (int k = 0; k < A; ++k) { mu[k] = p.mean[i * p.mean_rs + k]; a[k] = p.act[i * p.act_rs + k * p.act_cs]; }
            const float lp = gaussian_logp(mu, a, p.v, A);
            const float lpo = p.logp_old[i];
            const float adv = (p.adv[i] - n_am) * n_ai;
            const float rho = expf(lp - lpo);
            const float surr1 = rho * adv;
            const float surr2 = fminf(fmaxf(rho, lo), hi) * adv;
            // torch.min / torch.clamp subgradients: inside the clip range both branches carry A
            // (tie -> half each); outside, only the unclipped branch when it is the smaller one.
            const bool inside = (rho >= lo) && (rho <= hi);
            const float w = inside ? 1.0f : (surr1 < surr2 ? 1.0f : 0.0f);
            s_surr += (double)fminf(surr1, surr2);
            // d total / d logp
            float dlp = surr_coef * adv * rho * w;
            if (kl_coef != 0.0f) {
                const float eo = expf(lpo);
                s_kl += (double)(eo * (lpo - lp));
                dlp -= kl_coef * eo;
            }
#pragma unroll
            for (int k = 0; k < A; ++k) g[k] = dlp * (a[k] - mu[k]) * p.v.inv_var[k];   // d logp / d mu_k
            if (p.value != nullptr) {
                const float d = p.value[i] - (p.ret[i] - n_rm) * n_ri;
                s_crit += (double)(d * d);
                gv = critic_coef * 2.0f * d;
            }
            s_cnt += 1.0;
        }
#pragma unroll
        for (int k = 0; k < A; ++k) p.grad_mean[i * A + k] = g[k];
        if (p.grad_value != nullptr) p.grad_value[i] = gv;
    }
    double acc[4] = {s_surr, s_crit, s_kl, s_cnt};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[j][w] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x;
        p.work[(int64_t)blockIdx.x * 4 + j] = ((sh[j][0] + sh[j][1]) + sh[j][2]) + sh[j][3];
    }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ work, int nblocks, double* __restrict__ sums) {
    __shared__ double sh[4][4];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += work[(int64_t)b * 4 + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[j][w] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x;
        sums[j] = ((sh[j][0] + sh[j][1]) + sh[j][2]) + sh[j][3];
    }
}

static int make_var(const float* var, int A, Var8& v, const char* who) {
    if (A < 1 || A > 8) return set_error(TG_ERR_ARG, "%s: act_dim %d outside 1..8", who, A);
    double logdet = 0;
    for (int k = 0; k < 8; ++k) v.inv_var[k] = 0.f;
    for (int k = 0; k < A; ++k) {
        if (!(var[k] > 0.f)) return set_error(TG_ERR_ARG, "%s: var[%d] must be positive", who, k);
        v.inv_var[k] = 1.0f / var[k];
        logdet += log((double)var[k]);
    }
    v.logp_const = (float)(-0.5 * A * log(2.0 * 3.141592653589793) - 0.5 * logdet);
    return TG_OK;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_loss_work_blocks(void) { return kLossBlocks; }

int tg_gaussian_logp(const float* d_mean, int64_t mean_row_stride, const float* d_act, int64_t act_row_stride,
                     int64_t act_col_stride, const float* var, int act_dim, float* d_logp, int64_t M, void* stream) {
    TG_REQUIRE(d_mean && d_act && var && d_logp, "tg_gaussian_logp: null pointer");
    TG_REQUIRE(M >= 0 && mean_row_stride >= act_dim, "tg_gaussian_logp: bad sizes");
    Var8 v;
    int rc = make_var(var, act_dim, v, "tg_gaussian_logp");
    if (rc != TG_OK) return rc;
    if (M == 0) return TG_OK;
    const unsigned grid = (unsigned)(ceil_div(M, 256) < 4096 ? ceil_div(M, 256) : 4096);
    hipStream_t st = (hipStream_t)stream;
#define L(AA)                                                                                                          \
    case AA:                                                                                                           \
        hipLaunchKernelGGL(gaussian_logp_kernel<AA>, dim3(grid), dim3(256), 0, st, d_mean, mean_row_stride, d_act,      \
                           act_row_stride, act_col_stride, v, d_logp, M);                                              \
        break;
    switch (act_dim) { L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) }
#undef L
    TG_LAUNCH_CHECK("tg_gaussian_logp");
    return TG_OK;
}

int tg_surrogate_loss(const tg_loss_args* a, void* stream) {
    TG_REQUIRE(a != nullptr, "tg_surrogate_loss: null args");
    TG_REQUIRE(a->d_mean && a->d_act && a->d_logp_old && a->d_adv && a->d_grad_mean && a->d_sums && a->d_work,
               "tg_surrogate_loss: null pointer");
    TG_REQUIRE((a->d_value == nullptr) == (a->d_ret == nullptr), "tg_surrogate_loss: value and ret go together");
    TG_REQUIRE(a->d_value == nullptr || a->d_grad_value != nullptr, "tg_surrogate_loss: grad_value missing");
    TG_REQUIRE(a->M >= 0 && a->mean_row_stride >= a->act_dim, "tg_surrogate_loss: bad sizes");
    LossK k;
    int rc = make_var(a->var, a->act_dim, k.v, "tg_surrogate_loss");
    if (rc != TG_OK) return rc;
    k.mean = a->d_mean; k.mean_rs = a->mean_row_stride;
    k.act = a->d_act; k.act_rs = a->act_row_stride; k.act_cs = a->act_col_stride;
    k.logp_old = a->d_logp_old; k.adv = a->d_adv; k.value = a->d_value; k.ret = a->d_ret;
    k.mask = a->d_mask; k.norm = a->d_norm; k.coef = a->d_coef;
    k.epsilon = a->epsilon; k.surr_coef = a->surr_coef; k.critic_coef = a->critic_coef; k.kl_coef = a->kl_coef;
    k.grad_mean = a->d_grad_mean; k.grad_value = a->d_grad_value; k.work = a->d_work; k.M = a->M;
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = ceil_div(a->M > 0 ? a->M : 1, kLossThreads);
    const unsigned grid = (unsigned)(nb < kLossBlocks ? nb : kLossBlocks);
#define L(AA)                                                                                            \
    case AA:                                                                                             \
        hipLaunchKernelGGL(surrogate_loss_kernel<AA>, dim3(grid), dim3(kLossThreads), 0, st, k);         \
        break;
    switch (a->act_dim) { L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) }
#undef L
    TG_LAUNCH_CHECK("tg_surrogate_loss");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, a->d_work, (int)grid, a->d_sums);
    TG_LAUNCH_CHECK("tg_surrogate_loss(final)");
    return TG_OK;
}

}  // extern "C"
