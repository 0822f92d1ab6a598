// torch.optim.Adam's default update of ONE element, shared by the optimizer launch (optim_kernels.hip: tg_adam_step[_push]) and by
// the fp32 learner's gradient-reduction launch when the step rides on it (mlp_f32_chain.hip: tg_mlp_f32_weight_grad_adam) -- the
// same fp32 operation sequence, rounded where torch's separate foreach kernels round, so either launch leaves the weights bit-identical
// to `optimizer.step()` (pipelines/*: torch.optim.Adam(policy.parameters(), lr); algorithms/grpo.py:145, ppo.py:183).
#pragma once
#include <math.h>
#include "tg_common.hpp"

namespace tg {

struct AdamTensor { float* p; float* g; float* m; float* v; int64_t first; };   // first = index of element 0 in the launch
constexpr int kAdamMaxTensors = 64;

struct GatherSegment { void* dst; const int32_t* code; int64_t first; int32_t is_bf16; int32_t pad; };
constexpr int kGatherMaxSegments = 32;
constexpr int kPushSegShift = 26;          // a push destination = segment << 26 | element of the segment

struct AdamScalars { float w1, beta2, w2, bc2_sqrt, eps, step_size; };

// the scalars exactly as torch/optim/adam.py::_multi_tensor_adam forms them (Python doubles, cast to float by the kernels);
// step = the 1-based step number AFTER the increment
static inline AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = (lr / bc1) * -1.0, bc2_sqrt = pow(bc2, 0.5);
    return AdamScalars{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)bc2_sqrt, (float)eps, (float)step_size};
}

// returns the new parameter; m, v updated in place
__device__ static inline float adam_update(float g, float& m, float& v, float p, const AdamScalars& a) {
#pragma clang fp contract(off)
    // torch._foreach_lerp_(exp_avg, grad, 1 - beta1): weight < 0.5 -> self + weight * (end - self), the product fused into the sum
    m = fmaf(a.w1, g - m, m);
    // torch._foreach_mul_(exp_avg_sq, beta2); torch._foreach_addcmul_(exp_avg_sq, grad, grad, 1 - beta2): self + value * t1 * t2
    // (measured against torch 2.10's kernels on gfx950: the square is rounded, then value * square is fused into the sum)
    v = v * a.beta2;
    v = fmaf(a.w2, g * g, v);
    // sqrt -> / sqrt(bias_correction2) -> + eps: three kernels in torch, three roundings here.  The divisor comes in as a LIST of
    // scalars (one per tensor): that overload divides (a / float(b)); the single-scalar overload would multiply by float(1 / b)
    float s = sqrtf(v);
    s = s / a.bc2_sqrt;
    s = s + a.eps;
    // torch._foreach_addcdiv_(param, exp_avg, denom, step_size): self + value * (t1 / t2)
    return fmaf(a.step_size, m / s, p);
}

// the new value of element e of the launch's index space written into every derived layout position it appears in (inv_start /
// inv_dst: the gather's codes inverted, CSR over the element index)
__device__ static inline void adam_push(int64_t e, float p, const GatherSegment* __restrict__ seg, const int32_t* __restrict__ inv_start,
                                        const int32_t* __restrict__ inv_dst) {
    for (int32_t q = inv_start[e]; q < inv_start[e + 1]; ++q) {
        const int32_t dd = inv_dst[q];
        const GatherSegment s_ = seg[dd >> kPushSegShift];
        const int32_t j = dd & ((1 << kPushSegShift) - 1);
        if (s_.is_bf16) reinterpret_cast<__bf16*>(s_.dst)[j] = (__bf16)p;        // (the conversion tg_gather_streams applies)
        else reinterpret_cast<float*>(s_.dst)[j] = p;
    }
}

}  // namespace tg
