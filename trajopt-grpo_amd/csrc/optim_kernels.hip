// One launch per optimizer step, one more for every derived weight layout -- instead of torch.optim.Adam's ~8 multi-tensor
// launches plus cat / gather / convert per weight stream (pipelines/*: `torch.optim.Adam(policy.parameters(), lr=...)`,
// algorithms/grpo.py:145 / ppo.py:183 `optimizer.step()`): at C2's 0.45 ms per update those ~20 small launches were a fifth of
// the step.
//
//   tg_adam_step      torch.optim.Adam's default (foreach, non-capturable, no amsgrad / weight decay / maximize) update of up to 64
//                     tensors in one launch, the SAME fp32 operation sequence per element -- lerp, mul, addcmul, sqrt, div, add,
//                     addcdiv, each rounded where torch's separate kernels round -- so the weights stay bit-identical to
//                     `optimizer.step()` (tests/test_gpu_parity.py::test_fused_adam_is_bit_identical_to_torch);
//   tg_gather_streams every derived layout of the weights (the chain kernels' bf16 fragment streams and f32 bias tables, the fp32
//                     chain stream) rebuilt from the fp32 masters by one gather: element j of segment s = master
//                     tensor (code >> 24), offset (code & 0xFFFFFF), or zero.
#include "adam_update.hpp"

namespace tg {

// kPush: the thread that has just updated a parameter also writes it into every derived layout it appears in (adam_push) --
// tg_gather_streams folded into the optimizer step.
template <bool kPush>
__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor* __restrict__ table, int32_t n_tensors, int64_t total, AdamScalars a,
                                                   int32_t zero_grads, const GatherSegment* __restrict__ seg,
                                                   const int32_t* __restrict__ inv_start, const int32_t* __restrict__ inv_dst) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int k = 0;
    for (int t = 1; t < n_tensors; ++t)
        if (e >= table[t].first) k = t;
    const AdamTensor d = table[k];
    const int64_t i = e - d.first;
    const float g = d.g[i];
    float m = d.m[i], v = d.v[i];
    const float p = adam_update(g, m, v, d.p[i], a);
    d.m[i] = m; d.v[i] = v; d.p[i] = p;
    if (zero_grads) d.g[i] = 0.0f;          // the next step's optimizer.zero_grad(set_to_none=False), while the line is here
    if constexpr (kPush) adam_push(e, p, seg, inv_start, inv_dst);
}

// flag[0] |= 1 when any element of tensor pair (p, g) of the table differs bitwise (the learner's check that "old_policy is the
// policy" really held when it let the first update stand in for the old policy's pass)
__global__ __launch_bounds__(256) void params_differ_kernel(const AdamTensor* __restrict__ table, int32_t n_tensors, int64_t total,
                                                            int32_t* __restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int k = 0;
    for (int t = 1; t < n_tensors; ++t)
        if (e >= table[t].first) k = t;
    const AdamTensor d = table[k];
    const int64_t i = e - d.first;
    if (__float_as_uint(d.p[i]) != __float_as_uint(d.g[i])) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void gather_streams_kernel(const GatherSegment* __restrict__ seg, int32_t n_seg, int64_t total,
                                                             const AdamTensor* __restrict__ table) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int k = 0;
    for (int t = 1; t < n_seg; ++t)
        if (e >= seg[t].first) k = t;
    const GatherSegment s = seg[k];
    const int64_t j = e - s.first;
    const int32_t c = s.code[j];
    const float v = c < 0 ? 0.0f : table[c >> 24].p[c & 0xFFFFFF];
    if (s.is_bf16) reinterpret_cast<__bf16*>(s.dst)[j] = (__bf16)v;
    else reinterpret_cast<float*>(s.dst)[j] = v;
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_adam_step(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, double lr, double beta1, double beta2, double eps,
                 int64_t step, int32_t zero_grads, void* stream) {
    TG_REQUIRE(d_table, "tg_adam_step: null table");
    TG_REQUIRE(n_tensors >= 1 && n_tensors <= kAdamMaxTensors, "tg_adam_step: %d tensors outside 1..%d", n_tensors, kAdamMaxTensors);
    TG_REQUIRE(total >= 0 && step >= 1, "tg_adam_step: bad sizes (total %lld, step %lld)", (long long)total, (long long)step);
    TG_REQUIRE(1.0 - beta1 < 0.5, "tg_adam_step: beta1 = %g: lerp's other branch (weight >= 0.5) is not implemented", beta1);
    if (total == 0) return TG_OK;
    static_assert(sizeof(tg_adam_tensor) == sizeof(AdamTensor), "ABI struct and kernel struct must agree");
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const AdamTensor*>(d_table), n_tensors, total, adam_scalars(lr, beta1, beta2, eps, step), zero_grads,
                       nullptr, nullptr, nullptr);
    TG_LAUNCH_CHECK("tg_adam_step");
    return TG_OK;
}

int tg_adam_step_push(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, double lr, double beta1, double beta2, double eps,
                      int64_t step, int32_t zero_grads, const tg_gather_segment* d_segments, int32_t n_segments,
                      const int32_t* d_inv_start, const int32_t* d_inv_dst, void* stream) {
    TG_REQUIRE(d_table && d_segments && d_inv_start && d_inv_dst, "tg_adam_step_push: null pointer");
    TG_REQUIRE(n_tensors >= 1 && n_tensors <= kAdamMaxTensors, "tg_adam_step_push: %d tensors outside 1..%d", n_tensors, kAdamMaxTensors);
    TG_REQUIRE(n_segments >= 1 && n_segments <= kGatherMaxSegments, "tg_adam_step_push: %d segments outside 1..%d", n_segments, kGatherMaxSegments);
    TG_REQUIRE(total >= 0 && step >= 1, "tg_adam_step_push: bad sizes (total %lld, step %lld)", (long long)total, (long long)step);
    TG_REQUIRE(1.0 - beta1 < 0.5, "tg_adam_step_push: beta1 = %g: lerp's other branch (weight >= 0.5) is not implemented", beta1);
    if (total == 0) return TG_OK;
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const AdamTensor*>(d_table), n_tensors, total, adam_scalars(lr, beta1, beta2, eps, step), zero_grads,
                       reinterpret_cast<const GatherSegment*>(d_segments), d_inv_start, d_inv_dst);
    TG_LAUNCH_CHECK("tg_adam_step_push");
    return TG_OK;
}

int tg_params_differ(const tg_adam_tensor* d_table, int32_t n_tensors, int64_t total, int32_t* d_flag, void* stream) {
    TG_REQUIRE(d_table && d_flag, "tg_params_differ: null pointer");
    TG_REQUIRE(n_tensors >= 1 && n_tensors <= kAdamMaxTensors && total >= 0, "tg_params_differ: bad sizes");
    if (total == 0) return TG_OK;
    hipLaunchKernelGGL(params_differ_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const AdamTensor*>(d_table), n_tensors, total, d_flag);
    TG_LAUNCH_CHECK("tg_params_differ");
    return TG_OK;
}

int tg_gather_streams(const tg_gather_segment* d_segments, int32_t n_segments, int64_t total, const tg_adam_tensor* d_table, void* stream) {
    TG_REQUIRE(d_segments && d_table, "tg_gather_streams: null pointer");
    TG_REQUIRE(n_segments >= 1 && n_segments <= kGatherMaxSegments, "tg_gather_streams: %d segments outside 1..%d", n_segments, kGatherMaxSegments);
    TG_REQUIRE(total >= 0, "tg_gather_streams: negative size");
    if (total == 0) return TG_OK;
    static_assert(sizeof(tg_gather_segment) == sizeof(GatherSegment), "ABI struct and kernel struct must agree");
    hipLaunchKernelGGL(gather_streams_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const GatherSegment*>(d_segments), n_segments, total, reinterpret_cast<const AdamTensor*>(d_table));
    TG_LAUNCH_CHECK("tg_gather_streams");
    return TG_OK;
}

}  // extern "C"
