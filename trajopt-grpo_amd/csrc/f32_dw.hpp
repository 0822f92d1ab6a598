// Job descriptors of the fp32 learners' weight-gradient launches (mlp_f32_chain.hip: H = 64 / 128; mlp_f32_wide.hip: H = 256): one
// launcher (tg_mlp_f32_weight_grad_adam) lays the jobs out over the workgroups and builds the reduction's descriptors for all widths.
#pragma once
#include "tg_common.hpp"

namespace tg {

constexpr int kF32DwMaxJobs = 8;
enum : int32_t { F32DW_MM = 0, F32DW_HEAD = 1 };
struct F32DwJob {
    const float* p;         // MM: dZ f32 [rows][M = H];            HEAD: g f32 [rows][4]
    const float* q;         // MM: A  f32 [rows][N] (N = H, or the padded input width <= 32);  HEAD: A_top f32 [rows][H]
    int32_t kind, n;        // n: columns of q
    int32_t first_block, n_blocks, slab_len;
    int64_t slab_off;
    // wide job with rebuilt operands and riders (see f32_dw_fused): bit 0: Q = relu(W0 x + b0) from the net input rows (`q` = x
    // f32 [rows][in_pad]), the first layer's gradient rides; bit 1: P = (g . W_head) * mask from d loss / d output (`p` = g f32
    // [rows][4]) and the top layer's ReLU mask bits, the head's gradient rides
    int32_t recompute, in_pad, in_dim, act_dim;
    const float* w0;        // Linear 0 weight, f32 [H][in_dim] (the master tensor)
    const float* b0;        // Linear 0 bias f32 [H]
    const float* wh;        // head weight f32 [act_dim][H] (the master tensor)
    const uint32_t* mask;   // u32 [rows][4]: the top hidden layer's ReLU mask bits as tg_mlp_f32_forward_backward writes them
    const float* a_top;     // head rider: the top activation f32 [rows][H]
    const float* dz0;       // first-layer rider: the bottom dZ f32 [rows][H]
    int32_t ring_slots;     // 2 or 3
};
struct F32DwArgs { F32DwJob job[kF32DwMaxJobs]; int32_t n_jobs; };

// H = 256 (mlp_f32_wide.hip): the job kernel for one 8-wave workgroup per CU.  `grid` workgroups, job j owning blocks
// [first_block, first_block + n_blocks); slabs as the other widths write them ([256][N] row-major, then the bias sums).
int launch_f32_wide_dw(const F32DwArgs& args, int64_t rows, float* d_workspace, int grid, hipStream_t st);

}  // namespace tg
