// Weight gradients of the reference's ReLU MLP -- what `loss.backward()` (algorithms/ppo.py:181-183,
// algorithms/grpo.py:143-145) leaves in every Linear's weight.grad / bias.grad -- for 10^6..10^7 rows in ONE launch:
//
//     dW_l[m][n] = sum over rows r of  dZ_l[r][m] * A_{l-1}[r][n]          db_l[m] = sum over rows r of dZ_l[r][m]
//
// Every dZ (written by tg_mlp_backward_chain) and every stored activation (tg_mlp_forward_chain) is read exactly ONCE;
// the products are HBM-bound (128 flop per byte at H = 256), so the kernel is organised around the byte stream:
//
//   * a workgroup (8 waves, one per CU) owns ONE layer ("job") for a share of the rows and keeps that layer's whole
//     H x H fp32 gradient in its accumulator registers (256 KiB at H = 256: 128 registers per lane in each of the 8
//     waves) while the rows stream past; the CUs are split between the jobs in proportion to their bytes per row, so
//     all finish together and there are ~50 partial gradients ("slabs") per layer instead of one per CU and layer;
//   * 32-row stages of both operands flow HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging) through a
//     ring sized per job kind (DwRing: 4 slots x 32 KiB for the H x H jobs, 7-8 x 18 KiB for the narrow ones, so that every
//     CU keeps ~100 KB in flight): counted `s_waitcnt vmcnt`, raw `s_barrier` (see mfma_ring.hpp).  Every stage issues the
//     same number of DMA instructions per wave -- stages past the end re-read the last row -- so the count is a
//     compile-time constant;
//   * the contraction runs over ROWS, which are the slow axis of both operands in memory: the fragments come out of
//     LDS through the transposing read `ds_read_b64_tr_b16` (4 rows x 16 columns per 16 lanes, delivered column-major).
//     The LDS image of a panel is [row / 4][32-column group][row % 4][64 B]: a 32-lane half of a transposed read then
//     covers 256 contiguous bytes (conflict-free) and every address is lane constant + immediate;
//   * bias gradients ride along: one extra MFMA per k-step multiplies the dZ fragment with a matrix of ones (the
//     matrix pipe is ~1/3 busy in this kernel; on the vector ALU the same sums cost the backward chain 10 %);
//   * the first hidden layer's activations are not read at all (kind HR): they are a function of the 64-B input row,
//     so the workgroup recomputes each stage's 32 x H tile (4 small MFMAs per wave, same instruction sequence as the forward
//     chain: identical bits) one stage ahead of its use, straight into the LDS image of a Q panel -- 576 instead of
//     1024 B per row for that layer, and the forward pass no longer has to write them;
//   * the top hidden layer's dZ is not read either (kind RH): dZ_top = (dOut . W_head) * (a_top > 0) is a function of the 16-B
//     head-gradient row and 32 B of mask bits, rebuilt per stage like a0 (the backward chain's own head block: identical bits), so
//     tg_mlp_backward_chain need not write it.  Recomputed tiles keep chunk c of a row's 64-byte segment in slot c ^ (row / 4 & 3):
//     their writers hold 16 rows x one 16-B chunk per 16 lanes, which would be a 4-way bank conflict in the plain image;
//   * the H x H jobs read their fragments with raw `ds_read_b64_tr_b16` in the order of use and count lgkmcnt themselves (hipcc
//     waits for all 12 reads of a k-step before the first MFMA): LDS operations return in order, so foreign accesses in the
//     queue can only make a counted wait stricter;
//   * rows past the end: their (clamped, finite) data is multiplied by zeros -- the dZ fragments of the one stage that holds
//     the last rows are masked in registers (a separate instantiation of the stage: the loop over full stages has no branch);
//   * a second small kernel adds the slabs in a fixed order straight into the gradient windows (the learner's flat
//     all-reduce bucket): deterministic, no float atomics.
#include <stdlib.h>

#include <type_traits>

#include "mfma_ring.hpp"

namespace tg {

typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;

constexpr int kDwMaxJobs = 8;
constexpr int kDwStageRows = 32;

struct DwJob {
    const uint16_t* p;        // bf16 [rows][H] (HH, HX, HR) or [rows][8] (DH, RH)
    const uint16_t* q;        // bf16 [rows][H] (HH, DH, RH) or [rows][32] (HX, HR)
    const uint32_t* aux;      // RH: ReLU mask bits of the top hidden layer, u32 [rows][H/32]
    int32_t kind;
    int32_t first_block;      // workgroups [first_block, first_block + n_blocks) work on this job
    int32_t n_blocks;
    int32_t slab_len;         // floats per slab
    int64_t slab_off;         // float offset of this job's slab 0 in the workspace
};
struct DwArgs {
    DwJob job[kDwMaxJobs];
    int32_t n_jobs;
    const uint4* w0frag;      // HR: first-layer block of the forward chain's weight stream ([tile][k-step][64 lanes] x 16 B)
    const float* b0;          // HR: first-layer bias, f32 [H]
    const uint4* whfrag;      // RH: first block of the backward chain's weight stream (W_head^T: [tile][half][64 lanes] x 16 B)
};

// One MFMA operand fragment (8 consecutive ROWS of one column per lane) out of a row-major LDS panel: two transposing
// reads of 4 rows each.  `__restrict__` on an inlined function's pointer parameter attaches alias-scope metadata to the
// reads; without it hipcc waits for every outstanding LDS-DMA (vmcnt(0)) before an LDS read (mfma_ring.hpp).
__device__ static inline bf16x8 tr_frag(const char* __restrict__ base, int off_lo, int off_hi) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(base + off_lo));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(base + off_hi));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// The same fragment as two raw instructions on a 32-bit LDS address + compile-time offset, invisible to hipcc's waitcnt
// insertion (which puts lgkmcnt(0) in front of the first MFMA even when only the first 4 of 12 reads feed it): the square jobs
// count their own lgkmcnt (tr_wait) so that the products start as soon as their two fragments have landed.
template <int OFF_LO, int OFF_HI>
__device__ static inline bf16x8 tr_frag_raw(uint32_t addr_lo, uint32_t addr_hi) {
    i16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr_lo), "n"(OFF_LO));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr_hi), "n"(OFF_HI));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// all but the youngest N LDS operations of this wave have returned; ties the two fragments that are about to be used to the wait
template <int N>
__device__ static inline void tr_wait(bf16x8& x, bf16x8& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}
template <int I, int N, class F>
__device__ static inline void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ static inline uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ static inline uint4 lds_load16(const char* __restrict__ p) { return *reinterpret_cast<const uint4*>(p); }
__device__ static inline void lds_store16(char* __restrict__ p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ static inline uint32_t lds_load4(const char* __restrict__ p) { return *reinterpret_cast<const uint32_t*>(p); }

// zero the fragment elements whose row (first row of the lane's 8: `first`) is >= rows
__device__ static inline bf16x8 mask_rows(bf16x8 f, int64_t first, int64_t rows) {
    const int64_t left = rows - first;
    const int nv = left <= 0 ? 0 : (left >= 8 ? 8 : (int)left);
    uint4 u = __builtin_bit_cast(uint4, f);
    uint32_t d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] &= (2 * k + 1 < nv) ? 0xFFFFFFFFu : (2 * k < nv ? 0x0000FFFFu : 0u);
    return __builtin_bit_cast(bf16x8, uint4{d[0], d[1], d[2], d[3]});
}

enum : int32_t { DW_HH = 0, DW_HX = 1, DW_DH = 2, DW_HR = 3, DW_RH = 4 };

template <int H>
struct DwGeom {
    static constexpr int CG = H / 32;                 // 32-column groups per row of a wide panel
    static constexpr int ROWQ = CG * 256;             // bytes per 4-row group of a wide panel
    static constexpr int PANEL = 8 * ROWQ;            // 32 rows
    static constexpr int NPW = H / 128;               // 1-KiB DMA pieces per wave and wide panel
    static constexpr int MA = H / 128, NB = H / 64;   // HH: 32-row / 32-column output tiles per wave (4 x 2 waves)
    static constexpr int LDS_BYTES = 160 * 1024;      // the whole CU: one workgroup per CU
    static constexpr int ZERO = LDS_BYTES - 16;       // 16 zero bytes
};

// LDS ring of one job kind.  A CU's share of the HBM stream is (bytes it keeps in flight) / (latency under load, 2-3 us):
// the narrow kinds (18 KB per stage) get deeper rings so that every kind keeps ~100 KB in flight.
//   slot = [P panel][second operand]; HR keeps its recomputed Q tiles (2: the stage in use, the next one) behind the ring;
//   RH: slot = [Q panel][dOut panel][mask bits], the recomputed P tiles behind the ring.
template <int H, int KIND>
struct DwRing {
    using G = DwGeom<H>;
    static constexpr bool kRecomp = KIND == DW_HR || KIND == DW_RH;          // one operand is rebuilt on chip, one stage ahead
    static constexpr int P_OFF = 0;
    static constexpr int Q_OFF = (KIND == DW_DH) ? 1024 : (KIND == DW_RH ? 0 : G::PANEL);   // DH: the [32][8] panel takes 512 B
    static constexpr int AUX_OFF = G::PANEL;                                 // RH: [32][8] dOut panel, + 1024: the mask bits
    static constexpr int SLOT = (KIND == DW_HH) ? 2 * G::PANEL : (KIND == DW_DH ? G::PANEL + 1024 : G::PANEL + 2048);
    static constexpr int QTILES = kRecomp ? 2 * G::PANEL : 0;
    static constexpr int D_FIT = (G::ZERO - QTILES) / SLOT;
    static constexpr int D = D_FIT > 8 ? 8 : D_FIT;                          // slots; D - 1 stages in flight
    static constexpr int QT_OFF = D * SLOT;
    static constexpr int NG = (KIND == DW_HH) ? 2 * G::NPW : G::NPW + 1;     // DMA instructions per wave and stage
    static_assert(D >= 4 && QT_OFF + QTILES <= G::ZERO, "ring does not fit the LDS");
    static_assert((D - 1) * NG <= 63, "vmcnt is a 6-bit counter");
};

// ---- LDS-DMA of one stage (every call issues the same number of instructions per wave) ----
template <int H>
__device__ static inline void dma_wide(const uint16_t* __restrict__ g, int64_t row0, int64_t rows, char* panel, int wave, int lane) {
    using G = DwGeom<H>;
#pragma unroll
    for (int t = 0; t < G::NPW; ++t) {
        const int piece = wave * G::NPW + t;
        const int rquad = piece / (G::CG / 4), ch = piece % (G::CG / 4);
        int64_t r = row0 + 4 * rquad + ((lane >> 2) & 3);
        r = r < rows ? r : rows - 1; r = mem_row(r);
        const uint4* src = reinterpret_cast<const uint4*>(g) + r * (H / 8) + (4 * ch + (lane >> 4)) * 4 + (lane & 3);
        // aux = 2: non-temporal -- every byte of the wide panels is read once (same-box A/B over the cache-policy bits: -1.3 %)
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(panel + piece * 1024), 16, 0, TG_DW_LOAD_AUX);
    }
}
// kSwz (HR, whose recompute reads the panel as MFMA B fragments, 16 rows x one 16-B chunk per 16 lanes: 4 rows share a bank group
// in the plain image): chunk c of row r lands in slot c ^ ((r >> 2) & 3) of its 64 bytes -- the source address does the permutation.
template <bool kSwz>
__device__ static inline void dma_x(const uint16_t* __restrict__ g, int64_t row0, int64_t rows, char* xpanel, int wave, int lane) {
    const int pc = wave & 1;                                        // 2 pieces of 16 rows x 64 B; every wave moves one
    int64_t r = row0 + 16 * pc + (lane >> 2);
    r = r < rows ? r : rows - 1; r = mem_row(r);
    const int chunk = kSwz ? ((lane & 3) ^ ((lane >> 4) & 3)) : (lane & 3);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(g) + r * 4 + chunk, (lds_void*)(xpanel + pc * 1024), 16, 0, 0);
}
__device__ static inline void dma_d8(const uint16_t* __restrict__ g, int64_t row0, int64_t rows, char* panel, int lane) {
    if (lane < 32) {                                                // 32 rows x 16 B
        int64_t r = row0 + lane;
        r = r < rows ? r : rows - 1; r = mem_row(r);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(g) + r, (lds_void*)panel, 16, 0, 0);
    }
}

// RH: the stage's [32][8] dOut rows (even waves) or its mask bits (odd waves): one instruction per wave, like dma_x.
// Mask bits land row-major: H = 256: lane L fetches the 16 B of (row L >> 1, half L & 1); H = 128: a row's two halves are 16 B.
template <int H>
__device__ static inline void dma_rh_aux(const uint16_t* __restrict__ dout, const uint32_t* __restrict__ bits, int64_t row0, int64_t rows,
                                         char* aux, int wave, int lane) {
    constexpr int MT = H / 32;
    if ((wave & 1) == 0) {
        dma_d8(dout, row0, rows, aux, lane);
    } else if constexpr (MT == 8) {
        int64_t r = row0 + (lane >> 1);
        r = r < rows ? r : rows - 1; r = mem_row(r);
        __builtin_amdgcn_global_load_lds(bits + r * MT + (lane & 1) * (MT / 2), (lds_void*)(aux + 1024), 16, 0, 0);
    } else {
        if (lane < 32) {
            int64_t r = row0 + lane;
            r = r < rows ? r : rows - 1; r = mem_row(r);
            __builtin_amdgcn_global_load_lds(bits + r * MT, (lds_void*)(aux + 1024), 16, 0, 0);
        }
    }
}

template <int H, int KIND>
__device__ static inline void dw_issue(const DwJob& job, int64_t sg, int64_t rows, char* slot, int wave, int lane) {
    using R = DwRing<H, KIND>;
    const int64_t row0 = sg * kDwStageRows;
    if constexpr (KIND == DW_HH) {
        dma_wide<H>(job.p, row0, rows, slot + R::P_OFF, wave, lane);
        dma_wide<H>(job.q, row0, rows, slot + R::Q_OFF, wave, lane);
    } else if constexpr (KIND == DW_HX || KIND == DW_HR) {
        dma_wide<H>(job.p, row0, rows, slot + R::P_OFF, wave, lane);
        dma_x<KIND == DW_HR>(job.q, row0, rows, slot + R::Q_OFF, wave, lane);
    } else if constexpr (KIND == DW_RH) {
        dma_wide<H>(job.q, row0, rows, slot + R::Q_OFF, wave, lane);
        dma_rh_aux<H>(job.p, job.aux, row0, rows, slot + R::AUX_OFF, wave, lane);
    } else {
        dma_d8(job.p, row0, rows, slot + R::P_OFF, lane);
        dma_wide<H>(job.q, row0, rows, slot + R::Q_OFF, wave, lane);
    }
}

template <int H, int KIND>
__device__ static void dw_run(const DwArgs& args, const DwJob& job, int64_t rows, float* __restrict__ ws, char* lds_c) {
    using G = DwGeom<H>;
    using R = DwRing<H, KIND>;
    constexpr int D = R::D, P = D - 1, NG = R::NG;
    constexpr bool kSquare = KIND == DW_HH || KIND == DW_HR || KIND == DW_RH;       // H x H output, 4 x 2 waves
    constexpr int MA = kSquare ? G::MA : 1;
    constexpr int NB = kSquare ? G::NB : 1;
    constexpr int NT = H / 32;                                        // 32-wide tiles across a wide operand
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int my = (int)blockIdx.x - job.first_block, nb = job.n_blocks;
    const int64_t n_st = (rows + kDwStageRows - 1) / kDwStageRows;

    // wave -> output tiles.  HH / HR: 4 (m) x 2 (n) waves; HX: wave w = m-tile w; DH: wave w = n-tile w
    const int wm = wave >> 1, wn = wave & 1;
    const bool active = kSquare ? true : wave < NT;
    // lane part of a transposed fragment read: rows 8h + q (+4), columns 16 g1 + 4 p .. +3 of the tile's 32
    const int lane_wide = 2 * h * G::ROWQ + q4 * 64 + g1 * 32 + p4 * 8;
    const int lane_x = 2 * h * 256 + q4 * 64 + g1 * 32 + p4 * 8;
    // a recomputed tile (HR: Q, RH: P) keeps the 16-B chunk c of a row's 64-byte segment in slot c ^ (quad & 3) (quad = row / 4):
    // its writers hold 16 rows x one chunk per 16 lanes, which would put 4 rows on one bank group in the plain image.  The reader's
    // low / high transposing reads look at quads 2 h and 2 h + 1 (+ 4 ks): two lane addresses instead of one.
    const int lane_sw_lo = 2 * h * G::ROWQ + q4 * 64 + (((2 * g1 + (p4 >> 1)) ^ ((2 * h) & 3)) * 16) + (p4 & 1) * 8;
    const int lane_sw_hi = 2 * h * G::ROWQ + q4 * 64 + (((2 * g1 + (p4 >> 1)) ^ ((2 * h + 1) & 3)) * 16) + (p4 & 1) * 8;
    int offA, offB;                                                   // + slot (or Q tile) base + k-step / tile immediates
    int offR_lo = 0, offR_hi = 0;                                     // the recomputed operand's pair
    if constexpr (KIND == DW_HH) {
        offA = R::P_OFF + lane_wide + (MA * wm) * 256;
        offB = R::Q_OFF + lane_wide + (NB * wn) * 256;
    } else if constexpr (KIND == DW_HR) {
        offA = R::P_OFF + lane_wide + (MA * wm) * 256;
        offB = lane_wide + (NB * wn) * 256;                           // relative to the stage's recomputed Q tile
        offR_lo = lane_sw_lo + (NB * wn) * 256;
        offR_hi = lane_sw_hi + (NB * wn) * 256;
    } else if constexpr (KIND == DW_RH) {
        offA = lane_wide + (MA * wm) * 256;                           // relative to the stage's recomputed P tile
        offR_lo = lane_sw_lo + (MA * wm) * 256;
        offR_hi = lane_sw_hi + (MA * wm) * 256;
        offB = R::Q_OFF + lane_wide + (NB * wn) * 256;
    } else if constexpr (KIND == DW_HX) {
        offA = R::P_OFF + lane_wide + (wave % NT) * 256;
        offB = R::Q_OFF + lane_x;
    } else {
        offA = R::P_OFF + (8 * h + q4) * 16 + 8 * p4;                 // [32 rows][8 columns]: lanes g1 == 0, p < 2
        offB = R::Q_OFF + lane_wide + (wave % NT) * 256;
    }
    const bool d8_valid = g1 == 0 && p4 < 2;

    f32x16 acc[MA][NB];
    f32x16 accb = {};
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x16{};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // HR: this wave's 32 first-layer features (block `wave`): weight fragments + bias in registers, in the forward chain's own
    // layout (mlp.FragmentStream(layout="chain"): v_mfma_f32_16x16x32_bf16, half f of lane (i, g) = features 8 (i >> 2) + 4 f + (i & 3))
    bf16x8 w0[2] = {};
    f32x4 b0v[2] = {};
    if constexpr (KIND == DW_HR) {
        if (wave < NT) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                w0[f] = __builtin_bit_cast(bf16x8, args.w0frag[(wave * 2 + f) * 64 + lane]);
#pragma unroll
                for (int r = 0; r < 4; ++r) b0v[f][r] = args.b0[32 * wave + 8 * (lane >> 4) + 4 * f + r];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ordinary loads: none may be counted in the ring
        // ... and hipcc must SEE them complete here: its own wait insertion does not look into the asm above, would still count
        // these registers as pending at the loop header and drain the DMA ring (vmcnt(0)) at their first use in EVERY stage
        asm volatile("" : "+v"(w0[0]), "+v"(w0[1]), "+v"(b0v[0]), "+v"(b0v[1]));
    }
    // RH: this wave's 32 top-layer features (block `wave`) of W_head^T, in the backward chain's own layout
    if constexpr (KIND == DW_RH) {
        if (wave < NT) {
#pragma unroll
            for (int f = 0; f < 2; ++f) w0[f] = __builtin_bit_cast(bf16x8, args.whfrag[(wave * 2 + f) * 64 + lane]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(w0[0]), "+v"(w0[1]));                  // (see HR above)
    }

    int64_t sg_issue = my;
    int slot_issue = 0;
#pragma unroll 1
    for (int i = 0; i < P; ++i) {
        dw_issue<H, KIND>(job, sg_issue, rows, lds_c + slot_issue * R::SLOT, wave, lane);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
    }

    // HR: a0 tile = relu(W0 . x^T + b0) for a stage's 32 rows (x panel of slot `sb`), written as the image of a Q panel into
    // tile buffer `qt`.  It runs ONE stage ahead, beside the products of the current stage (the two are independent, so
    // the recompute's dependent chain -- LDS read, 2 MFMAs, pack, LDS write -- hides behind them; no second barrier)
    auto recompute_a0 = [&](const char* sb, char* qt) {
        if (wave < NT) {
            const int col16 = lane & 15, grp = lane >> 4;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = 16 * c + col16;
                const bf16x8 xb = __builtin_bit_cast(bf16x8, lds_load16(sb + R::Q_OFF + row * 64 + 16 * (grp ^ ((row >> 2) & 3))));
                // one k-step (K = 32), bias as the initial accumulator: the forward chain's first layer, instruction for instruction
                const f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[0], xb, b0v[0], 0, 0, 0);
                const f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[1], xb, b0v[1], 0, 0, 0);
                const bf16x8 o = relu_pack_bf16(t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]);
                lds_store16(qt + (row >> 2) * G::ROWQ + wave * 256 + (row & 3) * 64 + 16 * (grp ^ ((row >> 2) & 3)), __builtin_bit_cast(uint4, o));
            }
        }
    };
    // RH: dZ_top tile = (W_head^T . dOut^T) * keep bits for a stage's 32 rows (dOut panel + mask bits of slot `sb`), written as the
    // image of a P panel: the backward chain's head block, instruction for instruction (identical bits), one stage ahead like a0
    auto recompute_dz = [&](const char* sb, char* pt) {
        if (wave < NT) {
            constexpr int MT = H / 32;
            const int col16 = lane & 15, grp = lane >> 4;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = 16 * c + col16;
                const uint4 gq = lds_load16(sb + R::AUX_OFF + row * 16);
                const bf16x8 xb = grp ? bf16x8{} : __builtin_bit_cast(bf16x8, gq);      // k = 8 g + j: outputs 0..7 sit in the g = 0 lanes
                const f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[0], xb, f32x4{}, 0, 0, 0);
                const f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[1], xb, f32x4{}, 0, 0, 0);
                // the lane's mask word: (row, half g >> 1, word wave >> 1), shifted right by its nibble 4 (g & 1)
                const uint32_t mwv = lds_load4(sb + R::AUX_OFF + 1024 + row * (MT * 4) + (grp >> 1) * (MT * 2) + (wave >> 1) * 4);
                const bf16x8 o = masked_pack(t0, t1, mwv >> (4 * (grp & 1)), wave);
                lds_store16(pt + (row >> 2) * G::ROWQ + wave * 256 + (row & 3) * 64 + 16 * (grp ^ ((row >> 2) & 3)), __builtin_bit_cast(uint4, o));
            }
        }
    };
    // stages that must have landed at the top of an iteration: the current one; HR / RH also the next (its inputs are read)
    constexpr int kWait = (R::kRecomp ? P - 2 : P - 1) * NG;
    int parity = 0;                                                   // HR / RH: recomputed tile of the current stage
    if constexpr (R::kRecomp) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NG) : "memory");      // the first stage
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (KIND == DW_HR) recompute_a0(lds_c, lds_c + R::QT_OFF);
        else recompute_dz(lds_c, lds_c + R::QT_OFF);
    }

    int slot = 0;
    // One stage.  kPartial (the one stage that holds the last rows, if they do not fill it) masks the dZ fragments; it is a
    // separate instantiation so that the loop over full stages is ONE basic block per k-step: hipcc only then issues the next
    // k-step's fragment reads among the current MFMAs, and its counted lgkmcnt lets the first products start after 4 of 12 reads.
    auto stage = [&](const int64_t sg, auto partial_c) {
        constexpr bool kPartial = decltype(partial_c)::value;
        if constexpr (R::kRecomp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's tile writes are done
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWait) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dw_issue<H, KIND>(job, sg_issue, rows, lds_c + slot_issue * R::SLOT, wave, lane);
        sg_issue += nb;
        slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;

        const char* sb = lds_c + slot * R::SLOT;
        slot = slot + 1 == D ? 0 : slot + 1;
        const int64_t row0 = sg * kDwStageRows;
        const char* pb = sb;                                          // bases of the A / B operands' panels
        const char* qb = sb;
        if constexpr (KIND == DW_HR) {
            qb = lds_c + R::QT_OFF + parity * G::PANEL;
            parity ^= 1;
        } else if constexpr (KIND == DW_RH) {
            pb = lds_c + R::QT_OFF + parity * G::PANEL;
            parity ^= 1;
        }
        if constexpr (R::kRecomp) {
            if (sg + nb < n_st) {
                if constexpr (KIND == DW_HR) recompute_a0(lds_c + slot * R::SLOT, lds_c + R::QT_OFF + parity * G::PANEL);
                else recompute_dz(lds_c + slot * R::SLOT, lds_c + R::QT_OFF + parity * G::PANEL);
            }
        }

        if constexpr (kSquare) {
            // H x H jobs: both k-steps' fragments through raw transposing reads in the order of their use, a0 b0 b1 .. a1 .. (12
            // reads per k-step at H = 256), with the wave's own counted lgkmcnt: the products of (a0, b0) start when 4 reads have
            // returned, and the second k-step's reads go out before the last group of products of the first.  LDS operations
            // return in issue order, so other accesses of this wave in the queue (the recompute's, wherever hipcc puts them) can
            // only make a counted wait stricter, never weaker: "at most N outstanding" still covers everything older than the
            // youngest N, and the raw reads and waits keep their order among themselves (volatile).
            const uint32_t pa = lds_addr(pb) + (uint32_t)(KIND == DW_RH ? offR_lo : offA);
            const uint32_t pa_hi = KIND == DW_RH ? lds_addr(pb) + (uint32_t)offR_hi : pa;
            const uint32_t qa = lds_addr(qb) + (uint32_t)(KIND == DW_HR ? offR_lo : offB);
            const uint32_t qa_hi = KIND == DW_HR ? lds_addr(qb) + (uint32_t)offR_hi : qa;
            constexpr int TOTAL = 2 * (MA + NB);
            bf16x8 fa[2][MA], fb[2][NB];
            auto reads = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int K0 = 4 * ks * G::ROWQ, K1 = (4 * ks + 1) * G::ROWQ;
                fa[ks][0] = tr_frag_raw<K0, K1>(pa, pa_hi);
                static_for<0, NB>([&](auto n) {
                    constexpr int N_ = decltype(n)::value;
                    fb[ks][N_] = tr_frag_raw<K0 + N_ * 256, K1 + N_ * 256>(qa, qa_hi);
                });
                static_for<1, MA>([&](auto m) {
                    constexpr int M_ = decltype(m)::value;
                    fa[ks][M_] = tr_frag_raw<K0 + M_ * 256, K1 + M_ * 256>(pa, pa_hi);
                });
            };
            reads(std::integral_constant<int, 0>{});
            static_for<0, 2>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                // products of a[0]: b[n] is complete when 2 + 2 (n + 1) reads have returned
                static_for<0, NB>([&](auto n) {
                    constexpr int N_ = decltype(n)::value;
                    tr_wait<TOTAL - 2 - 2 * (N_ + 1)>(fa[ks][0], fb[ks][N_]);
                    if constexpr (kPartial && N_ == 0) fa[ks][0] = mask_rows(fa[ks][0], row0 + 16 * ks + 8 * h, rows);
                    if constexpr (MA == 1 && N_ == NB - 1 && ks == 0) reads(std::integral_constant<int, 1>{});
                    acc[0][N_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], fb[ks][N_], acc[0][N_], 0, 0, 0);
                });
                static_for<1, MA>([&](auto m) {
                    constexpr int M_ = decltype(m)::value;
                    tr_wait<TOTAL - 2 - 2 * NB - 2 * M_>(fa[ks][M_], fb[ks][0]);
                    if constexpr (kPartial) fa[ks][M_] = mask_rows(fa[ks][M_], row0 + 16 * ks + 8 * h, rows);
                    // every read of this k-step has returned: the next k-step's go out under the remaining products
                    if constexpr (M_ == MA - 1 && ks == 0) reads(std::integral_constant<int, 1>{});
                    static_for<0, NB>([&](auto n) {
                        constexpr int N_ = decltype(n)::value;
                        acc[M_][N_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][M_], fb[ks][N_], acc[M_][N_], 0, 0, 0);
                    });
                });
                // bias gradient of one m-tile per wave: H = 256: tile 2 wm + wn; H = 128: the wn == 0 waves
                if constexpr (MA == 2) {
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wn ? fa[ks][1] : fa[ks][0], ones, accb, 0, 0, 0);
                } else {
                    if (wn == 0) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], ones, accb, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if (active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[MA], b[NB];
                if constexpr (KIND == DW_DH) {
                    // lanes without a column of their own (columns 8..31 of the padded tile) read the zero word
                    const int o = d8_valid ? (int)(sb - lds_c) + offA + 256 * ks : G::ZERO;
                    a[0] = tr_frag(lds_c, o, d8_valid ? o + 64 : o);
                    b[0] = tr_frag(qb, offB + 4 * ks * G::ROWQ, offB + (4 * ks + 1) * G::ROWQ);
                } else {
                    a[0] = tr_frag(pb, offA + 4 * ks * G::ROWQ, offA + (4 * ks + 1) * G::ROWQ);
                    b[0] = tr_frag(qb, offB + 4 * ks * 256, offB + (4 * ks + 1) * 256);
                }
                if constexpr (kPartial) {
#pragma unroll
                    for (int m = 0; m < MA; ++m) a[m] = mask_rows(a[m], row0 + 16 * ks + 8 * h, rows);
                }
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
                if constexpr (KIND == DW_HX) {
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], ones, accb, 0, 0, 0);
                }
            }
        }
    };
    const int64_t n_full = rows / kDwStageRows;                       // stages below this index hold 32 rows
    int64_t sg = my;
#pragma unroll 1
    for (; sg < n_full; sg += nb) stage(sg, std::false_type{});
    if (sg < n_st) stage(sg, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // no LDS-DMA may outlive the workgroup's LDS allocation

    // ---- this workgroup's slab: [M][N] gradient (+ [M] bias sums) ----
    float* slab = ws + job.slab_off + (int64_t)my * job.slab_len;
    const int col = lane & 31;
    if (active) {
        if constexpr (kSquare) {
#pragma unroll
            for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int m0 = 32 * (MA * wm + m), n0 = 32 * (NB * wn + n);
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[(m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * H + n0 + col] = acc[m][n][r];
                }
            const bool has_bias = (MA == 2) || wn == 0;
            if (has_bias && col == 0) {
                const int m0 = 32 * (MA == 2 ? 2 * wm + wn : wm);
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[H * H + m0 + (r & 3) + 8 * (r >> 2) + 4 * h] = accb[r];
            }
        } else if constexpr (KIND == DW_HX) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + col] = acc[0][0][r];
            if (col == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[H * 32 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h] = accb[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(r + 4 * h) * H + 32 * wave + col] = acc[0][0][r];    // rows 0..7 of the padded 32
        }
    }
}

TG_CLOCK_PROBE_VAR(g_probe_weight_grad, attach_probe_weight_grad)

template <int H>
__global__ __launch_bounds__(512, 2) void dw_kernel(DwArgs args, int64_t rows, float* __restrict__ ws) {
    extern __shared__ uint4 lds[];
    char* lds_c = reinterpret_cast<char*>(lds);
    TG_CLOCK_PROBE_BEGIN(g_probe_weight_grad)
    if (threadIdx.x == 0) lds[DwGeom<H>::ZERO / 16] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    int j = 0;
#pragma unroll
    for (int t = 1; t < kDwMaxJobs; ++t)
        if (t < args.n_jobs && (int)blockIdx.x >= args.job[t].first_block) j = t;
    const DwJob job = args.job[j];
    switch (job.kind) {
        case DW_HH: dw_run<H, DW_HH>(args, job, rows, ws, lds_c); break;
        case DW_HX: dw_run<H, DW_HX>(args, job, rows, ws, lds_c); break;
        case DW_DH: dw_run<H, DW_DH>(args, job, rows, ws, lds_c); break;
        case DW_RH: dw_run<H, DW_RH>(args, job, rows, ws, lds_c); break;
        default: dw_run<H, DW_HR>(args, job, rows, ws, lds_c); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TG_CLOCK_PROBE_END(g_probe_weight_grad)
}

// grad[m][n] += sum over the job's slabs, in a fixed order.  A region whose slabs are few (the jobs' own: one per workgroup of the
// job, 40-75) takes one thread per output element -- consecutive threads read consecutive floats of a slab row; a region with many
// slabs (the chain kernels' riders: up to 4 per workgroup of a 256-workgroup launch) takes `lanes` consecutive lanes per element,
// lane `sub` adding slabs sub, sub + lanes, ... and a fixed shuffle tree adding the lanes: one thread walking 1,024 slabs was a
// 150-us chain of load latencies (C4: 6 % of an update).  Regions start on a multiple of 64 units, so a wavefront never straddles two.
struct DwFinishDesc {
    const float* slab;        // slab 0, already offset to the region ([M][N] gradient or [1][M] bias sums)
    float* grad;
    int64_t grad_ld;
    int32_t slab_len, n_slabs, N, m_out, n_out, first_elem, lanes;     // first_elem: in UNITS (an element = `lanes` units)
};
constexpr int kDwMaxExtra = 4;          // the chain kernels' own partial gradients riding on this launch (tg_mlp_weight_grad_ex)
constexpr int kDwMaxFinish = 2 * kDwMaxJobs + kDwMaxExtra;
struct DwFinishArgs {
    DwFinishDesc d[kDwMaxFinish]; int32_t n; int32_t total;
    // optional rider: loss_sums[k] += sum over rows [0, n_loss_rows) of loss_work[row][k], k < 4 (the forward chain's per-workgroup
    // f64 loss sums: this launch follows it in stream order): the workgroup behind the last region, one wavefront per k
    const double* loss_work; double* loss_sums; int32_t n_loss_rows;
};

__global__ __launch_bounds__(256) void dw_finish_all_kernel(DwFinishArgs fa) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= fa.total) {                                    // (fa.total is a multiple of 256: this is a whole workgroup)
        if (fa.loss_work != nullptr && e < fa.total + 256) {
            const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
            double t = 0.0;
            for (int r = lane; r < fa.n_loss_rows; r += 64) t += fa.loss_work[(int64_t)r * 4 + k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) fa.loss_sums[k] += t;
        }
        return;
    }
    int k = 0;
#pragma unroll
    for (int t = 1; t < kDwMaxFinish; ++t)
        if (t < fa.n && e >= fa.d[t].first_elem) k = t;
    const DwFinishDesc d = fa.d[k];                         // (wave-uniform: regions are 64-aligned)
    const int unit = e - d.first_elem;
    const int le = unit / d.lanes, sub = unit - le * d.lanes;
    const bool live = le < d.m_out * d.n_out;
    const int m = live ? le / d.n_out : 0, n = live ? le - m * d.n_out : 0;
    const float* src = d.slab + (int64_t)m * d.N + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = sub;
    const int step = d.lanes;
    if (live) {
        for (; b + 3 * step < d.n_slabs; b += 4 * step) {   // 4 loads in flight; the sum order stays fixed
            const float v0 = src[(int64_t)b * d.slab_len], v1 = src[(int64_t)(b + step) * d.slab_len];
            const float v2 = src[(int64_t)(b + 2 * step) * d.slab_len], v3 = src[(int64_t)(b + 3 * step) * d.slab_len];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; b < d.n_slabs; b += step) s0 += src[(int64_t)b * d.slab_len];
    }
    float t = (s0 + s1) + (s2 + s3);
    for (int off = d.lanes >> 1; off > 0; off >>= 1) t += __shfl_down(t, off, 64);    // (lanes = 1: no iteration)
    if (live && sub == 0) d.grad[(int64_t)m * d.grad_ld + n] += t;
}

static int dw_cus() { return device_cus(); }

// Relative cost of a row of each job kind = its bytes, times a per-kind factor for the kinds that are not purely
// byte-bound (HR recomputes a tile per stage).  The workgroups are split between the jobs in proportion to it.
static int64_t dw_row_cost(int H, int kind) {
    static const int pct[5] = {100, 100, 100, 260, 300};   // HR / RH: measured optimum (tools/dw_probe.py, 2^22 rows, the learner's job set)
    int64_t bytes;
    switch (kind) {
        case DW_HH: bytes = 4 * H; break;
        case DW_HX: case DW_HR: bytes = 2 * H + 64; break;
        case DW_RH: bytes = 2 * H + 16 + H / 8; break;
        default: bytes = 2 * H + 16; break;
    }
    return bytes * pct[kind];
}
static int dw_slab_len(int H, int kind) {
    switch (kind) {
        case DW_HH: case DW_HR: case DW_RH: return H * H + H;
        case DW_HX: return H * 32 + H;
        default: return 8 * H;
    }
}

}  // namespace tg

using namespace tg;

template <int H>
static int launch_dw(const DwArgs& args, int grid, int64_t rows, float* ws, hipStream_t st) {
    auto kern = dw_kernel<H>;
    const size_t shmem = DwGeom<H>::LDS_BYTES;
    static LdsOptIn opt_in;
    if (int rc = reserve_dynamic_lds((const void*)kern, shmem, opt_in, "tg_mlp_weight_grad")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), shmem, st, args, rows, ws);
    TG_LAUNCH_CHECK("tg_mlp_weight_grad");
    return TG_OK;
}

extern "C" {

int64_t tg_mlp_weight_grad_workspace(int32_t hidden) {
    if (hidden != 128 && hidden != 256) return 0;
    return (int64_t)dw_cus() * (hidden * hidden + hidden) * (int64_t)sizeof(float);
}

int tg_mlp_weight_grad(int32_t hidden, const tg_dw_job* jobs, int32_t n_jobs, int64_t rows, const void* d_w0frag, const float* d_b0,
                       const void* d_whfrag, void* d_workspace, int64_t workspace_bytes, void* stream) {
    return tg_mlp_weight_grad_ex(hidden, jobs, n_jobs, rows, d_w0frag, d_b0, d_whfrag, d_workspace, workspace_bytes, nullptr, 0, nullptr, 0,
                                 nullptr, stream);
}

int tg_mlp_weight_grad_ex(int32_t hidden, const tg_dw_job* jobs, int32_t n_jobs, int64_t rows, const void* d_w0frag, const float* d_b0,
                          const void* d_whfrag, void* d_workspace, int64_t workspace_bytes, const tg_slab_sum* extra, int32_t n_extra,
                          const double* d_loss_work, int32_t n_loss_rows, double* d_loss_sums, void* stream) {
    TG_REQUIRE(jobs && d_workspace, "tg_mlp_weight_grad: null pointer");
    TG_REQUIRE(n_extra >= 0 && n_extra <= kDwMaxExtra && (n_extra == 0 || extra), "tg_mlp_weight_grad_ex: %d extra reductions outside 0..%d", n_extra, kDwMaxExtra);
    TG_REQUIRE((d_loss_work == nullptr) == (d_loss_sums == nullptr) && n_loss_rows >= 0, "tg_mlp_weight_grad_ex: loss-sum rider needs both pointers");
    for (int x = 0; x < n_extra; ++x)
        TG_REQUIRE(extra[x].d_slab && extra[x].d_grad && extra[x].n_slabs >= 0 && extra[x].m_out >= 1 && extra[x].n_out >= 1 && extra[x].row_pitch >= extra[x].n_out &&
                   extra[x].slab_stride >= (int64_t)(extra[x].m_out - 1) * extra[x].row_pitch + extra[x].n_out && extra[x].slab_stride < ((int64_t)1 << 31),
                   "tg_mlp_weight_grad_ex: extra reduction %d is malformed", x);
    TG_REQUIRE(hidden == 128 || hidden == 256, "tg_mlp_weight_grad: hidden width %d unsupported (128, 256)", hidden);
    TG_REQUIRE(n_jobs >= 1 && n_jobs <= kDwMaxJobs, "tg_mlp_weight_grad: %d jobs outside 1..%d", n_jobs, kDwMaxJobs);
    TG_REQUIRE(rows >= 0, "tg_mlp_weight_grad: negative row count");
    TG_REQUIRE(workspace_bytes >= tg_mlp_weight_grad_workspace(hidden), "tg_mlp_weight_grad: workspace of %lld B is smaller than %lld B",
               (long long)workspace_bytes, (long long)tg_mlp_weight_grad_workspace(hidden));
    if (rows == 0) return TG_OK;
    const int H = hidden;
    int64_t wsum = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const tg_dw_job& jb = jobs[j];
        TG_REQUIRE(jb.kind >= TG_DW_HH && jb.kind <= TG_DW_RH, "tg_mlp_weight_grad: job %d has kind %d", j, jb.kind);
        TG_REQUIRE(jb.d_p && jb.d_q && jb.d_wgrad, "tg_mlp_weight_grad: job %d has a null pointer", j);
        const int M = jb.kind == TG_DW_DH ? 8 : H, N = (jb.kind == TG_DW_HX) ? 32 : H;
        TG_REQUIRE(jb.m_out >= 1 && jb.m_out <= M && jb.n_out >= 1 && jb.n_out <= N && jb.wgrad_ld >= jb.n_out,
                   "tg_mlp_weight_grad: job %d: window %d x %d (ld %lld) outside %d x %d", j, jb.m_out, jb.n_out, (long long)jb.wgrad_ld, M, N);
        TG_REQUIRE(jb.kind != TG_DW_DH || !jb.d_bgrad, "tg_mlp_weight_grad: job %d: the head's bias gradient comes from tg_head_prep", j);
        TG_REQUIRE(jb.kind != TG_DW_HR || (d_w0frag && d_b0), "tg_mlp_weight_grad: job %d recomputes the first layer: weights / bias missing", j);
        TG_REQUIRE(jb.kind != TG_DW_RH || (d_whfrag && jb.d_aux), "tg_mlp_weight_grad: job %d recomputes the top dZ: head weights / mask bits missing", j);
        wsum += dw_row_cost(H, jb.kind);
    }
    // workgroups per job in proportion to its bytes per row (largest remainders), at least one, at most one per 4 stages
    const int64_t n_st = ceil_div(rows, (int64_t)kDwStageRows);
    const int cap = (int)(n_st < 4 ? 1 : (n_st / 4 > 1 << 20 ? 1 << 20 : n_st / 4));
    const int cus = dw_cus();
    int alloc[kDwMaxJobs];
    int64_t rem[kDwMaxJobs];
    int used = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int64_t w = dw_row_cost(H, jobs[j].kind) * cus;
        alloc[j] = (int)(w / wsum);
        rem[j] = w % wsum;
        if (alloc[j] < 1) { alloc[j] = 1; rem[j] = 0; }
        used += alloc[j];
    }
    while (used < cus) {
        int best = 0;
        for (int j = 1; j < n_jobs; ++j)
            if (rem[j] > rem[best]) best = j;
        if (rem[best] == 0) break;
        ++alloc[best];
        rem[best] = 0;
        ++used;
    }
    DwArgs args{};
    args.n_jobs = n_jobs;
    args.w0frag = (const uint4*)d_w0frag;
    args.b0 = d_b0;
    args.whfrag = (const uint4*)d_whfrag;
    DwFinishArgs fa{};
    int grid = 0, elems = 0;
    int64_t off = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const tg_dw_job& jb = jobs[j];
        DwJob& dj = args.job[j];
        dj.p = (const uint16_t*)jb.d_p;
        dj.q = (const uint16_t*)jb.d_q;
        dj.aux = (const uint32_t*)jb.d_aux;
        dj.kind = jb.kind;
        dj.first_block = grid;
        dj.n_blocks = alloc[j] < cap ? alloc[j] : cap;
        dj.slab_len = dw_slab_len(H, jb.kind);
        dj.slab_off = off;
        grid += dj.n_blocks;
        const int N = (jb.kind == TG_DW_HX) ? 32 : H;
        DwFinishDesc& fd = fa.d[fa.n++];
        auto region = [&](DwFinishDesc& d) {                 // units of the region, rounded up to whole wavefronts
            d.lanes = d.n_slabs > 96 ? 32 : 1;
            d.first_elem = elems;
            elems += (d.m_out * d.n_out * d.lanes + 63) / 64 * 64;
        };
        fd = DwFinishDesc{(const float*)d_workspace + off, jb.d_wgrad, jb.wgrad_ld, dj.slab_len, dj.n_blocks, N, jb.m_out, jb.n_out, 0, 1};
        region(fd);
        if (jb.d_bgrad) {
            const int boff = jb.kind == TG_DW_HX ? H * 32 : H * H;
            DwFinishDesc& fb = fa.d[fa.n++];
            fb = DwFinishDesc{(const float*)d_workspace + off + boff, jb.d_bgrad, (int64_t)H, dj.slab_len, dj.n_blocks, H, 1, jb.m_out, 0, 1};
            region(fb);
        }
        off += (int64_t)dj.n_blocks * dj.slab_len;
    }
    for (int x = 0; x < n_extra; ++x) {
        DwFinishDesc& fd = fa.d[fa.n++];
        fd = DwFinishDesc{extra[x].d_slab, extra[x].d_grad, extra[x].grad_ld, (int32_t)extra[x].slab_stride, extra[x].n_slabs, extra[x].row_pitch,
                          extra[x].m_out, extra[x].n_out, 0, 1};
        fd.lanes = fd.n_slabs > 96 ? 32 : 1;
        fd.first_elem = elems;
        elems += (fd.m_out * fd.n_out * fd.lanes + 63) / 64 * 64;
    }
    fa.loss_work = d_loss_work; fa.loss_sums = d_loss_sums; fa.n_loss_rows = n_loss_rows;
    elems = (elems + 255) / 256 * 256;                      // (the loss rider is the workgroup behind the regions)
    fa.total = elems;
    TG_REQUIRE(grid <= cus, "tg_mlp_weight_grad: %d workgroups for %d CUs (the workspace holds one slab per CU)", grid, cus);
    hipStream_t st = (hipStream_t)stream;
    int rc = hidden == 256 ? launch_dw<256>(args, grid, rows, (float*)d_workspace, st) : launch_dw<128>(args, grid, rows, (float*)d_workspace, st);
    if (rc != TG_OK) return rc;
    hipLaunchKernelGGL(dw_finish_all_kernel, dim3((unsigned)(elems / 256 + (d_loss_work ? 1 : 0))), dim3(256), 0, st, fa);
    TG_LAUNCH_CHECK("tg_mlp_weight_grad (finish)");
    return TG_OK;
}

}  // extern "C"
