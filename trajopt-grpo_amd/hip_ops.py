"""Tensor-level wrappers over the C ABI for the returns / advantage / loss kernels.

Every function takes CUDA (ROCm) tensors and enqueues on torch's current stream; none has a
CPU fallback.  Layouts are the time-major SoA of the device trajectory: `[T][n]`, env fastest.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N


def _st(t):
    return N.stream_ptr(t.device)


def rtg_scan(rew: torch.Tensor, mask: torch.Tensor, gamma: float) -> torch.Tensor:
    """Reward-to-go (algorithms/grpo.py:66-74 == algorithms/ppo.py:100-111).  rew f32 [T][n], mask u8 [T][n]."""
    N.require_cuda(rew, mask)
    assert rew.dtype == torch.float32 and mask.dtype == torch.uint8 and rew.is_contiguous() and mask.is_contiguous()
    T, n = rew.shape
    out = torch.empty_like(rew)
    N.check(N.load().tg_rtg_scan(rew.data_ptr(), mask.data_ptr(), float(gamma), out.data_ptr(), n, T, _st(rew)), "tg_rtg_scan")
    return out


def gae_scan(rew, values, mask, gamma: float, lam: float):
    """GAE advantages and returns (algorithms/ppo.py:112-124)."""
    N.require_cuda(rew, values, mask)
    assert rew.dtype == values.dtype == torch.float32 and mask.dtype == torch.uint8
    assert rew.is_contiguous() and values.is_contiguous() and mask.is_contiguous()
    T, n = rew.shape
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    N.check(N.load().tg_gae_scan(rew.data_ptr(), values.data_ptr(), mask.data_ptr(), float(gamma), float(lam),
                                 adv.data_ptr(), ret.data_ptr(), n, T, _st(rew)), "tg_gae_scan")
    return adv, ret


def masked_moments(x: torch.Tensor, mask: torch.Tensor, group_size: int) -> torch.Tensor:
    """f64 [n/group_size][3] = (count, sum, sum of squares) over the valid entries of each group."""
    N.require_cuda(x, mask)
    assert x.dtype == torch.float32 and mask.dtype == torch.uint8 and x.is_contiguous() and mask.is_contiguous()
    T, n = x.shape
    out = torch.empty(n // group_size, 3, dtype=torch.float64, device=x.device)
    work = torch.empty(3 * n, dtype=torch.float64, device=x.device)
    N.check(N.load().tg_masked_moments(x.data_ptr(), mask.data_ptr(), n, T, int(group_size), out.data_ptr(),
                                       work.data_ptr(), _st(x)), "tg_masked_moments")
    return out


def group_normalize(x, mask, moments, mode: int, group_size: int) -> torch.Tensor:
    """mode 0: (x-mean_g)/std_g (GRPO, grpo.py:115); mode 1: /(std_g+1e-8) (PPO, ppo.py:138-139)."""
    N.require_cuda(x, mask, moments)
    assert moments.dtype == torch.float64 and moments.is_contiguous()
    T, n = x.shape
    out = torch.empty_like(x)
    N.check(N.load().tg_group_normalize(x.data_ptr(), mask.data_ptr(), moments.data_ptr(), int(mode), out.data_ptr(),
                                        n, T, int(group_size), _st(x)), "tg_group_normalize")
    return out


def returns_moments(rew: torch.Tensor, mask: torch.Tensor, gamma: float, group_size: int):
    """(rtg, moments) = (rtg_scan(rew, mask, gamma), masked_moments(rtg, mask, group_size)), bit-identical, in two launches built for
    rollouts of a few thousand envs (tg_returns_moments: LDS-staged strips, one lane per env on the recurrence)."""
    N.require_cuda(rew, mask)
    assert rew.dtype == torch.float32 and mask.dtype == torch.uint8 and rew.is_contiguous() and mask.is_contiguous()
    T, n = rew.shape
    rtg = torch.empty_like(rew)
    moments = torch.empty(n // group_size, 3, dtype=torch.float64, device=rew.device)
    work = torch.empty(3 * n, dtype=torch.float64, device=rew.device)
    N.check(N.load().tg_returns_moments(rew.data_ptr(), mask.data_ptr(), float(gamma), rtg.data_ptr(), n, T, int(group_size),
                                        moments.data_ptr(), work.data_ptr(), _st(rew)), "tg_returns_moments")
    return rtg, moments


def returns_moments_max_horizon() -> int:
    return int(N.load().tg_returns_moments_max_horizon())


def learn_count(mask: torch.Tensor, expected_rows: int, work: torch.Tensor, total: torch.Tensor) -> None:
    """tg_learn_count: per-chunk counts of the flat mask and their exclusive prefix into `work` (int32 buffer of
    learn_count_workspace(mask.numel()) bytes); total int64 [2] = (valid entries, 1 if != expected_rows >= 0)."""
    N.require_cuda(mask, work, total)
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and total.dtype == torch.int64 and total.numel() >= 2
    N.check(N.load().tg_learn_count(mask.data_ptr(), mask.numel(), int(expected_rows), work.data_ptr(), work.numel() * work.element_size(),
                                    total.data_ptr(), _st(mask)), "tg_learn_count")


def learn_count_workspace(entries: int) -> int:
    return int(N.load().tg_learn_count_workspace(int(entries)))


def learn_compact(traj, offsets, rows_cap: int, xin: torch.Tensor, ones_col: int, act_rows, idx, src0=None, dst0=None, src1=None, dst1=None,
                  moments=None, norm_mode: int = 0, group_size: int = 0) -> None:
    """tg_learn_compact on a DeviceTrajectory: the valid rows' flat indices, padded input rows, action rows and up to two per-row
    scalars (src0 optionally normalised with `moments`), in time-major order, in one launch."""
    N.require_cuda(traj.mask, traj.obs, traj.act, xin, act_rows, idx, src0, dst0, src1, dst1, moments)
    assert xin.dim() == 2 and xin.is_contiguous() and xin.dtype in (torch.bfloat16, torch.float32) and xin.shape[0] >= rows_cap
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() >= rows_cap
    assert traj.obs.is_contiguous() and traj.act.is_contiguous() and traj.mask.is_contiguous()
    a = N.CompactArgs()
    a.d_mask, a.d_offsets, a.n, a.T, a.S, a.A = traj.mask.data_ptr(), offsets.data_ptr(), traj.n, traj.T, traj.S, traj.A
    a.obs_dtype, a.d_obs, a.obs_feat_stride, a.d_act = N.dtype_code(traj.obs.dtype), traj.obs.data_ptr(), (traj.T + 1) * traj.n, traj.act.data_ptr()
    a.d_xin, a.in_pad, a.xin_bf16, a.ones_col, a.norm_mode = xin.data_ptr(), xin.shape[1], int(xin.dtype == torch.bfloat16), int(ones_col), int(norm_mode)
    if act_rows is not None:
        assert act_rows.dtype == torch.float32 and act_rows.is_contiguous() and act_rows.shape[0] >= rows_cap and act_rows.shape[1] == traj.A
    a.d_act_rows, a.d_idx = N.ptr(act_rows), idx.data_ptr()
    for s_, d_ in ((src0, dst0), (src1, dst1)):
        assert (s_ is None) == (d_ is None)
        if s_ is not None:
            assert s_.dtype == d_.dtype == torch.float32 and s_.is_contiguous() and d_.is_contiguous() and s_.numel() == traj.T * traj.n and d_.numel() >= rows_cap
    a.d_src0, a.d_dst0, a.d_src1, a.d_dst1 = N.ptr(src0), N.ptr(dst0), N.ptr(src1), N.ptr(dst1)
    if moments is not None:
        assert moments.dtype == torch.float64 and moments.is_contiguous() and moments.numel() == 3 * (traj.n // group_size)
    a.d_moments, a.group_size, a.rows_cap = N.ptr(moments), int(group_size), int(rows_cap)
    N.check(N.load().tg_learn_compact(C.byref(a), _st(xin)), "tg_learn_compact")


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> None:
    """dst.view(-1)[idx[r]] = src[r][0] (tg_scatter_rows): src f32 [rows][>= 1] with any row stride, idx int64 [rows], dst f32."""
    N.require_cuda(src, idx, dst)
    rows = idx.numel()
    assert src.dtype == dst.dtype == torch.float32 and idx.dtype == torch.int64 and idx.is_contiguous() and dst.is_contiguous()
    assert src.dim() == 2 and src.shape[0] >= rows
    N.check(N.load().tg_scatter_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), rows, dst.data_ptr(), _st(dst)), "tg_scatter_rows")


def ppo_returns(rew, values, mask, gamma: float, lam: float, monte_carlo: bool, adv: torch.Tensor, ret: torch.Tensor,
                work: torch.Tensor = None) -> torch.Tensor:
    """tg_ppo_returns: fills adv / ret f32 [T][n] (ppo.py:100-124) and returns the f64 [2][3] masked moments {count, sum, sum of
    squares} of the advantages and of the returns -- bit-identical to rtg_scan / `rtg - values` / gae_scan + masked_moments(group = n)."""
    N.require_cuda(rew, values, mask, adv, ret, work)
    T, n = rew.shape
    for t in (rew, values, adv, ret):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == T * n
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == T * n
    if work is None:
        work = torch.empty(6 * n, dtype=torch.float64, device=rew.device)
    assert work.dtype == torch.float64 and work.numel() >= 6 * n
    moments = torch.empty(2, 3, dtype=torch.float64, device=rew.device)
    N.check(N.load().tg_ppo_returns(rew.data_ptr(), values.data_ptr(), mask.data_ptr(), float(gamma), float(lam), 1 if monte_carlo else 0,
                                    adv.data_ptr(), ret.data_ptr(), n, T, moments.data_ptr(), work.data_ptr(), _st(rew)), "tg_ppo_returns")
    return moments


def ppo_norm(moments: torch.Tensor, c1: float, kl_coeff: float, out: torch.Tensor = None) -> torch.Tensor:
    """tg_ppo_norm: f32 [8] = {adv mean, 1 / (adv std + 1e-8), ret mean, 1 / (ret std + 1e-8), -1 / n, c1 / n, kl_coeff / n, n} on the
    device (what the loss heads read through `norm8=`)."""
    N.require_cuda(moments, out)
    assert moments.dtype == torch.float64 and moments.is_contiguous() and moments.numel() == 6
    if out is None:
        out = torch.empty(8, dtype=torch.float32, device=moments.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == 8
    N.check(N.load().tg_ppo_norm(moments.data_ptr(), float(c1), float(kl_coeff), out.data_ptr(), _st(moments)), "tg_ppo_norm")
    return out


def gather_rows2(idx: torch.Tensor, src0: torch.Tensor, dst0: torch.Tensor, src1: torch.Tensor = None, dst1: torch.Tensor = None) -> None:
    """dst0[r] = src0.view(-1)[idx[r]] (and dst1 / src1): tg_gather_rows2."""
    N.require_cuda(idx, src0, dst0, src1, dst1)
    rows = idx.numel()
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    for s_, d_ in ((src0, dst0), (src1, dst1)):
        assert (s_ is None) == (d_ is None)
        if s_ is not None:
            assert s_.dtype == d_.dtype == torch.float32 and s_.is_contiguous() and d_.is_contiguous() and d_.numel() >= rows
    N.check(N.load().tg_gather_rows2(idx.data_ptr(), rows, src0.data_ptr(), dst0.data_ptr(), N.ptr(src1), N.ptr(dst1), _st(idx)),
            "tg_gather_rows2")


def _var_array(var):
    v = [float(x) for x in var]
    return (C.c_float * len(v))(*v), len(v)


def gaussian_logp(mean: torch.Tensor, act: torch.Tensor, var, out: torch.Tensor = None) -> torch.Tensor:
    """log N(act; mean, diag(var)) per row (actor_critic.py:159-160).  mean [M][A] f32 (row stride free),
    act any 2-D strided [M][A] f32.  out: a contiguous f32 [M] to write into (e.g. a slice of a larger result)."""
    N.require_cuda(mean, act, out)
    assert mean.dtype == act.dtype == torch.float32 and mean.dim() == 2 and mean.stride(1) == 1
    M, A = mean.shape
    va, k = _var_array(var)
    assert k == A
    if out is None:
        out = torch.empty(M, dtype=torch.float32, device=mean.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == M
    N.check(N.load().tg_gaussian_logp(mean.data_ptr(), mean.stride(0), act.data_ptr(), act.stride(0), act.stride(1),
                                      va, A, out.data_ptr(), M, _st(mean)), "tg_gaussian_logp")
    return out


def surrogate_loss(mean, value, act, logp_old, adv, ret, mask, norm, var, epsilon, surr_coef, critic_coef, kl_coef,
                   want_total: bool = True, coef: torch.Tensor = None):
    """One launch of tg_surrogate_loss: returns (total f32 scalar, sums f64[4], d total/d mean, d total/d value|None).
    want_total=False skips the handful of scalar launches that combine the sums (the learners only use the sums).
    coef: device f32 [3] {surr_coef, critic_coef, kl_coef} used instead of the three host numbers (PPO: tg_ppo_norm's output [4:7])."""
    N.require_cuda(mean, act, logp_old, adv, coef)
    assert mean.dtype == torch.float32 and mean.dim() == 2 and mean.stride(1) == 1
    M, A = mean.shape
    a = N.LossArgs()
    a.d_mean, a.mean_row_stride = mean.data_ptr(), mean.stride(0)
    a.d_act, a.act_row_stride, a.act_col_stride = act.data_ptr(), act.stride(0), act.stride(1)
    a.d_logp_old, a.d_adv = logp_old.data_ptr(), adv.data_ptr()
    grad_value = None
    if value is not None:
        assert value.dtype == torch.float32 and value.is_contiguous() and ret is not None and ret.is_contiguous()
        grad_value = torch.empty_like(value)
        a.d_value, a.d_ret, a.d_grad_value = value.data_ptr(), ret.data_ptr(), grad_value.data_ptr()
    a.d_mask = N.ptr(mask)
    a.d_norm = N.ptr(norm)
    if coef is not None:
        assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.numel() >= 3 and not want_total
        a.d_coef = coef.data_ptr()
    va, k = _var_array(var)
    assert k == A
    for i in range(A):
        a.var[i] = va[i]
    a.act_dim, a.epsilon = A, float(epsilon)
    a.surr_coef, a.critic_coef, a.kl_coef = float(surr_coef), float(critic_coef), float(kl_coef)
    grad_mean = torch.empty(M, A, dtype=torch.float32, device=mean.device)
    sums = torch.empty(4, dtype=torch.float64, device=mean.device)
    work = torch.empty(4 * N.load().tg_loss_work_blocks(), dtype=torch.float64, device=mean.device)
    a.d_grad_mean, a.d_sums, a.d_work, a.M = grad_mean.data_ptr(), sums.data_ptr(), work.data_ptr(), M
    N.check(N.load().tg_surrogate_loss(C.byref(a), _st(mean)), "tg_surrogate_loss")
    total = (surr_coef * sums[0] + critic_coef * sums[1] + kl_coef * sums[2]).float() if want_total else None
    return total, sums, grad_mean, grad_value


class SurrogateLoss(torch.autograd.Function):
    """Fused clipped-surrogate (+ value MSE + KL-ish penalty) head: one kernel computes the loss sums
    AND d(total)/d(mean), d(total)/d(value); backward just hands those to autograd so the MLP
    forward/backward stay on PyTorch-ROCm.

        total = surr_coef * sum_i min(rho_i A_i, clip(rho_i) A_i)
              + critic_coef * sum_i (V_i - R_i)^2 + kl_coef * sum_i exp(lp_old_i)(lp_old_i - lp_i)

    GRPO (grpo.py:137-145): surr_coef=+1/G, descent on J as the reference writes it.
    PPO  (ppo.py:159-179): surr_coef=-1/n, critic_coef=c1/n, kl_coef=kl_coeff/n.
    Returns (total f32 scalar, sums f64[4] = [sum surrogate, sum sq err, sum kl, #valid]).
    """

    @staticmethod
    def forward(ctx, mean, value, act, logp_old, adv, ret, mask, norm, var, epsilon, surr_coef, critic_coef, kl_coef):
        total, sums, grad_mean, grad_value = surrogate_loss(mean, value, act, logp_old, adv, ret, mask, norm, var,
                                                            epsilon, surr_coef, critic_coef, kl_coef)
        ctx.save_for_backward(grad_mean, grad_value)
        ctx.has_value = value is not None
        ctx.mark_non_differentiable(sums)
        return total, sums

    @staticmethod
    def backward(ctx, g_total, g_sums):
        grad_mean, grad_value = ctx.saved_tensors
        gm = grad_mean * g_total
        gv = grad_value * g_total if ctx.has_value else None
        return (gm, gv) + (None,) * 11
