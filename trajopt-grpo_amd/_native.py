"""ctypes binding of libtrajopt_grpo_hip.so (the C ABI in include/trajopt_grpo_hip.h).

There is no CPU fallback: if the HIP library is missing, or an entry point is called
without a GPU tensor, this module raises.  `import torch` happens first on purpose so the
library resolves `libamdhip64.so.7` to the HIP runtime torch already loaded (one runtime
per process: device pointers and streams are shared with PyTorch-ROCm).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TG_NATIVE_LIB: another build of the same library (probe builds with different compile-time flags, tools/mall_probe.py)
LIB_PATH = os.environ.get("TG_NATIVE_LIB") or os.path.join(_HERE, "libtrajopt_grpo_hip.so")
ABI_VERSION = 12                     # TG_ABI_VERSION of include/trajopt_grpo_hip.h this binding was written for

TG_ENV_CARTPOLE, TG_ENV_QUADPOLE2D, TG_ENV_QUADPOLE, TG_ENV_QUADROTOR12, TG_ENV_PENDULUM = 0, 1, 2, 3, 4
TG_F32, TG_F64 = 0, 1
ENV_IDS = {"CartPole": TG_ENV_CARTPOLE, "QuadPole2D": TG_ENV_QUADPOLE2D, "QuadPole": TG_ENV_QUADPOLE,
           "Quadrotor": TG_ENV_QUADROTOR12, "Pendulum": TG_ENV_PENDULUM}


class NativeLibraryError(RuntimeError):
    pass


class EnvParams(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("max_steps", C.c_int32), ("time_trunc_step", C.c_int32),
                ("agents", C.c_int32), ("timestep", C.c_double), ("p", C.c_double * 12)]


class Traj(C.Structure):
    _fields_ = [("d_obs", C.c_void_p), ("d_act", C.c_void_p), ("d_rew", C.c_void_p), ("d_mask", C.c_void_p),
                ("d_len", C.c_void_p), ("d_counters", C.c_void_p), ("n", C.c_int64), ("horizon", C.c_int32),
                ("dtype", C.c_int32)]


class LossArgs(C.Structure):
    _fields_ = [("d_mean", C.c_void_p), ("mean_row_stride", C.c_int64),
                ("d_act", C.c_void_p), ("act_row_stride", C.c_int64), ("act_col_stride", C.c_int64),
                ("d_logp_old", C.c_void_p), ("d_adv", C.c_void_p), ("d_value", C.c_void_p), ("d_ret", C.c_void_p),
                ("d_mask", C.c_void_p), ("d_norm", C.c_void_p),
                ("var", C.c_float * 8), ("act_dim", C.c_int32), ("epsilon", C.c_float), ("surr_coef", C.c_float),
                ("critic_coef", C.c_float), ("kl_coef", C.c_float),
                ("d_grad_mean", C.c_void_p), ("d_grad_value", C.c_void_p), ("d_sums", C.c_void_p),
                ("d_work", C.c_void_p), ("M", C.c_int64), ("d_coef", C.c_void_p)]


class DwJob(C.Structure):
    _fields_ = [("d_p", C.c_void_p), ("d_q", C.c_void_p), ("d_wgrad", C.c_void_p), ("d_bgrad", C.c_void_p),
                ("wgrad_ld", C.c_int64), ("kind", C.c_int32), ("m_out", C.c_int32), ("n_out", C.c_int32), ("d_aux", C.c_void_p)]


TG_DW_HH, TG_DW_HX, TG_DW_DH, TG_DW_HR, TG_DW_RH = 0, 1, 2, 3, 4


class SlabSum(C.Structure):
    """tg_slab_sum (include/trajopt_grpo_hip.h)."""
    _fields_ = [("d_slab", C.c_void_p), ("slab_stride", C.c_int64), ("n_slabs", C.c_int32), ("row_pitch", C.c_int32),
                ("d_grad", C.c_void_p), ("grad_ld", C.c_int64), ("m_out", C.c_int32), ("n_out", C.c_int32)]


class F32DwJob(C.Structure):
    """tg_f32_dw_job (include/trajopt_grpo_hip.h)."""
    _fields_ = [("d_p", C.c_void_p), ("d_q", C.c_void_p), ("d_wgrad", C.c_void_p), ("d_bgrad", C.c_void_p),
                ("wgrad_ld", C.c_int64), ("kind", C.c_int32), ("n_cols", C.c_int32), ("m_out", C.c_int32), ("n_out", C.c_int32),
                ("recompute", C.c_int32), ("in_pad", C.c_int32), ("in_dim", C.c_int32), ("act_dim", C.c_int32),
                ("d_w0", C.c_void_p), ("d_b0", C.c_void_p), ("d_wh", C.c_void_p), ("d_maskbits", C.c_void_p),
                ("d_a_top", C.c_void_p), ("d_whgrad", C.c_void_p), ("d_bhgrad", C.c_void_p), ("whgrad_ld", C.c_int64),
                ("d_dz0", C.c_void_p), ("d_w0grad", C.c_void_p), ("d_b0grad", C.c_void_p), ("w0grad_ld", C.c_int64)]


TG_F32DW_MM, TG_F32DW_HEAD = 0, 1


class AdamTensor(C.Structure):
    """tg_adam_tensor (include/trajopt_grpo_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("first", C.c_int64)]


class AdamRider(C.Structure):
    """tg_adam_rider (include/trajopt_grpo_hip.h)."""
    _fields_ = [("h_table", C.POINTER(AdamTensor)), ("n_tensors", C.c_int32), ("zero_grads", C.c_int32), ("total", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int64),
                ("d_segments", C.c_void_p), ("n_segments", C.c_int32), ("pad", C.c_int32), ("d_inv_start", C.c_void_p),
                ("d_inv_dst", C.c_void_p)]


class ChainLoss(C.Structure):
    """tg_chain_loss (include/trajopt_grpo_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("act_dim", C.c_int32), ("d_act", C.c_void_p), ("act_row_stride", C.c_int64),
                ("act_col_stride", C.c_int64), ("d_logp_old", C.c_void_p), ("d_adv", C.c_void_p), ("d_ret", C.c_void_p),
                ("norm_mean", C.c_float), ("norm_inv", C.c_float), ("var", C.c_float * 4), ("epsilon", C.c_float), ("surr_coef", C.c_float),
                ("critic_coef", C.c_float), ("kl_coef", C.c_float), ("d_dout8", C.c_void_p), ("d_head_slabs", C.c_void_p),
                ("d_work", C.c_void_p), ("d_bias_partial", C.c_void_p), ("d_logp_old_out", C.c_void_p), ("d_norm8", C.c_void_p)]

class CompactArgs(C.Structure):
    """tg_compact_args (include/trajopt_grpo_hip.h)."""
    _fields_ = [("d_mask", C.c_void_p), ("d_offsets", C.c_void_p), ("n", C.c_int64), ("T", C.c_int32), ("S", C.c_int32), ("A", C.c_int32),
                ("obs_dtype", C.c_int32), ("d_obs", C.c_void_p), ("obs_feat_stride", C.c_int64), ("d_act", C.c_void_p),
                ("d_xin", C.c_void_p), ("in_pad", C.c_int32), ("xin_bf16", C.c_int32), ("ones_col", C.c_int32), ("norm_mode", C.c_int32),
                ("d_act_rows", C.c_void_p), ("d_idx", C.c_void_p), ("d_src0", C.c_void_p), ("d_dst0", C.c_void_p),
                ("d_src1", C.c_void_p), ("d_dst1", C.c_void_p), ("d_moments", C.c_void_p), ("group_size", C.c_int64),
                ("rows_cap", C.c_int64)]


# name -> (restype, argtypes); every symbol include/trajopt_grpo_hip.h declares
_P, _I32, _I64, _U64, _F, _VP = C.POINTER, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
SIGNATURES = {
    "tg_last_error": (C.c_char_p, []),
    "tg_abi_version": (C.c_int, []),
    "tg_env_dims": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_int)]),
    "tg_env_default_params": (C.c_int, [C.c_int, C.c_int, _P(EnvParams)]),
    "tg_env_finalize_params": (C.c_int, [_P(EnvParams)]),
    "tg_env_reset": (C.c_int, [_P(EnvParams), C.c_int, _VP, _I64, _I64, _U64, _U64, _I64, _I64, _VP]),
    "tg_env_step": (C.c_int, [_P(EnvParams), C.c_int, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP]),
    "tg_quadrotor12_dynamics": (C.c_int, [_P(EnvParams), C.c_int, _VP, _I64, _VP, _I64, _VP, _I64, _I64, _VP]),
    "tg_rollout_begin": (C.c_int, [_P(Traj), C.c_int, C.c_int, _VP]),
    "tg_rollout_step": (C.c_int, [_P(EnvParams), _P(Traj), _I32, _VP, _I64, _P(_F), _VP, _I64, _VP]),
    "tg_rollout_forced": (C.c_int, [_P(EnvParams), _P(Traj), _I32, _I32, _VP]),
    "tg_rollout_finish": (C.c_int, [_P(Traj), _VP]),
    "tg_rollout_finish_stats_workspace": (C.c_int, []),
    "tg_rollout_finish_stats": (C.c_int, [_P(Traj), _VP, _VP, _VP, _VP]),
    "tg_fused_rollout": (C.c_int, [_P(EnvParams), _P(Traj), _VP, _VP, _I32, _I32, _P(_F), _VP, _I64, _I32, _I32, _VP]),
    "tg_fused_rollout_f32_supported": (C.c_int, [_I32, _I32]),
    "tg_fused_rollout_f32_block_envs": (C.c_int, [_I64, _I32]),
    "tg_fused_rollout_f32": (C.c_int, [_P(EnvParams), _P(Traj), _VP, _VP, _I32, _I32, _I32, _P(_F), _VP, _I64, _I32, _I32, _VP]),
    "tg_rng_advance": (C.c_int, [_VP, _VP]),
    "tg_colsum_finish": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP]),
    "tg_head_prep_blocks": (C.c_int, []),
    "tg_head_prep": (C.c_int, [_VP, _I64, _I32, _I32, _I32, _VP, _VP, _VP]),
    "tg_dw_finish": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP, _I32, _I32, _VP, _I64, _I32, _I32, _VP]),
    "tg_rtg_scan": (C.c_int, [_VP, _VP, _F, _VP, _I64, _I32, _VP]),
    "tg_gae_scan": (C.c_int, [_VP, _VP, _VP, _F, _F, _VP, _VP, _I64, _I32, _VP]),
    "tg_masked_moments": (C.c_int, [_VP, _VP, _I64, _I32, _I64, _VP, _VP, _VP]),
    "tg_group_normalize": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _I64, _I32, _I64, _VP]),
    "tg_gaussian_logp": (C.c_int, [_VP, _I64, _VP, _I64, _I64, _P(_F), C.c_int, _VP, _I64, _VP]),
    "tg_loss_work_blocks": (C.c_int, []),
    "tg_surrogate_loss": (C.c_int, [_P(LossArgs), _VP]),
    "tg_relu_bwd_bias_blocks": (C.c_int, []),
    "tg_relu_bwd_bias": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    "tg_head_bwd_relu_bias": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    "tg_dx_relu_bias_supported": (C.c_int, [_I32, _I32]),
    "tg_dx_relu_bias_blocks": (C.c_int, []),
    "tg_dx_pack_weights": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "tg_dx_relu_bias": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    "tg_mlp_backward_chain_blocks": (C.c_int, []),
    "tg_mlp_backward_chain": (C.c_int, [_VP, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _VP]),
    "tg_mlp_backward_chain_w0": (C.c_int, [_VP, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _VP, _I64,
                                           C.POINTER(C.c_int32), _VP]),
    "tg_mlp_weight_grad_workspace": (C.c_int64, [_I32]),
    "tg_mlp_weight_grad": (C.c_int, [_I32, C.POINTER(DwJob), _I32, _I64, _VP, _VP, _VP, _VP, _I64, _VP]),
    "tg_mlp_weight_grad_ex": (C.c_int, [_I32, C.POINTER(DwJob), _I32, _I64, _VP, _VP, _VP, _VP, _I64, C.POINTER(SlabSum), _I32, _VP, _I32, _VP, _VP]),
    "tg_mlp_forward_chain": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _I32,
                                       _VP]),
    "tg_mlp_forward_chain_blocks": (C.c_int, []),
    "tg_mlp_forward_chain_loss": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(ChainLoss), _VP]),
    "tg_mlp_f32_stream_floats": (C.c_int64, [_I32, _I32, _I32]),
    "tg_mlp_f32_blocks": (C.c_int, []),
    "tg_mlp_f32_forward": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I64, _VP, _VP]),
    "tg_mlp_f32_forward_backward": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP,
                                              C.POINTER(ChainLoss), _VP]),
    "tg_mlp_f32w_stream_floats": (C.c_int64, [_I32]),
    "tg_mlp_f32w_table_floats": (C.c_int64, []),
    "tg_mlp_f32w_blocks": (C.c_int, []),
    "tg_mlp_f32w_forward": (C.c_int, [_VP, _I32, _VP, _VP, _I32, _I64, _VP, _VP]),
    "tg_mlp_f32w_forward_backward": (C.c_int, [_VP, _I32, _VP, _VP, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(ChainLoss), _VP]),
    "tg_mlp_f32r_supported": (C.c_int, [_I32, _I32, _I32]),
    "tg_mlp_f32r_stream_floats": (C.c_int64, [_I32, _I32]),
    "tg_mlp_f32r_w0_floats": (C.c_int64, [_I32, _I32]),
    "tg_mlp_f32r_table_floats": (C.c_int64, [_I32]),
    "tg_mlp_f32r_grid": (C.c_int, [_I64]),
    "tg_mlp_f32r_forward": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I32, _I32, _I32, _I64, _VP, _VP]),
    "tg_mlp_f32r_forward_backward": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I32, _I32, _I64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP,
                                               C.POINTER(ChainLoss), _VP]),
    "tg_mlp_f32_weight_grad_workspace": (C.c_int64, [_I32]),
    "tg_mlp_f32_weight_grad": (C.c_int, [_I32, C.POINTER(F32DwJob), _I32, _I64, _VP, _I64, _VP, _I32, _VP, _VP]),
    "tg_mlp_f32_weight_grad_adam": (C.c_int, [_I32, C.POINTER(F32DwJob), _I32, _I64, _VP, _I64, _VP, _I32, _VP, C.POINTER(AdamRider), _VP]),
    "tg_adam_step": (C.c_int, [_VP, _I32, _I64, C.c_double, C.c_double, C.c_double, C.c_double, _I64, _I32, _VP]),
    "tg_gather_streams": (C.c_int, [_VP, _I32, _I64, _VP, _VP]),
    "tg_params_differ": (C.c_int, [_VP, _I32, _I64, _VP, _VP]),
    "tg_adam_step_push": (C.c_int, [_VP, _I32, _I64, C.c_double, C.c_double, C.c_double, C.c_double, _I64, _I32, _VP, _I32, _VP, _VP, _VP]),
    "tg_returns_moments_max_horizon": (C.c_int, []),
    "tg_returns_moments": (C.c_int, [_VP, _VP, _F, _VP, _I64, _I32, _I64, _VP, _VP, _VP]),
    "tg_learn_count_workspace": (C.c_int64, [_I64]),
    "tg_learn_count": (C.c_int, [_VP, _I64, _I64, _VP, _I64, _VP, _VP]),
    "tg_learn_compact": (C.c_int, [C.POINTER(CompactArgs), _VP]),
    "tg_scatter_rows": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP]),
    "tg_ppo_returns": (C.c_int, [_VP, _VP, _VP, _F, _F, C.c_int, _VP, _VP, _I64, _I32, _VP, _VP, _VP]),
    "tg_ppo_norm": (C.c_int, [_VP, C.c_double, C.c_double, _VP, _VP]),
    "tg_gather_rows2": (C.c_int, [_VP, _I64, _VP, _VP, _VP, _VP, _VP]),
    "tg_clock_probe_attach": (C.c_int, [_I32, _VP]),
    "tg_mfma_sustained_probe_blocks": (C.c_int, []),
    "tg_mfma_sustained_probe_flops": (C.c_double, [_I32, _I32]),
    "tg_mfma_sustained_probe": (C.c_int, [_I32, _I32, _VP, _VP, _VP, _VP]),
}

# tg_clock_probe_attach families, and the size of a probe buffer in uint64
(TG_PROBE_FWD_CHAIN, TG_PROBE_BWD_CHAIN, TG_PROBE_WEIGHT_GRAD, TG_PROBE_F32_CHAIN, TG_PROBE_F32_WEIGHT_GRAD, TG_PROBE_MFMA_LOOP,
 TG_PROBE_FWD_CHAIN_PLAIN) = range(7)
TG_CLOCK_PROBE_U64 = 4 + 2 * 4096

_lib = None


def load():
    """Load the HIP library (once).  Raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the GPU rollout path.")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    if lib.tg_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {lib.tg_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().tg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'trajopt_grpo_hip'} failed ({rc}): {msg}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeLibraryError("the HIP rollout path needs tensors on an MI355X (cuda) device; got a CPU tensor. "
                                     "There is no CPU fallback.")


def ptr(t):
    return 0 if t is None else t.data_ptr()


# Bumped by every launch that writes parameters through raw pointers (optim.FusedAdam.step): torch's own version counters do not
# see those writes, so anything that caches a function of the weights keys its freshness on (versions, this counter).
RAW_PARAM_WRITES = [0]
ALWAYS_REBUILD = os.environ.get("TG_ALWAYS_REBUILD", "0") == "1"    # 1: derived weight layouts never count as fresh (rounds 1-2 behaviour)
# Writes through `param.data` bypass torch's version counters, so at the ENTRY of every learn() / rollout the layouts are rebuilt
# whatever the keys say (one gather launch when the optimizer step is the fused one); between the updates of a learn(), where every
# write is the build's own, the keys decide.  1: trust the keys at the entries too (saves the two launches per iteration).
TRUST_KEYS = os.environ.get("TG_TRUST_VERSION_KEYS", "0") == "1"


def event_pair():
    """Two timing events for one launch (profiling runs: bench.py)."""
    return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def dtype_code(dt):
    if dt == torch.float32:
        return TG_F32
    if dt == torch.float64:
        return TG_F64
    raise ValueError(f"state dtype must be float32 or float64, got {dt}")


def default_params(env_id, max_steps):
    p = EnvParams()
    check(load().tg_env_default_params(env_id, int(max_steps), C.byref(p)), "tg_env_default_params")
    return p


def env_dims(env_id):
    s, a = C.c_int(), C.c_int()
    check(load().tg_env_dims(env_id, C.byref(s), C.byref(a)), "tg_env_dims")
    return s.value, a.value
