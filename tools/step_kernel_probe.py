#!/usr/bin/env python3
"""Launch the dynamics kernel (tg_rollout_step) in isolation: 48 consecutive time steps of a QuadPole
rollout with nobody terminating (bounds opened), sampling from a fixed mean buffer.  Used under
rocprofv3 (--kernel-trace --stats, then --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes) to get
the kernel's duration and HBM-side traffic per launch at a given env count.

    python3 tools/step_kernel_probe.py [n_envs] [env_name]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402

N_ = tg._native
BYTES = {"CartPole": 57, "QuadPole2D": 101, "QuadPole": 189}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    name = sys.argv[2] if len(sys.argv) > 2 else "QuadPole"
    T = 49
    dev = torch.device("cuda", 0)
    env = tg.environments.ENV_CLASSES[name](max_steps=T)
    if hasattr(env, "spatial_bounds"):
        env.spatial_bounds = tuple((-1e9, 1e9) for _ in env.spatial_bounds)
    pol = tg.GaussianActor_NeuralNetwork(env.obs_dim, env.act_dim, (8,), cov=0.3, device=dev)
    eng = tg.DeviceRollout(env, pol, n // 256, 256, seed=1)
    eng._seed_host, eng._stream_host = 1, 0
    lib, tr, st, p = N_.load(), eng.traj.native(), N_.stream_ptr(dev), C.byref(eng.params)
    mean = torch.zeros(n, 4, device=dev)                 # 16-B mean rows (tg_mlp_forward_chain's output for <= 4 outputs)
    eng._enqueue_prepare(None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for t in range(T - 1):
        N_.check(lib.tg_rollout_step(p, C.byref(tr), t, mean.data_ptr(), 4, eng._sigma, eng.rng.data_ptr(), 0, st))
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (T - 1)
    assert int((eng.traj.len == 0).sum()) == n
    print(f"{name} n={n}: {us:.2f} us per launch back-to-back, {BYTES[name] * n / us / 1e3:.1f} GB/s algorithmic "
          f"({BYTES[name]} B/env-step)")


if __name__ == "__main__":
    main()
