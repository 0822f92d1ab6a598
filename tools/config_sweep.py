#!/usr/bin/env python3
"""One-GPU throughput of every BASELINE.json config shape (per-GPU shard for the 8-GPU ones).

    python3 tools/config_sweep.py > profiles/r01_config_sweep.json

Each entry: rollout + learn iterations on the device path, env-steps/s over valid (mask = 1) env-steps.
C4 / C5 are sized as ONE rank's shard of the 8-GPU configuration (the rollout needs no collective; the only
exchange is the flat gradient all-reduce, measured by the driver's multi-GPU bench)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402

dev = torch.device("cuda", 0)


def run(name, env_fn, policy, algo_fn, G, E, restart, cdt, iters=3, **mgr_kw):
    mgr = tg.RolloutManager(env_fn, policy, restart=restart, num_workers=G, num_episodes_per_worker=E, seed=1,
                            compute_dtype=cdt, **mgr_kw)
    buf = tg.Rollout_Buffer(mgr)
    algo = algo_fn(policy)
    buf.sample(); algo.learn(buf)                       # warm-up
    torch.cuda.synchronize()
    steps, t_roll = 0, 0.0
    t0 = time.perf_counter()
    for _ in range(iters):
        r0 = time.perf_counter()
        buf.sample()
        t_roll += time.perf_counter() - r0
        steps += buf.device_traj.env_steps()
        algo.learn(buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"config": name, "envs": mgr.engine.n, "horizon": mgr.engine.T, "fused_rollout": mgr.engine.fused,
            "env_steps_per_iter": steps / iters, "ms_per_iter": 1e3 * dt / iters, "rollout_ms": 1e3 * t_roll / iters,
            "env_steps_per_s": steps / dt, "rollout_only_env_steps_per_s": steps / t_roll,
            "avg_return": float(buf.avg_reward[-1])}


def grpo(lr=3e-4, cdt=None, gamma=0.5):
    return lambda pol: tg.GRPO(epsilon=0.15, beta=0.5, gamma=gamma, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=lr),
                               updates_per_iter=1, autocast_dtype=cdt)


def ppo(updates, lr=3e-4, cdt=None, gamma=0.999):
    return lambda pol: tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=lr), ref_model=None,
                              updates_per_iter=updates, gamma=gamma, batch_size=None, autocast_dtype=cdt)


def main():
    out = []
    bf = torch.bfloat16
    torch.manual_seed(0)
    # C1: CartPole GRPO, 4 envs x 128 steps (plumbing case of the reference script)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    out.append(run("C1 CartPole GRPO 4 envs x 128, fp32", lambda: tg.CartPole(max_steps=128), pol, grpo(), 2, 2, True, None))
    # C2: CartPole GRPO, 4,096 envs, fp32, 2-layer MLP
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    out.append(run("C2 CartPole GRPO 4096 envs x 500, fp32 2-layer MLP", lambda: tg.CartPole(max_steps=500), pol, grpo(), 64, 64,
                   True, None))
    # C3: the bench workload
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
    out.append(run("C3 QuadPole PPO 65536 envs x 256, bf16, 32 updates", lambda: tg.QuadPole(max_steps=256), pol, ppo(32, cdt=bf),
                   256, 256, False, bf, iters=2))
    # C4: QuadPole GRPO, 262,144 envs over 8 GPUs -> 32,768 envs (128 groups x 256) per GPU, restart groups
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
    out.append(run("C4 QuadPole GRPO shard: 32768 envs x 256 (128 restart groups), bf16", lambda: tg.QuadPole(max_steps=256), pol,
                   grpo(cdt=bf, gamma=0.99), 128, 256, True, bf))
    # C5: swarm, 32,768 envs x 8 agents over 8 GPUs -> 4,096 envs x 8 bodies per GPU
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
    out.append(run("C5 QuadPoleSwarm GRPO shard: 4096 envs x 8 agents x 256 (64 groups x 64), bf16",
                   lambda: tg.QuadPoleSwarm(n_agents=8, max_steps=256), pol, grpo(cdt=bf, gamma=0.99), 64, 64, True, bf))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
