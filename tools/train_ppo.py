#!/usr/bin/env python3
"""End-to-end learning check: PPO with the reference factories' hyper-parameters but 4,096 parallel episodes per
epoch instead of 50-80.

    python3 tools/train_ppo.py [epochs] [CartPole|QuadPole2D|QuadPole] [bf16|fp32]

CartPole / QuadPole2D (pipelines/cartpole_pipeline_ppo.py, quadpole2d_pipeline_ppo.py): 128x3 actor-critic, cov 0.5,
eps 0.2, gamma 0.99, 24 full-batch updates, Adam 2e-4 (published curves: -37 -> ~800 and -70 -> ~1047).
QuadPole (quadpole_pipeline_ppo.py): 256x5, cov 0.3, gamma 0.999, 32 updates, Adam 3e-4, bf16 policy compute."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    name = sys.argv[2] if len(sys.argv) > 2 else "CartPole"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    if name == "QuadPole":
        S, A, hidden, cov, lr, upd, gamma, cdt = 20, 4, (256,) * 5, 0.3, 3e-4, 32, 0.999, torch.bfloat16
        if len(sys.argv) > 3 and sys.argv[3] == "fp32":          # the reference's own precision: the H = 256 fp32 chain learner
            cdt = None
    else:
        S, A = (5, 1) if name == "CartPole" else (10, 2)
        hidden, cov, lr, upd, gamma, cdt = (128, 128, 128), 0.5, 2e-4, 24, 0.99, None
        if len(sys.argv) > 3 and sys.argv[3] == "bf16":          # bf16 policy compute: fused bf16 rollout + chain kernels at H = 128
            cdt = torch.bfloat16
    pol = tg.GaussianActorCritic_NeuralNetwork(S, A, hidden, cov=cov, device=dev)
    mgr = tg.RolloutManager(lambda: tg.environments.ENV_CLASSES[name](), pol, num_workers=64, num_episodes_per_worker=64,
                            seed=0, compute_dtype=cdt)
    buf = tg.Rollout_Buffer(mgr)
    algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=lr), ref_model=None,
                  updates_per_iter=upd, c1=0.5, kl_coeff=0.5, gamma=gamma, lam=0.95, entropy=0.01, batch_size=None,
                  autocast_dtype=cdt)
    t0 = time.time()
    for ep in range(epochs):
        buf.sample()
        algo.learn(buf)
        if ep % 10 == 0 or ep == epochs - 1:
            print(f"epoch {ep:4d}  avg return {float(buf.avg_reward[-1]):9.2f}  mean len {float(buf.device_traj.len.float().mean()):6.1f}  "
                  f"elapsed {time.time() - t0:6.1f}s", flush=True)
    print("first -> last:", float(buf.avg_reward[0]), "->", float(buf.avg_reward[-1]), " max", float(max(buf.avg_reward)))


if __name__ == "__main__":
    main()
