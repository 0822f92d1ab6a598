#!/usr/bin/env python3
"""End-to-end learning check: CartPole swing-up PPO with the reference factory's hyper-parameters
(pipelines/cartpole_pipeline_ppo.py: 5-128x3-1 actor-critic, cov 0.5, eps 0.2, gamma 0.99, 24 full-batch
updates, Adam 2e-4) but 4,096 parallel episodes per epoch instead of 80.  Prints avg episode return per epoch
(the reference's published curve goes from -37 to ~800 in ~800 epochs of 80 episodes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = tg.GaussianActorCritic_NeuralNetwork(5, 1, (128, 128, 128), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(), pol, num_workers=64, num_episodes_per_worker=64, seed=0)
    buf = tg.Rollout_Buffer(mgr)
    algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=2e-4), ref_model=None,
                  updates_per_iter=24, c1=0.5, kl_coeff=0.5, gamma=0.99, lam=0.95, entropy=0.01, batch_size=None)
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
