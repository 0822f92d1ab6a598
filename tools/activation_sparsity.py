#!/usr/bin/env python3
"""How sparse are the stored activations?  Per hidden layer, the mean / maximum per-row fraction of non-zero (post-ReLU)
activations of the actor and the critic on the valid rows of C3-shaped rollouts, over a few PPO iterations.

    python3 tools/activation_sparsity.py
"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
bf = torch.bfloat16
pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
mgr = tg.RolloutManager(lambda: tg.QuadPole(max_steps=256), pol, num_workers=64, num_episodes_per_worker=256, seed=1234, compute_dtype=bf)
buf = tg.Rollout_Buffer(mgr)
algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None, updates_per_iter=32,
              c1=0.5, kl_coeff=0.5, gamma=0.999, lam=0.95, entropy=0.01, batch_size=None, autocast_dtype=bf)
for it in range(6):
    buf.sample()
    algo.learn(buf)
    tr = buf.device_traj
    idx = tr.mask.reshape(-1).nonzero().squeeze(1)[:500000]
    X = tr.obs_rows().index_select(0, idx).float()
    out = []
    for name, net in (("actor", pol.actor), ("critic", pol.critic)):
        h = X
        fr = []
        for m in net.network:
            h = m(h)
            if isinstance(m, torch.nn.ReLU):
                nz = (h > 0).float().mean(1)              # per-row fraction of non-zeros
                fr.append((float(nz.mean()), float(nz.max())))
        out.append(name + " " + " ".join(f"{a:.2f}/{b:.2f}" for a, b in fr))
    print(f"iter {it} avg return {float(buf.avg_reward[-1]):8.1f}  non-zero fraction mean/max per layer: " + " | ".join(out), flush=True)
