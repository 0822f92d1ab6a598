#!/usr/bin/env python3
"""C2's rollout (4,096 CartPole envs x 500 steps, fp32 5-128-128-1) on the fp32 fused kernel with 32 and with 16 envs per workgroup:
launch time (HIP events around run()), longest episode, time per step of the longest episode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
for be in (32, 16, 32, 16):
    eng = tg.DeviceRollout(tg.CartPole(max_steps=500), pol, 64, 64, seed=3)
    eng.f32_block_envs = be
    for _ in range(3):
        tr = eng.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts, mx = [], []
    for _ in range(20):
        e0.record(); tr = eng.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3); mx.append(int(tr.len.max()))
    t, m = sum(ts) / len(ts), sum(mx) / len(mx)
    print(f"block_envs {be}: run() {t:.1f} us, longest episode {m:.0f} steps (mean {float(tr.len.float().mean()):.1f}), {1e3 * t / m:.0f} ns per step of the longest episode")
