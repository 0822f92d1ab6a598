#!/usr/bin/env python3
"""Is the update launch-bound?  Times when the host has ENQUEUED the 32 PPO updates of a C3-shaped learn() against when the
GPU has finished them, then cProfiles the host side (where it blocks on queue back-pressure).

    python3 tools/host_ahead_probe.py
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
bf = torch.bfloat16
pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
mgr = tg.RolloutManager(lambda: tg.QuadPole(max_steps=256), pol, num_workers=256, num_episodes_per_worker=256, seed=1234, compute_dtype=bf)
buf = tg.Rollout_Buffer(mgr)
algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None, updates_per_iter=32,
              c1=0.5, kl_coeff=0.5, gamma=0.999, lam=0.95, entropy=0.01, batch_size=None, autocast_dtype=bf)
for it in range(4):
    buf.sample()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    algo.learn(buf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iter {it}: learn() returned after {1e3*(t1-t0):.1f} ms, GPU done after {1e3*(t2-t0):.1f} ms, rows {buf.device_traj.env_steps()}", flush=True)

# host-side enqueue time of the 32 updates (no sync inside _step)
orig = algo._step
stamps = []
def timed_step(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    stamps.append((t, time.perf_counter()))
    return r
algo._step = timed_step
buf.sample(); torch.cuda.synchronize()
t0 = time.perf_counter()
algo.learn(buf)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"first _step entered at {1e3*(stamps[0][0]-t0):.1f} ms; last _step enqueued at {1e3*(stamps[-1][1]-t0):.1f} ms; GPU done {1e3*(t2-t0):.1f} ms; "
      f"host per _step {1e3*sum(b-a for a,b in stamps)/len(stamps):.2f} ms; rows {buf.device_traj.env_steps()}")

import cProfile, pstats, io
algo._step = orig
buf.sample(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
algo.learn(buf)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
