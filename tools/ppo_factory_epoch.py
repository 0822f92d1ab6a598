#!/usr/bin/env python3
"""One epoch of CartPole PPO at the reference factory's own size (pipelines/cartpole_pipeline_ppo.py:54-79: 10 workers x 8 episodes,
500-step horizon, 128x3 actor-critic fp32, 24 full-batch updates, Adam 2e-4): rollout + learn(), wall clock per epoch (host
synchronised at the end of every epoch, as a training loop that logs its reward is).  At this size the learner's prologue and
its host round trips ARE the step.

    python tools/ppo_factory_epoch.py [--tree DIR] [--epochs 60] [--gae]

--tree: the directory that holds the `trajopt-grpo_amd` package to time (default: this repository); tools/final_r05.sh points it
at an unpacked round-4 tree for the before / after pair."""
import argparse
import importlib.util
import json
import os
import sys
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--tree", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap.add_argument("--epochs", type=int, default=60)
ap.add_argument("--gae", action="store_true")
ap.add_argument("--envs", type=int, nargs=2, default=[10, 8], metavar=("WORKERS", "EPISODES"))
a = ap.parse_args()

spec = importlib.util.spec_from_file_location("trajopt_grpo_amd", os.path.join(a.tree, "trajopt-grpo_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(a.tree, "trajopt-grpo_amd")])
tg = importlib.util.module_from_spec(spec)
sys.modules["trajopt_grpo_amd"] = tg
spec.loader.exec_module(tg)

dev = torch.device("cuda", 0)
torch.manual_seed(0)
pol = tg.GaussianActorCritic_NeuralNetwork(5, 1, (128, 128, 128), cov=0.5, device=dev)
mgr = tg.RolloutManager(lambda: tg.CartPole(), pol, num_workers=a.envs[0], num_episodes_per_worker=a.envs[1], seed=0)
buf = tg.Rollout_Buffer(mgr)
algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=2e-4), ref_model=None, updates_per_iter=24,
              c1=0.5, kl_coeff=0.5, gamma=0.99, lam=0.95, entropy=0.01, batch_size=None, monte_carlo=not a.gae)
times, launches = [], None
for ep in range(a.epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf.sample()
    algo.learn(buf)
    r = float(buf.avg_reward[-1])
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
# launches of one more epoch, counted by the profiler's own kernel table
try:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        buf.sample()
        algo.learn(buf)
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    launches = len(ev)
    native = sum(1 for e in ev if e.name.startswith("tg::") or "tg::" in e.name)
except Exception as exc:                                    # (no profiler on this box: the timing stands on its own)
    launches, native = None, None
times = sorted(times[5:])
print(json.dumps({"tree": a.tree, "envs": a.envs[0] * a.envs[1], "horizon": 500, "updates": 24, "advantages": "gae" if a.gae else "monte carlo",
                  "epochs_timed": len(times), "ms_per_epoch_median": 1e3 * times[len(times) // 2], "ms_per_epoch_min": 1e3 * times[0],
                  "device_launches_per_epoch": launches, "of_which_this_library": native, "valid_rows_last": int(buf.device_traj.env_steps()),
                  "avg_return_last": r}))
