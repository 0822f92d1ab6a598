#!/usr/bin/env python3
"""Rollout time of the fp32 fused kernel (tg_fused_rollout_f32) against the per-step path replayed as one hipGraph,
for fp32 policies of the reference's sizes.

    python3 tools/fused_f32_probe.py [--iters 5]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402

CASES = [("CartPole", (128, 128), 4096, 500), ("CartPole", (128, 128, 128), 4096, 500), ("CartPole", (128, 128, 128, 128), 4096, 500),
         ("QuadPole2D", (128, 128, 128), 4096, 500), ("QuadPole", (128, 128, 128), 4096, 256), ("CartPole", (64, 64), 1024, 500),
         ("CartPole", (128, 128, 128), 16384, 500), ("CartPole", (128, 128, 128), 65536, 500), ("QuadPole", (128, 128, 128), 65536, 256)]


def timed(eng, iters):
    eng.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    for _ in range(iters):
        steps += eng.run().env_steps()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters, steps / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = []
    for name, hidden, n, T in CASES:
        S, A = tg.environments.ENV_CLASSES[name](max_steps=T).obs_dim, tg.environments.ENV_CLASSES[name](max_steps=T).act_dim
        torch.manual_seed(0)
        pol = tg.GaussianActor_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
        mk = lambda: tg.environments.ENV_CLASSES[name](max_steps=T)
        row = {"env": name, "hidden": list(hidden), "envs": n, "horizon": T}
        for tag, kw in (("fused_f32", dict(fused=True)), ("graph", dict(fused=False, use_graph=True))):
            eng = tg.DeviceRollout(mk(), pol, n // 64, 64, seed=5, **kw)
            ms, steps = timed(eng, args.iters)
            row[tag + "_ms"] = ms
            row[tag + "_env_steps_per_s"] = steps / ms * 1e3
            row["env_steps"] = steps
            del eng
        row["speedup"] = row["graph_ms"] / row["fused_f32_ms"]
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


if __name__ == "__main__":
    main()
