#!/usr/bin/env python3
"""One fused persistent rollout (tg_fused_rollout) of 65,536 QuadPole envs x 256 steps with nobody terminating
(bounds opened), bf16 20-256x5-4 actor.  Used under rocprofv3 (--kernel-trace --stats; --pmc FETCH_SIZE /
--pmc WRITE_SIZE; SQ_* counters) for the kernel's duration, HBM-side traffic and pipe utilisation.

    python3 tools/fused_rollout_probe.py [n_envs]        (TG_FUSED_NT=1|2 selects the tiles-per-wave variant)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
    T = 256
    env = tg.QuadPole(max_steps=T)
    env.spatial_bounds = tuple((-1e9, 1e9) for _ in range(3))
    eng = tg.DeviceRollout(env, pol, n // 256, 256, seed=1, compute_dtype=torch.bfloat16, fused=True)
    eng.run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        eng.step_events = []
        eng.run()
        torch.cuda.synchronize()
        _, a, b = eng.step_events[0]
        best = min(best, a.elapsed_time(b))
    steps = eng.traj.env_steps()
    n_par = sum(p.numel() for p in pol.actor.parameters())
    print(f"fused rollout n={n} T={T}: {best:.2f} ms kernel, {best * 1e3 / T:.1f} us/step, {steps / best / 1e3:.1f} M env-steps/s, "
          f"{2 * n_par * steps / best / 1e9:.1f} TFLOP/s actor")


if __name__ == "__main__":
    main()
