#!/usr/bin/env python3
"""Rollout-only timing of the three rollout paths (fused persistent kernel, hipGraph replay of the per-step
launches, eager per-step launches) at 16k / 65k / 262k QuadPole envs x 256 steps, bf16 20-256x5-4 actor."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import trajopt_grpo_amd as tg
dev=torch.device('cuda',0)
torch.manual_seed(0)
pol = tg.GaussianActorCritic_NeuralNetwork(20,4,(256,)*5,cov=0.3,device=dev)
T=256
def timeit(eng, reps=3):
    eng.run(); torch.cuda.synchronize()
    best=1e9; steps=0
    for _ in range(reps):
        t=time.perf_counter(); tr=eng.run(); torch.cuda.synchronize(); dt=time.perf_counter()-t
        best=min(best,dt); steps=tr.env_steps()
    return best*1e3, steps
for n in (16384, 65536, 262144):
    for label, kw in (('fused', dict(fused=True)), ('graph', dict(fused=False, use_graph=True)), ('eager', dict(fused=False))):
        eng = tg.DeviceRollout(tg.QuadPole(max_steps=T), pol, n//256, 256, seed=1, compute_dtype=torch.bfloat16, **kw)
        ms, steps = timeit(eng)
        print(f"n={n:7d} {label:6s}: {ms:8.2f} ms/rollout  env-steps {steps}  {steps/ms/1e3:8.1f} M env-steps/s")
        del eng
# all-alive variant (bounds opened): pure throughput of the fused kernel
env = tg.QuadPole(max_steps=T); env.spatial_bounds = tuple((-1e9,1e9) for _ in range(3))
eng = tg.DeviceRollout(env, pol, 256, 256, seed=1, compute_dtype=torch.bfloat16, fused=True)
ms, steps = timeit(eng)
print(f"n=65536 fused all-alive: {ms:.2f} ms  {steps/ms/1e3:.1f} M env-steps/s  {539144*steps/ms/1e9:.1f} TFLOP/s actor")
