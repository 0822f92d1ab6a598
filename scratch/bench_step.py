"""Isolated timing of tg_rollout_step (sample mode with a fixed mean buffer, and forced mode)."""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, '.')
import trajopt_grpo_amd as tg
N_ = tg._native
dev = torch.device('cuda', 0)
name = os.environ.get('ENV', 'QuadPole')
cls = tg.environments.ENV_CLASSES[name]
T = 64
def run(n, sample=True, alive_frac=1.0, reps=3):
    env = cls(max_steps=T)
    if hasattr(env, 'spatial_bounds'):
        env.spatial_bounds = tuple((-1e9, 1e9) for _ in env.spatial_bounds)   # nobody terminates
    pol = tg.GaussianActor_NeuralNetwork(env.obs_dim, env.act_dim, (8,), cov=0.3, device=dev)
    eng = tg.DeviceRollout(env, pol, n // 256, 256, seed=1)
    lib = N_.load(); tr = eng.traj.native(); st = N_.stream_ptr(dev); p = C.byref(eng.params)
    mean = torch.zeros(n, 8, device=dev)
    best = 1e9
    for rep in range(reps):
        eng._stream_host = 0; eng._seed_host = 1
        eng._enqueue_prepare(None)
        if alive_frac < 1.0:
            dead = torch.rand(n, device=dev) > alive_frac
            eng.traj.len[dead] = 1
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(T - 1):
            if sample:
                lib.tg_rollout_step(p, C.byref(tr), t, mean.data_ptr(), 8, eng._sigma, eng.rng.data_ptr(), 0, st)
            else:
                lib.tg_rollout_step(p, C.byref(tr), t, None, 0, None, None, 0, st)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (T - 1))
    alive = int((eng.traj.len == 0).sum()) if alive_frac < 1 else n
    return best, alive
bytes_per = {'CartPole': 57, 'QuadPole2D': 101, 'QuadPole': 189}[name]
for n in (65536, 262144, 1 << 20, 1 << 22):
    for sample in (True, False):
        us, alive = run(n, sample)
        print(f"{name} n={n:8d} sample={sample} block={os.environ.get('TG_STEP_BLOCK','auto')}: {us:8.2f} us/launch  {bytes_per*alive/us/1e3:8.1f} GB/s algorithmic")
for frac in (0.5, 0.1):
    us, alive = run(65536, True, frac)
    print(f"{name} n=65536 alive={alive}: {us:8.2f} us/launch  {bytes_per*alive/us/1e3:8.1f} GB/s")
