#!/usr/bin/env python3
"""Benchmark of the hot path: GPU-resident rollout + PPO / GRPO update on QuadPole ("quadrotor_env.py").

    python bench.py --gpus N --steps K --warmup W [--config c3|c4|c5] [--scaling weak|strong]

One "step" = one pass of the hot path over one batch: a rollout of every env of this rank's shard (one persistent kernel:
actor MLP on the matrix cores + sampling + dynamics + recording) followed by `learn` on that buffer.

  --config c3 (default)  BASELINE.json configs[2]: QuadPole PPO, 65,536 envs x 256 steps, the reference factory's
                         hyper-parameters (pipelines/quadpole_pipeline_ppo.py: 20-256x5-{4,1} actor-critic, cov 0.3,
                         gamma 0.999, 32 full-batch updates, Adam 3e-4), bf16 policy.  The metric is quoted on this one.
  --config c4            configs[3]: QuadPole GRPO, restart groups (E episodes share the group's initial state),
                         32,768 envs per GPU (128 groups x 256; 262,144 = 1,024 groups on 8 GPUs), actor 20-256x5-4.
  --config c5            configs[4]: QuadPoleSwarm (8 bodies per env), GRPO with the group statistics across the bodies,
                         4,096 envs x 8 agents per GPU (64 groups x 64 episodes; 32,768 envs on 8 GPUs).
  --scaling weak         (default) the per-GPU shard above on every rank;  strong: the config's TOTAL env count (c3: 65,536;
                         c4: 262,144; c5: 32,768) divided over the ranks.

Metric: env-steps/s, an env-step being one valid (mask == 1) Env.step.  Gradients are all-reduced once per optimizer step
(RCCL); the rollout needs no collective.

`--gpus N` with N > 1 from a bare invocation (WORLD_SIZE unset): this process -- which never touches the GPU -- starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and exits with its code; rank 0 of the child
prints the JSON line.  Under torchrun (WORLD_SIZE set) the process is a rank.

Besides the contract fields the JSON line carries
  roofline          the hand-written kernel FAMILY with the most GPU time in the step (weight gradients, backward-data chain
                    or forward chain), algorithmic bytes of every launch of the timed steps / its duration by HIP events on
                    the launch stream; `kernels` has the same figures for all three families.
  rollout_kernel    the fused rollout kernel against the MFMA roofline (2 x actor parameters flop per valid env-step).
  dynamics_kernel   the stand-alone dynamics kernel (tg_rollout_step) at this env count, HBM roofline, timed after the run.
  fixed work        `update_ns_per_valid_row` (learn time per valid row: does not depend on how long the policy survives) and
                    `fixed_work_ms_per_step` (one step with the bounds opened: every env runs the full horizon, 16.8 M rows).
  roofline.other_configs   (plain `--config c3` runs) BASELINE.json configs[1], [3], [4] -- C2, and C4's / C5's per-GPU shards -- run in the
                    same process after the headline's timed region: value, ms_per_step, steps, env_steps_per_step, dtype, and the
                    dominant kernel's roofline fraction of each (launches untimed in their timed regions; extra steps carry the events).
  cpu_baseline      the CPU port of the reference path (oracle/) timed on this box's host cores over bounded samples:
                    all cores (<= 16), exactly 8 cores (sched_setaffinity, comparable with BASELINE.md section 2), and C1 exactly.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HIDDEN = (256, 256, 256, 256, 256)
ALGO_BYTES = {"CartPole": 57, "QuadPole2D": 101, "QuadPole": 189}    # SURVEY 8(d), compact variant
HBM_PEAK_GBS = 8000.0                                                 # MI355X_MICROARCH.md: 8.0 TB/s spec
# PMC traffic per row of the three learner kernels at 2^22 rows (profiles/r04_*_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE)
PMC_JSON = {"dw": "r04_dw_probe_pmc.json", "bwd": "r04_bwd_chain_probe_pmc.json", "fwd": "r04_fwd_chain_probe_pmc.json"}
CONFIGS = {
    #        env           algo    groups/GPU  episodes  agents  restart  total envs (strong scaling)
    "c2": ("CartPole",      "grpo", 64,         64,       1,      False,   4096),
    "c3": ("QuadPole",      "ppo",  256,        256,      1,      False,   65536),
    "c4": ("QuadPole",      "grpo", 128,        256,      1,      True,    262144),
    "c5": ("QuadPoleSwarm", "grpo", 64,         64,       8,      True,    32768),
}


# ------------------------------------------------------------------------------------------------
# CPU baseline (runs BEFORE the GPU is initialised: it forks worker processes)
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    import numpy as np
    import torch
    from oracle import learner as L
    wid, sd, T, episodes = args
    torch.set_num_threads(1)
    torch.manual_seed(1000 + wid)
    # as the reference's worker runs it: a MultivariateNormal built per env step, under autograd (actor_critic.py:279-284,
    # rollout_worker.py:55) -- the closed-form sampling under no_grad that this leg used through round 2 ran 3.6 x the genuine
    # reference's rollout rate
    pol = L.OraclePolicy(20, 4, HIDDEN, cov=0.3, critic=True, per_step_distribution=True)
    pol.load_state_dict(sd)
    env = L.OracleEnv("QuadPole", max_steps=T, rng=np.random.default_rng(1000 + wid))
    return L.run_episodes(env, pol, episodes, restart=False)


def _cpu_leg(T, updates, workers, episodes):
    """Reduced C3: `workers` forked rollout processes (1 thread each, like OMP_NUM_THREADS=1 in BASELINE.md section 2), then
    PPO.learn on the CPU with `workers` threads."""
    import multiprocessing as mp
    import torch
    from oracle import learner as L
    torch.manual_seed(0)
    pol = L.OraclePolicy(20, 4, HIDDEN, cov=0.3, critic=True)
    sd = pol.state_dict()
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        outs = pool.map(_cpu_worker, [(w, sd, T, episodes) for w in range(workers)])
    t_roll = time.perf_counter() - t0
    obs, act, rew, ln, mask = (torch.stack([o[i] for o in outs]) for i in range(5))
    steps = int(mask.sum())
    torch.set_num_threads(workers)
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    t1 = time.perf_counter()
    L.ppo_learn(pol, opt, obs, act, rew, mask, epsilon=0.2, gamma=0.999, lam=0.95, c1=0.5, kl_coeff=0.5,
                entropy_coeff=0.01, updates_per_iter=updates)
    t_learn = time.perf_counter() - t1
    return {"value": steps / (t_roll + t_learn), "unit": "env-steps/s", "cores": workers, "kind": "port",
            "sample": f"QuadPole T={T}, {workers} forked workers x {episodes} episodes ({steps} env-steps), "
                      f"256x5 actor-critic, PPO {updates} full-batch updates; rollout {t_roll:.1f}s "
                      f"({steps / t_roll:.0f} env-steps/s) + learn {t_learn:.1f}s",
            "rollout_value": steps / t_roll}


def _in_child(q, fn, cores, *a):
    if cores is not None:
        os.sched_setaffinity(0, cores)
    q.put(fn(*a))


def _run_in_forked_child(fn, cores, *a):
    """A leg that runs multi-threaded torch code must not share a process with a later fork (a forked child of a
    process whose OpenMP pool is up deadlocks in its first parallel region): every secondary leg gets a child of its
    own, forked while this process is still single-threaded."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_in_child, args=(q, fn, cores) + a)
    p.start()
    out = q.get()
    p.join()
    return out


def _cpu_leg_c1():
    """BASELINE.json configs[0] exactly: CartPole GRPO, 4 envs (2 workers x 2 episodes), 128-step horizon, the GRPO factory's
    hyper-parameters (pipelines/cartpole_pipeline_grpo.py:54-76: 5-128x4-1 actor, cov 0.5, eps 0.15, gamma 0.5, 1 update),
    in-process like the reference's `use_multiprocessing=False` path; 10 iterations."""
    import numpy as np
    import torch
    from oracle import learner as L
    torch.set_num_threads(1)
    torch.manual_seed(0)
    pol = L.OraclePolicy(5, 1, (128,) * 4, cov=0.5, per_step_distribution=True)
    old = L.OraclePolicy(5, 1, (128,) * 4, cov=0.5)
    old.load_state_dict(pol.state_dict())
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    mk = lambda: L.OracleEnv("CartPole", max_steps=128, rng=np.random.default_rng(0))
    steps, t0 = 0, time.perf_counter()
    for _ in range(10):
        obs, act, rew, ln, mask = L.rollout(mk, pol, 2, 2, restart=False)
        L.grpo_learn(pol, old, opt, obs, act, rew, mask, epsilon=0.15, gamma=0.5, updates_per_iter=1)
        steps += int(mask.sum())
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"C1 exactly: CartPole GRPO, 2 workers x 2 episodes x 128 steps in-process, 10 iterations ({steps} env-steps)"}


def cpu_baseline(T, updates):
    cores = sorted(os.sched_getaffinity(0))
    legs = {}
    if len(cores) >= 8:
        legs["pinned_8_cores"] = _run_in_forked_child(_cpu_leg, set(cores[:8]), T, updates, 8, 16)
    legs["c1_exact"] = _run_in_forked_child(_cpu_leg_c1, None)
    main_leg = _cpu_leg(T, updates, max(1, min(len(cores), 16)), 32)      # last: it brings up this process's thread pool
    main_leg["legs"] = legs
    return main_leg


def dynamics_kernel_probe(tg, dev, n, launches=64):
    """The stand-alone dynamics kernel (tg_rollout_step, sampling mode) at `n` envs with nobody terminating:
    `launches` consecutive time steps bracketed by one HIP event pair on the launch stream.  HBM roofline:
    SURVEY 8(d) algorithmic bytes (189 B / QuadPole env-step) / time per launch (launch-to-launch, so it
    includes the ~1.5 us kernel boundary)."""
    import ctypes as C
    import torch
    N_ = tg._native
    env = tg.QuadPole(max_steps=launches + 1)
    env.spatial_bounds = tuple((-1e9, 1e9) for _ in env.spatial_bounds)
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (8,), cov=0.3, device=dev)
    eng = tg.DeviceRollout(env, pol, n // 256, 256, seed=1)
    eng._seed_host, eng._stream_host = 1, 0
    lib, tr, st, p = N_.load(), eng.traj.native(), N_.stream_ptr(dev), C.byref(eng.params)
    mean = torch.zeros(n, 4, device=dev)                 # 16-B mean rows, as tg_mlp_forward_chain writes them for <= 4 outputs
    best = None
    for _ in range(3):
        eng._enqueue_prepare(None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(launches):
            N_.check(lib.tg_rollout_step(p, C.byref(tr), t, mean.data_ptr(), 4, eng._sigma, eng.rng.data_ptr(), 0, st))
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / launches
        best = us if best is None else min(best, us)
    gbs = ALGO_BYTES["QuadPole"] * n / best / 1e3
    # yardstick for a launch this small: a bare device copy that moves the same number of bytes (half read, half
    # written), launched back to back the same way
    half = ALGO_BYTES["QuadPole"] * n // 8
    src, dst = torch.zeros(half, device=dev), torch.empty(half, device=dev)
    copy_us = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(launches):
            dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / launches
        copy_us = us if copy_us is None else min(copy_us, us)
    del eng, mean, src, dst
    torch.cuda.empty_cache()
    pmc = {65536: ("r03_step_kernel_65536_pmc.json", None), 4194304: ("r03_step_kernel_4194304_pmc.json", None)}.get(n)
    traffic = None
    if pmc:
        try:
            with open(os.path.join(REPO, "profiles", pmc[0])) as f:
                traffic = json.load(f)["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    return {"kernel": "tg::rollout_step_kernel<QuadPoleEnv<float>,float,true>", "bound": "hbm", "n_envs": n,
            "us_per_launch": best, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_env_step": ALGO_BYTES["QuadPole"], "same_bytes_device_copy_us": copy_us,
            "traffic": traffic,
            "traffic_source": f"profiles/{pmc[0]} (2 x FETCH_SIZE + WRITE_SIZE per launch)" if traffic else None}


def forced_rollout_probe(tg, dev, n, T=256):
    """The dynamics kernel in its one-launch form (tg_rollout_forced: every time step of a teacher-forced replay in one launch, the
    state in registers between steps) at `n` QuadPole envs x `T` steps with nobody terminating.  Against the HBM roofline twice: with
    SURVEY 8(d)'s algorithmic bytes (189 B / env-step: the per-step kernel's state read + write) and with the bytes this form really
    moves (the action read, the next observation, reward and mask byte written: 101 B) -- the second is the honest fraction."""
    import ctypes as C
    import torch
    N_ = tg._native
    env = tg.QuadPole(max_steps=T)
    env.spatial_bounds = tuple((-1e9, 1e9) for _ in env.spatial_bounds)
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (8,), cov=0.3, device=dev)
    eng = tg.DeviceRollout(env, pol, n // 256, 256, seed=1, fused=False)
    eng._seed_host, eng._stream_host = 1, 0
    eng.params = env.native_params()
    lib, st = N_.load(), N_.stream_ptr(dev)
    best = None
    for _ in range(4):
        eng._enqueue_prepare(None)                       # zeroed trajectory (actions 0 = hover thrust), fresh initial states
        tr = eng.traj.native()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N_.check(lib.tg_rollout_forced(C.byref(eng.params), C.byref(tr), 0, T, st), "tg_rollout_forced")
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    steps = int(eng.traj.len.sum().item())
    real = 4 * 20 + 4 * 4 + 4 + 1
    del eng
    torch.cuda.empty_cache()
    return {"kernel": "tg::rollout_forced_kernel<QuadPoleEnv<float>,float>", "bound": "hbm", "n_envs": n, "horizon": T, "ms_per_launch": best,
            "env_steps": steps, "env_steps_per_s": steps / best * 1e3,
            "achieved": ALGO_BYTES["QuadPole"] * steps / best / 1e6, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": ALGO_BYTES["QuadPole"] * steps / best / 1e6 / HBM_PEAK_GBS, "bytes_per_env_step": ALGO_BYTES["QuadPole"],
            "real_bytes_per_env_step": real, "real_GBps": real * steps / best / 1e6, "frac_of_real_traffic": real * steps / best / 1e6 / HBM_PEAK_GBS,
            "note": "`frac` prices the launch with SURVEY 8(d)'s 189 B / env-step, which this form does not move (the state stays in registers: "
                    "it can exceed 1); `frac_of_real_traffic` with the 101 B it does move"}


def _pmc_bytes_per_row(family):
    """HBM traffic per row by the PMC counters of this family's 2^22-row probe (profiles/), or None."""
    try:
        with open(os.path.join(REPO, "profiles", PMC_JSON[family])) as f:
            d = json.load(f)
        return d["traffic_bytes_per_launch"] / d["rows"], PMC_JSON[family]
    except Exception:
        return None, None


def _f32_pmc_bytes_per_row(family, wide=False):
    """HBM traffic per row of the fp32 chain learner's kernels (5-128-128-1, 2^20-row probe: profiles/r05_f32_chain_pmc.json; wide: the
    H = 256 learner at 20-256x5-4, profiles/r05_f32_wide_pmc.json)."""
    if wide:
        try:
            with open(os.path.join(REPO, "profiles", "r05_f32_wide_pmc.json")) as f:
                d = json.load(f)
            k = d["kernels"]["void tg::mlp_f32_wide_kernel<true>" if family == "fwd" else "tg::mlp_f32_wide_dw_kernel"]
            return k["bytes_per_row"], "r05_f32_wide_pmc.json"
        except Exception:
            return None, None
    for name in ("r05_f32_chain_pmc.json", "r03_f32_chain_pmc.json"):          # (round 5: the resident 16-row kernel; round 3: the 32-row chain kernel)
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                d = json.load(f)
            k = d["kernels"]["forward_backward" if family == "fwd" else "weight_grad"]
            return k["traffic_bytes_per_launch"] / d["rows"], name
        except Exception:
            continue
    return None, None


class PowerSampler:
    """Socket power (and the driver's sclk reading) of the device this rank runs on, sampled from a thread while a region runs:
    hwmon's power1_input / freq1_input of the PCI device torch reports (microwatts / Hz; the files rocm-smi itself reads), or
    `rocm-smi --showpower --json` when sysfs is not readable.  The update sits on the package power limit (power1_cap): the line
    carries the evidence instead of a builder-kept profile."""

    def __init__(self, torch, dev_index, period_s=0.025):
        import glob
        import threading
        self.period, self.samples, self.sclk, self.marks = period_s, [], [], []
        self._stop, self._thread, self._threading = threading.Event(), None, threading
        self.hwmon = self.cap_W = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hits = glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/power1_input")
            if hits:
                self.hwmon = os.path.dirname(hits[0])
                with open(os.path.join(self.hwmon, "power1_cap")) as f:
                    self.cap_W = int(f.read()) * 1e-6
        except Exception:
            self.hwmon = None
        self.source = f"{self.hwmon}/power1_input" if self.hwmon else "rocm-smi --showpower --json"

    def _read(self):
        if self.hwmon:
            with open(os.path.join(self.hwmon, "power1_input")) as f:
                w = int(f.read()) * 1e-6
            try:
                with open(os.path.join(self.hwmon, "freq1_input")) as f:
                    self.sclk.append(int(f.read()) * 1e-9)
            except Exception:
                pass
            return w
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        return float(next(v for k, v in card.items() if "Power" in k))

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(), self._read()))
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        self._thread = self._threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)

    def window(self, t0, t1):
        """Statistics of the samples taken in [t0, t1] (perf_counter times)."""
        w = [v for t, v in self.samples if t0 <= t <= t1]
        if not w:
            return None
        w.sort()
        return {"mean": sum(w) / len(w), "median": w[len(w) // 2], "max": w[-1], "min": w[0], "samples": len(w), "cap": self.cap_W,
                "source": self.source}


def clock_probe_result(buf):
    """(GHz, workgroups stamped) from a tg_clock_probe_attach buffer: shader-clock ticks / 100-MHz ticks, summed over workgroups."""
    clk, real, n = (int(v) for v in buf[:3].tolist())
    return (0.1 * clk / real if real > 0 else None), n


def sustained_mfma_peak(tg, torch, dev, dtype_code, sampler=None):
    """tg_mfma_sustained_probe run back to back: ~0.5 s to let the clock settle under the power limit, then ~100 ms timed by one HIP
    event pair, its own in-kernel clock and the socket power beside it.  -> dict (TFLOP/s, GHz, W)."""
    N_ = tg._native
    lib = N_.load()
    blocks = lib.tg_mfma_sustained_probe_blocks()
    g = torch.Generator(device=dev).manual_seed(1)
    if dtype_code == 0:
        w = (torch.rand(32 * 1024, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        x = (torch.rand(blocks * 512 * 128, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        iters = 12000
    else:
        w = torch.rand(16 * 1024, device=dev, generator=g) * 2 - 1
        x = torch.rand(blocks * 512 * 64, device=dev, generator=g) * 2 - 1
        iters = 800
    out = torch.empty(blocks * 512, device=dev)
    probe = torch.zeros(N_.TG_CLOCK_PROBE_U64, dtype=torch.int64, device=dev)
    st = N_.stream_ptr(dev)
    launch = lambda: N_.check(lib.tg_mfma_sustained_probe(dtype_code, iters, w.data_ptr(), x.data_ptr(), out.data_ptr(), st), "tg_mfma_sustained_probe")
    torch.cuda.synchronize()
    launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); b.record()
    torch.cuda.synchronize()
    one_ms = a.elapsed_time(b)
    n_warm, n_timed = max(4, int(500.0 / one_ms)), max(4, int(100.0 / one_ms))
    for _ in range(n_warm):
        launch()
    torch.cuda.synchronize()
    N_.check(lib.tg_clock_probe_attach(N_.TG_PROBE_MFMA_LOOP, probe.data_ptr()), "tg_clock_probe_attach")
    t0 = time.perf_counter()
    a.record()
    for _ in range(n_timed):
        launch()
    b.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    N_.check(lib.tg_clock_probe_attach(N_.TG_PROBE_MFMA_LOOP, None), "tg_clock_probe_attach")
    ms = a.elapsed_time(b)
    ghz, _ = clock_probe_result(probe)
    flops = lib.tg_mfma_sustained_probe_flops(dtype_code, iters) * n_timed
    return {"TFLOPs": flops / ms / 1e9, "clock_GHz": ghz, "timed_ms": ms, "warm_ms": one_ms * n_warm,
            "power_W": sampler.window(t0, t1) if sampler is not None else None,
            "kernel": "tg::mfma_loop_bf16_kernel (v_mfma_f32_16x16x32_bf16)" if dtype_code == 0 else "tg::mfma_loop_f32_kernel (v_mfma_f32_32x32x2_f32)",
            "what": "the chain kernels' bare inner loop (LDS A fragments, register B fragments, 2 waves per SIMD, no global traffic) run "
                    "back to back: the matrix rate the package sustains under its power limit"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (whole groups); default: the config's shard")
    ap.add_argument("--horizon", type=int, default=None, help="default 256 (c2: CartPole's own 500)")
    ap.add_argument("--updates", type=int, default=None,
                    help="updates_per_iter (c3: the PPO factory's 32; c4 / c5: GRPO's constructor default 10, grpo.py:26-35)")
    ap.add_argument("--policy-dtype", default=None, choices=["bf16", "fp32"], help="default bf16 (c2: fp32, as BASELINE says)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fixed-work", action="store_true", help="skip the all-alive (bounds opened) step after the timed region")
    ap.add_argument("--graph", action="store_true", help="replay the T-step rollout loop as one hipGraph")
    ap.add_argument("--no-fused", action="store_true",
                    help="per-step launches (actor GEMMs + tg_rollout_step) instead of the fused persistent rollout kernel")
    ap.add_argument("--event-every", type=int, default=0,
                    help="time the hot kernels' launches on every N-th step of the timed region (0: every step when a step takes >= 20 ms, else every 8th)")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="do not bracket the rollout / learner launches with HIP events (A/B of the measurement's own cost; no roofline objects)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only for rehearsing the multi-rank path on a single GPU (ranks share the device)")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="with --config c3: also run C2 and the C4 / C5 per-GPU shards after the headline's timed region and nest their "
                         "results under roofline.other_configs (auto: when no flag reshapes the workload)")
    ap.add_argument("--check", action="store_true",
                    help="with --gpus N > 1: before the timed run every rank runs the small rank-count cases of tests/dist_product_worker.py "
                         "twice -- as a rank of the N-rank group and alone (a one-rank subgroup) -- and asserts that its trajectory "
                         "shard, PPO's global moments and the post-step weights agree (SURVEY 8e); the result rides in the JSON line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare invocation: one child process per GPU through torchrun.  This parent has made no GPU call (torch is not
        # even imported) and only waits; it never replaces itself.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    # The JSON line must be the only thing on stdout, but libraries write there too (RCCL prints a version banner at
    # communicator set-up): keep a private handle on the real stdout and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")


    def progress(msg):                       # stderr only: the JSON line is the one thing on stdout
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.config == "c3":
        progress("cpu baseline (3 legs, ~40 s) ...")
        cpu = cpu_baseline(args.horizon if args.horizon is not None else 256, 32)
        progress(f"cpu baseline: {cpu['value']:.0f} env-steps/s on {cpu['cores']} cores")

    import torch
    import torch.distributed as dist
    import trajopt_grpo_amd as tg

    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPUs visible")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    in_group = world > 1 or "WORLD_SIZE" in os.environ        # under torchrun even one rank forms a process group
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    n_ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    rank_check = None
    if args.check:
        if world not in (2, 4, 8):
            raise SystemExit("--check compares an N-rank run with a one-rank run of cases with max(4, N) groups: needs --gpus 2, 4 or 8")
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import dist_product_worker as W
        solo = None
        for r in range(world):                                  # every rank takes part in every new_group call
            g = dist.new_group([r])
            if r == rank:
                solo = g
        names = [n for n in W.CASES if not n.endswith("ragged") or world == 2]
        with torch.cuda.device(dev):
            # (the worker runs on cuda:0 of the process; under nccl every rank has its own device, so make it current)
            many = W.run_cases(names, rank, world, None, device=dev, groups=max(4, world))
            one = W.run_cases(names, 0, 1, solo, device=dev, emulate_world=world, groups=max(4, world))
        rank_check = {n: W.check_case(one[n], [many[n]], n) for n in names}
        flag = torch.ones(1, device=dev)
        dist.all_reduce(flag)                                   # every rank got here: nobody raised
        assert int(flag.item()) == world
        progress_early = f"rank-count check passed on {world} ranks: {rank_check}"
        if rank == 0:
            print(f"[bench] {progress_early}", file=sys.stderr, flush=True)

    ctx = {"rank": rank, "world": world, "dev": dev, "in_group": in_group, "n_ranks_seen": n_ranks_seen, "rank_check": rank_check,
           "cpu": cpu, "torch": torch, "dist": dist, "tg": tg, "progress": progress}
    out = run_config(args, ctx)

    # ---- the other GPU configs of BASELINE.json (C2, C4's and C5's per-GPU shards) on the same box, in the same process, after the
    # headline's timed region: their launches untimed, then a few steps with the hot kernels' launches bracketed for the roofline.
    # Nested under `roofline` (where the driver's parser keeps objects whole).  Only for the plain headline command: any flag that
    # reshapes the workload (--envs, --horizon, ...) is a probe run of ONE config.
    plain = (args.envs is None and args.horizon is None and args.updates is None and args.policy_dtype is None and not args.graph
             and not args.no_fused and not args.no_launch_events)
    if args.config == "c3" and (args.other_configs == "on" or (args.other_configs == "auto" and plain)):
        others = {}
        for name, steps, warm, ev_steps in (("c2", 200, 5, 8), ("c4", 20, 3, 2), ("c5", 20, 3, 2)):
            a2 = argparse.Namespace(**vars(args))
            a2.config, a2.steps, a2.warmup, a2.event_steps = name, steps, warm, ev_steps
            a2.envs = a2.horizon = a2.updates = a2.policy_dtype = None
            a2.no_fixed_work = True
            progress(f"other config {name}: {warm} + {steps} steps ...")
            try:
                o = run_config(a2, ctx, secondary=True)
            except Exception as e:           # (a leg that cannot run must not take the headline with it)
                import traceback
                traceback.print_exc()
                o = {"error": repr(e)} if rank == 0 else None
            if rank == 0:
                if "error" in o:
                    others[name] = o
                    continue
                r = o.get("roofline") or {}
                others[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                                "n_gpus": o["n_gpus"], "env_steps_per_step": o["env_steps_per_step"], "dtype": o["dtype"],
                                "workload": o["config"]["workload"], "update_ns_per_valid_row": o["update_ns_per_valid_row"],
                                "kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"),
                                "roofline_unit": r.get("unit"), "frac": r.get("frac"), "frac_of_sustained": r.get("frac_of_sustained"),
                                "kernel_launches_timed": r.get("launches"), "kernel_avg_launch_ms": r.get("avg_launch_ms"),
                                "launch_events": o["launch_events"],
                                "kernels": {k: {"kernel": v["kernel"], "bound": v["bound"], "frac": v["frac"], "avg_launch_ms": v["avg_launch_ms"]}
                                            for k, v in (o.get("kernels") or {}).items()},
                                "rollout_kernel": ({k: (o["rollout_kernel"].get(k)) for k in ("kernel", "bound", "frac", "avg_launch_ms")}
                                                   if o.get("rollout_kernel") else None)}
        if rank == 0:
            out.setdefault("roofline", {})["other_configs"] = others
            out["roofline"]["other_configs_note"] = (
                "BASELINE.json configs[1], [3], [4] (per-GPU shards) run in this process after the headline's timed region: `value` from "
                "`steps` steps bracketed by barrier + synchronize with NO per-launch events, `kernel` / `frac` from extra steps with the hot "
                "kernels' launches bracketed by HIP events (launch_events.timed_steps)")
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if in_group:
        dist.destroy_process_group()


def run_config(args, ctx, secondary=False):
    """One config's bench: warm-up, the timed region, the per-kernel figures.  -> the JSON object on rank 0 (None elsewhere).
    secondary: one of the other configs riding behind the headline -- no CPU baseline, no stand-alone probes; the timed region runs
    with no per-launch events and `args.event_steps` extra steps afterwards carry them."""
    rank, world, dev, in_group = ctx["rank"], ctx["world"], ctx["dev"], ctx["in_group"]
    n_ranks_seen, rank_check, cpu = ctx["n_ranks_seen"], ctx["rank_check"], (None if secondary else ctx["cpu"])
    torch, dist, tg, progress = ctx["torch"], ctx["dist"], ctx["tg"], ctx["progress"]
    args = argparse.Namespace(**vars(args))          # (the defaults filled in below are this config's)
    if secondary:
        import gc
        gc.collect()
        torch.cuda.empty_cache()                     # (the previous config's workspaces: tens of GB)
    env_name, algo_name, G_shard, E, agents, restart, total_envs = CONFIGS[args.config]
    cartpole = env_name == "CartPole"
    if args.horizon is None:
        args.horizon = 500 if cartpole else 256
    if args.policy_dtype is None:
        args.policy_dtype = "fp32" if cartpole else "bf16"
    # C2: actor 5-128-128-1, cov 0.5 and the hyper-parameters of pipelines/cartpole_pipeline_grpo.py:54-68
    obs_dim, act_dim, hidden, cov = (5, 1, (128, 128), 0.5) if cartpole else (20, 4, HIDDEN, 0.3)
    if args.envs is not None:
        G_local = args.envs // E
    elif args.scaling == "strong":
        if (total_envs // E) % world:
            raise SystemExit(f"{total_envs // E} groups do not divide over {world} ranks")
        G_local = total_envs // E // world
    else:
        G_local = G_shard
    envs_local = G_local * E
    updates = args.updates if args.updates is not None else (32 if algo_name == "ppo" else 10)
    G_global = G_local * world
    T = args.horizon
    cdt = torch.bfloat16 if args.policy_dtype == "bf16" else None
    torch.manual_seed(0)                                      # identical random-init weights on every rank
    if algo_name == "ppo":
        policy = tg.GaussianActorCritic_NeuralNetwork(obs_dim, act_dim, hidden, cov=cov, device=dev)
    else:
        policy = tg.GaussianActor_NeuralNetwork(obs_dim, act_dim, hidden, cov=cov, device=dev)

    def make_env(open_bounds=False):
        if cartpole:
            return tg.CartPole(max_steps=T)
        env = tg.QuadPoleSwarm(n_agents=agents, max_steps=T) if agents > 1 else tg.QuadPole(max_steps=T)
        if open_bounds:
            env.spatial_bounds = tuple((-1e9, 1e9) for _ in env.spatial_bounds)
        return env

    mgr = tg.RolloutManager(make_env, policy, restart=restart, num_workers=G_global, num_episodes_per_worker=E,
                            dtype=torch.float32, seed=1234, compute_dtype=cdt, use_graph=bool(args.graph),
                            fused=False if args.no_fused else None)
    buf = tg.Rollout_Buffer(mgr)
    if algo_name == "ppo":
        algo = tg.PPO(epsilon=0.2, policy=policy, optimizer=torch.optim.Adam(policy.parameters(), lr=3e-4), ref_model=None,
                      updates_per_iter=updates, c1=0.5, kl_coeff=0.5, gamma=0.999, lam=0.95, entropy=0.01,
                      batch_size=None, autocast_dtype=cdt)
    else:
        algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5 if cartpole else 0.99, policy=policy, optimizer=torch.optim.Adam(policy.parameters(), lr=3e-4),
                       updates_per_iter=updates, autocast_dtype=cdt)

    def barrier():
        if in_group:
            dist.barrier()
        torch.cuda.synchronize()

    warm_ms = 1e9                        # duration of the last warm-up step (decides how densely launches are timed below)
    for i in range(args.warmup):
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        buf.sample()
        algo.learn(buf)
        torch.cuda.synchronize()
        warm_ms = 1e3 * (time.perf_counter() - w0)
        progress(f"warm-up step {i + 1}/{args.warmup}")
    # event-pair overhead (no kernel in between), for the per-launch timing below
    pairs = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ev_overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]

    env_steps = 0
    t_roll = t_learn = 0.0
    launches = []            # (duration ms, env-steps in that launch)
    launch_units = []
    nets = [policy.actor] + ([policy.critic] if algo_name == "ppo" else [])
    learner_mlps = [m for m in (algo._mlp(n_) for n_ in nets) if m is not None]
    # HIP-event pairs around every hot-kernel launch.  An event is a packet on the queue (~6 us between two kernels): on a step
    # of a few milliseconds (C2: ~45 timed launches in 3.5 ms) timing EVERY launch of EVERY step slows the step by 8 %, so such
    # steps are timed on every `event_every`-th step of the timed region (8th: a timed C2 step is ~15 % longer, so the line loses ~2 %;
    # `--no-launch-events` gives the undisturbed figure); steps >= 20 ms (C3) time all of them.
    event_every = args.event_every if args.event_every > 0 else (1 if warm_ms >= 20.0 else 8)
    event_lists = {id(m): ([], [], []) for m in learner_mlps}
    timed_steps = 0

    def set_events(on):
        mgr.engine.step_events = [] if (on and not args.graph) else None
        for m in learner_mlps:
            m.dx_events, m.dw_events, m.fwd_events = event_lists[id(m)] if on else (None, None, None)

    # Phase split of a step on the GPU's own timeline: one event at the start of the step, one after sample() has enqueued the
    # rollout, one after learn() has enqueued the update.  The host is NOT synchronised between steps: the learner reads its loss
    # statistics lazily, so the next rollout is enqueued while the last updates still run (the timed region is bracketed by a
    # barrier + synchronize on both sides, as the contract says).
    # The cyclic garbage collector is off inside the timed region (as `timeit` does): the per-launch event objects this script keeps
    # -- 600 pairs per C4 step -- otherwise trigger a full collection somewhere in the region, a 70-150 ms host pause with nothing
    # queued on the GPU (C4, 20 steps: 43 against 49 M env-steps/s; the pause never happens without the events)
    # ---- evidence that rides in the line: in-kernel clocks of the update's kernel families, socket power, collectives ----
    N_ = tg._native
    f32_learner = any(m._f32 is not None for m in learner_mlps)
    chain_learner = any(m._chain is not None and m._bchain is not None for m in learner_mlps)
    probe_fams = ({"fwd": N_.TG_PROBE_F32_CHAIN, "dw": N_.TG_PROBE_F32_WEIGHT_GRAD} if f32_learner else
                  ({"fwd": N_.TG_PROBE_FWD_CHAIN, "bwd": N_.TG_PROBE_BWD_CHAIN, "dw": N_.TG_PROBE_WEIGHT_GRAD} if chain_learner else {}))
    probe_bufs = {}
    ev_in_region = not args.no_launch_events and not secondary      # (secondary: events and clock stamps on extra steps AFTER the region)

    def attach_probes():
        if rank == 0:
            torch.cuda.synchronize()
            for fam, code in probe_fams.items():
                probe_bufs[fam] = torch.zeros(N_.TG_CLOCK_PROBE_U64, dtype=torch.int64, device=dev)
                N_.check(N_.load().tg_clock_probe_attach(code, probe_bufs[fam].data_ptr()), "tg_clock_probe_attach")

    def detach_probes():
        for fam, code in probe_fams.items():
            if fam in probe_bufs:
                clocks[fam] = clock_probe_result(probe_bufs[fam])
                N_.check(N_.load().tg_clock_probe_attach(code, None), "tg_clock_probe_attach")

    def collect_rollout_events(n_steps_now):
        nonlocal launches
        if mgr.engine.step_events:
            if mgr.engine.fused:
                launches += [(a.elapsed_time(b), n_steps_now) for _k, a, b in mgr.engine.step_events]
                launch_units.append(n_steps_now)
            else:
                alive = buf.device_traj.mask.sum(1, dtype=torch.int64).tolist()
                launches += [(a.elapsed_time(b), alive[t]) for t, a, b in mgr.engine.step_events]
            mgr.engine.step_events = []

    clocks = {}
    if ev_in_region:
        attach_probes()
    sampler = PowerSampler(torch, dev.index).start() if rank == 0 else None
    tg.distributed.COLLECTIVE_LOG = coll_log = []
    import gc
    gc.collect()
    gc.disable()
    phase_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    step_units = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed = ev_in_region and _ % event_every == 0
        set_events(timed)
        timed_steps += timed
        r0 = time.perf_counter()
        phase_ev[_][0].record()
        buf.sample()
        phase_ev[_][1].record()
        r1 = time.perf_counter()
        algo.learn(buf)
        phase_ev[_][2].record()
        n_steps_now = buf.device_traj.env_steps()           # (on the host since learn() asked for it: no wait here)
        env_steps += n_steps_now
        step_units.append(n_steps_now)
        collect_rollout_events(n_steps_now)
        if os.environ.get("TG_BENCH_STEP_TIMES"):
            print(f"step {_} timed={timed} sample {1e3 * (r1 - r0):.3f} ms (host) learn enqueue {1e3 * (time.perf_counter() - r1):.3f} ms (host)", file=sys.stderr, flush=True)
        if (_ + 1) % max(5, args.steps // 4) == 0:
            progress(f"step {_ + 1}/{args.steps}")
    barrier()
    t_end = time.perf_counter()
    dt = t_end - t0
    tg.distributed.COLLECTIVE_LOG = None
    power_timed = sampler.window(t0, t_end) if sampler is not None else None
    if ev_in_region:
        detach_probes()
    elif secondary and not args.no_launch_events:
        # the hot kernels' launches of `event_steps` further steps, bracketed by HIP events and stamped (outside the timed region)
        attach_probes()
        for _ in range(args.event_steps):
            set_events(True)
            buf.sample()
            algo.learn(buf)
            collect_rollout_events(buf.device_traj.env_steps())
            timed_steps += 1
        torch.cuda.synchronize()
        detach_probes()
    gc.enable()
    # collectives of the timed region on this rank: count, bytes and stream time per tag
    coll = {}
    for tag, nbytes, a, b in coll_log:
        c = coll.setdefault(tag, {"count": 0, "bytes": 0, "ms": 0.0})
        c["count"] += 1; c["bytes"] += nbytes; c["ms"] += a.elapsed_time(b)
    set_events(not args.no_launch_events)       # (the lists the code below reads)
    t_roll = sum(e[0].elapsed_time(e[1]) for e in phase_ev) * 1e-3
    t_learn = sum(e[1].elapsed_time(e[2]) for e in phase_ev) * 1e-3
    if os.environ.get("TG_BENCH_STEP_TIMES"):
        for k, e in enumerate(phase_ev):
            print(f"step {k} on the GPU: rollout {e[0].elapsed_time(e[1]):.3f} ms, learn {e[1].elapsed_time(e[2]):.3f} ms, env-steps {step_units[k]}", file=sys.stderr, flush=True)

    fam_launches = {"bwd": [], "dw": [], "fwd": []}     # (ms, algorithmic bytes, rows, kernel name) per launch
    for m in learner_mlps:
        for fam, attr in (("bwd", "dx_events"), ("dw", "dw_events"), ("fwd", "fwd_events")):
            fam_launches[fam] += [(a.elapsed_time(b), rows * bpr, rows, name) for a, b, rows, bpr, name in (getattr(m, attr) or [])]
            setattr(m, attr, None)

    # ---- fixed work: one step with nobody terminating (every env runs the whole horizon) ----
    progress(f"timed region done: {1e3 * dt / args.steps:.0f} ms/step")
    fixed = None
    if not args.no_fixed_work and hasattr(make_env(), "spatial_bounds"):
        mgr_open = tg.RolloutManager(lambda: make_env(True), policy, restart=restart, num_workers=G_global,
                                     num_episodes_per_worker=E, dtype=torch.float32, seed=4321, compute_dtype=cdt,
                                     fused=False if args.no_fused else None)
        buf_open = tg.Rollout_Buffer(mgr_open)
        buf_open.sample(); algo.learn(buf_open)                 # grows the workspaces to the full row count
        barrier()
        f0 = time.perf_counter()
        buf_open.sample(); algo.learn(buf_open)
        barrier()
        fixed = (time.perf_counter() - f0, buf_open.device_traj.env_steps())
        del mgr_open, buf_open

    dyn = dynamics_kernel_probe(tg, dev, envs_local * agents) if rank == 0 and agents == 1 and not cartpole and not secondary else None
    if dyn is not None and args.config == "c3":
        # north_star's ">= 60 % of the HBM roofline on the dynamics kernel": reachable where the stream is HBM-resident.  At 65,536
        # envs one launch moves 12.4 MB in ~5 us: the launch is a kernel boundary (~1.5 us) plus ONE wave of work per SIMD lane
        # group -- 1,024 waves on 1,024 SIMDs, each a dependent load -> ~300 flops -> store chain -- so it is latency-, not
        # bandwidth-bound (a bare device copy of the same bytes takes 3.7-4.0 us); at 4,194,304 envs the same kernel streams.
        big = dynamics_kernel_probe(tg, dev, 4194304, launches=16)
        dyn["at_4194304_envs"] = big
        dyn["one_launch_form"] = forced_rollout_probe(tg, dev, envs_local * agents, T)
        dyn["note"] = ("65,536 envs = 12.4 MB per launch: one wave per SIMD and a kernel boundary per step, latency-bound (a bare copy of "
                       "the same bytes: same_bytes_device_copy_us); the same kernel at 4,194,304 envs (at_4194304_envs) is HBM-resident; "
                       "one_launch_form = the same dynamics with the time loop INSIDE the launch (tg_rollout_forced, the teacher-forced "
                       "replay path): at 65,536 envs a plain write stream of the trajectory")
    fused_all_alive = None
    if rank == 0 and mgr.engine.fused and agents == 1 and not cartpole and not secondary:
        # the fused kernel with nobody terminating (bounds opened): its matrix-core rate without idle lanes
        eng = tg.DeviceRollout(make_env(True), policy, G_local, E, restart=restart, seed=7, compute_dtype=cdt, fused=True)
        eng.run()
        eng.step_events = []
        eng.run()
        torch.cuda.synchronize()
        _, a, b = eng.step_events[0]
        ms = a.elapsed_time(b)
        n_par = sum(p.numel() for p in policy.actor.parameters())
        fused_all_alive = {"ms_per_rollout": ms, "env_steps": eng.traj.env_steps(),
                           "env_steps_per_s": eng.traj.env_steps() / ms * 1e3,
                           "achieved_TFLOPs": 2.0 * n_par * eng.traj.env_steps() / ms / 1e9,
                           "frac_of_2500_TFLOPs": 2.0 * n_par * eng.traj.env_steps() / ms / 1e9 / 2500.0}
        del eng
    coll_ms = sum(c["ms"] for c in coll.values())
    tot = torch.tensor([float(env_steps), dt, t_roll, t_learn, fixed[0] if fixed else 0.0, float(fixed[1]) if fixed else 0.0, coll_ms],
                       dtype=torch.float64, device=dev)
    steps_minmax = (float(env_steps), float(env_steps))
    coll_ms_max = coll_ms
    if in_group:
        mx, mn = tot.clone(), tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_steps, dt, t_roll, t_learn = float(tot[0]), float(mx[1]), float(mx[2]), float(mx[3])
        fixed = (float(mx[4]), float(tot[5])) if fixed else None
        steps_minmax, coll_ms_max = (float(mn[0]), float(mx[0])), float(mx[6])
    else:
        total_steps = float(env_steps)
    sustained = None
    if rank == 0 and probe_fams and not args.no_launch_events:
        code = 1 if f32_learner else 0
        if code not in ctx.setdefault("sustained", {}):        # (once per process and dtype: the other configs' legs reuse it)
            progress("sustained matrix-rate probe (~1 s) ...")
            ctx["sustained"][code] = sustained_mfma_peak(tg, torch, dev, code, sampler)
        sustained = ctx["sustained"][code]
    if sampler is not None:
        sampler.stop()

    if rank == 0:
        n_nets = len(nets)
        workload = {"c2": f"C2: CartPole GRPO, {envs_local} envs/GPU ({G_local} groups x {E}) x {T}-step horizon, actor 5-128-128-1, "
                          f"{updates} updates/iter, {args.policy_dtype} policy",
                    "c3": f"C3: QuadPole (quadrotor_env.py) PPO, {envs_local} envs/GPU x {T}-step horizon, natural termination, "
                          f"actor-critic 20-256x5-{{4,1}}, {updates} full-batch updates/iter, {args.policy_dtype} policy",
                    "c4": f"C4: QuadPole GRPO, {envs_local} envs/GPU ({G_local} restart groups x {E}) x {T}-step horizon, actor "
                          f"20-256x5-4, {updates} updates/iter, {args.policy_dtype} policy",
                    "c5": f"C5: QuadPoleSwarm GRPO, {envs_local} envs x {agents} agents per GPU ({G_local} groups x {E} episodes) x "
                          f"{T}-step horizon, shared actor 20-256x5-4, {updates} updates/iter, {args.policy_dtype} policy"}[args.config]
        out = {
            "metric": "env-steps/sec at 65k parallel quadrotor envs (rollout + PPO update)" if args.config == "c3" else
                      f"env-steps/sec, {args.config} (rollout + GRPO update)",
            "value": total_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            # the arithmetic the step computes in: the MLP passes (96 % of the GPU time at C3) multiply bf16 operands into f32
            # accumulators; environment dynamics, returns / advantages, loss head and Adam (f32 master weights) are f32
            "dtype": "bf16 (f32 accumulate) MLP; f32 env / returns / loss / optimizer" if args.policy_dtype == "bf16" else "f32",
            "policy_dtype": args.policy_dtype, "data": "synthetic",
            "n_ranks_seen": n_ranks_seen,
            "rank_count_check": rank_check,
            "config": {"workload": workload, "name": args.config, "envs_per_gpu": envs_local, "agents_per_env": agents,
                       "envs_total": envs_local * world, "horizon": T, "updates_per_iter": updates,
                       "parallelism": f"env-shard x{world} (whole groups per rank), 1 grad all-reduce per optimizer step"},
            "rollout_only_env_steps_per_s": total_steps / t_roll if t_roll > 0 else None,
            "rollout_ms": 1e3 * t_roll / args.steps,
            "launch_events": None if args.no_launch_events else
                             ({"after_the_timed_region": True, "timed_steps": timed_steps} if secondary else
                              {"every_nth_step": event_every, "timed_steps": timed_steps}),
            "env_steps_per_step": total_steps / args.steps,
            # per rank: the smallest / largest number of valid env-steps a rank processed per step (ranks wait for the slowest at
            # every gradient all-reduce)
            "env_steps_per_step_per_rank": {"min": steps_minmax[0] / args.steps, "max": steps_minmax[1] / args.steps},
            # every collective of the timed region (rank 0's log; events on the stream the collective is ordered on, so `ms`
            # includes waiting for the slowest rank): "grad" = ONE flat gradient all-reduce per optimizer step
            "collectives": {"backend": (args.backend if in_group else None),
                            "per_step": sum(c["count"] for c in coll.values()) / args.steps,
                            "allreduce_ms_per_step": coll_ms / args.steps,
                            "allreduce_ms_per_step_max_over_ranks": coll_ms_max / args.steps,
                            "allreduce_bytes_per_step": sum(c["bytes"] for c in coll.values()) / args.steps,
                            "by_tag": {k: {"per_step": v["count"] / args.steps, "bytes_each": v["bytes"] / max(v["count"], 1),
                                           "ms_per_step": v["ms"] / args.steps} for k, v in sorted(coll.items())}},
            # fixed work: independent of how long the policy survives (the number of valid rows grows as it learns)
            "update_ns_per_valid_row": 1e9 * t_learn * world / total_steps if total_steps else None,
            "update_ns_per_row_update_net": 1e9 * t_learn * world / total_steps / updates / n_nets if total_steps else None,
        }
        if fixed:
            out["fixed_work_ms_per_step"] = 1e3 * fixed[0]
            out["fixed_work_env_steps_per_s"] = fixed[1] / fixed[0]
            out["fixed_work_note"] = "one step with the spatial bounds opened: every env runs the whole horizon (all ranks' rows)"
        fused = mgr.engine.fused
        out["rollout_path"] = "fused persistent kernel (tg_fused_rollout)" if fused else "per-step launches (GEMMs + tg_rollout_step)"
        if launches and fused:
            # one launch per rollout: actor MLP on the matrix cores + sampling + dynamics + recording.
            # algorithmic flops = 2 * actor parameters per valid env-step (539 kflop for 20-256x5-4)
            n_par = sum(p.numel() for p in policy.actor.parameters())
            dur = sum(d for d, _ in launches) * 1e-3
            ach = 2.0 * n_par * sum(launch_units) / dur / 1e12
            f32 = args.policy_dtype == "fp32"
            peak = 157.3 if f32 else 2500.0          # dense fp32 / bf16 matrix peaks (MI355X_MICROARCH.md)
            out["rollout_kernel"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                     "traffic": 1706642656 if envs_local == 65536 and agents == 1 else None,
                                     "traffic_source": "profiles/r01_fused_rollout_pmc.json (all-alive launch: 2 x FETCH_SIZE + "
                                                       "WRITE_SIZE = the 101 B/env-step trajectory record; weights stay in L2)",
                                     "kernel": (f"tg::fused_rollout_f32{'x16' if mgr.engine._f32_block_envs == 16 else ''}_kernel"
                                                f"<{env_name}Env<float>,{hidden[0]},{len(hidden) - 1}>") if f32 else
                                               "tg::fused_rollout_kernel<QuadPoleEnv<float>,256,1,8>",
                                     "flops_per_env_step": 2 * n_par, "launches": len(launches),
                                     "avg_launch_ms": 1e3 * dur / len(launches),
                                     "note": "valid env-steps only (natural termination: ended envs idle their lanes)",
                                     "all_alive": fused_all_alive}
        elif launches:
            dur = sum(max(d - ev_overhead_ms, 1e-4) for d, _ in launches) * 1e-3
            units = sum(u for _, u in launches)
            ach = ALGO_BYTES["QuadPole"] * units / dur / 1e9
            out["rollout_kernel"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ach / HBM_PEAK_GBS, "traffic": 14334976 if envs_local == 65536 else None,
                                     "traffic_source": "profiles/r01_step_kernel_65536_pmc.json (all-alive launch)",
                                     "kernel": "tg::rollout_step_kernel<QuadPoleEnv<float>,float,true>",
                                     "bytes_per_env_step": ALGO_BYTES["QuadPole"], "launches": len(launches),
                                     "avg_launch_us": 1e6 * dur / len(launches), "event_pair_overhead_us": 1e3 * ev_overhead_ms}
        # ---- the three hand-written learner kernel families: every launch of the timed steps, HIP events on the launch stream ----
        kernels = {}
        notes = {"dw": "tg_mlp_weight_grad: every weight + hidden bias gradient of a net in one launch; algorithmic bytes = each stored dZ "
                       "and activation read once (the first activation is recomputed from the 64-B input row, the top layer's dZ "
                       "from the 16-B head gradient + 32 B of mask bits; the first layer's gradient is formed inside the backward chain and the "
                       "head's inside the forward chain: 3184 B per row.  With a stored top dZ and the first layer's job (round 2's form) it read 4752 B "
                       "per row at a higher byte rate and the step took 5 % longer)",
                 "bwd": "tg_mlp_backward_chain: the dZ of all hidden layers in one launch; 16 B + per layer 32 B of mask bits read "
                        "and, for every layer but the top and the bottom one, 512 B of dZ written; the bottom layer's dZ is contracted with the "
                        "64-B input row on chip (the first layer's weight and bias gradient)",
                 "fwd": "tg_mlp_forward_chain_loss (training passes): 64 B + the per-row loss inputs read; per stored layer 512 B of "
                        "activations + 32 B of mask bits written (neither the first nor the top activation is stored), 16 B of "
                        "d loss / d output; the loss head and the head's weight gradient are formed inside"}
        for fam, ls in fam_launches.items():
            if not ls:
                continue
            if "mlp_f32" in ls[0][3]:
                # the fp32 chain learner (the reference's own precision): bound by the fp32 matrix pipe (v_mfma_f32_32x32x2_f32,
                # 157.3 TFLOP/s dense = 1/16 of the bf16 rate); the events carry matrix-core flops per row, not bytes
                dur = sum(d for d, _, _, _ in ls) * 1e-3
                nflop = sum(b for _, b, _, _ in ls)
                nrows = sum(r for _, _, r, _ in ls)
                ach = nflop / dur / 1e12
                # (HBM traffic of these matrix-bound kernels, for the record: PMC bytes per row of the 5-128-128-1 probe x rows per launch;
                #  only quoted for that net shape)
                pmc_row, pmc_file = (_f32_pmc_bytes_per_row(fam) if (args.config == "c2" and hidden == (128, 128)) else
                                     (_f32_pmc_bytes_per_row(fam, wide=True) if "mlp_f32_wide" in ls[0][3] else (None, None)))
                kernels[fam] = {"bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
                                "traffic": pmc_row * nrows / len(ls) if pmc_row else None,
                                "traffic_source": (f"profiles/{pmc_file}: 2 x FETCH_SIZE + WRITE_SIZE per row of the 2^20-row probe, times this "
                                                   "run's average rows per launch (the kernel is bound by the matrix pipe, not by these)") if pmc_row else None,
                                "kernel": ls[0][3], "flops_per_row": nflop / nrows, "launches": len(ls),
                                "avg_launch_ms": 1e3 * dur / len(ls), "avg_rows_per_launch": nrows / len(ls),
                                "total_ms_per_step": 1e3 * dur / max(timed_steps, 1),
                                "note": {"fwd": "tg_mlp_f32_forward_backward: forward + loss head + backward-data pass of every row in one launch; "
                                                "flops = the algorithmic count, un-padded (2 H in + 4 (L - 1) H^2 + 4 H out per row; rounds 2-3 "
                                                "counted the first layer padded to 8 columns and no head: +0.4 % at C2)",
                                         "dw": "tg_mlp_f32_weight_grad: every weight / bias gradient of the net in one launch + the fixed-order slab "
                                               "reduction launch (which also carries the optimizer step); flops = the algorithmic count, un-padded "
                                               "(2 (L - 1) H^2 + 2 H in + 2 H out per row; rounds 2-3 counted the first layer padded to 32 columns: "
                                               "x 1.19 at C2 -- their 0.39 is 0.33 in this convention)"}.get(fam)}
                continue
            dur = sum(d for d, _, _, _ in ls) * 1e-3
            nbytes = sum(b for _, b, _, _ in ls)
            nrows = sum(r for _, _, r, _ in ls)
            ach = nbytes / dur / 1e9
            pmc, src = _pmc_bytes_per_row(fam)
            # the same launches against the matrix-core roofline (nominal 2 x weights flop per row; the backward-data chain has
            # no product for the first layer): the chain kernels load both resources at once under one power limit
            lins = [[m_ for m_ in n_.network if isinstance(m_, torch.nn.Linear)] for n_ in nets]
            w_all = sum(sum(l.weight.numel() for l in ls_) for ls_ in lins) / len(lins)
            w_first = sum(ls_[0].weight.numel() for ls_ in lins) / len(lins)
            tflops = 2.0 * (w_all - (w_first if fam == "bwd" else 0.0)) * nrows / dur / 1e12
            kernels[fam] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "matrix_TFLOPs": tflops, "matrix_frac_of_2500": tflops / 2500.0,
                            "traffic": pmc * nrows / len(ls) if pmc else None,
                            "traffic_source": f"profiles/{src}: 2 x FETCH_SIZE + WRITE_SIZE per row of the 2^22-row probe, times "
                                              "this run's average rows per launch" if pmc else None,
                            "kernel": ls[0][3], "bytes_per_row": nbytes / nrows, "launches": len(ls),
                            "avg_launch_ms": 1e3 * dur / len(ls), "avg_rows_per_launch": nrows / len(ls),
                            "total_ms_per_step": 1e3 * dur / max(timed_steps, 1), "note": notes[fam]}
        # ---- the power-limit evidence, in the line: per family the clock its kernels ran at (in-kernel stamps of
        # the timed region), the socket power over the timed region, and the matrix rate the package sustains at that limit ----
        for fam, k in kernels.items():
            ghz, n_wg = clocks.get(fam, (None, 0))
            k["clock_GHz"] = ghz
            k["clock_source"] = (f"tg_clock_probe_attach: s_memtime / s_memrealtime at entry and exit of every workgroup of every launch of "
                                 f"the timed region ({n_wg} workgroups)") if ghz else None
            k["power_W"] = power_timed
            if sustained is not None:
                mine = k["achieved"] if k["bound"] == "mfma" else k["matrix_TFLOPs"]
                k["sustained_peak"] = sustained["TFLOPs"]
                k["sustained_peak_unit"] = "TFLOP/s"
                k["frac_of_sustained"] = mine / sustained["TFLOPs"]
                k["sustained_peak_clock_GHz"] = sustained["clock_GHz"]
                k["sustained_peak_power_W"] = sustained["power_W"]["mean"] if sustained["power_W"] else None
        if sustained is not None:
            out["sustained_matrix_rate"] = sustained
            rk = out.get("rollout_kernel")
            if rk and rk.get("bound") == "mfma" and (args.policy_dtype == "fp32") == ("mfma_loop_f32" in sustained["kernel"]):
                # the fused rollout kernel against the same ceiling (its all-alive launch: no idle lanes)
                rk["sustained_peak"] = sustained["TFLOPs"]
                rk["frac_of_sustained"] = rk["achieved"] / sustained["TFLOPs"]
                if rk.get("all_alive"):
                    rk["all_alive"]["frac_of_sustained"] = rk["all_alive"]["achieved_TFLOPs"] / sustained["TFLOPs"]
        if kernels:
            top = max(kernels, key=lambda k: kernels[k]["total_ms_per_step"])
            out["roofline"] = dict(kernels[top], family=top,
                                   why="the hand-written kernel family with the most GPU time in the step")
            out["kernels"] = kernels
        elif "rollout_kernel" in out:
            out["roofline"] = out["rollout_kernel"]
        out["dynamics_kernel"] = dyn
        if cpu is not None:
            out["cpu_baseline"] = cpu
        return out
    return None


if __name__ == "__main__":
    main()
