#!/usr/bin/env python3
"""Benchmark of the hot path: GPU-resident rollout + PPO update on QuadPole ("quadrotor_env.py").

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: a 65,536-env x 256-step QuadPole rollout (one persistent
kernel: actor MLP on the matrix cores + sampling + dynamics + recording) followed by PPO.learn on that buffer with
the reference factory's hyper-parameters (pipelines/quadpole_pipeline_ppo.py: 20-256x5-{4,1} actor-critic, cov 0.3,
gamma 0.999, 32 full-batch updates, Adam 3e-4), i.e. BASELINE.json configs[2] (C3).  Metric: env-steps/s, where an
env-step is one valid (mask == 1) Env.step.  Weak scaling: every rank runs 65,536 envs; gradients are all-reduced
once per optimizer step (RCCL).

Besides the contract fields, the JSON line carries
  roofline         the kernel with the most GPU time in the step (29 %): tg_mlp_backward_chain, the backward-data pass of
                   all hidden layers in one launch.  HBM-bound (writes); algorithmic bytes per row = 16 (dOut) +
                   5 x 32 (1-bit ReLU masks) + 5 x 512 (the dZ it must write for the weight gradients) = 2736 B at
                   20-256x5-4; EVERY launch of the timed steps is bracketed by HIP events on the launch stream.
                   (Shapes without that kernel report tg_dx_relu_bias; with an fp32 policy `roofline` is the rollout
                   kernel's.)
  rollout_kernel   the fused rollout kernel against the MFMA roofline: 2 x actor parameters flop per valid env-step
                   (539 kflop), HIP events around each rollout's launch; `all_alive` = the same kernel with
                   nobody terminating.  With --no-fused: the per-step dynamics kernel against the HBM roofline (189 B
                   per env-step, SURVEY 8d state-in-trajectory variant).
  dynamics_kernel  the stand-alone dynamics kernel (tg_rollout_step) at this env count, HBM roofline, timed after the run;
  learner_kernel   tg_dx_relu_bias (the per-layer form of the backward-data pass) on random data at 2^22 rows, timed after
                   the run;
  cpu_baseline     the CPU port of the reference path (oracle/: scalar fp64 env + batch-1 torch policy per step in
                   forked worker processes, then PPO.learn on CPU) timed on this box's host cores over a bounded
                   sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HIDDEN = (256, 256, 256, 256, 256)
ALGO_BYTES = {"CartPole": 57, "QuadPole2D": 101, "QuadPole": 189}    # SURVEY 8(d), compact variant
BWD_PMC_BYTES_PER_ROW = 11481697627 / 4194304                        # profiles/r01_bwd_chain_probe_pmc.json
HBM_PEAK_GBS = 8000.0                                                 # MI355X_MICROARCH.md: 8.0 TB/s spec


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
    pol = L.OraclePolicy(20, 4, HIDDEN, cov=0.3, critic=True)
    pol.load_state_dict(sd)
    env = L.OracleEnv("QuadPole", max_steps=T, rng=np.random.default_rng(1000 + wid))
    with torch.no_grad():
        return L.run_episodes(env, pol, episodes, restart=False)


def cpu_baseline(T, updates, budget_workers=None, episodes=32):
    import multiprocessing as mp
    import torch
    from oracle import learner as L
    cores = len(os.sched_getaffinity(0))
    workers = budget_workers or max(1, min(cores, 16))
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
    mean = torch.zeros(n, 8, device=dev)
    best = None
    for _ in range(3):
        eng._enqueue_prepare(None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(launches):
            N_.check(lib.tg_rollout_step(p, C.byref(tr), t, mean.data_ptr(), 8, eng._sigma, eng.rng.data_ptr(), 0, st))
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
    return {"kernel": "tg::rollout_step_kernel<QuadPoleEnv<float>,float,true>", "bound": "hbm", "n_envs": n,
            "us_per_launch": best, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_env_step": ALGO_BYTES["QuadPole"], "same_bytes_device_copy_us": copy_us,
            "traffic": 14334976 if n == 65536 else None,
            "traffic_source": "profiles/r01_step_kernel_65536_pmc.json (2 x FETCH_SIZE + WRITE_SIZE per launch)"}


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU (groups of 256)")
    ap.add_argument("--horizon", type=int, default=256)
    ap.add_argument("--updates", type=int, default=32, help="PPO updates_per_iter (reference factory: 32)")
    ap.add_argument("--policy-dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the T-step rollout loop as one hipGraph")
    ap.add_argument("--no-fused", action="store_true",
                    help="per-step launches (actor GEMMs + tg_rollout_step) instead of the fused persistent rollout kernel")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="do not bracket the rollout / backward-chain launches with HIP events (A/B of the measurement's own cost; no roofline objects)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only for rehearsing the multi-rank path on a single GPU (ranks share the device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.horizon, args.updates)

    import torch
    import torch.distributed as dist
    import trajopt_grpo_amd as tg

    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPUs visible")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    E = 256
    G_local = args.envs // E
    G_global = G_local * world
    T = args.horizon
    cdt = torch.bfloat16 if args.policy_dtype == "bf16" else None
    torch.manual_seed(0)                                      # identical random-init weights on every rank
    policy = tg.GaussianActorCritic_NeuralNetwork(20, 4, HIDDEN, cov=0.3, device=dev)
    mk = lambda: tg.QuadPole(max_steps=T)
    mgr = tg.RolloutManager(mk, policy, num_workers=G_global, num_episodes_per_worker=E, dtype=torch.float32,
                            seed=1234, compute_dtype=cdt, use_graph=bool(args.graph), fused=False if args.no_fused else None)
    buf = tg.Rollout_Buffer(mgr)
    algo = tg.PPO(epsilon=0.2, policy=policy, optimizer=torch.optim.Adam(policy.parameters(), lr=3e-4), ref_model=None,
                  updates_per_iter=args.updates, c1=0.5, kl_coeff=0.5, gamma=0.999, lam=0.95, entropy=0.01,
                  batch_size=None, autocast_dtype=cdt)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        buf.sample()
        algo.learn(buf)

    for _ in range(args.warmup):
        one_step()
    # event-pair overhead (no kernel in between), for the per-launch timing below
    pairs = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ev_overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]

    env_steps = 0
    t_roll = 0.0
    launches = []            # (duration ms, env-steps in that launch)
    launch_units = []
    if not args.graph and not args.no_launch_events:
        mgr.engine.step_events = []
    learner_mlps = [m for m in (algo._mlp(policy.actor), algo._mlp(policy.critic)) if m is not None]
    for m in (learner_mlps if not args.no_launch_events else []):
        m.dx_events = []                    # HIP-event pairs around every tg_dx_relu_bias launch of the timed steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r0 = time.perf_counter()
        buf.sample()                        # ends with a host read of avg_reward -> rollout is complete here
        t_roll += time.perf_counter() - r0
        env_steps += buf.device_traj.env_steps()
        if mgr.engine.step_events:
            if mgr.engine.fused:
                launches += [(a.elapsed_time(b), buf.device_traj.env_steps()) for _, a, b in mgr.engine.step_events]
                launch_units.append(buf.device_traj.env_steps())
            else:
                alive = buf.device_traj.mask.sum(1, dtype=torch.int64).tolist()
                launches += [(a.elapsed_time(b), alive[t]) for t, a, b in mgr.engine.step_events]
            mgr.engine.step_events = []
        algo.learn(buf)
    barrier()
    dt = time.perf_counter() - t0

    dx_launches = []                        # (ms, algorithmic bytes, rows) per launch, the update's dominant kernel
    for m in learner_mlps:
        dx_launches += [(a.elapsed_time(b), rows * bpr, rows, name) for a, b, rows, bpr, name in (m.dx_events or [])]
        m.dx_events = None
    dyn = dynamics_kernel_probe(tg, dev, args.envs) if rank == 0 else None
    relu_probe = None
    if rank == 0 and cdt is not None:
        # the hand-written kernel with the most GPU time in the update: a hidden layer's backward-data product fused
        # with the ReLU backward + bias gradient below it (tg_dx_relu_bias), on random data at the learner's chunk size.
        # Algorithmic traffic as the learner runs it: read dZ (512 B) and the 1-bit ReLU masks (32 B), write dZ_below
        # (512 B) per row at 256 bf16 features.
        N_ = tg._native
        lib_ = N_.load()
        rows, cols = 1 << 22, 256
        dZ = (torch.randn(rows, cols, device=dev) * 0.5).to(cdt)
        bits_ = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, cols // 32), dtype=torch.int32, device=dev)
        W_ = (torch.randn(cols, cols, device=dev) / 16).to(cdt)
        frag = torch.empty(cols * cols, dtype=cdt, device=dev)
        out_ = torch.empty_like(dZ)
        part = torch.empty(lib_.tg_dx_relu_bias_blocks(), cols, dtype=torch.float32, device=dev)
        st = N_.stream_ptr(dev)
        N_.check(lib_.tg_dx_pack_weights(W_.data_ptr(), frag.data_ptr(), cols, cols, st))
        run = lambda: N_.check(lib_.tg_dx_relu_bias(dZ.data_ptr(), frag.data_ptr(), None, bits_.data_ptr(), out_.data_ptr(), rows, cols,
                                                    cols, part.data_ptr(), st))
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 10
        bpr = 2 * cols * 2 + cols // 8
        gbs = float(bpr) * rows / us / 1e3
        relu_probe = {"kernel": "tg::dx_relu_bias_kernel<256,256,1,8,bits>", "bound": "hbm", "rows": rows, "cols": cols,
                      "us_per_launch": us, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                      "bytes_per_row": bpr, "TFLOPs": 2.0 * rows * cols * cols / us / 1e6}
        del dZ, bits_, W_, frag, out_, part
    fused_all_alive = None
    if rank == 0 and mgr.engine.fused:
        # the fused kernel with nobody terminating (bounds opened): its matrix-core rate without idle lanes
        env_open = tg.QuadPole(max_steps=T)
        env_open.spatial_bounds = tuple((-1e9, 1e9) for _ in env_open.spatial_bounds)
        eng = tg.DeviceRollout(env_open, policy, G_local, E, seed=7, compute_dtype=cdt, fused=True)
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
    tot = torch.tensor([float(env_steps), dt, t_roll], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_steps, dt, t_roll = float(tot[0]), float(mx[1]), float(mx[2])
    else:
        total_steps = float(env_steps)

    if rank == 0:
        out = {
            "metric": "env-steps/sec at 65k parallel quadrotor envs (rollout + PPO update)",
            "value": total_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "policy_dtype": args.policy_dtype, "data": "synthetic",
            "config": {"workload": f"C3: QuadPole (quadrotor_env.py) PPO, {args.envs} envs/GPU x {T}-step horizon, "
                                   f"natural termination, actor-critic 20-256x5-{{4,1}}, {args.updates} full-batch "
                                   f"updates/iter, {args.policy_dtype} policy",
                       "envs_per_gpu": args.envs, "horizon": T, "updates_per_iter": args.updates,
                       "parallelism": f"env-shard x{world}, 1 grad all-reduce per optimizer step"},
            "rollout_only_env_steps_per_s": total_steps / t_roll if t_roll > 0 else None,
            "rollout_ms": 1e3 * t_roll / args.steps,
            "env_steps_per_step": total_steps / args.steps,
        }
        fused = mgr.engine.fused
        out["rollout_path"] = "fused persistent kernel (tg_fused_rollout)" if fused else "per-step launches (GEMMs + tg_rollout_step)"
        if launches and fused:
            # one launch per rollout: actor MLP on the matrix cores + sampling + dynamics + recording.
            # algorithmic flops = 2 * actor parameters per valid env-step (539 kflop for 20-256x5-4)
            n_par = sum(p.numel() for p in policy.actor.parameters())
            dur = sum(d for d, _ in launches) * 1e-3
            ach = 2.0 * n_par * sum(launch_units) / dur / 1e12
            out["rollout_kernel"] = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                               "traffic": 1706642656 if args.envs == 65536 else None,
                               "traffic_source": "profiles/r01_fused_rollout_pmc.json (all-alive launch: 2 x FETCH_SIZE + "
                                                 "WRITE_SIZE = the 101 B/env-step trajectory record; weights stay in L2)",
                               "kernel": "tg::fused_rollout_kernel<QuadPoleEnv<float>,256,1,8>",
                               "flops_per_env_step": 2 * n_par, "launches": len(launches),
                               "avg_launch_ms": 1e3 * dur / len(launches),
                               "note": "valid env-steps only (natural termination: ended envs idle their lanes)",
                               "all_alive": fused_all_alive}
        elif launches:
            dur = sum(max(d - ev_overhead_ms, 1e-4) for d, _ in launches) * 1e-3
            units = sum(u for _, u in launches)
            full = [(d, u) for d, u in launches if u == args.envs]
            ach = ALGO_BYTES["QuadPole"] * units / dur / 1e9
            traffic = 14334976 if args.envs == 65536 else None
            out["rollout_kernel"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": "profiles/r01_step_kernel_65536_pmc.json (all-alive launch)",
                               "kernel": "tg::rollout_step_kernel<QuadPoleEnv<float>,float,true>",
                               "bytes_per_env_step": ALGO_BYTES["QuadPole"], "launches": len(launches),
                               "avg_launch_us": 1e6 * dur / len(launches),
                               "event_pair_overhead_us": 1e3 * ev_overhead_ms}
            if full:
                d_full = sum(max(d - ev_overhead_ms, 1e-4) for d, _ in full) * 1e-3 / len(full)
                out["rollout_kernel"]["full_launch_us"] = 1e6 * d_full
                out["rollout_kernel"]["full_launch_GBs"] = ALGO_BYTES["QuadPole"] * args.envs / d_full / 1e9
        if dx_launches:
            # the kernel with the most GPU time in the step (29 %, profiles/r01_learner_bench_kernel_stats.csv): the
            # backward-data pass (tg_mlp_backward_chain; per-layer tg_dx_relu_bias for shapes without it).  Algorithmic
            # bytes per row come with each event record (mlp.GemmMLP.dx_events).
            dur = sum(d for d, _, _, _ in dx_launches) * 1e-3
            nbytes = sum(b for _, b, _, _ in dx_launches)
            nrows = sum(r for _, _, r, _ in dx_launches)
            ach = nbytes / dur / 1e9
            # PMC traffic of the 2^22-row probe launch (profiles/r01_dx_kernel_probe_pmc.json), scaled to the average launch
            traffic = BWD_PMC_BYTES_PER_ROW * nrows / len(dx_launches) if "bwd_chain" in dx_launches[0][3] else None
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic,
                               "traffic_source": "profiles/r01_bwd_chain_probe_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE = 2737 B/row "
                                                 "(1.001 x algorithmic), times this run's average rows per launch",
                               "kernel": dx_launches[0][3], "bytes_per_row": nbytes / nrows,
                               "launches": len(dx_launches), "avg_launch_ms": 1e3 * dur / len(dx_launches),
                               "avg_rows_per_launch": nrows / len(dx_launches),
                               "TFLOPs": 2.0 * 256 * 256 * nrows / dur / 1e12,
                               "note": "dominant kernel by GPU time; every launch of the timed steps, HIP events on the launch stream"}
        elif "rollout_kernel" in out:
            out["roofline"] = out["rollout_kernel"]
        out["dynamics_kernel"] = dyn
        out["learner_kernel"] = relu_probe
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
