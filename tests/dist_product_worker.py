"""One rank of the product-path rank-count test (tests/test_distributed_gpu.py; bench.py --check reuses `run_cases`).

Started as a fresh child process: `python dist_product_worker.py RANK WORLD PORT OUT.pt`.  World > 1: gloo process group, all
ranks share cuda:0, each owns a contiguous range of whole groups.  Every case runs rollout -> Rollout_Buffer.sample
-> PPO.learn / GRPO.learn on the REAL kernels and records the local trajectory shard, PPO's global moments and the post-step
weights as CPU tensors."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# name -> (algorithm, env, env kwargs, obs, act, hidden, compute dtype, groups G, episodes E, restart, updates, batch_size)
CASES = {
    # bf16 chain kernels (forward chain + loss head, backward chain, weight gradients), full batch: C3's learner in small
    "ppo_bf16_full": ("ppo", "QuadPole", dict(max_steps=24), 20, 4, (256, 256, 256), torch.bfloat16, 4, 32, False, 2, None),
    # fp32 learner, minibatches of 64 over EQUAL row counts per rank (no env ends in 8 steps): comparable with one rank
    "ppo_f32_minibatch_equal": ("ppo", "QuadPole", dict(max_steps=8), 20, 4, (64, 64), None, 4, 8, False, 1, 64),
    # ... and over UNEQUAL row counts (an untrained QuadPole policy leaves the +-1.5 m box after ~90 steps, each env at its own
    # time): ranks must agree with each other and take the same number of steps
    "ppo_f32_minibatch_ragged": ("ppo", "QuadPole", dict(max_steps=160), 20, 4, (64, 64), None, 4, 8, False, 1, 64),
    "grpo_bf16": ("grpo", "QuadPole", dict(max_steps=24), 20, 4, (128, 128, 128), torch.bfloat16, 4, 32, True, 2, None),
    # the reference's own precision and factory shape (cartpole_pipeline_grpo.py:54-76 in small)
    "grpo_f32": ("grpo", "CartPole", dict(max_steps=32), 5, 1, (128, 128), None, 4, 16, False, 2, None),
}


def local_permutation(m_local, rank, world, device):
    """This rank's minibatch permutation: a function of (rank, world, row count) only, so that a one-rank run can rebuild the
    union of the ranks' slices (test_distributed_gpu.py)."""
    g = torch.Generator().manual_seed(1000 + 17 * rank + world)
    return torch.randperm(m_local, generator=g).to(device)


def union_permutation(m_global, T, n_envs, emulate_world, batch_size, device):
    """The one-rank permutation that reproduces a `emulate_world`-rank minibatch run when every row is valid: step k of the
    global batch = the ranks' local slices k side by side.  Rows are time-major: global row t * n_envs + n; rank r owns the envs
    [r * n_envs / W, (r + 1) * n_envs / W), its local row i = (t, n_local) = divmod(i, n_envs / W)."""
    assert m_global == T * n_envs, "every env must run the whole horizon for this emulation"
    per = n_envs // emulate_world
    h = -(-batch_size // emulate_world)
    perms = [local_permutation(T * per, r, emulate_world, "cpu") for r in range(emulate_world)]
    steps = -(-(T * per) // h)
    out = []
    for k in range(steps):
        for r, pr in enumerate(perms):
            i = pr[k * h:(k + 1) * h]
            out.append((i // per) * n_envs + r * per + (i % per))
    return torch.cat(out).to(device)


def run_cases(names, rank, world, group=None, device=None, emulate_world=2, groups=None):
    """group=None: the default process group (or no group at all); a one-rank subgroup runs the cases alone inside a larger job.
    groups: run every case with this many groups instead of its own 4 (bench.py --check on 8 ranks: one group per rank)."""
    import trajopt_grpo_amd as tg
    dev = device if device is not None else torch.device("cuda", 0)
    out = {}
    for name in names:
        algo_name, env_name, env_kw, S, A, hidden, cdt, G, E, restart, updates, bs = CASES[name]
        G = G if groups is None else int(groups)
        torch.manual_seed(1234)                                   # identical initial weights on every rank
        cls = tg.GaussianActorCritic_NeuralNetwork if algo_name == "ppo" else tg.GaussianActor_NeuralNetwork
        pol = cls(S, A, hidden, cov=0.3, device=dev)
        env_cls = getattr(tg, env_name)
        mgr = tg.RolloutManager(lambda: env_cls(**env_kw), pol, restart=restart, num_workers=G, num_episodes_per_worker=E, seed=7,
                                compute_dtype=cdt, process_group=group)
        buf = tg.Rollout_Buffer(mgr)
        buf.sample()
        tr = buf.device_traj
        rec = {"obs": tr.obs.cpu(), "act": tr.act.cpu(), "rew": tr.rew.cpu(), "mask": tr.mask.cpu(), "len": tr.len.cpu(),
               "groups": (mgr.group_lo, mgr.group_hi), "avg_reward": float(buf.avg_reward[-1])}
        opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
        steps = {"n": 0}
        plain = opt.step

        def counted(*a, **k):
            steps["n"] += 1
            return plain(*a, **k)

        opt.step = counted
        if algo_name == "ppo":
            algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=updates, gamma=0.99, batch_size=bs,
                          autocast_dtype=cdt, process_group=group)
            if bs is not None:
                if world == 1 and name.endswith("_equal"):        # one rank walking the union of a two-rank run's minibatches
                    algo.permutation_fn = lambda m, d, T=env_kw["max_steps"], n=G * E, b=bs: union_permutation(m, T, n, emulate_world, b, d)
                else:
                    algo.permutation_fn = lambda m, d, r=rank, w=world: local_permutation(m, r, w, d)
        else:
            algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=opt, updates_per_iter=updates, autocast_dtype=cdt,
                           process_group=group)
        w0 = [p.detach().clone() for p in pol.parameters()]
        algo.learn(buf)
        torch.cuda.synchronize()
        rec["weights"] = [p.detach().cpu() for p in pol.parameters()]
        rec["delta"] = [(p.detach() - q).cpu() for p, q in zip(pol.parameters(), w0)]
        rec["optimizer_steps"] = steps["n"]
        rec["stats"] = {k: v for k, v in algo.last_stats.items()}
        if algo_name == "ppo":
            rec["moments"] = algo.norm8[:4].tolist()        # (the global normalisation constants, as the loss heads read them)
        rec["n_valid_local"] = int(tr.mask.sum().item())
        m = algo._mlp(pol.actor)
        rec["learner_path"] = ("chain" if (m is not None and m._chain is not None and m._bchain is not None) else
                               ("f32chain" if (m is not None and getattr(m, "_f32", None) is not None) else
                                ("gemm" if m is not None else "autograd")))
        out[name] = rec
    return out


def _rel(a, b):
    return float((a.double() - b.double()).norm()) / (float(b.double().norm()) + 1e-30)


def check_case(one, two, case):
    """Assert that the two ranks' records `two` equal the one-rank record `one` (see test_distributed_gpu.py); returns the
    largest relative weight / update differences found.  `two` may hold only this rank's record (bench.py --check)."""
    assert all(rec["learner_path"] == one["learner_path"] for rec in two)
    if "bf16" in case:
        assert one["learner_path"] == "chain"                       # the hot kernels are what is being compared
    # ---- rollout: each rank's shard is bit-for-bit the one-rank trajectory of its groups (Philox keyed by global indices) ----
    n_total = one["len"].numel()
    for rec in two:
        lo, hi = rec["groups"]
        per_group = n_total // one["groups"][1]
        sl = slice(lo * per_group, hi * per_group)
        assert torch.equal(rec["len"], one["len"][sl]) and torch.equal(rec["mask"], one["mask"][:, sl])
        assert torch.equal(rec["obs"], one["obs"][:, :, sl]) and torch.equal(rec["act"], one["act"][:, :, sl])
        assert torch.equal(rec["rew"], one["rew"][:, sl])
        assert abs(rec["avg_reward"] - one["avg_reward"]) <= 1e-6 * abs(one["avg_reward"])
    if len(two) > 1:
        assert sum(rec["n_valid_local"] for rec in two) == one["n_valid_local"]
        if case.endswith("ragged"):
            assert two[0]["n_valid_local"] != two[1]["n_valid_local"], "the case is meant to have unequal row counts"
        # ---- all ranks hold bit-identical weights after learn(), and took the same number of optimizer steps ----
        for rec in two[1:]:
            assert rec["optimizer_steps"] == two[0]["optimizer_steps"] and rec["avg_reward"] == two[0]["avg_reward"]
            for a, b in zip(two[0]["weights"], rec["weights"]):
                assert torch.equal(a, b)
    assert two[0]["optimizer_steps"] > 0 and all(torch.isfinite(a).all() for a in two[0]["weights"])
    if "moments" in one:                                            # PPO: global advantage / return moments (ppo.py:138-139)
        for rec in two:
            for x, y in zip(rec["moments"], one["moments"]):
                assert abs(x - y) <= 1e-6 * (abs(y) + 1e-6), (rec["moments"], one["moments"])
    if case.endswith("ragged"):
        # unequal row counts: a one-rank run walks different minibatches (the permutation is rank-local); what must
        # hold is the schedule -- ceil(max rows / ceil(64 / world)) steps on every rank
        if len(two) > 1:
            local_bs = -(-64 // len(two))
            assert two[0]["optimizer_steps"] == max(-(-rec["n_valid_local"] // local_bs) for rec in two)
        return {"weights": None, "update": None}
    # ---- ... and they equal the one-rank weights up to the all-reduce's summation order ----
    assert two[0]["optimizer_steps"] == one["optimizer_steps"]
    # 1e-6 relative (L2) on every weight tensor; on the UPDATE itself (weights after - before) 1e-4 for the fp32 learner and 1e-3
    # for the bf16 chain kernels, whose second update re-rounds fp32 masters that may differ in the last bit
    # (measured: weights 4e-9..7e-8, updates 6e-7..3e-6)
    w_tol, d_tol = (1e-6, 1e-3) if "bf16" in case else (1e-6, 1e-4)
    worst_w = worst_d = 0.0
    for a, b, da, db in zip(two[0]["weights"], one["weights"], two[0]["delta"], one["delta"]):
        worst_w = max(worst_w, _rel(a, b))
        assert _rel(a, b) <= w_tol, (_rel(a, b), a.shape)
        if float(db.norm()) > 0:
            worst_d = max(worst_d, _rel(da, db))
            assert _rel(da, db) <= d_tol, (_rel(da, db), a.shape)
    for k in ("J", "total_loss", "actor_loss", "critic_loss"):
        if k in one["stats"]:
            for x, y in zip(two[0]["stats"][k], one["stats"][k]):
                assert abs(x - y) <= 2e-3 * (abs(y) + 1e-3), (k, x, y)
    return {"weights": worst_w, "update": worst_d}


def main():
    rank, world, port, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    names = sys.argv[5].split(",") if len(sys.argv) > 5 else list(CASES)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = run_cases(names, rank, world)
        torch.save(res, path)
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
