"""CPU, world_size 2, gloo: the N>1 host logic (group sharding, the single flat gradient all-reduce,
sum-based global statistics).  The per-rank gradients come from the oracle's loss restatement on each
rank's shard; the product's reduction helpers must turn them into exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import trajopt_grpo_amd as tg
from oracle import learner as L

D = tg.distributed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ragged(rng, G, E, T, S, A):
    lens = rng.integers(2, T + 1, size=(G, E))
    mask = (np.arange(T)[None, None] < lens[..., None]).astype(np.float32)
    obs = (rng.normal(size=(G, E, T, S)) * mask[..., None]).astype(np.float32)
    act = (rng.normal(size=(G, E, T, A)) * 0.5 * mask[..., None]).astype(np.float32)
    rew = (rng.normal(size=(G, E, T)) * mask).astype(np.float32)
    return tuple(torch.from_numpy(x) for x in (obs, act, rew, mask))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        G, E, T, S, A = max(4, world), 3, 12, 5, 2
        obs, act, rew, mask = _ragged(np.random.default_rng(0), G, E, T, S, A)
        lo, hi = D.shard_groups(G, rank, world)
        assert (lo, hi) == (rank * G // world, (rank + 1) * G // world) and D.rank_world() == (rank, world)
        sl = slice(lo, hi)
        torch.manual_seed(1)
        pol = L.OraclePolicy(S, A, (16, 16), cov=0.4, critic=True)
        old = L.OraclePolicy(S, A, (16, 16), cov=0.4, critic=True)
        old.load_state_dict(pol.state_dict())
        with torch.no_grad():
            for p in old.parameters():
                p.add_(0.01)
        # ---- GRPO: local J uses 1/G_global (grpo.py:140), then ONE flat all-reduce ---------------------------
        rtg = L.rtg_scan(rew, mask, 0.9)
        bucket = D.GradBucket(pol.parameters())
        bucket.zero_()
        J_local = L.grpo_objective(pol, old, obs[sl], act[sl], rtg[sl], mask[sl], 0.15) * (hi - lo) / G
        J_local.backward()
        bucket.allreduce()
        grpo_grad = bucket.flat.clone()
        # ---- PPO: global moments from all-reduced (count, sum, sumsq), losses scaled by 1/n_global -----------
        with torch.no_grad():
            values = pol.value(obs.reshape(-1, S)).reshape(G, E, T)
        adv_raw = (rtg - values)
        m_loc = mask[sl].bool()
        stats = torch.tensor([[m_loc.sum(), adv_raw[sl][m_loc].double().sum(), (adv_raw[sl][m_loc].double() ** 2).sum()],
                              [m_loc.sum(), rtg[sl][m_loc].double().sum(), (rtg[sl][m_loc].double() ** 2).sum()]],
                             dtype=torch.float64)
        D.allreduce_sum_(stats)
        (a_mean, a_std), (r_mean, r_std) = (D.unbiased_moments(*row.tolist()) for row in stats)
        n_global = float(stats[0, 0])
        o = obs[sl].reshape(-1, S)[m_loc.reshape(-1)]
        a = act[sl].reshape(-1, A)[m_loc.reshape(-1)]
        adv = ((adv_raw[sl][m_loc] - a_mean) / (a_std + 1e-8)).float()
        ret = ((rtg[sl][m_loc] - r_mean) / (r_std + 1e-8)).float()
        with torch.no_grad():
            old_lp, _ = pol.log_prob(o, a)
            old_lp = old_lp + 0.05 * torch.sin(torch.arange(len(old_lp), dtype=torch.float32) + rank)  # ratio != 1
        lp, _ = pol.log_prob(o, a)
        ratio = torch.exp(lp - old_lp)
        total = (-torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).sum()
                 + 0.5 * ((pol.value(o) - ret) ** 2).sum()
                 + 0.5 * (torch.exp(old_lp) * (old_lp - lp)).sum()) / n_global
        bucket.zero_()
        total.backward()
        bucket.allreduce()
        out[rank] = dict(grpo=grpo_grad.numpy(), ppo=bucket.flat.clone().numpy(),
                         moments=np.array([a_mean, a_std, r_mean, r_std, n_global]), old_lp=old_lp.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_two_rank_gradients_equal_single_process(world):
    """(world 4 and 8: one group per rank.  Eight ranks of the PRODUCT path cannot be rehearsed on a one-GPU box -- the pool admits six
    GPU processes -- so this is where shard_groups / GradBucket / the sum-based moments first run at the driver's largest rank count.)"""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = {k: out[k] for k in range(world)}
    # all ranks hold the same reduced gradient
    for key in ("grpo", "ppo", "moments"):
        for r in range(1, world):
            np.testing.assert_array_equal(res[0][key], res[r][key])
    # single-process reference on the full batch
    torch.set_num_threads(1)
    G, E, T, S, A = max(4, world), 3, 12, 5, 2
    obs, act, rew, mask = _ragged(np.random.default_rng(0), G, E, T, S, A)
    torch.manual_seed(1)
    pol = L.OraclePolicy(S, A, (16, 16), cov=0.4, critic=True)
    old = L.OraclePolicy(S, A, (16, 16), cov=0.4, critic=True)
    old.load_state_dict(pol.state_dict())
    with torch.no_grad():
        for p in old.parameters():
            p.add_(0.01)
    rtg = L.rtg_scan(rew, mask, 0.9)
    for p in pol.parameters():
        p.grad = None
    L.grpo_objective(pol, old, obs, act, rtg, mask, 0.15).backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in pol.parameters()])
    np.testing.assert_allclose(res[0]["grpo"], flat.numpy(), rtol=1e-5, atol=1e-6)
    # PPO: global normalisation must equal torch's unbiased mean/std over the whole valid batch (ppo.py:138-139)
    with torch.no_grad():
        values = pol.value(obs.reshape(-1, S)).reshape(G, E, T)
    mb = mask.bool()
    adv_raw, n = (rtg - values)[mb], int(mb.sum())
    np.testing.assert_allclose(res[0]["moments"], [adv_raw.mean(), adv_raw.std(), rtg[mb].mean(), rtg[mb].std(), n],
                               rtol=1e-5)
    adv = (adv_raw - adv_raw.mean()) / (adv_raw.std() + 1e-8)
    ret = (rtg[mb] - rtg[mb].mean()) / (rtg[mb].std() + 1e-8)
    o, a = obs[mb], act[mb]
    old_lp = torch.from_numpy(np.concatenate([res[r]["old_lp"] for r in range(world)]))   # rank order == group order
    for p in pol.parameters():
        p.grad = None
    lp, _ = pol.log_prob(o, a)
    ratio = torch.exp(lp - old_lp)
    total = (-torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean() + 0.5 * ((pol.value(o) - ret) ** 2).mean()
             + 0.5 * (torch.exp(old_lp) * (old_lp - lp)).mean())
    total.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
    np.testing.assert_allclose(res[0]["ppo"], flat.numpy(), rtol=2e-4, atol=2e-6)


def test_shard_groups_and_bucket_single_process():
    assert D.shard_groups(256, 3, 8) == (96, 128) and D.shard_groups(4, 0, 1) == (0, 4)
    with pytest.raises(ValueError):
        D.shard_groups(10, 0, 4)
    with pytest.raises(ValueError):
        D.shard_groups(8, 8, 8)
    lin = torch.nn.Linear(3, 2)
    b = D.GradBucket(lin.parameters())
    lin(torch.ones(4, 3)).sum().backward()
    assert b.flat.numel() == 8 and torch.equal(b.flat[:6].view(2, 3), lin.weight.grad) and float(b.flat[6]) == 4.0
    lin.zero_grad(set_to_none=True)         # what torch optimizers do by default
    b.zero_()
    assert lin.weight.grad is not None and lin.weight.grad.data_ptr() == b.flat.data_ptr()
    b.allreduce()                            # world 1: no-op
    m, s = D.unbiased_moments(4.0, 10.0, 30.0)
    assert m == 2.5 and s == pytest.approx(np.std([1, 2, 3, 4], ddof=1))


def _minibatch_worker(rank, world, port, out):
    """Minibatch PPO's schedule (algorithms/ppo.py:147-157) with UNEQUAL numbers of valid rows per rank: every rank must
    take the same number of optimizer steps (each one is a gradient all-reduce), the rank that runs out of rows joins
    the remaining ones with an empty slice, and each step's gradient must equal the single-process mean over the union
    of the ranks' slices."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        S, A = 5, 2
        m_local = (11, 3)[rank]                                  # rank 1 runs out after 2 of 6 steps
        g = torch.Generator().manual_seed(10 + rank)
        o = torch.randn(m_local, S, generator=g)
        a = torch.randn(m_local, A, generator=g) * 0.5
        adv, ret = torch.randn(m_local, generator=g), torch.randn(m_local, generator=g)
        torch.manual_seed(1)
        pol = L.OraclePolicy(S, A, (16, 16), cov=0.4, critic=True)
        with torch.no_grad():
            old_lp, _ = pol.log_prob(o, a)
            old_lp = old_lp + 0.05 * torch.sin(torch.arange(m_local, dtype=torch.float32) + rank)
        perm = torch.randperm(m_local, generator=g)
        local_bs, n_steps, sizes = D.minibatch_schedule(m_local, 4, None, None)
        assert (local_bs, n_steps, sizes) == (2, 6, [4, 3, 2, 2, 2, 1])
        bucket = D.GradBucket(pol.parameters())
        grads, slices = [], []
        for k in range(n_steps):
            b = perm[k * local_bs:(k + 1) * local_bs]
            bucket.zero_()
            if b.numel():
                lp, _ = pol.log_prob(o[b], a[b])
                ratio = torch.exp(lp - old_lp[b])
                total = (-torch.min(ratio * adv[b], torch.clamp(ratio, 0.8, 1.2) * adv[b]).sum()
                         + 0.5 * ((pol.value(o[b]) - ret[b]) ** 2).sum()
                         + 0.5 * (torch.exp(old_lp[b]) * (old_lp[b] - lp)).sum()) / sizes[k]
                total.backward()
            bucket.allreduce()                                   # EVERY rank, EVERY step (a missing one would hang)
            grads.append(bucket.flat.clone().numpy())
            slices.append(b.numpy())
        out[rank] = dict(grads=grads, slices=slices, o=o.numpy(), a=a.numpy(), adv=adv.numpy(), ret=ret.numpy(),
                         old_lp=old_lp.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_minibatch_schedule_with_unequal_shards():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_minibatch_worker, args=(2, port, out), nprocs=2, join=True)
        res = {k: out[k] for k in (0, 1)}
    torch.set_num_threads(1)
    torch.manual_seed(1)
    pol = L.OraclePolicy(5, 2, (16, 16), cov=0.4, critic=True)
    seen = [np.concatenate(res[r]["slices"]) for r in (0, 1)]
    assert sorted(seen[0].tolist()) == list(range(11)) and sorted(seen[1].tolist()) == list(range(3))   # every row once
    for k in range(6):
        np.testing.assert_array_equal(res[0]["grads"][k], res[1]["grads"][k])
        # single process: the union of both ranks' slices as one minibatch, mean losses (ppo.py:165-179)
        parts = [{n: torch.from_numpy(res[r][n][res[r]["slices"][k]]) for n in ("o", "a", "adv", "ret", "old_lp")} for r in (0, 1)]
        cat = {n: torch.cat([p[n] for p in parts]) for n in parts[0]}
        for p in pol.parameters():
            p.grad = None
        lp, _ = pol.log_prob(cat["o"], cat["a"])
        ratio = torch.exp(lp - cat["old_lp"])
        total = (-torch.min(ratio * cat["adv"], torch.clamp(ratio, 0.8, 1.2) * cat["adv"]).mean()
                 + 0.5 * ((pol.value(cat["o"]) - cat["ret"]) ** 2).mean()
                 + 0.5 * (torch.exp(cat["old_lp"]) * (cat["old_lp"] - lp)).mean())
        total.backward()
        flat = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
        np.testing.assert_allclose(res[0]["grads"][k], flat.numpy(), rtol=2e-4, atol=2e-6)
    # world size 1: the reference's own walk over the permutation
    assert D.minibatch_schedule(10, 4) == (4, 3, [4, 4, 2]) and D.minibatch_schedule(0, 64) == (64, 0, [])
