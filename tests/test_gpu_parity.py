"""GPU parity tests (run on an MI355X with `-m gpu`): the HIP path, called through the C ABI,
against the CPU oracle on seeded inputs and against the golden fixtures generated from the
real reference.  Tolerances: bit-exact for indices / lengths / masks / flags; fp64 kernels
<= 1e-11 on states and rewards; fp32 kernels within a few fp32 ulp per step (stated inline);
fp32 returns within 1e-5 (the north-star bar)."""
import ctypes as C

import copy
import os

import numpy as np
import pytest
import torch

from conftest import check_pinned, keep_first_step_gradients, load_golden
from oracle import envs as E
from oracle import learner as L

pytestmark = pytest.mark.gpu

ENVS = ["CartPole", "QuadPole2D", "QuadPole"]
DIMS = {"CartPole": (5, 1), "QuadPole2D": (10, 2), "QuadPole": (20, 4), "Pendulum": (3, 1)}


@pytest.fixture(scope="module")
def tg():
    import trajopt_grpo_amd as tg
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return tg


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def native_step(tg, name, state, action, steps, tb, max_steps, dtype, dev, **ctor):
    """tg_env_step through the C ABI on SoA device arrays."""
    Nn = tg._native
    env = tg.environments.ENV_CLASSES[name](max_steps=max_steps, **ctor)
    p = env.native_params()
    n = state.shape[0]
    st = torch.as_tensor(np.ascontiguousarray(state.T), dtype=dtype, device=dev)
    ac = torch.as_tensor(np.ascontiguousarray(action.T), dtype=torch.float32, device=dev)
    nx = torch.empty_like(st)
    sp = torch.as_tensor(np.asarray(steps), dtype=torch.int32, device=dev)
    tbt = torch.as_tensor(np.asarray(tb), dtype=dtype, device=dev)
    rw = torch.empty(n, dtype=dtype, device=dev)
    tr = torch.empty(n, dtype=torch.uint8, device=dev)
    Nn.check(Nn.load().tg_env_step(C.byref(p), Nn.dtype_code(dtype), st.data_ptr(), n, ac.data_ptr(), n, nx.data_ptr(), n,
                                   sp.data_ptr(), tbt.data_ptr(), rw.data_ptr(), tr.data_ptr(), n, Nn.stream_ptr(dev)))
    torch.cuda.synchronize()
    return (nx.cpu().numpy().T.astype(np.float64), rw.cpu().numpy().astype(np.float64), tr.cpu().numpy().astype(bool),
            sp.cpu().numpy(), tbt.cpu().numpy().astype(np.float64))


# --------------------------------------------------------------------------------------------
# single-step maps
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ENVS)
def test_step_fp64_matches_reference_golden(tg, dev, name):
    g = load_golden(f"env_step_{name.lower()}.npz")
    nx, rw, tr, sp, tb = native_step(tg, name, g["state"], g["action"], g["steps"], g["time_balanced"],
                                     int(g["max_steps"]), torch.float64, dev)
    # measured: <= 4.4e-16 on the states (1 ulp; 96 % of the entries bit-identical), <= 7.1e-15 on the rewards
    np.testing.assert_allclose(nx, g["next_state"], rtol=1e-14, atol=2e-15)
    np.testing.assert_allclose(rw, g["reward"], rtol=1e-14, atol=5e-14)
    assert np.array_equal(tr, g["truncated"])
    assert np.array_equal(sp, g["steps"] + 1)
    if name != "QuadPole":
        np.testing.assert_allclose(tb, g["time_balanced_after"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag,name", [("cartpole_custom", "CartPole"), ("quadpole2d_custom", "QuadPole2D")])
def test_step_with_non_default_constructor_arguments(tg, dev, tag, name):
    g = load_golden(f"env_step_{tag}.npz")
    kw = {k[len("param_"):]: float(g[k]) for k in g if k.startswith("param_")}
    nx, rw, tr, sp, _ = native_step(tg, name, g["state"], g["action"], g["steps"], g["time_balanced"],
                                    int(g["max_steps"]), torch.float64, dev, **kw)
    np.testing.assert_allclose(nx, g["next_state"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(rw, g["reward"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(tr, g["truncated"])


@pytest.mark.parametrize("name", ENVS)
def test_step_fp32_matches_reference_golden(tg, dev, name):
    g = load_golden(f"env_step_{name.lower()}.npz")
    nx, rw, tr, sp, tb = native_step(tg, name, g["state"], g["action"], g["steps"], g["time_balanced"],
                                     int(g["max_steps"]), torch.float32, dev)
    # one step in fp32: inputs are rounded to fp32 (6e-8 relative) and |state| <= ~20
    np.testing.assert_allclose(nx, g["next_state"], rtol=2e-6, atol=5e-6)
    scale = np.maximum(1.0, np.abs(g["reward"]))
    assert np.all(np.abs(rw - g["reward"]) <= 2e-5 * scale)
    # flags may only differ where the fp64 position sits within fp32 rounding of a threshold
    diff = tr != g["truncated"]
    if diff.any():
        pos = np.abs(g["next_state"][diff][:, :3 if name == "QuadPole" else 2 if name == "QuadPole2D" else 1])
        bound = {"CartPole": 1.0, "QuadPole2D": 2.0, "QuadPole": 1.5}[name]
        assert np.all(np.min(np.abs(pos - bound), axis=1) < 1e-5)


@pytest.mark.parametrize("name", ENVS)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_step_matches_oracle_on_seeded_batch(tg, dev, name, dtype, tol):
    rng = np.random.default_rng(123)
    S, A = DIMS[name]
    n = 5000                                       # ragged vs the 64-lane wavefront on purpose
    st = rng.normal(size=(n, S)) * 0.5
    if name == "QuadPole":
        for sl in (slice(6, 10), slice(13, 17)):
            st[:, sl] /= np.linalg.norm(st[:, sl], axis=1, keepdims=True)
    else:
        for i0 in ((2,) if name == "CartPole" else (4, 7)):
            ang = rng.uniform(-np.pi, np.pi, n)
            st[:, i0], st[:, i0 + 1] = np.sin(ang), np.cos(ang)
    if dtype == torch.float32:
        st = st.astype(np.float32).astype(np.float64)
    act = (rng.normal(size=(n, A)) * 0.8).astype(np.float32)
    steps = rng.integers(0, 99, n)
    tb0 = rng.choice([0.0, 0.02], n)
    ref = E.ENV_SPECS[name]["step"](st, act, steps, tb0, max_steps=100)
    nx, rw, tr, sp, tb = native_step(tg, name, st, act, steps, tb0, 100, dtype, dev)
    np.testing.assert_allclose(nx, ref[0], rtol=tol, atol=tol)
    np.testing.assert_allclose(rw, ref[1], rtol=tol, atol=tol * 50)
    if dtype == torch.float64:
        assert np.array_equal(tr, ref[2])
    else:
        assert (tr != ref[2]).mean() < 1e-3


def test_quadrotor12_dynamics(tg, dev):
    g = load_golden("quadrotor_dynamics.npz")
    q = tg.Quadrotor(device=dev)
    np.testing.assert_allclose(q._dynamics(g["state"], g["control"]), g["next_state"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(q._dynamics(g["state"][0], g["control"][0]), g["next_state"][0], rtol=1e-11, atol=1e-12)


def test_scalar_env_dropin_api(tg, dev):
    """reset / restart / step of one env instance: the reference's return conventions."""
    env = tg.CartPole(max_steps=10, device=dev)
    obs, info = env.reset()
    assert obs.shape == (5,) and obs.dtype == np.float64 and info == {"time_balanced": 0}
    assert obs[0] == 0 and obs[1] == 0 and obs[4] == 0 and abs(obs[2] ** 2 + obs[3] ** 2 - 1) < 1e-12
    env.set_state([0, 0, 0, 1, 0.0])
    o, r, term, trunc, info = env.step(np.zeros(1, np.float32))      # reference tests/test_cartpole.py:91-104
    assert 0 < r < 5 and r == pytest.approx(2.8, abs=1e-12) and term is False and trunc is False
    assert env._time_balanced == pytest.approx(0.02) and info["time_balanced"] == 0   # info is pre-update
    o0, _ = env.restart()
    np.testing.assert_array_equal(o0, [0, 0, 0, 1, 0])
    for k in range(10):
        o, r, term, trunc, _ = env.step(np.zeros(1, np.float32))
    assert env._steps == 10
    np.testing.assert_array_equal(env.state_dict["cartpole"], o)


# --------------------------------------------------------------------------------------------
# teacher-forced rollouts against the reference's in-process rollout
# --------------------------------------------------------------------------------------------
ROLL = [("CartPole", 128), ("QuadPole2D", 128), ("QuadPole", 256)]


@pytest.mark.parametrize("name,T", ROLL)
@pytest.mark.parametrize("tag", ["reset", "restart"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_teacher_forced_rollout_matches_reference(tg, dev, name, T, tag, dtype):
    g = load_golden(f"rollout_{name.lower()}.npz")
    obs, act, rew, ln, mask = (g[f"{tag}_{k}"] for k in ("obs", "act", "rew", "len", "mask"))
    G, Eps = ln.shape
    S, A = DIMS[name]
    pol = tg.GaussianActor_NeuralNetwork(S, A, (8,), cov=0.5, device=dev)      # unused: actions are forced
    mgr = tg.RolloutManager(lambda: tg.environments.ENV_CLASSES[name](max_steps=T), pol, restart=(tag == "restart"),
                            num_workers=G, num_episodes_per_worker=Eps, dtype=dtype)
    init = obs[:, :, 0, :].reshape(G * Eps, S)
    o2, a2, r2, l2, m2 = mgr.rollout(initial_states=init, forced_actions=act)
    assert o2.shape == obs.shape and a2.shape == act.shape and r2.shape == rew.shape       # shape contract
    assert o2.dtype == torch.float32 and l2.dtype == torch.float32 and m2.dtype == torch.float32
    assert np.array_equal(l2.numpy(), ln), "episode lengths must be bit-exact"
    assert np.array_equal(m2.numpy(), mask), "masks must be bit-exact"
    assert np.array_equal(a2.numpy(), act)
    assert np.all(o2.numpy()[mask == 0] == 0) and np.all(r2.numpy()[mask == 0] == 0)       # zero padding
    if dtype == torch.float64:
        # the fixture's initial state went through float32; the fp64 kernel then tracks the fp64 reference
        np.testing.assert_allclose(o2.numpy(), obs, rtol=0, atol=5e-4)
        np.testing.assert_allclose(r2.numpy(), rew, rtol=2e-4, atol=2e-4)
    else:
        # fp32 state over up to 256 chaotic steps: compare where the episode is young, bound the rest
        early = np.zeros_like(mask, dtype=bool)
        early[:, :, :16] = True
        np.testing.assert_allclose(o2.numpy()[early], obs[early], rtol=0, atol=2e-4)
        np.testing.assert_allclose(r2.numpy()[early[..., ]], rew[early], rtol=1e-3, atol=1e-3)
    assert mgr.engine.traj.env_steps() == int(mask.sum())
    assert list(mgr.episodes_completed) == [Eps] * G


def test_rollout_worker_dropin(tg, dev):
    g = load_golden("rollout_cartpole.npz")
    obs, act, ln, mask = g["reset_obs"], g["reset_act"], g["reset_len"], g["reset_mask"]
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (8,), cov=0.5, device=dev)
    done = [0, 0]
    w = tg.RolloutWorker(1, tg.CartPole(max_steps=128), pol, done, dtype=torch.float64)
    o, a, r, l, m = w.run_episodes(2, initial_states=obs[1, :, 0, :], forced_actions=act[1])
    assert o.shape == (2, 128, 5) and l.dtype == torch.int32 and done == [0, 2]
    assert np.array_equal(l.numpy(), ln[1].astype(np.int32)) and np.array_equal(m.numpy(), mask[1])


# --------------------------------------------------------------------------------------------
# sampled rollouts: invariants, determinism, independence of sharding
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ENVS)
def test_sampled_rollout_invariants_and_oracle_replay(tg, dev, name):
    S, A = DIMS[name]
    T, G, Eps = 64, 3, 70                 # n = 210: not a multiple of the wavefront
    torch.manual_seed(1)
    pol = tg.GaussianActor_NeuralNetwork(S, A, (32, 32), cov=0.4, device=dev)
    mk = lambda: tg.environments.ENV_CLASSES[name](max_steps=T)
    mgr = tg.RolloutManager(mk, pol, num_workers=G, num_episodes_per_worker=Eps, dtype=torch.float64, seed=5)
    obs, act, rew, ln, mask = mgr.rollout()
    m = mask.numpy()
    assert np.array_equal(m, (np.arange(T)[None, None, :] < ln.numpy()[..., None]).astype(np.float32))
    assert np.all(obs.numpy()[m == 0] == 0) and np.all(act.numpy()[m == 0] == 0) and np.all(rew.numpy()[m == 0] == 0)
    assert mgr.engine.traj.env_steps() == int(m.sum())
    # the oracle replays the same initial states and actions
    init = mgr.engine.traj.obs[:, 0, :].t().reshape(G, Eps, S).cpu().numpy()
    o2, a2, r2, l2, m2 = L.rollout(lambda: L.OracleEnv(name, max_steps=T), None, G, Eps, initial_states=init,
                                   forced_actions=act.numpy())
    assert torch.equal(l2, ln) and torch.equal(m2, mask)
    np.testing.assert_allclose(obs.numpy(), o2.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(rew.numpy(), r2.numpy(), rtol=1e-5, atol=1e-5)
    # initial-state distributions of the reference resets
    i0 = init.reshape(-1, S)
    if name == "CartPole":
        assert np.all(i0[:, [0, 1, 4]] == 0) and np.allclose(i0[:, 2] ** 2 + i0[:, 3] ** 2, 1)
    elif name == "QuadPole2D":
        assert np.all(i0[:, [0, 1, 2, 3, 4, 6, 9]] == 0) and np.all(i0[:, 5] == 1)
        assert np.allclose(i0[:, 7] ** 2 + i0[:, 8] ** 2, 1)
    else:
        assert np.all(i0[:, :6] == 0) and np.all(i0[:, 6] == 1) and np.all(i0[:, 7:13] == 0) and np.all(i0[:, 17:] == 0)
        assert np.allclose((i0[:, 13:17] ** 2).sum(1), 1) and np.all(np.abs(i0[:, 16]) <= np.sin(0.5) ** 2 + 1e-12)
    # sampled actions are mean + sigma*eps: eps has unit variance
    mean0 = pol.actor(torch.as_tensor(init.reshape(-1, S), dtype=torch.float32, device=dev)).detach().cpu().numpy()
    eps = (act.numpy()[:, :, 0, :].reshape(-1, A) - mean0) / np.sqrt(0.4)
    assert abs(eps.mean()) < 0.2 and 0.7 < eps.std() < 1.3


def test_rollout_is_deterministic_and_sharding_independent(tg, dev):
    T, G, Eps = 32, 4, 64
    torch.manual_seed(2)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (32, 32), cov=0.3, device=dev)
    mk = lambda: tg.QuadPole(max_steps=T)
    for restart in (False, True):
        full = tg.DeviceRollout(mk(), pol, G, Eps, restart, seed=9).run()
        ref = [t.clone() for t in (full.obs, full.act, full.rew, full.mask, full.len)]
        again = tg.DeviceRollout(mk(), pol, G, Eps, restart, seed=9).run()
        for a, b in zip(ref, (again.obs, again.act, again.rew, again.mask, again.len)):
            assert torch.equal(a, b)
        # two "ranks", each owning half the groups, reproduce the single-GPU rollout exactly
        n_half = G // 2 * Eps
        for r in range(2):
            part = tg.DeviceRollout(mk(), pol, G // 2, Eps, restart, seed=9, group_offset=r * G // 2).run()
            sl = slice(r * n_half, (r + 1) * n_half)
            assert torch.equal(part.obs[:, 0, :], ref[0][:, 0, sl])
            assert torch.equal(part.act, ref[1][:, :, sl]) and torch.equal(part.len, ref[4][sl])
            assert torch.equal(part.rew, ref[2][:, sl])
        if restart:     # the E episodes of a group share the group's initial state
            init = ref[0][:, 0, :].reshape(20, G, Eps)
            assert torch.equal(init, init[:, :, :1].expand_as(init))
            assert not torch.equal(init[:, 0, 0], init[:, 1, 0])
    second = tg.DeviceRollout(mk(), pol, G, Eps, False, seed=10).run()
    assert not torch.equal(second.act, ref[1])


def test_graph_replay_matches_eager(tg, dev):
    T, G, Eps = 16, 2, 128
    torch.manual_seed(3)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (64, 64), cov=0.3, device=dev)
    eager = tg.DeviceRollout(tg.QuadPole(max_steps=T), pol, G, Eps, seed=4)
    graph = tg.DeviceRollout(tg.QuadPole(max_steps=T), pol, G, Eps, seed=4, use_graph=True)
    for it in range(3):             # replay 3 times: the RNG stream id lives in device memory
        a, b = eager.run(), graph.run()
        torch.cuda.synchronize()
        assert torch.equal(a.len, b.len) and torch.equal(a.mask, b.mask)
        assert torch.allclose(a.act, b.act, atol=1e-5) and torch.allclose(a.rew, b.rew, atol=1e-4)
    assert a.env_steps() == b.env_steps() > 0


@pytest.mark.parametrize("how", ["torch_adam", "fused_adam"])
@pytest.mark.parametrize("hidden,cd", [((64, 64), None), ((128, 128, 128), torch.bfloat16), ((48, 48), None)])
def test_captured_rollout_graph_follows_weight_updates(tg, dev, hidden, cd, how):
    """the captured per-step graph holds the forward launches only, so the operands they read must be rebuilt
    in front of every replay -- a replay after an optimizer step (torch's, or the raw-pointer fused one) has to act with the NEW
    weights.  An eager engine with the same seed is the witness; the first action is also checked against the policy itself."""
    T, G, Eps = 16, 2, 128
    torch.manual_seed(5)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, hidden, cov=1e-10, device=dev)
    kw = dict(seed=4, fused=False, compute_dtype=cd)
    eager = tg.DeviceRollout(tg.QuadPole(max_steps=T), pol, G, Eps, use_graph=False, **kw)
    graph = tg.DeviceRollout(tg.QuadPole(max_steps=T), pol, G, Eps, use_graph=True, **kw)
    opt = torch.optim.Adam(pol.parameters(), lr=2e-2)
    fused = tg.optim.FusedAdam(opt)
    x_fix = torch.randn(8, 20, device=dev)
    before = pol.actor(x_fix).detach().clone()
    for it in range(3):
        a, b = eager.run(), graph.run()
        torch.cuda.synchronize()
        assert graph._graph is not None
        assert torch.equal(a.len, b.len) and torch.equal(a.mask, b.mask)
        assert torch.allclose(a.act, b.act, atol=1e-5), f"replay {it} acted with other weights than the eager engine"
        mean0 = pol.actor(b.obs[:, 0, :].t().float())
        tol = 5e-2 if cd == torch.bfloat16 else 1e-4
        assert float((b.act[:, 0, :].t() - mean0).abs().max()) < tol, f"replay {it}: first action is not the current policy's mean"
        for p in pol.parameters():                       # a visible step: the mean moves by far more than the tolerances
            p.grad = torch.randn_like(p)
        if how == "fused_adam":
            assert fused.step(), "the fused optimizer step did not apply"
        else:
            opt.step()
    assert float((pol.actor(x_fix) - before).abs().max()) > 1e-2, "the updates did not move the policy: the test proves nothing"


# --------------------------------------------------------------------------------------------
# returns / advantages
# --------------------------------------------------------------------------------------------
def _to_dev_tn(x, dev, dtype=torch.float32):
    """(G,E,T) reference layout -> [T][n] device layout."""
    G, Eps, T = x.shape
    return torch.as_tensor(np.ascontiguousarray(x.reshape(G * Eps, T).T), dtype=dtype, device=dev)


def _from_dev_tn(x, G, Eps):
    return x.t().reshape(G, Eps, -1).cpu().numpy()


@pytest.mark.parametrize("gamma", [0.5, 0.99, 0.999])
def test_rtg_and_advantages_match_reference_golden(tg, dev, gamma):
    K = tg.hip_ops
    g = load_golden("rtg_adv.npz")
    G, Eps, T = g["rew"].shape
    rew, mask = _to_dev_tn(g["rew"], dev), _to_dev_tn(g["mask"], dev, torch.uint8)
    tag = f"g{gamma}"
    rtg = K.rtg_scan(rew, mask, gamma)
    np.testing.assert_allclose(_from_dev_tn(rtg, G, Eps), g[f"{tag}_rtg"], rtol=0, atol=1e-5)   # north star: 1e-5
    # GRPO group-relative advantage
    adv = K.group_normalize(rtg, mask, K.masked_moments(rtg, mask, Eps), 0, Eps)
    adv_ref = _from_dev_tn(adv, G, Eps)
    m = g["mask"].astype(bool)
    for i in range(G):
        np.testing.assert_allclose(adv_ref[i][m[i]], g[f"{tag}_grpo_adv_{i}"], rtol=1e-5, atol=1e-5)
    assert np.all(adv_ref[~m] == 0)
    # PPO: A = R - V (MC) or GAE, then global normalisation of A and R
    V = _to_dev_tn(g[f"{tag}_values"], dev)
    for kind in ("mc", "gae"):
        if kind == "mc":
            a_full, r_full = rtg - V, rtg
        else:
            a_full, r_full = K.gae_scan(rew, V, mask, gamma, 0.95)
        n = G * Eps
        a_n = K.group_normalize(a_full, mask, K.masked_moments(a_full, mask, n), 1, n)
        r_n = K.group_normalize(r_full, mask, K.masked_moments(r_full, mask, n), 1, n)
        np.testing.assert_allclose(_from_dev_tn(a_n, G, Eps)[m], g[f"{tag}_ppo_{kind}_adv"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(_from_dev_tn(r_n, G, Eps)[m], g[f"{tag}_ppo_{kind}_ret"], rtol=1e-5, atol=1e-5)


def test_rtg_scan_is_bit_exact_vs_oracle_and_known_answer(tg, dev):
    K = tg.hip_ops
    rng = np.random.default_rng(7)
    G, Eps, T = 5, 67, 300                        # n = 335 (ragged), T not a multiple of the 32-step chunk
    lens = rng.integers(1, T + 1, size=(G, Eps))
    mask = (np.arange(T)[None, None] < lens[..., None]).astype(np.float32)
    rew = (rng.normal(size=(G, Eps, T)) * 3).astype(np.float32) * mask
    for gamma in (0.5, 0.999, 1.0):
        got = _from_dev_tn(K.rtg_scan(_to_dev_tn(rew, dev), _to_dev_tn(mask, dev, torch.uint8), gamma), G, Eps)
        want = L.rtg_scan(torch.from_numpy(rew), torch.from_numpy(mask), gamma).numpy()
        assert np.array_equal(got, want), f"reward-to-go not bit-exact at gamma={gamma}"
    # reference tests/test_rollout_buffer.py:76-92 (gamma 0.99, no masks)
    r = np.array([[[1, 2, 3], [0, 1, 2]], [[3, 2, 1], [1, 0, 1]]], dtype=np.float32)
    got = _from_dev_tn(K.rtg_scan(_to_dev_tn(r, dev), _to_dev_tn(np.ones_like(r), dev, torch.uint8), 0.99), 2, 2)
    exp = np.zeros_like(r)
    for j in range(2, -1, -1):
        exp[:, :, j] = r[:, :, j] + (0.99 * exp[:, :, j + 1] if j < 2 else 0)
    np.testing.assert_allclose(got, exp, atol=1e-5)


def test_returns_properties_at_scale(tg, dev):
    """Size-independent properties at the bench size (65,536 envs x 256 steps)."""
    K = tg.hip_ops
    n, T, Eg = 65536, 256, 256
    gen = torch.Generator(device=dev).manual_seed(0)
    lens = torch.randint(1, T + 1, (n,), device=dev, generator=gen)
    mask = (torch.arange(T, device=dev)[:, None] < lens[None, :]).to(torch.uint8)
    rew = torch.randn(T, n, device=dev, generator=gen) * mask
    r1 = K.rtg_scan(rew, mask, 1.0)
    # gamma = 1: R[0] is the episode return (sum of masked rewards)
    assert torch.allclose(r1[0], rew.sum(0), rtol=1e-4, atol=1e-3)
    # linearity in the rewards
    r2 = K.rtg_scan(2 * rew, mask, 0.9)
    assert torch.equal(r2, 2 * K.rtg_scan(rew, mask, 0.9))
    # zero beyond the episode; group-normalised advantages have zero mean / unit (unbiased) std per group
    assert torch.all(r1[mask == 0] == 0)
    mom = K.masked_moments(r1, mask, Eg)
    assert torch.equal(mom[:, 0].long().sum(), lens.sum())
    adv = K.group_normalize(r1, mask, mom, 0, Eg)
    a = adv.t().reshape(n // Eg, Eg * T)
    mk = mask.t().reshape(n // Eg, Eg * T).bool()
    for gi in (0, 17, 255):
        v = a[gi][mk[gi]]
        assert abs(float(v.mean())) < 1e-4 and abs(float(v.std()) - 1) < 1e-4


# --------------------------------------------------------------------------------------------
# policy log-prob and the fused loss head
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["actor", "actorcritic"])
def test_gaussian_logp_matches_reference_golden(tg, dev, kind):
    g = load_golden(f"policy_{kind}.npz")
    mean = torch.as_tensor(g["mean"], device=dev)
    act = torch.as_tensor(g["action"], device=dev)
    lp = tg.hip_ops.gaussian_logp(mean, act, g["cov"])
    np.testing.assert_allclose(lp.cpu().numpy(), g["logp_eval"], rtol=0, atol=5e-6)
    lp_t = tg.hip_ops.gaussian_logp(mean, act.t().contiguous().t(), g["cov"])          # strided actions
    assert torch.equal(lp, lp_t)
    # the policy object: load the reference checkpoint layout and evaluate
    cls = tg.GaussianActorCritic_NeuralNetwork if kind == "actorcritic" else tg.GaussianActor_NeuralNetwork
    pol = cls(20, 4, (64, 64), cov=[float(c) for c in g["cov"]], device=dev)
    sd = {k[len("policy."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("policy.")}
    if kind == "actorcritic":
        pol.load_state_dict({"actor": {k[6:]: v for k, v in sd.items() if k.startswith("actor.")},
                             "critic": {k[7:]: v for k, v in sd.items() if k.startswith("critic.")}})
    else:
        pol.load_state_dict(sd)
    lp2, ent = pol.log_prob(g["obs"], g["action"])
    np.testing.assert_allclose(lp2.detach().cpu().numpy(), g["logp_eval"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), g["entropy"], rtol=0, atol=1e-6)
    if kind == "actorcritic":
        np.testing.assert_allclose(pol.value(g["obs"]).detach().cpu().numpy(), g["value_squeezed"], rtol=0, atol=2e-5)
    a, lp3, v = pol(g["obs"][0])
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.shape == (4,)


@pytest.mark.parametrize("A,ppo", [(1, False), (4, False), (2, True), (4, True)])
def test_fused_loss_matches_torch_autograd(tg, dev, A, ppo):
    """The fused HIP loss head against plain torch fp32 autograd of the same formulae
    (oracle.learner.grpo_objective / ppo_loss restate algorithms/grpo.py:115-140, ppo.py:159-179)."""
    K = tg.hip_ops
    gen = torch.Generator(device=dev).manual_seed(11)
    M = 10007
    var = [0.3, 0.2, 0.5, 0.1][:A]
    mean = torch.randn(M, A, device=dev, generator=gen, requires_grad=True)
    act = (mean.detach() + 0.6 * torch.randn(M, A, device=dev, generator=gen))
    vt = torch.tensor(var, device=dev)
    c = -0.5 * A * np.log(2 * np.pi) - 0.5 * float(torch.log(vt).sum())
    logp_old = (-0.5 * (((act - mean.detach()) ** 2) / vt).sum(1) + c) + 0.25 * torch.randn(M, device=dev, generator=gen)
    adv = torch.randn(M, device=dev, generator=gen)
    mask = (torch.rand(M, device=dev, generator=gen) < 0.8).to(torch.uint8)
    eps = 0.2
    value = torch.randn(M, device=dev, generator=gen, requires_grad=True) if ppo else None
    ret = torch.randn(M, device=dev, generator=gen) if ppo else None
    norm = torch.tensor([0.1, 1.7, -0.2, 0.6], device=dev) if ppo else None
    nv = float(mask.sum())
    coefs = (-1.0 / nv, 0.5 / nv, 0.5 / nv) if ppo else (1.0 / 3, 0.0, 0.0)
    total, sums = K.SurrogateLoss.apply(mean, value, act, logp_old, adv, ret, mask, norm, var, eps, *coefs)
    total.backward()
    g_mean, g_val = mean.grad.clone(), (value.grad.clone() if ppo else None)
    mean.grad = None
    # torch reference
    mb = mask.bool()
    lp = -0.5 * (((act - mean) ** 2) / vt).sum(1) + c
    a_n = (adv - norm[0]) * norm[1] if ppo else adv
    ratio = torch.exp(lp - logp_old)
    surr = torch.min(ratio * a_n, torch.clamp(ratio, 1 - eps, 1 + eps) * a_n)[mb].sum()
    ref_total = coefs[0] * surr
    if ppo:
        value.grad = None
        r_n = (ret - norm[2]) * norm[3]
        sq = ((value - r_n) ** 2)[mb].sum()
        kl = (torch.exp(logp_old) * (logp_old - lp))[mb].sum()
        ref_total = ref_total + coefs[1] * sq + coefs[2] * kl
        np.testing.assert_allclose(float(sums[1]), float(sq), rtol=1e-5)
        np.testing.assert_allclose(float(sums[2]), float(kl), rtol=1e-4, atol=1e-4)
    ref_total.backward()
    np.testing.assert_allclose(float(sums[0]), float(surr), rtol=1e-5, atol=1e-3)
    assert float(sums[3]) == nv
    np.testing.assert_allclose(float(total), float(ref_total), rtol=1e-4, atol=1e-6)
    scale = float(mean.grad.abs().max())
    assert float((g_mean - mean.grad).abs().max()) <= 1e-5 * scale + 1e-9
    assert torch.all(g_mean[~mb] == 0)
    if ppo:
        assert float((g_val - value.grad).abs().max()) <= 1e-5 * float(value.grad.abs().max()) + 1e-9


# --------------------------------------------------------------------------------------------
# learn(): optimizer steps against the reference
# --------------------------------------------------------------------------------------------
class _Buf:
    device_traj = None


def _buffer_from_golden(g):
    b = _Buf()
    b.group_observations, b.group_actions = torch.from_numpy(g["obs"]), torch.from_numpy(g["act"])
    b.group_rewards, b.group_masks = torch.from_numpy(g["rew"]), torch.from_numpy(g["mask"])
    return b


@pytest.mark.parametrize("n_upd", [1, 2])
def test_grpo_learn_matches_reference(tg, dev, n_upd):
    g = load_golden(f"grpo_step_u{n_upd}.npz")
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (32, 32), cov=float(g["cov"]), device=dev)
    pol.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")})
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    algo = tg.GRPO(epsilon=float(g["epsilon"]), beta=0.5, gamma=float(g["gamma"]), policy=pol, optimizer=opt,
                   updates_per_iter=n_upd)
    algo.old_policy.load_state_dict({k[9:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("old_init.")})
    algo.learn(_buffer_from_golden(g))
    np.testing.assert_allclose(algo.last_stats["J"], g["J"], rtol=2e-4, atol=5e-6)
    for k, p in pol.actor.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"lastgrad.{k}"], rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final.{k}"], rtol=0, atol=2e-5)
    # old_policy <- policy after learn (grpo.py:148)
    for a, b in zip(algo.old_policy.parameters(), pol.parameters()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_upd", [1, 2])
def test_ppo_learn_matches_reference(tg, dev, n_upd):
    g = load_golden(f"ppo_step_u{n_upd}.npz")
    pol = tg.GaussianActorCritic_NeuralNetwork(10, 2, (32, 32), cov=float(g["cov"]), device=dev)
    sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    pol.load_state_dict({"actor": {k[6:]: v for k, v in sd.items() if k.startswith("actor.")},
                         "critic": {k[7:]: v for k, v in sd.items() if k.startswith("critic.")}})
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    algo = tg.PPO(epsilon=float(g["epsilon"]), policy=pol, optimizer=opt, ref_model=None, updates_per_iter=n_upd,
                  c1=float(g["c1"]), kl_coeff=float(g["kl_coeff"]), gamma=float(g["gamma"]), lam=0.95,
                  entropy=float(g["entropy_coeff"]), batch_size=None)
    algo.learn(_buffer_from_golden(g))
    st = algo.last_stats
    np.testing.assert_allclose(st["total_loss"], g["total_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(st["actor_loss"], g["actor_loss"], rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(st["critic_loss"], g["critic_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(st["kl_div"], g["kl_div"], rtol=1e-3, atol=1e-7)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"lastgrad.{net}.{k}"], rtol=2e-3, atol=2e-6)
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-5)


def test_ppo_learn_with_gae_and_per_dimension_covariance(tg, dev):
    g = load_golden("ppo_gae_step_u2.npz")
    pol = tg.GaussianActorCritic_NeuralNetwork(10, 2, (32, 32), cov=[float(c) for c in g["cov"]], device=dev)
    sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    pol.load_state_dict({"actor": {k[6:]: v for k, v in sd.items() if k.startswith("actor.")},
                         "critic": {k[7:]: v for k, v in sd.items() if k.startswith("critic.")}})
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    algo = tg.PPO(epsilon=float(g["epsilon"]), policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2,
                  c1=float(g["c1"]), kl_coeff=float(g["kl_coeff"]), gamma=float(g["gamma"]), lam=float(g["lam"]),
                  entropy=float(g["entropy_coeff"]), batch_size=None, monte_carlo=False)
    algo.learn(_buffer_from_golden(g))
    st = algo.last_stats
    np.testing.assert_allclose(st["total_loss"], g["total_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(st["critic_loss"], g["critic_loss"], rtol=2e-5, atol=2e-6)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-5)


def test_pipeline_trains_and_checkpoints(tg, dev, tmp_path, monkeypatch):
    """C1 plumbing: CartPole GRPO, 4 envs (2x2), 128-step horizon through the Pipeline API; resume from the checkpoint."""
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    mk = lambda: tg.CartPole(max_steps=128)
    mgr = tg.RolloutManager(mk, pol, num_workers=2, num_episodes_per_worker=2, restart=True)
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4),
                   updates_per_iter=1)
    pipe = tg.create_cartpole_pipeline_grpo("t", "001", env_fn=mk, policy=pol, algorithm=algo, rollout_manager=mgr)
    pipe.train(3)
    assert len(pipe.buffer.avg_reward) == 3 and np.isfinite(pipe.buffer.avg_reward).all()
    ck = tmp_path / "archive" / "CartPole" / "t" / "001"
    for f in ("policy.pt", "optimizer.pth", "reward.csv", "metadata.json"):
        assert (ck / f).exists()
    obs = pipe.buffer.group_observations
    assert obs.shape == (2, 2, 128, 5) and pipe.buffer.group_lengths.shape == (2, 2)
    pipe.save_trajectory()
    assert (ck / "trajectory.csv").exists()
    pipe.save(pipe.archive_path)
    pol2 = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    mgr2 = tg.RolloutManager(mk, pol2, num_workers=2, num_episodes_per_worker=2)
    algo2 = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol2, optimizer=torch.optim.Adam(pol2.parameters(), lr=3e-4),
                    updates_per_iter=1)
    pipe2 = tg.create_cartpole_pipeline_grpo("t", "002", env_fn=mk, policy=pol2, algorithm=algo2, rollout_manager=mgr2,
                                             load_path=str(ck))
    for a, b in zip(pol.parameters(), pol2.parameters()):
        assert torch.equal(a, b)
    # GRPO deep-copies the policy at construction (grpo.py:48), before the checkpoint is loaded: the resumed learner must take
    # its first old-log-probs from the LOADED weights, not from the random-init copy
    for a, b in zip(algo2.old_policy.parameters(), pol2.parameters()):
        assert torch.equal(a, b)
    assert len(pipe2.buffer.avg_reward) >= 1
    # ... so one more iteration on the same rollout gives the same objective in both learners (ratio = pi / pi_old)
    pipe.buffer.sample()
    algo.learn(pipe.buffer)
    algo2.learn(pipe.buffer)
    assert algo2.last_stats["J"] == algo.last_stats["J"]
    pipe2.shutdown()


def test_rollout_noise_follows_the_policy_covariance_of_the_moment(tg, dev):
    """The reference reads `self.cov` on every forward (policies/actor_critic.py:107-138): a covariance changed after the
    manager exists (annealed exploration noise, a restored checkpoint) must reach the next rollout."""
    torch.manual_seed(0)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (64, 64), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=16), pol, num_workers=2, num_episodes_per_worker=64, seed=3)
    t1 = mgr.rollout_device()
    a1 = t1.act_rows().float().clone()
    mgr2 = tg.RolloutManager(lambda: tg.CartPole(max_steps=16), pol, num_workers=2, num_episodes_per_worker=64, seed=3)
    pol.cov = 1e-8 * torch.eye(1)
    t2 = mgr2.rollout_device()
    a2 = t2.act_rows().float()
    with torch.no_grad():
        mean0 = pol.actor(t2.obs_rows()[:128].float())              # step 0 of all 128 envs: same initial states in both runs
    assert float((a2[:128] - mean0).abs().max()) < 1e-3              # sigma = 1e-4: the action is the mean
    assert float((a1[:128] - mean0).abs().max()) > 0.1               # sigma = 0.71


def test_captured_rollout_graph_follows_a_changed_covariance(tg, dev):
    """tg_rollout_step takes sigma by value, so a captured hipGraph replays the sigma of its capture: the SAME manager (per-step
    path, use_graph=True) must re-capture when the covariance changed between two rollouts (the learner would
    otherwise form its ratios with the new variance against actions drawn with the old one)."""
    torch.manual_seed(0)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (64, 64), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=16), pol, num_workers=2, num_episodes_per_worker=64, seed=3,
                            fused=False, use_graph=True)
    t1 = mgr.rollout_device()
    assert mgr.engine._graph is not None and not mgr.engine.fused
    g1 = mgr.engine._graph
    with torch.no_grad():
        mean0 = pol.actor(t1.obs_rows()[:128].float())
    assert float((t1.act_rows().float()[:128] - mean0).abs().max()) > 0.1     # sigma = 0.71
    mgr.rollout_device()
    assert mgr.engine._graph is g1                                   # unchanged covariance: the capture is reused
    pol.cov = 1e-8 * torch.eye(1)
    t3 = mgr.rollout_device()
    assert mgr.engine._graph is not g1
    with torch.no_grad():
        mean3 = pol.actor(t3.obs_rows()[:128].float())
    assert float((t3.act_rows().float()[:128] - mean3).abs().max()) < 1e-3     # sigma = 1e-4: the action is the mean


# --------------------------------------------------------------------------------------------
# hand-scheduled MLP (GEMM chain + tg_relu_bwd_bias) against torch autograd
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dims", [(20, 4, (256, 256, 256)), (5, 1, (128, 64)), (10, 2, (32,)), (10, 2, (128, 128, 128)), (3, 1, (256,))])
def test_gemm_mlp_matches_autograd(tg, dev, cd, dims):
    """Reference = torch fp32 autograd of the same module.  Gradients are sums over ~25k rows through ReLU masks: a
    single mask flip from rounding moves a hidden-layer gradient by ~5e-4 relative (also seen between torch fp32 and
    fp64), so hidden layers are compared in relative L2 norm; the fp32 head (no mask above it) is compared tightly."""
    S, A, hidden = dims
    torch.manual_seed(4)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    assert tg.mlp.supports(net)
    m = tg.mlp.GemmMLP(net, cd)
    rows = 3 * 8192 + 777                                     # three split-K blocks + a ragged tail
    X = torch.randn(rows, S, device=dev)
    g = torch.randn(rows, A, device=dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    out = m.forward(m.prepare_input(X), keep=True)
    m.backward(g)
    got = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    bf = cd == torch.bfloat16
    ref = net(X)                                              # fp32 autograd: the truth both bf16 pipelines approximate
    ref.backward(g)
    truth = [p.grad.clone() for p in net.parameters()]
    assert float((out - ref.detach()).abs().max()) <= (2e-2 if bf else 2e-6) * float(ref.abs().max())
    names = [n for n, _ in net.named_parameters()]
    if bf:
        # bf16: the forward runs as one chain kernel with fp32 biases, the backward on the fused kernels; torch's own bf16
        # pipeline (autocast + autograd) is the yardstick: random upstream gradients cancel heavily in this test, so both
        # sit several per cent from fp32 -- ours must not be further away than autocast's (+30 %, + 2e-3)
        for p in net.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ac = net(X)
        ac.float().backward(g)
        for n, a, t, p in zip(names, got, truth, net.parameters()):
            err, err_ac = float((a - t).norm() / t.norm()), float((p.grad - t).norm() / t.norm())
            assert err <= 1.3 * err_ac + 2e-3, (n, err, err_ac)
    else:
        for n, a, t in zip(names, got, truth):
            rel = float((a - t).norm() / t.norm())
            head = n.startswith(f"network.{2 * len(hidden)}.")
            assert rel <= (5e-6 if head else 3e-3), (n, rel)
    pad = m.forward(m.prepare_input(X), keep=False, padded=True)
    # (an fp32 net of the chain learner's shapes runs its no-grad pass on tg_mlp_f32_forward, its keep=True pass on the per-layer
    # GEMMs: same values to fp32 rounding, not the same bits)
    same = torch.equal(pad[:, :A].contiguous(), out) if m._f32 is None else \
        float((pad[:, :A] - out).abs().max()) <= 2e-6 * float(out.abs().max())
    assert pad.shape[1] % 4 == 0 and same and torch.all(pad[:, A:] == 0)
    assert not tg.mlp.supports(tg.NeuralNetwork(S, A, hidden, "Tanh"))


@pytest.mark.parametrize("width", [64, 128, 256])
@pytest.mark.parametrize("rows", [1, 1000, 70001])
def test_fused_backward_data_relu_bias_kernel(tg, dev, width, rows):
    """tg_dx_relu_bias against the two-pass formulation it replaces (dA = dZ @ W; dZ_below = dA * (A > 0); column sums)."""
    Nn = tg._native
    lib = Nn.load()
    assert lib.tg_dx_relu_bias_supported(width, width) and not lib.tg_dx_relu_bias_supported(width, 2 * width)
    gen = torch.Generator(device="cpu").manual_seed(width + rows)
    dz = (torch.randn(rows, width, generator=gen) * 0.5).to(torch.bfloat16).to(dev)
    W = (torch.randn(width, width, generator=gen) / width ** 0.5).to(torch.bfloat16).to(dev)
    act = torch.relu(torch.randn(rows, width, generator=gen)).to(torch.bfloat16).to(dev)
    act[0, :5] = torch.tensor([0.0, -0.0, 1e-30, 1.0, 0.0], dtype=torch.bfloat16)      # signed zero, tiny positive
    frag = torch.empty(width * width, dtype=torch.bfloat16, device=dev)
    Nn.check(lib.tg_dx_pack_weights(W.data_ptr(), frag.data_ptr(), width, width, Nn.stream_ptr(dev)))
    out = torch.full((rows, width), float("nan"), dtype=torch.bfloat16, device=dev)
    partial = torch.full((lib.tg_dx_relu_bias_blocks(), width), float("nan"), dtype=torch.float32, device=dev)
    Nn.check(lib.tg_dx_relu_bias(dz.data_ptr(), frag.data_ptr(), act.data_ptr(), None, out.data_ptr(), rows, width, width,
                                 partial.data_ptr(), Nn.stream_ptr(dev)))
    torch.cuda.synchronize()
    exact = (dz.double() @ W.double()) * (act > 0)
    got = out.double()
    assert torch.isfinite(got).all()
    assert torch.equal(got == 0, exact.to(torch.bfloat16) == 0) or ((got == 0) == (exact == 0)).float().mean() > 0.9999
    # one bf16 rounding of an fp32-accumulated sum: <= 2^-8 relative (+ accumulation noise far below that)
    assert torch.all((got - exact).abs() <= 2.0 ** -8 * exact.abs() + 1e-6)
    assert torch.all(got[act <= 0] == 0)
    # the bias-gradient partials add up to the column sums of what was written
    np.testing.assert_allclose(partial.sum(0).cpu().numpy(), got.sum(0).float().cpu().numpy(), rtol=2e-5, atol=2e-4 * rows ** 0.5)
    if width >= 128:
        # the same launch reading 1 bit per activation (tg_mlp_forward_chain's mask layout) instead of the activations
        bits = _pack_mask_bits(act)
        out2 = torch.full_like(out, float("nan"))
        partial2 = torch.full_like(partial, float("nan"))
        Nn.check(lib.tg_dx_relu_bias(dz.data_ptr(), frag.data_ptr(), None, bits.data_ptr(), out2.data_ptr(), rows, width, width,
                                     partial2.data_ptr(), Nn.stream_ptr(dev)))
        torch.cuda.synchronize()
        assert torch.equal(out2, out) and torch.equal(partial2, partial)


def _pack_mask_bits(act):
    """[rows][H] activations -> [rows][H/32] int32 ReLU-mask words in tg_mlp_forward_chain's layout: per row
    [lane half h][H/64 words]; feature 32 mt + 16 h + r is bit (mt&1)*8 + (r>>1) + 16*(r&1) of word mt>>1."""
    rows, H = act.shape
    f = torch.arange(H, device=act.device)
    mt, h, r = f >> 5, (f >> 4) & 1, f & 15
    word = h * (H // 64) + (mt >> 1)
    bit = (mt & 1) * 8 + (r >> 1) + 16 * (r & 1)
    out = torch.zeros(rows, H // 32, dtype=torch.int64, device=act.device)
    out.index_add_(1, word, (act > 0).to(torch.int64) << bit)
    return torch.where(out >= 2 ** 31, out - 2 ** 32, out).to(torch.int32)


@pytest.mark.parametrize("dims", [(20, 4, (256,) * 5), (10, 2, (128, 128)), (5, 1, (256,)), (32, 12, (128,) * 3)])
@pytest.mark.parametrize("rows", [1, 257, 70001])
def test_forward_chain_kernel_matches_layer_by_layer(tg, dev, dims, rows):
    """tg_mlp_forward_chain (all layers in one launch) against the per-layer GEMM path of the same GemmMLP and against
    fp64: stored activations and head output."""
    from trajopt_grpo_amd.mlp import GemmMLP
    S, A, hidden = dims
    torch.manual_seed(rows + S)
    net = tg.policies.NeuralNetwork(S, A, hidden).to(dev)
    for p in net.parameters():
        p.requires_grad_(False)
    mlp = GemmMLP(net, torch.bfloat16)
    assert mlp._chain is not None
    xp = mlp.prepare_input(torch.randn(rows, S, device=dev))
    bchain, mlp._bchain = mlp._bchain, None                              # per-layer backward mode: every activation is stored
    out_c = mlp.forward(xp, keep=True, padded=True)
    acts_c = mlp._acts
    assert len(acts_c) == len(hidden) + 1 and acts_c[0] is xp
    # the ReLU mask bits written beside the activations are exactly (activation > 0), in the documented layout
    assert len(mlp._bits) == len(acts_c) and mlp._bits[0] is None
    for a_l, b_l in zip(acts_c[1:], mlp._bits[1:]):
        assert torch.equal(b_l, _pack_mask_bits(a_l))
    mlp_bits_c = [None if b is None else b.clone() for b in mlp._bits]
    acts_c = [a if a is xp else a.clone() for a in acts_c]                # the workspace buffers are reused by later passes
    out_nokeep = mlp.forward(xp, keep=False, padded=True)
    assert mlp._acts is None and torch.equal(out_nokeep, out_c)          # same arithmetic with and without the stores
    chain, mlp._chain = mlp._chain, None
    out_l = mlp.forward(xp, keep=True, padded=True)
    acts_l = mlp._acts
    mlp._chain = chain
    torch.cuda.synchronize()
    # fp64 evaluation on the same bf16 weights / input, bf16 activations between layers as both paths store them
    h = xp[:, :S].double()
    for i, lin in enumerate(mlp.linears[:-1]):
        h = torch.relu(h @ lin.weight.to(torch.bfloat16).double().t() + lin.bias.double()).to(torch.bfloat16)
        # one bf16 rounding apart (+ the per-layer path rounds the bias to bf16): compare to the chain's activations
        a = acts_c[i + 1].double()
        assert torch.all((a - h.double()).abs() <= 2.0 ** -7 * h.double().abs() + 2e-3), f"layer {i}"
        assert torch.all((a - acts_l[i + 1].double()).abs() <= 2.0 ** -6 * a.abs() + 4e-3)
        h = acts_c[i + 1].double()                                      # follow the chain's own activations
    ref = h @ mlp.linears[-1].weight.to(torch.bfloat16).double().t() + mlp.linears[-1].bias.double()
    np.testing.assert_allclose(out_c[:, :A].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert torch.all(out_c[:, A:] == 0)
    assert out_c.shape[1] == (4 if A <= 4 else mlp.out_pad)             # <= 4 outputs: a 16-B row (tg_rollout_step's mean read)
    assert float((out_c[:, :A] - out_l[:, :A]).abs().max()) <= 2e-2 * float(out_l.abs().max()) + 1e-3
    if bchain is not None:
        # with the backward chain active the first activation is left out (tg_mlp_weight_grad recomputes it); everything
        # else -- later activations, ALL mask bits, the output -- is bit-identical
        bits_c = [None if b is None else b.clone() for b in mlp_bits_c]
        acts_keep = [None if a is None else a.clone() for a in acts_c]
        mlp._bchain = bchain
        out_s = mlp.forward(xp, keep=True, padded=True)
        assert mlp._acts[1] is not None                 # no gradient buffers: backward() could not take the chain path, a0 is kept
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        out_s = mlp.forward(xp, keep=True, padded=True)
        assert mlp._acts[1] is None and torch.equal(out_s, out_c)
        for a_s, a_k in zip(mlp._acts[2:], acts_keep[2:]):
            assert torch.equal(a_s, a_k)
        for b_s, b_k in zip(mlp._bits[1:], bits_c[1:]):
            assert torch.equal(b_s, b_k)


@pytest.mark.parametrize("dims,rows", [((20, 4, (256,) * 5), 70001), ((20, 1, (256,) * 3), 257), ((10, 2, (256,) * 4), 1),
                                       ((5, 1, (128,) * 3), 33333), ((20, 4, (128,) * 6), 255)])
def test_backward_chain_kernel_matches_layer_by_layer(tg, dev, dims, rows):
    """tg_mlp_backward_chain (every hidden layer's dZ in one launch) against the per-layer kernels of the same GemmMLP:
    the same gradients up to bf16 rounding of the head's product, bit-identical from run to run."""
    from trajopt_grpo_amd.mlp import GemmMLP
    S, A, hidden = dims
    torch.manual_seed(rows + S)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    mlp = GemmMLP(net, torch.bfloat16)
    assert mlp._bchain is not None
    xp = mlp.prepare_input(torch.randn(rows, S, device=dev))
    g = torch.randn(rows, A, device=dev)

    def grads(use_chain):
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        keep, mlp._bchain = mlp._bchain, (mlp._bchain if use_chain else None)
        mlp.forward(xp, keep=True)
        mlp.backward(g)
        mlp._bchain = keep
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    a, b, a2 = grads(True), grads(False), grads(True)
    for x, y in zip(a, a2):
        assert torch.equal(x, y)                                   # deterministic reductions
    for (n, _), x, y in zip(net.named_parameters(), a, b):
        denom = float(y.norm()) + 1e-12
        # (anchored elementwise to fp64 in test_backward_chain_kernel_matches_fp64; here the two HIP paths differ by the head's
        # product -- bf16 x bf16 in the chain, fp32 dout in the per-layer path -- and one rounding per layer)
        assert float((x - y).norm()) / denom < 1e-2, n


@pytest.mark.parametrize("dims", [(20, 4, (256,) * 5), (5, 1, (128,) * 3), (10, 2, (256,) * 3), (20, 4, (128,) * 6), (12, 3, (256,) * 4)])
@pytest.mark.parametrize("rows", [1, 257, 70001, 300000])
def test_backward_chain_kernel_matches_fp64(tg, dev, dims, rows):
    """tg_mlp_backward_chain / _w0 (torch autograd's backward-data pass, algorithms/ppo.py:181-183) against fp64: from the same
    bf16 weights, the stored ReLU mask bits and the bf16 d loss / d output, every layer's dZ = (dZ_above . W) * mask evaluated in
    fp64 from the chain's OWN stored dZ of the layer above, one bf16 rounding per layer: <= 1 bf16 ulp (2^-7 relative) + the fp32
    accumulation slack, on every stored element.  rows > 65,536: several rounds per workgroup.  The fused first-layer gradient
    (_w0: dW0 | db0 = dZ_bottom^T . [x | 1]) against the fp64 product of the fp64-derived bottom dZ."""
    from trajopt_grpo_amd import mlp as M, _native as N
    S, A, hidden = dims
    H, nh = hidden[0], len(hidden)
    torch.manual_seed(rows + S + nh)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    mlp = M.GemmMLP(net, torch.bfloat16)
    assert mlp._bchain is not None
    lib = N.load()
    xp = mlp.prepare_input(torch.randn(rows, S, device=dev))
    mlp.forward(xp, keep=True)
    bits = [None if b is None else b.clone() for b in mlp._bits]            # bits[i + 1] masks hidden layer i
    acts_mask = []
    # masks as booleans from the stored bits (layout: _pack_mask_bits)
    f = torch.arange(H, device=dev)
    mt, hh, r = f >> 5, (f >> 4) & 1, f & 15
    word, bit = hh * (H // 64) + (mt >> 1), (mt & 1) * 8 + (r >> 1) + 16 * (r & 1)
    for i in range(nh):
        w = bits[i + 1].to(torch.int64) & 0xFFFFFFFF
        acts_mask.append(((w[:, word] >> bit) & 1).bool())
    dzh = torch.zeros(rows, 8, dtype=torch.bfloat16, device=dev)
    dzh[:, :A] = (torch.randn(rows, A, device=dev) * 0.05).bfloat16()
    mlp._fresh("bchain")
    m_ptrs = (N.C.c_void_p * nh)(*[bits[nh - j].data_ptr() for j in range(nh)])
    dzs = [torch.full((rows, H), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(nh)]      # top hidden layer first
    ptrs = (N.C.c_void_p * nh)(*[t.data_ptr() for t in dzs])
    N.check(lib.tg_mlp_backward_chain(dzh.data_ptr(), mlp._bchain.stream.data_ptr(), H, nh, rows, ptrs, m_ptrs, None,
                                      N.stream_ptr(dev)), "tg_mlp_backward_chain")
    torch.cuda.synchronize()
    lin = mlp.linears                                                        # lin[i]: layer i; lin[nh] the head
    above = dzh[:, :A].double()
    for j in range(nh):
        i = nh - 1 - j                                                       # dzs[j] = dZ of hidden layer i
        W = lin[i + 1].weight.detach().to(torch.bfloat16).double()           # [out][in]: dA_i = dZ_{i+1} . W_{i+1}
        ref = (above @ W) * acts_mask[i]
        slack = 2e-6 * (above.abs() @ W.abs())                               # fp32 accumulation of K <= 256 products
        got = dzs[j].double()
        bad = (got - ref).abs() > 2.0 ** -7 * ref.abs() + slack + 1e-30
        assert not bool(bad.any()), (j, int(bad.sum()), float((got - ref).abs().max()))
        assert torch.all(got[~acts_mask[i]] == 0)                           # masked entries are exact zeros
        above = got                                                          # follow the chain's own stored values
    # ---- the fused first-layer gradient: the bottom dZ is not written; dW0 | db0 from the chain's registers ----
    if S < 32:
        slabs = torch.full((2 * lib.tg_mlp_backward_chain_blocks() * H * 32,), float("nan"), dtype=torch.float32, device=dev)
        nsl = N.C.c_int32(0)
        dz2 = [torch.full((rows, H), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(nh)]
        ptrs2 = (N.C.c_void_p * nh)(*[(t.data_ptr() if 0 < j < nh - 1 else None) for j, t in enumerate(dz2)])
        N.check(lib.tg_mlp_backward_chain_w0(dzh.data_ptr(), mlp._bchain.stream.data_ptr(), H, nh, rows, ptrs2, m_ptrs, xp.data_ptr(),
                                             slabs.data_ptr(), slabs.numel(), N.C.byref(nsl), N.stream_ptr(dev)), "tg_mlp_backward_chain_w0")
        torch.cuda.synchronize()
        for j in range(1, nh - 1):
            assert torch.equal(dz2[j], dzs[j])                               # the stored layers do not depend on the fusion
        got0 = slabs[:nsl.value * H * 32].view(nsl.value, H, 32).double().sum(0)
        ref0 = dzs[nh - 1].double().t() @ xp.double()                        # [H][32]: columns < S = dW0, column 31 = db0
        tol = 2e-5 * (float(ref0.abs().max()) + 1e-3) * max(1.0, (rows / 1000) ** 0.5)
        assert float((got0 - ref0).abs().max()) < tol
        assert float(got0[:, S:31].abs().max()) == 0.0


def test_weight_gradient_split_k_paths_agree(tg, dev):
    """GemmMLP._dw: the fixed-batch-count split (>= 2^19 rows) and the fixed-block split below it against one fp32 GEMM."""
    from trajopt_grpo_amd.mlp import GemmMLP
    net = tg.NeuralNetwork(20, 4, (64, 64), "ReLU").to(dev)
    m = GemmMLP(net, torch.bfloat16)
    gen = torch.Generator(device="cpu").manual_seed(3)
    for rows in (128 * 4096 + 1777, 3 * 8192 + 5, 100):
        dz = torch.randn(rows, 64, generator=gen).to(torch.bfloat16).to(dev)
        a = torch.randn(rows, 32, generator=gen).to(torch.bfloat16).to(dev)
        ref = dz.double().t() @ a.double()
        got = m._dw(dz, a).double()
        assert float((got - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_weight_gradient_epilogue_kernel(tg, dev, cdt):
    """GemmMLP._dw_into: split-K batched GEMM + tg_dw_finish (partial sums + row tail + accumulation into a window of
    the gradient) against one fp64 product; identical from run to run."""
    from trajopt_grpo_amd.mlp import GemmMLP
    net = tg.NeuralNetwork(20, 4, (64, 64), "ReLU").to(dev)
    m = GemmMLP(net, cdt)
    gen = torch.Generator(device="cpu").manual_seed(4)
    for rows, M, K, mo, ko in ((128 * 4096 + 77, 64, 64, 64, 64), (128 * 4096, 64, 32, 64, 20), (128 * 4100 + 127, 8, 64, 4, 64)):
        dz = torch.randn(rows, M, generator=gen).to(cdt).to(dev)
        a = torch.randn(rows, K, generator=gen).to(cdt).to(dev)
        g0 = torch.randn(mo, ko, generator=gen).to(dev)
        flat = torch.zeros(mo * ko + 5, device=dev)                       # a view into a flat bucket, like GradBucket's
        grad = flat[5:].view(mo, ko)
        grad.copy_(g0)
        m._dw_into(grad, dz, a)
        ref = g0.double() + (dz.double().t() @ a.double())[:mo, :ko]
        tol = 1e-5 if cdt == torch.bfloat16 else 1e-4                     # fp32 operands: the GEMM's own fp32 summation
        assert float((grad.double() - ref).norm() / ref.norm()) < tol
        assert torch.all(flat[:5] == 0)
        again = g0.clone()
        m._dw_into(again, dz, a)
        assert torch.equal(again, grad)
    # below the split threshold the torch path is used; same contract
    dz = torch.randn(5000, 64, generator=gen).to(cdt).to(dev)
    a = torch.randn(5000, 64, generator=gen).to(cdt).to(dev)
    grad = torch.ones(64, 64, device=dev)
    m._dw_into(grad, dz, a)
    ref = 1.0 + dz.double().t() @ a.double()
    assert float((grad.double() - ref).norm() / ref.norm()) < 1e-4


def test_learners_fall_back_to_autograd_for_non_relu_nets(tg, dev):
    """A Tanh policy cannot use the GEMM chain; learn() must still run (torch autograd path) and move the weights."""
    torch.manual_seed(5)
    pol = tg.GaussianActorCritic_NeuralNetwork(5, 1, (32, 32), activation="Tanh", cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=32), pol, num_workers=2, num_episodes_per_worker=64)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=1e-3), ref_model=None,
                  updates_per_iter=2, batch_size=None)
    before = [p.detach().clone() for p in pol.parameters()]
    algo.learn(buf)
    assert all(not torch.equal(a, b) for a, b in zip(before, pol.parameters()))
    assert np.isfinite(algo.last_stats["total_loss"]).all()
    # minibatch mode (batch_size set) takes several optimizer steps per epoch
    algo2 = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=1e-3), ref_model=None,
                   updates_per_iter=1, batch_size=1024)
    algo2.learn(buf)
    assert len(algo2.last_stats["total_loss"]) == -(-int(algo2.last_stats["n_valid"]) // 1024)


# --------------------------------------------------------------------------------------------
# published checkpoints of the reference (reports/*/001): load the reference's artefact formats and
# reproduce the published learning-curve end points with a fresh GPU rollout
# --------------------------------------------------------------------------------------------
PUBLISHED = [
    # name, env, actor-critic?, dims, published last avg_reward, reference-CPU re-evaluation (20 episodes), band
    ("cartpole_nn_ppo", "CartPole", True, (5, 1, (128, 128, 128)), 800.79, 819.7, (650.0, 950.0)),
    ("quadpole2d_nn_ppo", "QuadPole2D", True, (10, 2, (128, 128, 128)), 1047.64, 906.4, (750.0, 1100.0)),
    ("cartpole_nn_grpo", "CartPole", False, (5, 1, (128, 128, 128, 128)), -62.20, -46.4, (-75.0, -30.0)),
]


@pytest.mark.parametrize("name,env_name,critic,dims,published,ref_eval,band", PUBLISHED)
def test_published_checkpoints_reproduce_on_gpu(tg, dev, name, env_name, critic, dims, published, ref_eval, band):
    import json
    import os
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "published", name)
    meta = json.load(open(os.path.join(path, "metadata.json")))
    assert meta["policy"]["hidden_dims"] == list(dims[2]) and meta["buffer"]["avg_reward"] == pytest.approx(published, abs=0.01)
    cls = tg.GaussianActorCritic_NeuralNetwork if critic else tg.GaussianActor_NeuralNetwork
    cov = [row[i] for i, row in enumerate(meta["policy"]["cov"])]
    pol = cls(dims[0], dims[1], dims[2], cov=cov, device=dev)
    pol.load(path)                                                       # reference `policy.pt` format
    assert pol.metadata()["num_parameters"] == meta["policy"]["num_parameters"]
    mk = lambda: tg.environments.ENV_CLASSES[env_name]()                 # default max_steps = 500, as published
    mgr = tg.RolloutManager(mk, pol, num_workers=16, num_episodes_per_worker=256, seed=0)
    buf = tg.Rollout_Buffer(mgr)
    assert buf.load(path) == len(open(os.path.join(path, "reward.csv")).read().split())   # reward.csv resume
    buf.sample()
    avg = float(buf.avg_reward[-1])
    assert band[0] < avg < band[1], f"{name}: avg return {avg:.1f}; published {published}, reference re-eval {ref_eval}"


# --------------------------------------------------------------------------------------------
# fused persistent rollout kernel (MFMA actor + sample + step in one launch)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,hidden", [("QuadPole", (256, 256, 256)), ("QuadPole", (128, 128)),
                                         ("CartPole", (128, 128, 128)), ("QuadPole2D", (256, 256))])
def test_fused_rollout_matches_unfused_path(tg, dev, name, hidden):
    S, A = DIMS[name]
    T, G, Eps = 40, 3, 200                      # 600 envs: 2 full workgroups + a ragged one
    torch.manual_seed(6)
    pol = tg.GaussianActor_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
    mk = lambda: tg.environments.ENV_CLASSES[name](max_steps=T)
    fused = tg.DeviceRollout(mk(), pol, G, Eps, seed=21, compute_dtype=torch.bfloat16, fused=True)
    plain = tg.DeviceRollout(mk(), pol, G, Eps, seed=21, compute_dtype=torch.bfloat16, fused=False)
    assert fused.fused and not plain.fused
    tf = fused.run()
    fo, fa, fr, fm, fl = (x.clone() for x in (tf.obs, tf.act, tf.rew, tf.mask, tf.len))
    # (1) one step from identical states: same Philox draw, means equal up to bf16 / summation order
    tp = plain.run()
    assert torch.equal(tp.obs[:, 0, :], fo[:, 0, :])                           # same reset draw
    std = float(np.sqrt(0.3))
    da = (tp.act[:, 0, :] - fa[:, 0, :]).abs().max()
    assert float(da) < 0.03 * max(1.0, float(fa[:, 0, :].abs().max())), float(da)
    assert float((fa[:, 0, :] - tp.act[:, 0, :]).abs().mean()) < 5e-3
    # (2) replaying the fused rollout's initial states and actions through the teacher-forced step kernel
    #     (same fp32 dynamics code) reproduces every recorded quantity
    replay = plain.run(initial_states=fo[:, 0, :].t().cpu().numpy(), forced_actions=fa.permute(2, 1, 0).cpu().numpy())
    assert torch.equal(replay.len, fl) and torch.equal(replay.mask, fm)
    assert torch.equal(replay.obs, fo) and torch.equal(replay.rew, fr)     # dynamics compiled without FMA contraction: same bits
    # (3) invariants
    m = fm.bool()
    assert torch.equal(fm.sum(0, dtype=torch.int32), fl) and tf.env_steps() == int(fm.sum())
    assert torch.all(fr[~m] == 0) and torch.all(fa[:, ~m] == 0)
    assert torch.all(fo[:, :T][:, ~m] == 0)
    eps = (fa[:, 0, :] - tp.act[:, 0, :])                                       # noise cancels: pure mean difference
    assert float(eps.abs().max()) < 0.05
    # (4) noise statistics of the sampled actions (unit-variance eps)
    mean0 = pol.actor(fo[:, 0, :].t()).detach()
    z = ((fa[:, 0, :].t() - mean0) / std).cpu().numpy()
    assert abs(z.mean()) < 0.15 and 0.8 < z.std() < 1.2
    # (5) the actor at LATER steps, on the states the fused kernel itself reached: its recorded action must be the GEMM path's
    #     mean of the recorded observation plus sigma x the Philox draw of (env, t).  The draw is read back from the step kernel
    #     (mean 0, sigma 1 on a scratch engine with the same seed and stream id: the action it records IS eps).
    import ctypes as C
    Nn = tg._native
    scratch = tg.DeviceRollout(mk(), pol, G, Eps, seed=21, compute_dtype=torch.bfloat16, fused=False)
    scratch._seed_host, scratch._stream_host = 21, 0
    zeros, ones = torch.zeros(G * Eps, A, device=dev), (C.c_float * A)(*([1.0] * A))
    from trajopt_grpo_amd.mlp import GemmMLP
    mlp = GemmMLP(pol.actor, torch.bfloat16)
    for t in (1, 7, T // 2, T - 1):
        alive = fm[t].bool()
        if int(alive.sum()) == 0:
            continue
        scratch._enqueue_prepare(None)
        Nn.check(Nn.load().tg_rollout_step(C.byref(scratch.params), C.byref(scratch.traj.native()), t, zeros.data_ptr(), A, ones,
                                           scratch.rng.data_ptr(), 0, Nn.stream_ptr(dev)), "tg_rollout_step")
        eps_t = scratch.traj.act[:, t, :].t()                                   # [n][A]
        mean_t = mlp.forward(mlp.prepare_input(fo[:, t, :].t()), keep=False)    # GEMM-path actor on the fused kernel's own states
        want = mean_t + std * eps_t
        d = (fa[:, t, :].t() - want)[alive]
        assert float(d.abs().max()) < 0.03 * max(1.0, float(want[alive].abs().max())), (t, float(d.abs().max()))
        assert float(d.abs().mean()) < 5e-3, (t, float(d.abs().mean()))


@pytest.mark.parametrize("name,hidden", [("CartPole", (128, 128)), ("CartPole", (128, 128, 128, 128)), ("QuadPole2D", (128, 128, 128)),
                                         ("QuadPole", (64, 64)), ("QuadPole", (128,)), ("Pendulum", (64, 64, 64))])
@pytest.mark.parametrize("block_envs", [16, 32])
def test_fused_f32_rollout_matches_unfused_path(tg, dev, name, hidden, block_envs):
    """tg_fused_rollout_f32 (fp32 products on the matrix cores, register-resident weights; both forms: 32 envs per workgroup on
    v_mfma_f32_32x32x2_f32 and 16 on v_mfma_f32_16x16x4_f32) against the per-step path: same reset and Philox draws, means equal
    to fp32 summation order; its recorded trajectory replays bit-exactly through the teacher-forced step kernel; a rollout split
    into two launches equals the unsplit one."""
    S, A = DIMS[name]
    T, G, Eps = 40, 3, 43                       # 129 envs: full workgroups + one with a single env
    torch.manual_seed(8)
    pol = tg.GaussianActor_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
    mk = lambda: tg.environments.ENV_CLASSES[name](max_steps=T)
    fused = tg.DeviceRollout(mk(), pol, G, Eps, seed=22)
    fused.f32_block_envs = block_envs
    plain = tg.DeviceRollout(mk(), pol, G, Eps, seed=22, fused=False)
    assert fused.fused and fused._fused_f32 and not plain.fused
    tf = fused.run()
    assert fused._frag.block_envs == block_envs
    fo, fa, fr, fm, fl = (x.clone() for x in (tf.obs, tf.act, tf.rew, tf.mask, tf.len))
    tp = plain.run()
    assert torch.equal(tp.obs[:, 0, :], fo[:, 0, :])                           # same reset draw
    scale = max(1.0, float(fa[:, 0, :].abs().max()))
    assert float((tp.act[:, 0, :] - fa[:, 0, :]).abs().max()) < 2e-5 * scale    # same noise, fp32 means
    mean0 = pol.actor(fo[:, 0, :].t()).detach().double()
    std = float(np.sqrt(0.3))
    z = ((fa[:, 0, :].t().double() - mean0) / std).cpu().numpy()
    assert abs(z.mean()) < 0.25 and 0.7 < z.std() < 1.3
    # later steps: the two paths stay together until rounding flips a termination (rare at T = 40)
    both = (fm.bool() & tp.mask.bool())
    assert float(both.float().mean()) > 0.9 * float(fm.float().mean())
    assert float(((fr - tp.rew).abs() * both).max()) < 1e-2
    # teacher-forced replay of the recorded actions through the step kernel (same fp32 dynamics code)
    replay = plain.run(initial_states=fo[:, 0, :].t().cpu().numpy(), forced_actions=fa.permute(2, 1, 0).cpu().numpy())
    assert torch.equal(replay.len, fl) and torch.equal(replay.mask, fm)
    assert torch.equal(replay.obs, fo) and torch.equal(replay.rew, fr)     # dynamics compiled without FMA contraction: same bits
    m = fm.bool()
    assert torch.equal(fm.sum(0, dtype=torch.int32), fl) and tf.env_steps() == int(fm.sum())
    assert torch.all(fr[~m] == 0) and torch.all(fa[:, ~m] == 0) and torch.all(fo[:, :T][:, ~m] == 0)
    # split launch: [0, 13) then [13, T) gives the same bits as one launch
    split = tg.DeviceRollout(mk(), pol, G, Eps, seed=22)
    split.f32_block_envs = block_envs
    split._seed_host, split._stream_host = 22, 0
    with torch.cuda.device(dev):
        split._enqueue_prepare(None)
        split._enqueue_fused(0, 13)
        split.rng[1] -= 1                                   # _enqueue_fused advanced the stream id; same rollout continues
        split._enqueue_fused(13, T)
    torch.cuda.synchronize()
    assert torch.equal(split.traj.obs, fo) and torch.equal(split.traj.act, fa) and torch.equal(split.traj.rew, fr)
    assert torch.equal(split.traj.mask, fm)


def test_fused_rollout_auto_selection_and_manager(tg, dev):
    torch.manual_seed(7)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256, 256), cov=0.3, device=dev)
    mk = lambda: tg.QuadPole(max_steps=24)
    assert tg.DeviceRollout(mk(), pol, 1, 64, compute_dtype=torch.bfloat16).fused            # auto
    assert not tg.DeviceRollout(mk(), pol, 1, 64).fused                                       # fp32 policy: GEMM path
    assert not tg.DeviceRollout(mk(), pol, 1, 64, dtype=torch.float64, compute_dtype=torch.bfloat16).fused
    with pytest.raises(ValueError):
        tg.DeviceRollout(mk(), tg.GaussianActor_NeuralNetwork(20, 4, (64, 64), device=dev), 1, 64,
                         compute_dtype=torch.bfloat16, fused=True)
    mgr = tg.RolloutManager(mk, pol, num_workers=2, num_episodes_per_worker=128, compute_dtype=torch.bfloat16, seed=3)
    a = mgr.rollout()
    b = mgr.rollout()                      # second rollout: new Philox stream id, new initial states
    assert a[0].shape == (2, 128, 24, 20) and not torch.equal(a[1], b[1])
    assert torch.equal(a[4].sum(2), a[3])


# --------------------------------------------------------------------------------------------
# swarm (BASELINE config 5; build-defined semantics, no reference oracle beyond n_agents = 1)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused,cdt,block_envs", [(False, torch.bfloat16, None), (True, torch.bfloat16, None), (True, None, 16), (True, None, 32)])
def test_swarm_termination_couples_the_bodies_of_an_env(tg, dev, fused, cdt, block_envs):
    """cdt None + fused: the fp32 fused rollout kernel (tg_fused_rollout_f32, 16 / 32 envs per workgroup); bf16 + fused: tg_fused_rollout."""
    T, G, Eps, K = 48, 2, 16, 8
    torch.manual_seed(8)
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (128, 128), cov=0.3, device=dev)
    kw = dict(seed=5, compute_dtype=cdt, fused=fused)

    class _Roll:                                               # DeviceRollout with the fp32 kernel's block size fixed before run()
        def __call__(self, *args, **kwargs):
            eng = tg.DeviceRollout(*args, **kwargs)
            eng.f32_block_envs = block_envs
            return eng
    roll = _Roll()
    swarm = roll(tg.QuadPoleSwarm(n_agents=K, max_steps=T), pol, G, Eps, **kw).run()
    s_len, s_act, s_obs = swarm.len.clone(), swarm.act.clone(), swarm.obs.clone()
    assert swarm.n == G * Eps * K and swarm.E == Eps * K
    # the same env slots stepped as independent QuadPole bodies: same Philox keys, same initial states
    indep = roll(tg.QuadPole(max_steps=T), pol, G, Eps * K, **kw).run()
    assert torch.equal(indep.obs[:, 0, :], s_obs[:, 0, :])
    # every body of an env stops with the env, at the first step any of its bodies would have stopped alone
    per_env = s_len.view(-1, K)
    assert torch.equal(per_env, per_env[:, :1].expand_as(per_env))
    assert torch.equal(per_env[:, 0], indep.len.view(-1, K).min(dim=1).values)
    # until then the bodies evolve exactly as the independent ones
    t_keep = (torch.arange(T, device=dev)[:, None] < s_len[None, :])
    assert torch.equal(s_act[:, t_keep], indep.act[:, t_keep])
    assert (per_env[:, 0] < T).any() and (per_env[:, 0] > 1).all()
    # n_agents = 1 is the plain env, bit for bit
    one = roll(tg.QuadPoleSwarm(n_agents=1, max_steps=T), pol, G, Eps * K, **kw).run()
    for a, b in ((one.obs, indep.obs), (one.act, indep.act), (one.rew, indep.rew), (one.len, indep.len)):
        assert torch.equal(a, b)


def test_grpo_on_a_swarm_buffer_uses_group_statistics_across_bodies(tg, dev):
    T, G, Eps, K = 32, 4, 8, 8
    torch.manual_seed(9)
    pol = tg.GaussianActor_NeuralNetwork(20, 4, (128, 128), cov=0.3, device=dev)
    mgr = tg.RolloutManager(lambda: tg.QuadPoleSwarm(n_agents=K, max_steps=T), pol, restart=True, num_workers=G,
                            num_episodes_per_worker=Eps, compute_dtype=torch.bfloat16, seed=2)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    assert buf.group_observations.shape == (G, Eps * K, T, 20)
    tr = buf.device_traj
    rtg = tg.hip_ops.rtg_scan(tr.rew, tr.mask, 0.9)
    adv = tg.hip_ops.group_normalize(rtg, tr.mask, tg.hip_ops.masked_moments(rtg, tr.mask, tr.E), 0, tr.E)
    a = adv.t().reshape(G, -1)
    m = tr.mask.t().reshape(G, -1).bool()
    for g in range(G):                      # one mean / std per group, across all bodies of its episodes
        v = a[g][m[g]]
        assert abs(float(v.mean())) < 1e-4 and abs(float(v.std()) - 1) < 1e-3
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.9, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4),
                   updates_per_iter=1, autocast_dtype=torch.bfloat16)
    before = [p.detach().clone() for p in pol.parameters()]
    algo.learn(buf)
    assert np.isfinite(algo.last_stats["J"]).all()
    assert any(not torch.equal(x, y) for x, y in zip(before, pol.parameters()))


@pytest.mark.parametrize("env_name,kw,dtype", [("CartPole", dict(max_steps=300), torch.float32), ("QuadPole", dict(max_steps=120), torch.float32),
                                               ("QuadPole", dict(max_steps=40), torch.float64), ("QuadPole2D", dict(max_steps=50), torch.float32),
                                               ("Pendulum", dict(max_steps=130), torch.float64), ("QuadPoleSwarm", dict(n_agents=4, max_steps=64), torch.float32)])
def test_forced_rollout_in_one_launch_is_bit_identical_to_the_per_step_launches(tg, dev, env_name, kw, dtype):
    """tg_rollout_forced (every time step of a teacher-forced replay in one launch, the state in registers) against T launches of
    tg_rollout_step on the same recorded actions: every tensor of the trajectory bit for bit -- ragged episode ends, the swarm's
    segmented termination, Pendulum's balanced-step count carried in `len`, f32 and f64 -- also when the range is split in two."""
    from trajopt_grpo_amd import rollout as RO
    Nn = tg._native
    env_cls = getattr(tg, env_name)
    torch.manual_seed(2)
    if env_name == "Pendulum":
        kw = dict(kw, gravity=0.0)                               # (no gravity: a near-silent policy balances and TERMINATES)
    env = env_cls(**kw)
    pol = tg.GaussianActor_NeuralNetwork(env.obs_dim, env.act_dim, (32, 32), cov=1.5 if env_name != "Pendulum" else 1e-4, device=dev)
    G, E = 5, 52                                                 # 260 envs: not a multiple of 64
    src = tg.DeviceRollout(env, pol, G, E, dtype=dtype, seed=11, fused=False)
    rec = src.run()
    init = rec.obs[:, 0, :].t().cpu().numpy()
    acts = rec.act.permute(2, 1, 0).cpu().numpy()                # (n, T, A)
    if env_name.startswith("QuadPole") and dtype == torch.float32:
        assert int(rec.len.min()) < env.max_steps, "the policy is meant to end some episodes early"

    def replay(per_step, split=False):
        if True:
            eng = tg.DeviceRollout(env, pol, G, E, dtype=dtype, seed=11, fused=False)
            eng.forced_per_step = per_step
            if not split:
                tr = eng.run(initial_states=init, forced_actions=acts)
            else:                                                # two launches: [0, T/2) then [T/2, T)
                eng.params = env.native_params()
                eng._seed_host, eng._stream_host = 11, 0
                with torch.cuda.device(dev):
                    eng._enqueue_prepare(init)
                    eng.traj.act.copy_(torch.as_tensor(acts, dtype=torch.float32).permute(2, 1, 0).to(dev))
                    st, nat = Nn.stream_ptr(dev), eng.traj.native()
                    half = eng.T // 2
                    Nn.check(Nn.load().tg_rollout_forced(C.byref(eng.params), C.byref(nat), 0, half, st))
                    Nn.check(Nn.load().tg_rollout_forced(C.byref(eng.params), C.byref(nat), half, eng.T, st))
                tr = eng.traj
            torch.cuda.synchronize()
            return [t.clone() for t in (tr.obs, tr.rew, tr.mask, tr.len)]

    a, b, c2 = replay(True), replay(False), replay(False, split=True)
    for x, y, z, name in zip(a, b, c2, ("obs", "rew", "mask", "len")):
        assert torch.equal(x, y), f"{name}: one launch differs from the per-step launches"
        assert torch.equal(x, z), f"{name}: a split range differs"
    assert torch.equal(a[2], rec.mask) and torch.equal(a[3], rec.len)      # (and both replay the sampled rollout's lengths)


# --------------------------------------------------------------------------------------------
# the learner's prologue as native launches (csrc/learn_kernels.hip)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,n,E", [(1, 1, 1), (63, 31, 31), (64, 32, 8), (65, 33, 11), (128, 100, 25), (500, 4096, 64), (256, 96, 32), (300, 4100, 4100),
                                   (1023, 70, 35)])     # 1023 = tg_returns_moments_max_horizon(): the whole 160 KiB of LDS
@pytest.mark.parametrize("gamma", [0.5, 0.999])
@pytest.mark.parametrize("padding", ["garbage", "zeros_short"])
def test_returns_moments_is_bit_identical_to_the_standalone_kernels(tg, dev, T, n, E, gamma, padding):
    """tg_returns_moments (LDS-staged strips, one lane per env on the recurrence: the form for a few thousand envs) against
    tg_rtg_scan + tg_masked_moments, which are pinned to the reference (grpo.py:66-74,110-115): same bits, ragged episode lengths,
    strips that do not divide the horizon, env counts that do not fill a workgroup, masks with holes.  padding = zeros_short: short
    episodes with zero rewards behind them, as a rollout writes them -- the kernel's scans then stop at each block's last live
    step; garbage: non-zero rewards under a zero mask everywhere (every step is live)."""
    K = tg.hip_ops
    g = torch.Generator(device="cpu").manual_seed(T * 1000 + n)
    rew = torch.randn(T, n, generator=g)
    lens = torch.randint(0, (T + 1) if padding == "garbage" else max(T // 5, 1) + 1, (n,), generator=g)
    mask = (torch.arange(T).view(T, 1) < lens.view(1, n))
    if padding == "zeros_short":
        rew = rew * mask
        rew[0, ::5] = -0.0 if T > 1 else rew[0, ::5]             # (a negative zero under a set mask bit is live too)
    rew = rew.to(dev)
    if T > 2:
        mask[T // 2 if padding == "garbage" else max(T // 10, 1) - 1, ::7] = False     # a hole: the reference's recurrence cuts the carry there
    mask = mask.to(torch.uint8).to(dev)
    rtg0 = K.rtg_scan(rew, mask, gamma)
    mom0 = K.masked_moments(rtg0, mask, E)
    rtg1, mom1 = K.returns_moments(rew, mask, gamma, E)
    torch.cuda.synchronize()
    assert torch.equal(rtg0, rtg1)
    assert torch.equal(mom0.view(torch.int64), mom1.view(torch.int64))       # bit for bit (NaN-safe)


@pytest.mark.parametrize("T,n", [(1, 1), (7, 3), (64, 65), (129, 300), (500, 80), (256, 4100)])
@pytest.mark.parametrize("monte_carlo", [True, False])
def test_ppo_prologue_kernels_are_bit_identical_to_the_sequence_they_replace(tg, dev, T, n, monte_carlo):
    """PPO's prologue without a host round trip (tg_scatter_rows / tg_ppo_returns / tg_ppo_norm / tg_gather_rows2) against what
    learn() ran before: zeros + index_copy_, tg_rtg_scan + `rtg - V` or tg_gae_scan (both pinned to the reference: ppo.py:100-124),
    two tg_masked_moments, torch's fp64 / fp32 arithmetic for mean / 1 / (std + 1e-8) (ppo.py:138-139) and the host's 1 / n, two
    index_selects -- same bits everywhere, ragged lengths, masks with holes."""
    K = tg.hip_ops
    g = torch.Generator(device="cpu").manual_seed(T * 131 + n)
    lens = torch.randint(0, T + 1, (n,), generator=g)
    lens[0] = T
    mask = (torch.arange(T).view(T, 1) < lens.view(1, n))
    if T > 2:
        mask[T // 2, ::7] = False
    rew = (torch.randn(T, n, generator=g) * mask).to(dev)
    mask = mask.to(torch.uint8).to(dev)
    idx = mask.reshape(-1).nonzero().squeeze(1)
    rows = idx.numel()
    v_rows = torch.randn(rows, 4, generator=g).to(dev)                     # (the critic's padded output: column 0 is the value)
    gamma, lam, c1, klc = 0.99, 0.95, 0.5, 0.3
    # ---- before
    V0 = torch.zeros(T * n, device=dev)
    V0.index_copy_(0, idx, v_rows[:, 0].contiguous())
    V0 = V0.view(T, n)
    if monte_carlo:
        ret0 = K.rtg_scan(rew, mask, gamma)
        adv0 = ret0 - V0
    else:
        adv0, ret0 = K.gae_scan(rew, V0, mask, gamma, lam)
    m0 = torch.cat([K.masked_moments(adv0, mask, n), K.masked_moments(ret0, mask, n)])
    cnt, s1, s2 = m0[:, 0], m0[:, 1], m0[:, 2]
    mean = s1 / cnt
    std = torch.sqrt(torch.clamp((s2 - s1 * mean) / (cnt - 1.0), min=0.0)).float()
    inv = 1.0 / (std + 1e-8)
    norm0 = torch.stack([mean[0].float(), inv[0], mean[1].float(), inv[1]])
    n_glob = float(cnt[0].item())
    coef0 = torch.tensor([-1.0 / n_glob, c1 / n_glob, klc / n_glob, n_glob], dtype=torch.float64).float() if n_glob > 0 else None
    # ---- after
    V1 = torch.zeros(T, n, device=dev)
    K.scatter_rows(v_rows, idx, V1)
    adv1, ret1 = torch.empty(T, n, device=dev), torch.empty(T, n, device=dev)
    m1 = K.ppo_returns(rew, V1, mask, gamma, lam, monte_carlo, adv1, ret1)
    norm8 = K.ppo_norm(m1, c1, klc)
    a_rows, r_rows = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    K.gather_rows2(idx, adv1, a_rows, ret1, r_rows)
    torch.cuda.synchronize()
    bits = lambda t: t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int64)
    assert torch.equal(V0, V1)
    assert torch.equal(bits(adv0), bits(adv1)) and torch.equal(bits(ret0), bits(ret1))
    assert torch.equal(bits(m0), bits(m1))
    assert torch.equal(bits(norm0), bits(norm8[:4])), (norm0.tolist(), norm8.tolist())       # (NaN-safe: one valid row -> NaN std, like torch)
    assert torch.equal(bits(coef0.to(dev)), bits(norm8[4:]))
    assert torch.equal(a_rows, adv0.reshape(-1).index_select(0, idx)) and torch.equal(r_rows, ret0.reshape(-1).index_select(0, idx))


def test_returns_moments_horizon_limit(tg, dev):
    """The advertised limit launches (the kernel's LDS is dynamic to the last byte), one step more is refused with a message --
    and GRPO.learn() at that horizon takes tg_rtg_scan + tg_masked_moments instead of raising."""
    K = tg.hip_ops
    Tmax = K.returns_moments_max_horizon()
    assert Tmax == 1023
    rew = torch.randn(Tmax + 1, 64, device=dev)
    mask = torch.ones(Tmax + 1, 64, dtype=torch.uint8, device=dev)
    K.returns_moments(rew[:Tmax].contiguous(), mask[:Tmax].contiguous(), 0.9, 32)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="horizon"):
        K.returns_moments(rew, mask, 0.9, 32)
    torch.manual_seed(0)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (64, 64), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=Tmax + 1), pol, num_workers=2, num_episodes_per_worker=8, seed=1)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    algo = tg.GRPO(0.15, 0.5, 0.5, pol, torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=1)
    algo.learn(buf)
    assert all(map(lambda v: v == v, algo.last_stats["J"]))


@pytest.mark.parametrize("env_name,S,A,hidden,cdt,dtype", [("CartPole", 5, 1, (128, 128), None, torch.float32),
                                                           ("QuadPole", 20, 4, (256, 256, 256), torch.bfloat16, torch.float32),
                                                           ("QuadPole", 20, 4, (64, 64), None, torch.float64),
                                                           ("QuadPole2D", 10, 2, (48, 48), None, torch.float32)])
def test_learn_prepare_matches_the_torch_prologue(tg, dev, env_name, S, A, hidden, cdt, dtype):
    """tg_learn_count + tg_learn_compact against what they replace (mask.nonzero(), three index_selects, prepare_input, the [T][n]
    group normalisation): the same rows in the same (time-major) order, the same bits -- bf16 chain input with its ones column,
    fp32 chain input padded to 8, per-layer fp32 input padded to 32, f64 trajectories; then the count check."""
    from trajopt_grpo_amd import mlp as M
    K = tg.hip_ops
    T, G, E = 48, 6, 20
    torch.manual_seed(1)
    pol = tg.GaussianActor_NeuralNetwork(S, A, hidden, cov=0.6, device=dev)
    env_cls = getattr(tg, env_name)
    mgr = tg.RolloutManager(lambda: env_cls(max_steps=T), pol, num_workers=G, num_episodes_per_worker=E, seed=3, dtype=dtype,
                            compute_dtype=cdt, fused=False if dtype == torch.float64 else None)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    tr = buf.device_traj
    tr.mask[T // 3, ::5] = 0                                     # holes too: the compaction follows the mask, not the lengths
    rows_true = int(tr.mask.sum())
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.9, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=1,
                   autocast_dtype=cdt)
    m = algo._mlp(pol.actor)
    rew = tr.rew.float()
    rtg = K.rtg_scan(rew, tr.mask, 0.9)
    mom = K.masked_moments(rtg, tr.mask, tr.E)
    tr.host_valid_rows = None                                    # (the statistic predates the holes: count on the device, read it back)
    idx, xin, act, adv, ret = algo._prepare(tr, m, src0=rtg, moments=mom, norm_mode=0, group_size=tr.E, src1=rtg)
    torch.cuda.synchronize()
    idx0 = tr.mask.reshape(-1).nonzero().squeeze(1)
    assert idx.numel() == rows_true and torch.equal(idx, idx0)
    X0 = tr.obs_rows().index_select(0, idx0).float()
    xin0 = m.prepare_input(X0)
    assert xin.dtype == xin0.dtype and xin.shape == xin0.shape
    assert torch.equal(xin.view(torch.int16 if xin.dtype == torch.bfloat16 else torch.int32), xin0.view(torch.int16 if xin.dtype == torch.bfloat16 else torch.int32))
    assert M.has_ones_column(xin) == M.has_ones_column(xin0)
    assert torch.equal(act, tr.act_rows().index_select(0, idx0))
    adv0 = K.group_normalize(rtg, tr.mask, mom, 0, tr.E).reshape(-1).index_select(0, idx0)
    assert torch.equal(adv.view(torch.int32), adv0.view(torch.int32)) and torch.equal(ret, rtg.reshape(-1).index_select(0, idx0))
    # the host-known count: right -> no complaint; wrong -> the NEXT learn() entry raises, nothing is written out of bounds
    tr.host_valid_rows = lambda: rows_true
    algo._prepare(tr, m)
    algo._check_row_count()
    tr.host_valid_rows = lambda: rows_true - 3
    idx_short = algo._prepare(tr, m)[0]
    assert idx_short.numel() == rows_true - 3 and torch.equal(idx_short, idx0[:rows_true - 3])
    with pytest.raises(RuntimeError, match="valid rows"):
        algo._check_row_count()


def test_learn_compaction_at_c3_size(tg, dev):
    """tg_learn_count + tg_learn_compact on a full BASELINE configs[2] rollout (65,536 QuadPole envs x 256 steps, ~6 M valid rows of
    16.8 M entries) through properties that do not need a second copy of the gather: the row indices are strictly increasing, all
    valid, as many as the mask holds (hence exactly `mask.nonzero()`); sampled input rows are the bf16 observation with the ones
    column and zero padding; sampled action rows are the recorded actions; the count agrees with the rollout's statistic."""
    T, G, Eps = 256, 256, 256
    torch.manual_seed(3)
    pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
    mgr = tg.RolloutManager(lambda: tg.QuadPole(max_steps=T), pol, num_workers=G, num_episodes_per_worker=Eps, seed=5, compute_dtype=torch.bfloat16)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    tr = buf.device_traj
    algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None, updates_per_iter=1, gamma=0.999,
                  batch_size=None, autocast_dtype=torch.bfloat16)
    idx, xin, act, _, _ = algo._prepare(tr, algo._mlp(pol.actor))
    algo._check_row_count()                                     # (the mask's own count equals the rollout's statistic)
    rows = int(tr.mask.sum())
    assert idx.numel() == rows == buf.valid_rows() and 0 < rows < T * G * Eps
    assert bool((idx[1:] > idx[:-1]).all()) and int(idx[0]) >= 0 and int(idx[-1]) < T * G * Eps
    assert bool(tr.mask.reshape(-1)[idx].all())
    pick = torch.randint(0, rows, (200000,), device=dev)
    want = tr.obs_rows().index_select(0, idx[pick]).to(torch.bfloat16)
    got = xin[pick]
    assert torch.equal(got[:, :20].view(torch.int16), want.view(torch.int16))
    assert bool((got[:, 20:31] == 0).all()) and bool((got[:, 31] == 1).all())
    assert torch.equal(act[pick], tr.act_rows().index_select(0, idx[pick]))


@pytest.mark.parametrize("kind,cdt,hidden", [("grpo", None, (128, 128)), ("grpo", torch.bfloat16, (128, 128, 128)), ("ppo", torch.bfloat16, (256, 256, 256)),
                                             ("ppo", None, (64, 64)), ("ppo", None, (40, 40)), ("ppo_gae", None, (64, 64)),
                                             ("ppo_gae", torch.bfloat16, (128, 128, 128))])
def test_learn_is_bit_identical_with_and_without_the_native_prologue(tg, dev, kind, cdt, hidden):
    """learn() on the native prologue (tg_returns_moments / tg_learn_count / tg_learn_compact) against the same learn() on the torch
    prologue that nets outside its gate take (mask.nonzero(), index_selects, prepare_input; forced here by patching the gate shut):
    identical weights, bit for bit, after two updates."""
    from trajopt_grpo_amd import algorithms as Alg
    gate = Alg._GpuLearner._prepare_enqueue

    def run(native):
        if not native:
            Alg._GpuLearner._prepare_enqueue = lambda self, *a, **k: None
        try:
            torch.manual_seed(11)
            cls = tg.GaussianActorCritic_NeuralNetwork if kind.startswith("ppo") else tg.GaussianActor_NeuralNetwork
            pol = cls(20, 4, hidden, cov=0.3, device=dev)
            mgr = tg.RolloutManager(lambda: tg.QuadPole(max_steps=120), pol, num_workers=6, num_episodes_per_worker=40, seed=5, compute_dtype=cdt,
                                    restart=kind == "grpo")
            buf = tg.Rollout_Buffer(mgr)
            buf.sample()
            opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
            if kind.startswith("ppo"):
                algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, gamma=0.99, batch_size=None, autocast_dtype=cdt,
                              monte_carlo=kind == "ppo")
            else:
                algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.9, policy=pol, optimizer=opt, updates_per_iter=2, autocast_dtype=cdt)
            algo.learn(buf)
            buf.sample()
            algo.learn(buf)
            torch.cuda.synchronize()
            return [p.detach().clone() for p in pol.parameters()], algo.last_stats
        finally:
            Alg._GpuLearner._prepare_enqueue = gate

    (wa, sa), (wb, sb) = run(True), run(False)
    assert sa["n_valid"] == sb["n_valid"] and sa["n_valid"] < 6 * 40 * 120          # (episodes do end early: ragged rows)
    for a, b in zip(wa, wb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cdt,hidden", [(None, (128, 128)), (torch.bfloat16, (128, 128, 128))])
@pytest.mark.parametrize("kind", ["grpo", "ppo"])
def test_first_update_can_stand_in_for_the_old_policy_pass(tg, dev, kind, cdt, hidden):
    """When old_policy still IS the policy (grpo.py:148 copied it and nothing touched either since; PPO takes the old log-probabilities
    from the current policy anyway, ppo.py:142-143) the fp32 chain learner's first update writes the old log-probabilities instead of
    a no-grad pass computing them: the ratio of that update is exactly 1 -- as in the reference, whose two passes are the same
    arithmetic.  Same result as the explicit pass up to the last bits of that ratio; a perturbed old policy is NOT folded."""
    from trajopt_grpo_amd import algorithms as Alg
    K = tg.hip_ops

    def run(fold, perturb_old=False, iters=2):
        Alg._FOLD_OLD_LOGP = fold
        try:
            torch.manual_seed(21)
            cls = tg.GaussianActorCritic_NeuralNetwork if kind == "ppo" else tg.GaussianActor_NeuralNetwork
            pol = cls(5, 1, hidden, cov=0.5, device=dev)
            mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=60), pol, num_workers=8, num_episodes_per_worker=32, seed=9, compute_dtype=cdt)
            buf = tg.Rollout_Buffer(mgr)
            opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
            if kind == "ppo":
                algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=3, gamma=0.99, batch_size=None, autocast_dtype=cdt)
            else:
                algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=opt, updates_per_iter=3, autocast_dtype=cdt)
            m_ = algo._mlp(pol.actor)
            assert (m_._f32 is not None) if cdt is None else (m_._chain is not None and m_._bchain is not None)
            seen = []
            plain = Alg._GpuLearner._logp_nograd
            algo._logp_nograd = lambda *a, **k: (seen.append(1), plain(algo, *a, **k))[1]
            for it in range(iters):
                buf.sample()
                if perturb_old and kind == "grpo":
                    with torch.no_grad():
                        for p_ in algo.old_policy.parameters():
                            p_.add_(1e-3)
                algo.learn(buf)
            torch.cuda.synchronize()
            return [p.detach().clone() for p in pol.parameters()], algo.last_stats, len(seen)
        finally:
            Alg._FOLD_OLD_LOGP = True

    (wf, sf, nf), (we, se, ne) = run(True), run(False)
    assert nf == 0 and ne == 2, "the folded run must not run the no-grad pass, the explicit run one per learn()"
    for a, b in zip(wf, we):
        assert float((a - b).norm()) <= (2e-6 if cdt is None else 5e-4) * float(b.norm()) + 1e-9
    key = "J" if kind == "grpo" else "total_loss"
    np.testing.assert_allclose(sf[key], se[key], rtol=1e-5 if cdt is None else 2e-3, atol=1e-7 if cdt is None else 1e-5)
    if kind == "grpo":
        (wp, _, n_p), (wq, _, n_q) = run(True, perturb_old=True), run(False, perturb_old=True)
        assert n_p == n_q == 2, "an old policy that was written since the copy must get its own pass"
        for a, b in zip(wp, wq):
            assert torch.equal(a, b)


@pytest.mark.parametrize("kind,cdt,hidden,graph", [("grpo", None, (128, 128), None), ("ppo", torch.bfloat16, (256, 256, 256), None), ("ppo", None, (40, 40), None),
                                                   ("ppo", None, (40, 40), False),          # the EAGER per-step engine (it trusted the keys)
                                                   ("ppo", None, (256, 256), False)])       # ... with the H = 256 fp32 chain's no-grad pass under it
def test_weights_written_through_data_are_seen_by_the_next_rollout_and_learn(tg, dev, kind, cdt, hidden, graph):
    """a write through `param.data` moves no version counter, so keys alone would leave every derived layout (the
    learner's streams, the fused fp32 rollout's register stream) stale.  Layouts are rebuilt at every learn() / rollout entry
    whatever the keys say: a run whose weights are clamped through `.data` after every learn() must equal, bit for bit, the run
    that clamps through torch (which does move the counters)."""
    def run(through_data):
        torch.manual_seed(31)
        cls = tg.GaussianActorCritic_NeuralNetwork if kind == "ppo" else tg.GaussianActor_NeuralNetwork
        S, A, env = (5, 1, lambda: tg.CartPole(max_steps=40)) if cdt is None else (20, 4, lambda: tg.QuadPole(max_steps=40))
        pol = cls(S, A, hidden, cov=0.4, device=dev)
        mgr = tg.RolloutManager(env, pol, num_workers=4, num_episodes_per_worker=32, seed=2, compute_dtype=cdt, use_graph=graph)
        buf = tg.Rollout_Buffer(mgr)
        opt = torch.optim.Adam(pol.parameters(), lr=1e-2)
        algo = (tg.PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, gamma=0.99, batch_size=None, autocast_dtype=cdt)
                if kind == "ppo" else tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=opt, updates_per_iter=2, autocast_dtype=cdt))
        trajs = []
        for it in range(3):
            buf.sample()
            trajs.append(buf.device_traj.act.clone())
            algo.learn(buf)
            with torch.no_grad():
                for p in pol.parameters():                 # weight surgery after every update, the way `.data` users do it
                    if through_data:
                        p.data.mul_(0.5)
                    else:
                        p.mul_(0.5)
            if kind == "grpo":
                algo.sync_old_policy()                     # (old_policy follows: this test is about the layouts, the next one about the fold)
        buf.sample()
        trajs.append(buf.device_traj.act.clone())
        torch.cuda.synchronize()
        return trajs, [p.detach().clone() for p in pol.parameters()]

    (ta, wa), (tb, wb) = run(True), run(False)
    for it, (x, y) in enumerate(zip(ta, tb)):
        assert torch.equal(x, y), f"rollout {it} acted with stale weights after a write through .data"
    for a, b in zip(wa, wb):
        assert torch.equal(a, b)


def test_folded_old_policy_pass_notices_weights_changed_behind_the_keys(tg, dev):
    """GRPO lets the first update stand in for the old policy's pass while the version keys say old_policy is the policy.  A write
    through `.data` breaks that silently: the bitwise comparison enqueued with the fold reports it at the end of that learn() (the
    optimizer steps have been applied by then: the error says so; ADVICE r04 asked for it no later than that)."""
    torch.manual_seed(3)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=40), pol, num_workers=4, num_episodes_per_worker=32, seed=2)
    buf = tg.Rollout_Buffer(mgr)
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=2)
    buf.sample(); algo.learn(buf)
    assert np.isfinite(algo.last_stats["J"]).all()           # (old_policy is the policy: no complaint)
    buf.sample(); algo.learn(buf)
    algo.last_stats
    with torch.no_grad():
        pol.actor.network[0].weight.data.add_(0.25)          # behind the version counters
    buf.sample()
    with pytest.raises(RuntimeError, match="different weights"):
        algo.learn(buf)


def test_env_dynamics_members_match_the_reference(tg, dev):
    """`Env._dynamics(state, control)` / `_propegate*` as callable members (environments/cartpole_env.py:52-100,
    quadrotor_env.py:417-528, :578-585, :1024-1130; the reference's tests/test_cartpole.py:36-40 calls `env._dynamics`): one f64
    tg_env_step on temporaries, against inputs / outputs read off the imported reference (oracle/tools/gen_members.py)."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_public_members.json")) as f:
        calls = json.load(f)["calls"]
    for name in ("CartPole", "QuadPole", "QuadPole2D"):
        env = getattr(tg, name)(device=dev)
        env.reset()
        before = env._get_obs().copy()
        for c in calls[name]:
            st, u, want = np.array(c["state"]), np.array(c["wrapped"], dtype=np.float32), np.array(c["next"])
            got = env._dynamics(st, u)
            assert isinstance(got, np.ndarray) and got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13, err_msg=name)
        assert np.array_equal(env._get_obs(), before), "_dynamics must not move the env"
        # the propagate helper = state_dict <- _dynamics(own state, control)
        u = env._wrap_action(np.zeros(env.act_dim, dtype=np.float32) + np.float32(0.25))
        want = env._dynamics(before, u)
        {"CartPole": lambda: env._propegate_cartpole(env.state_dict["cartpole"], u), "QuadPole": lambda: env._propegate(u),
         "QuadPole2D": lambda: env._propogate(u)}[name]()
        np.testing.assert_array_equal(env._get_obs(), want)
    env = tg.CartPole(device=dev)
    assert env._dynamics(np.array([0, 0, 0, 1, 0]), 0.0).shape == (5,)         # (a scalar control, as the reference's own test passes it)


# --------------------------------------------------------------------------------------------
# edge shapes
# --------------------------------------------------------------------------------------------
def test_degenerate_shapes(tg, dev):
    """1 env x 1 step; horizon 1; empty launches through the C ABI."""
    K, Nn = tg.hip_ops, tg._native
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (8,), cov=0.5, device=dev)
    tr = tg.DeviceRollout(tg.CartPole(max_steps=1), pol, 1, 1, seed=1).run()
    assert tr.len.tolist() == [1] and tr.mask.tolist() == [[1]] and tr.env_steps() == 1
    obs, act, rew, ln, mask = tr.to_reference()
    assert obs.shape == (1, 1, 1, 5) and ln.shape == (1, 1) and float(ln) == 1.0
    # a single valid step per episode: RTG == reward; GRPO's unbiased std over one sample is NaN, exactly like torch
    r = K.rtg_scan(tr.rew, tr.mask, 0.9)
    assert torch.equal(r, tr.rew)
    adv = K.group_normalize(r, tr.mask, K.masked_moments(r, tr.mask, 1), 0, 1)
    assert torch.isnan(adv).all() and torch.isnan(torch.std(r.reshape(-1) + 1e-8))
    # n == 0 is a no-op, not an error
    lib = Nn.load()
    z = torch.empty(0, device=dev)
    assert lib.tg_rtg_scan(z.data_ptr() or 1, z.data_ptr() or 1, 0.5, z.data_ptr() or 1, 0, 4, Nn.stream_ptr(dev)) == 0
    p = tg.CartPole(max_steps=4).native_params()
    assert lib.tg_env_reset(C.byref(p), 0, 1, 0, 0, 1, 1, 0, 1, Nn.stream_ptr(dev)) == 0
    # argument validation happens on the host
    assert lib.tg_rollout_step(C.byref(p), C.byref(tr.native()), 7, None, 0, None, None, 0, Nn.stream_ptr(dev)) == -1
    assert b"outside horizon" in lib.tg_last_error()


def test_reference_learner_can_consume_the_device_buffer(tg, dev):
    """Mixing components across the seam: the buffer serves the reference-layout CPU tensors, so a CPU learner written
    against the reference's attribute names (here: the oracle's restatement of GRPO.learn) runs on a GPU rollout."""
    torch.manual_seed(10)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (16, 16), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=24), pol, num_workers=2, num_episodes_per_worker=8, seed=4)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    cpu_pol = L.OraclePolicy(5, 1, (16, 16), cov=0.5)
    cpu_pol.load_state_dict({k: v.cpu() for k, v in pol.state_dict().items()})
    old = L.OraclePolicy(5, 1, (16, 16), cov=0.5)
    old.load_state_dict(cpu_pol.state_dict())
    opt = torch.optim.Adam(cpu_pol.parameters(), lr=3e-4)
    Js = L.grpo_learn(cpu_pol, old, opt, buf.group_observations, buf.group_actions, buf.group_rewards, buf.group_masks,
                      epsilon=0.15, gamma=0.5, updates_per_iter=1)
    # ... and the GPU learner on the same buffer and weights reaches the same J and the same step
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4),
                   updates_per_iter=1)
    algo.learn(buf)
    np.testing.assert_allclose(algo.last_stats["J"], Js, rtol=1e-3, atol=1e-5)
    for (k, p), q in zip(pol.actor.named_parameters(), cpu_pol.actor.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=0, atol=5e-5)


# --------------------------------------------------------------------------------------------
# Pendulum: the env whose episodes terminate (time_balanced > 5)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["default", "custom"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_pendulum_step_matches_reference_golden(tg, dev, tag, dtype, tol):
    g = load_golden(f"env_step_pendulum_{tag}.npz")
    kw = {k[len("param_"):]: float(g[k]) for k in g if k.startswith("param_")}
    nx, rw, tr, sp, tb = native_step(tg, "Pendulum", g["state"], g["action"], g["steps"], g["time_balanced"],
                                     int(g["max_steps"]), dtype, dev, **kw)
    scale = 1.0 if dtype == torch.float64 else 20.0          # |thetadot| <= 10 + one step
    np.testing.assert_allclose(nx, g["next_state"], rtol=tol, atol=tol * scale)
    assert np.all(np.abs(rw - g["reward"]) <= tol * 50 * np.maximum(1.0, np.abs(g["reward"])))
    assert np.array_equal(tr, g["truncated"]) and np.array_equal(sp, g["steps"] + 1)
    if dtype == torch.float64:
        np.testing.assert_allclose(tb, g["time_balanced_after"], rtol=0, atol=1e-12)
        assert np.array_equal(tb > 5.0, g["terminated"])


def test_pendulum_scalar_api_returns_truncated_before_terminated(tg, dev):
    g = load_golden("env_step_pendulum_default.npz")
    env = tg.Pendulum(max_steps=int(g["max_steps"]))
    assert env.observation_space.shape == (3,) and env.action_space.shape == (1,)
    i = int(np.flatnonzero(g["terminated"])[0])
    k = int(round(g["time_balanced"][i] / 0.05))
    env.set_state(g["state"][i])
    env._steps = int(g["steps"][i])
    env._steps_t.fill_(int(g["steps"][i]))
    tbv = 0
    for _ in range(k):
        tbv = tbv + 0.05
    env._tb_t.fill_(tbv)
    obs, rew, truncated, terminated, info = env.step(g["action"][i])
    np.testing.assert_allclose(obs, g["next_state"][i], rtol=1e-11, atol=1e-12)
    assert rew == pytest.approx(float(g["reward"][i]), abs=1e-10)
    assert (truncated, terminated) == (bool(g["truncated"][i]), True)
    assert info["time_balanced"] == pytest.approx(float(g["info_time_balanced"][i]), abs=1e-12) and info["time_balanced"] > 5
    o0, _ = env.reset()
    assert abs(np.arctan2(o0[0], o0[1])) > np.pi - 0.0501 and o0[2] == 0          # near upright unless swingup


@pytest.mark.parametrize("tag,T,ends", [("fall", 64, 64), ("hold", 140, 101)])
@pytest.mark.parametrize("fused", [False, True])
def test_pendulum_rollout_matches_reference(tg, dev, tag, T, ends, fused):
    """Teacher-forced (per-step kernels) and sampled fused rollouts.  'hold' (no gravity, near-silent policy) ends by
    TERMINATION after 101 consecutive balanced steps; the count lives in len as a negative number while running."""
    g = load_golden("rollout_pendulum.npz")
    obs, act, rew, ln, mask = (g[f"{tag}_{k}"] for k in ("obs", "act", "rew", "len", "mask"))
    G, Eps = ln.shape
    env = tg.Pendulum(max_steps=T, gravity=float(g[f"{tag}_gravity"]))
    pol = tg.GaussianActor_NeuralNetwork(3, 1, (32, 32), cov=float(g[f"{tag}_cov"][0]), device=dev)
    pol.load_state_dict({k[len(f"{tag}_policy."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"{tag}_policy.")})
    init = obs[:, :, 0, :].reshape(G * Eps, 3).astype(np.float64)
    if not fused:
        eng = tg.DeviceRollout(env, pol, G, Eps, dtype=torch.float64, seed=0)
        tr = eng.run(initial_states=init, forced_actions=act.reshape(G * Eps, T, 1))
        o2, a2, r2, l2, m2 = tr.to_reference()
        assert np.array_equal(l2.numpy(), ln) and np.array_equal(m2.numpy(), mask) and np.all(ln == ends)
        np.testing.assert_allclose(o2.numpy(), obs, rtol=0, atol=5e-4)
        np.testing.assert_allclose(r2.numpy(), rew, rtol=2e-4, atol=2e-4)
    else:
        # sampled actions on the fused kernel (128-wide policy so that it is supported): same dynamics, own noise
        pol = tg.GaussianActor_NeuralNetwork(3, 1, (128, 128), cov=float(g[f"{tag}_cov"][0]), device=dev)
        with torch.no_grad():
            for prm in pol.actor.network[-1].parameters():
                prm.mul_(1e-3)
        runs = []
        for use_fused in (True, False):
            eng = tg.DeviceRollout(env, pol, 4, 96, seed=3, compute_dtype=torch.bfloat16, fused=use_fused)
            tr = eng.run(initial_states=np.tile(init, (64, 1)))
            torch.cuda.synchronize()
            runs.append(tr)
            lens = tr.len.cpu().numpy()
            assert np.all(lens > 0)
            if tag == "hold":
                # terminated by balance on both paths (a few envs may drift out of the band under their own noise and
                # run on: then they end later, never earlier)
                assert np.mean(lens == 101) > 0.9 and np.all(lens >= 101)
            else:
                assert np.all(lens == T)
        # the two paths draw the same noise but round the policy differently (bf16 MFMA chain vs bf16 GEMMs)
        assert float((runs[0].len == runs[1].len).float().mean()) > 0.97
        same = (runs[0].len == runs[1].len).cpu().numpy()
        np.testing.assert_allclose(runs[0].rew.cpu().numpy()[:, same], runs[1].rew.cpu().numpy()[:, same], rtol=0, atol=5e-2)


def test_pendulum_balanced_count_survives_a_split_fused_rollout(tg, dev):
    """tg_fused_rollout in two segments [0, 50) + [50, T): the consecutive-balanced-step count crosses the boundary in
    d_len (negative while the episode runs), so the termination step and every recorded byte are those of one launch."""
    g = load_golden("rollout_pendulum.npz")
    T = 140
    env = tg.Pendulum(max_steps=T, gravity=0.0)
    pol = tg.GaussianActor_NeuralNetwork(3, 1, (128, 128), cov=1e-4, device=dev)
    with torch.no_grad():
        for prm in pol.actor.network[-1].parameters():
            prm.mul_(1e-3)
    init = np.tile(g["hold_obs"][:, :, 0, :].reshape(6, 3).astype(np.float64), (64, 1))
    out = []
    for split in (None, 50):
        eng = tg.DeviceRollout(env, pol, 4, 96, seed=3, compute_dtype=torch.bfloat16, fused=True)
        if split is None:
            tr = eng.run(initial_states=init)
        else:
            eng.params = env.native_params()
            with torch.cuda.device(dev):
                eng._enqueue_prepare(init)
                eng._enqueue_fused(0, split)
                torch.cuda.synchronize()
                mid = eng.traj.len.clone()
                eng.rng[1] -= 1                              # _enqueue_fused advanced the rollout id; same keys for part two
                eng._enqueue_fused(split, T)
            tr = eng.traj
            assert int((mid < 0).sum()) > 0.9 * mid.numel() and int(mid.min()) == -split      # running counts, negative
        torch.cuda.synchronize()
        out.append([t.clone() for t in (tr.obs, tr.act, tr.rew, tr.mask, tr.len)])
    assert float((out[0][4] == 101).float().mean()) > 0.9
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_bias_and_head_epilogue_kernels(tg, dev, cdt):
    """tg_head_prep (padded compute-dtype copy of dout + per-workgroup column sums) and tg_colsum_finish (several bias
    gradients in one launch) against torch."""
    N = tg._native
    lib = N.load()
    st = N.stream_ptr(dev)
    gen = torch.Generator(device="cpu").manual_seed(6)
    for rows, A, pad in ((70001, 4, 8), (3, 1, 8), (0, 2, 8), (5000, 8, 16 if cdt == torch.bfloat16 else 8)):
        dout = torch.randn(rows, A, generator=gen).to(dev)
        dz = torch.full((rows, pad), 7.0, dtype=cdt, device=dev)
        part = torch.empty(lib.tg_head_prep_blocks(), A, device=dev)
        N.check(lib.tg_head_prep(dout.data_ptr(), rows, A, pad, 1 if cdt == torch.bfloat16 else 0, dz.data_ptr(), part.data_ptr(), st))
        assert torch.equal(dz[:, :A], dout.to(cdt)) and torch.all(dz[:, A:] == 0)
        g = torch.ones(A, device=dev)
        ptrs = (N.C.c_void_p * 1)(g.data_ptr())
        N.check(lib.tg_colsum_finish(part.data_ptr(), part.shape[0], 1, A, ptrs, st))
        ref = 1.0 + dout.double().sum(0)
        assert float((g.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    # several vectors at once, deterministic
    part = torch.randn(256, 5, 256, generator=gen).to(dev)
    flat = torch.zeros(5 * 256 + 3, device=dev)
    outs = [flat[3 + 256 * v: 3 + 256 * (v + 1)] for v in range(5)]
    ptrs = (N.C.c_void_p * 5)(*[o.data_ptr() for o in outs])
    N.check(lib.tg_colsum_finish(part.data_ptr(), 256, 5, 256, ptrs, st))
    ref = part.double().sum(0)
    assert float((torch.stack(outs).double() - ref).abs().max()) < 1e-4 and torch.all(flat[:3] == 0)
    first = torch.stack(outs).clone()
    flat.zero_()
    N.check(lib.tg_colsum_finish(part.data_ptr(), 256, 5, 256, ptrs, st))
    assert torch.equal(torch.stack(outs), first)


@pytest.mark.parametrize("name,hidden,G,Eps,T,cdt", [("QuadPole", (256,) * 5, 256, 256, 256, torch.bfloat16),      # C3: tg_fused_rollout
                                                   ("CartPole", (128, 128), 64, 64, 500, None)])               # C2: tg_fused_rollout_f32
def test_fused_rollouts_at_baseline_sizes(tg, dev, name, hidden, G, Eps, T, cdt):
    """BASELINE.json's full rollout sizes through size-independent properties: masks are prefixes of length len, padding
    is zero, the counters agree, the recorded trajectory replays BIT-EXACTLY through the teacher-forced step kernel (the
    dynamics are compiled without FMA contraction, so every kernel that instantiates them computes the same bits), and a
    second run of the same stream reproduces the same bits."""
    S, A = DIMS[name]
    torch.manual_seed(11)
    pol = tg.GaussianActor_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
    mk = lambda: tg.environments.ENV_CLASSES[name](max_steps=T)
    eng = tg.DeviceRollout(mk(), pol, G, Eps, seed=31, compute_dtype=cdt)
    assert eng.fused and eng._fused_f32 == (cdt is None)
    tr = eng.run()
    obs, act, rew, mask, ln = (x.clone() for x in (tr.obs, tr.act, tr.rew, tr.mask, tr.len))
    n = G * Eps
    assert torch.equal(mask, (torch.arange(T, device=dev)[:, None] < ln[None, :]).to(torch.uint8))
    assert int(ln.min()) >= 1 and int(ln.max()) <= T and tr.env_steps() == int(ln.sum())
    m = mask.bool()
    assert torch.all(rew[~m] == 0) and torch.all(act[:, ~m] == 0) and torch.all(obs[:, :T][:, ~m] == 0)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    from trajopt_grpo_amd import rollout as RO
    plain = tg.DeviceRollout(mk(), pol, G, Eps, seed=31, compute_dtype=cdt, fused=False)
    init, forced = obs[:, 0, :].t().cpu().numpy(), act.permute(2, 1, 0).cpu().numpy()
    for per_step in (True, False):              # T launches of the golden-pinned step kernel, then the one-launch form (tg_rollout_forced)
        plain.forced_per_step = per_step
        replay = plain.run(initial_states=init, forced_actions=forced)
        assert torch.equal(replay.len, ln) and torch.equal(replay.mask, mask), per_step
        assert torch.equal(replay.rew, rew) and torch.equal(replay.obs, obs), per_step
    again = tg.DeviceRollout(mk(), pol, G, Eps, seed=31, compute_dtype=cdt).run()
    assert torch.equal(again.obs, obs) and torch.equal(again.act, act) and torch.equal(again.len, ln)


@pytest.mark.parametrize("H", [256, 128])
@pytest.mark.parametrize("rows", [1, 33, 1000, 70001])
def test_weight_gradient_kernel_matches_fp64(tg, dev, H, rows):
    """tg_mlp_weight_grad (torch autograd's dW = dZ^T A and db = sum dZ of every Linear, algorithms/ppo.py:181-183): all
    job kinds in one launch against fp64 products, ragged row tails, accumulation into windows of a flat bucket with
    the neighbouring words untouched, bit-identical from run to run."""
    from trajopt_grpo_amd import mlp as M, _native as N
    gen = torch.Generator(device="cpu").manual_seed(rows + H)

    def rnd(c, valid=None):
        t = torch.randn(rows, c, generator=gen)
        if valid is not None:
            t[:, valid:] = 0
        return t.to(torch.bfloat16).to(dev)

    dz1, a0, dz0, x, dh, a2 = rnd(H), rnd(H), rnd(H), rnd(32, 20), rnd(8, 4), rnd(H)
    sizes = [H * H, H, H * 20, H, 4 * H]
    ws = M.weight_grad_workspace(H, dev)

    def run():
        flat = torch.arange(sum(sizes) + 7, dtype=torch.float32, device=dev) * 1e-3
        o, views = 3, []
        for n in sizes:
            views.append(flat[o:o + n])
            o += n
        w1, b1, w0, b0, wh = views[0].view(H, H), views[1], views[2].view(H, 20), views[3], views[4].view(4, H)
        M.weight_grad(H, [(N.TG_DW_HH, dz1, a0, w1, b1), (N.TG_DW_HX, dz0, x, w0, b0), (N.TG_DW_DH, dh, a2, wh, None)], rows, ws)
        torch.cuda.synchronize()
        return flat, (w1, b1, w0, b0, wh)

    flat, got = run()
    base = (torch.arange(sum(sizes) + 7, dtype=torch.float32, device=dev) * 1e-3).double()
    ref = [dz1.double().t() @ a0.double(), dz1.double().sum(0), (dz0.double().t() @ x.double())[:, :20], dz0.double().sum(0),
           (dh.double().t() @ a2.double())[:4]]
    o = 3
    for r, g, n in zip(ref, got, sizes):
        want = base[o:o + n].view_as(r) + r
        scale = float(r.abs().max()) + 1.0
        assert float((g.double() - want).abs().max()) < 2e-5 * scale * max(1.0, (rows / 1000) ** 0.5), (n, rows)
        o += n
    assert torch.equal(flat[:3].double(), base[:3]) and torch.equal(flat[-4:].double(), base[-4:])     # neighbours untouched
    flat2, _ = run()
    assert torch.equal(flat, flat2)


@pytest.mark.parametrize("H,layers", [(256, 5), (128, 3)])
@pytest.mark.parametrize("rows", [1, 31, 255, 40000])
def test_weight_gradient_kernel_recomputes_the_first_activation(tg, dev, H, layers, rows):
    """Kind HR of tg_mlp_weight_grad rebuilds relu(W0 x + b0) on chip from the 64-B input row instead of reading the
    stored activation: bit-identical to the HH job on what tg_mlp_forward_chain stored."""
    from trajopt_grpo_amd import mlp as M, _native as N
    torch.manual_seed(rows + H)
    net = tg.NeuralNetwork(20, 4, (H,) * layers, "ReLU").to(dev)
    mlp = M.GemmMLP(net, torch.bfloat16)
    xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
    keep_chain, mlp._bchain = mlp._bchain, None                   # per-layer backward mode: the forward pass stores a0
    mlp.forward(xp, keep=True)
    mlp._bchain = keep_chain
    a0 = mlp._acts[1]
    dz = torch.randn(rows, H, device=dev).to(torch.bfloat16)
    ws = M.weight_grad_workspace(H, dev)
    w_a, b_a = torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)
    w_b, b_b = torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)
    M.weight_grad(H, [(N.TG_DW_HH, dz, a0, w_a, b_a)], rows, ws)
    M.weight_grad(H, [(N.TG_DW_HR, dz, xp, w_b, b_b)], rows, ws, mlp._chain.stream, mlp._chain.bias[0])
    torch.cuda.synchronize()
    ref = dz.double().t() @ a0.double()
    assert float((w_a.double() - ref).abs().max()) < 2e-5 * (float(ref.abs().max()) + 1.0) * max(1.0, (rows / 1000) ** 0.5)
    assert torch.equal(w_a, w_b) and torch.equal(b_a, b_b)


@pytest.mark.parametrize("H,layers,A", [(256, 5, 4), (128, 3, 1), (256, 3, 8)])
@pytest.mark.parametrize("rows", [1, 31, 255, 40000])
def test_weight_gradient_kernel_recomputes_the_top_layer_dz(tg, dev, H, layers, A, rows):
    """Kind RH of tg_mlp_weight_grad rebuilds dZ_top = (dOut . W_head) * (a_top > 0) on chip from the 16-B head gradient row and
    the layer's mask bits instead of reading a stored dZ; tg_mlp_backward_chain then leaves that store out (d_dz[0] = NULL).
    Bit-identical to the HH job on what the backward chain stores, and the lower layers' dZ do not depend on the store."""
    from trajopt_grpo_amd import mlp as M, _native as N
    torch.manual_seed(rows + H + A)
    net = tg.NeuralNetwork(20, A, (H,) * layers, "ReLU").to(dev)
    mlp = M.GemmMLP(net, torch.bfloat16)
    lib = N.load()
    xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
    mlp.forward(xp, keep=True)
    acts, bits = mlp._acts, mlp._bits
    nh = layers
    dzh = torch.zeros(rows, 8, dtype=torch.bfloat16, device=dev)
    dzh[:, :A] = torch.randn(rows, A, device=dev).bfloat16()
    mlp._fresh("bchain")
    m_ptrs = (N.C.c_void_p * nh)(*[bits[nh - j].data_ptr() for j in range(nh)])

    def chain(store_top):
        dzs = [torch.full((rows, H), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(nh)]
        ptrs = (N.C.c_void_p * nh)(*[(t.data_ptr() if (j > 0 or store_top) else None) for j, t in enumerate(dzs)])
        N.check(lib.tg_mlp_backward_chain(dzh.data_ptr(), mlp._bchain.stream.data_ptr(), H, nh, rows, ptrs, m_ptrs, None,
                                          N.stream_ptr(dev)), "tg_mlp_backward_chain")
        torch.cuda.synchronize()
        return dzs

    full, lean = chain(True), chain(False)
    assert float(lean[0].float().min()) == 7.0 == float(lean[0].float().max())          # untouched
    for a, b in zip(full[1:], lean[1:]):
        assert torch.equal(a, b)
    a_below = acts[nh - 1]                                        # input of the top hidden-to-hidden layer
    ws = M.weight_grad_workspace(H, dev)
    w_a, b_a = torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)
    w_b, b_b = torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)
    M.weight_grad(H, [(N.TG_DW_HH, full[0], a_below, w_a, b_a)], rows, ws)
    M.weight_grad(H, [(N.TG_DW_RH, dzh, a_below, w_b, b_b, bits[nh])], rows, ws, whfrag=mlp._bchain.stream)
    torch.cuda.synchronize()
    ref = full[0].double().t() @ a_below.double()
    assert float((w_a.double() - ref).abs().max()) < 2e-5 * (float(ref.abs().max()) + 1.0) * max(1.0, (rows / 1000) ** 0.5)
    assert torch.equal(w_a, w_b) and torch.equal(b_a, b_b)
    assert float(w_b.abs().max()) > 0 and float(b_b.abs().max()) > 0


@pytest.mark.parametrize("H,layers,S,A", [(256, 5, 20, 4), (128, 3, 5, 1), (256, 3, 10, 2)])
@pytest.mark.parametrize("rows", [1, 255, 777, 40000, 70001, 300000])     # > 65,536 rows: several rounds per workgroup
def test_backward_chain_forms_the_first_layer_gradient(tg, dev, H, layers, S, A, rows):
    """tg_mlp_backward_chain_w0 contracts the bottom layer's dZ with the net input inside the chain (the dZ is not written, the
    weight-gradient kernel has no HX job; the bias gradient is the ones column of the input): same gradients as the stored form
    within fp32 summation order, and dW0 / db0 against an fp64 product of the stored dZ."""
    from trajopt_grpo_amd import mlp as M
    torch.manual_seed(rows + H + S)
    net = tg.NeuralNetwork(S, A, (H,) * layers, "ReLU").to(dev)
    X = torch.randn(rows, S, device=dev)
    g = torch.randn(rows, A, device=dev)

    def run(fuse):
        mlp = M.GemmMLP(net, torch.bfloat16)
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        xp = mlp.prepare_input(X)
        assert float(xp[:, 31].float().min()) == 1.0 and float(xp[:, S:31].float().abs().max()) == 0.0
        if not fuse:
            M.set_ones_column(xp, False)       # (an input nobody vouches for: the stored form, kind HX job)
        mlp.forward(xp, keep=True)
        mlp.backward(g)
        torch.cuda.synchronize()
        dz_bottom = None if fuse else mlp._ws.get(f"z{layers - 1}", rows, H, torch.bfloat16, dev).clone()
        return [p.grad.clone() for p in net.parameters()], dz_bottom, xp

    stored, dz_bottom, xp = run(False)
    fused, _, _ = run(True)
    names = [n for n, _ in net.named_parameters()]
    for n, a, b in zip(names, stored, fused):
        # (not bit-identical even above the first layer: without the HX job the weight-gradient kernel splits its CUs -- and with
        # them the fixed-order partial sums of every layer -- differently)
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-5 * scale * max(1.0, (rows / 1000) ** 0.5), n
    ref = dz_bottom.double().t() @ xp.double()                  # [H][32]: columns < S = dW0, column 31 = db0
    w0, b0 = fused[0].double(), fused[1].double()
    tol = 2e-5 * (float(ref.abs().max()) + 1.0) * max(1.0, (rows / 1000) ** 0.5)
    assert float((w0 - ref[:, :S]).abs().max()) < tol and float((b0 - ref[:, 31]).abs().max()) < tol


def test_caller_padded_input_keeps_the_first_layer_bias_gradient(tg, dev):
    """the fused first-layer gradient takes db0 from the ones column prepare_input() writes.  An input the caller
    padded with zeros itself (the documented layout before that fusion) must not get a silently zero bias gradient: it takes
    the HX job of tg_mlp_weight_grad instead."""
    from trajopt_grpo_amd import mlp as M
    torch.manual_seed(5)
    rows, S, A, H = 3000, 20, 4, 256
    net = tg.NeuralNetwork(S, A, (H,) * 3, "ReLU").to(dev)
    X, g = torch.randn(rows, S, device=dev), torch.randn(rows, A, device=dev)

    def run(own_padding):
        mlp = M.GemmMLP(net, torch.bfloat16)
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        if own_padding:
            xp = torch.zeros(rows, 32, dtype=torch.bfloat16, device=dev)
            xp[:, :S] = X
        else:
            xp = mlp.prepare_input(X)
        mlp.forward(xp, keep=True)
        mlp.backward(g)
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    a, b = run(False), run(True)
    assert float(a[1].abs().max()) > 0
    for (n, _), x, y in zip(net.named_parameters(), a, b):
        assert float((x - y).abs().max()) <= 2e-5 * (float(x.abs().max()) + 1e-6) * 2, n


@pytest.mark.parametrize("H,layers,S,A,kind", [(256, 5, 20, 4, 0), (256, 5, 20, 1, 1), (128, 3, 5, 1, 0), (128, 3, 5, 1, 1), (256, 3, 10, 2, 0)])
@pytest.mark.parametrize("rows", [1, 255, 777, 40000, 70001, 300000])     # > 65,536 rows: several rounds per workgroup
def test_forward_chain_with_the_loss_head_inside(tg, dev, H, layers, S, A, kind, rows):
    """tg_mlp_forward_chain_loss: the clipped-surrogate (kind 0) / squared-error (kind 1) gradient formed in the forward kernel, the
    head's weight gradient contracted on chip with the top activation (never written): against forward + tg_surrogate_loss +
    backward of the same rows -- d loss / d output equal up to 1 bf16 ulp, loss sums to 1e-6, gradients within fp32 summation order."""
    from trajopt_grpo_amd import mlp as M, hip_ops as K
    torch.manual_seed(rows + H + S + kind)
    net = tg.NeuralNetwork(S, A, (H,) * layers, "ReLU").to(dev)
    X = torch.randn(rows, S, device=dev)
    act = torch.randn(rows, A, device=dev)
    lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0).contiguous()
    adv = torch.randn(rows, device=dev)
    ret = torch.randn(rows, device=dev)
    norm = torch.tensor([0.1, 1.3, -0.2, 0.7], device=dev)
    var = torch.full((A,), 0.3)
    eps, sc, cc, kc = 0.2, -1.0 / rows, 0.5 / rows, 0.5 / rows

    def grads():
        return [p.grad.clone() for p in net.parameters()]

    mlp = M.GemmMLP(net, torch.bfloat16)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    xp = mlp.prepare_input(X)
    out = mlp.forward(xp, keep=True)
    if kind == 0:
        _, s_ref, g_mean, _ = K.surrogate_loss(out, None, act, lpo, adv, None, None, norm, var, eps, sc, 0.0, kc, want_total=False)
        mlp.backward(g_mean)
    else:
        value = out.reshape(-1).contiguous()
        dummy = torch.zeros(rows, 1, device=dev)
        _, s_ref, _, g_val = K.surrogate_loss(dummy, value, dummy, lpo, adv, ret, None, norm, torch.ones(1), eps, 0.0, cc, 0.0,
                                              want_total=False)
        mlp.backward(g_val.view(rows, 1))
    dz_ref = mlp._ws.get("z_head", rows, mlp.out_pad, torch.bfloat16, dev).clone()
    torch.cuda.synchronize()
    ref = grads()

    mlp2 = M.GemmMLP(net, torch.bfloat16)
    if True:
        assert mlp2.can_fuse_head()
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        if kind == 0:
            s = mlp2.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, norm=norm.tolist()[0:2], var=var, epsilon=eps, surr_coef=sc,
                                  kl_coef=kc)
        else:
            s = mlp2.forward_loss(xp, 1, ret=ret, norm=norm.tolist()[2:4], critic_coef=cc)
        dz = mlp2._dz_head.clone()
        mlp2.backward_fused()
        torch.cuda.synchronize()
    got = grads()
    ddz = (dz.float() - dz_ref.float()).abs()
    assert float((ddz > 2.0 ** -7 * dz_ref.float().abs() + 1e-30).float().mean()) < 1e-3 and float(ddz.max()) <= 2.0 ** -6 * float(dz_ref.float().abs().max())
    idx = [0, 2, 3] if kind == 0 else [1, 3]
    for j in idx:
        # (the two kernels contract a * b + c differently here and there: 1-ulp differences per row)
        assert abs(float(s[j]) - float(s_ref[j])) <= 1e-6 * (abs(float(s_ref[j])) + 1.0), (j, float(s[j]), float(s_ref[j]))
    for (n, _), a, b in zip(net.named_parameters(), ref, got):
        scale = float(a.abs().max()) + 1e-9
        assert float((a - b).abs().max()) <= 2e-5 * scale * max(1.0, (rows / 1000) ** 0.5), n


# --------------------------------------------------------------------------------------------
# optimizer step + derived weight layouts as two launches (csrc/optim_kernels.hip)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lr,betas,eps", [(3e-4, (0.9, 0.999), 1e-8), (2e-4, (0.8, 0.99), 1e-6)])
def test_fused_adam_is_bit_identical_to_torch(tg, dev, lr, betas, eps):
    """optim.FusedAdam (tg_adam_step: one launch on the optimizer's own state tensors) against torch.optim.Adam.step() -- the
    reference's optimizer (pipelines/*_pipeline_*.py) -- over 10 steps with fresh random gradients: weights, both moments and the
    step counters BIT-identical, and the state dict loads back into a plain torch Adam."""
    from trajopt_grpo_amd import optim as O
    torch.manual_seed(7)
    pol_a = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256, 256, 256), cov=0.3, device=dev)
    pol_b = copy.deepcopy(pol_a)
    opt_a = torch.optim.Adam(pol_a.parameters(), lr=lr, betas=betas, eps=eps)
    opt_b = torch.optim.Adam(pol_b.parameters(), lr=lr, betas=betas, eps=eps)
    fused = O.FusedAdam(opt_a)
    gen = torch.Generator(device=dev).manual_seed(3)
    for it in range(10):
        for pa, pb in zip(pol_a.parameters(), pol_b.parameters()):
            g = torch.randn(pa.shape, device=dev, generator=gen) * (10.0 ** float(torch.randint(-6, 2, (1,)).item()))
            if it == 3:
                g[g.abs() < 0.5 * g.abs().mean()] = 0.0                      # exact zeros: sqrt(0) + eps paths
            pa.grad, pb.grad = g.clone(), g.clone()
        assert fused.step(zero_grads=it % 2 == 1)       # odd steps: the launch also zeroes the gradients it consumed
        assert fused.grads_zeroed == (it % 2 == 1)
        opt_b.step()
        for pa, pb in zip(pol_a.parameters(), pol_b.parameters()):
            assert torch.equal(pa, pb), it
            assert torch.equal(pa.grad, torch.zeros_like(pa) if it % 2 == 1 else pb.grad), it
            sa, sb = opt_a.state[pa], opt_b.state[pb]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), it
            assert float(sa["step"]) == float(sb["step"]) == it + 1
    opt_c = torch.optim.Adam(pol_b.parameters(), lr=lr, betas=betas, eps=eps)
    opt_c.load_state_dict(opt_a.state_dict())                               # the checkpoint format is torch's own
    # anything but a plain default Adam keeps torch's own step()
    assert not O.FusedAdam(torch.optim.Adam(pol_a.parameters(), lr=lr, amsgrad=True)).usable()
    assert not O.FusedAdam(torch.optim.AdamW(pol_a.parameters(), lr=lr)).usable()
    patched = torch.optim.Adam(pol_a.parameters(), lr=lr)
    patched.step = lambda *a, **k: None
    assert not O.FusedAdam(patched).usable()


@pytest.mark.parametrize("cdt,hidden", [(torch.bfloat16, (256,) * 5), (torch.bfloat16, (128,) * 3), (None, (128, 128)), (None, (64,) * 4)])
def test_stream_refresher_equals_the_per_stream_refresh(tg, dev, cdt, hidden, monkeypatch):
    """optim.StreamRefresher (tg_gather_streams: every derived weight layout of actor and critic in one launch) leaves exactly the
    bytes FragmentStream.refresh() / F32ChainStream.refresh() build, and learn() with the fused optimizer step equals learn() with
    torch's step bit for bit."""
    from trajopt_grpo_amd import mlp as M, optim as O, algorithms as ALG
    torch.manual_seed(11)
    S, A = (20, 4) if cdt is not None else (5, 1)
    pol = tg.GaussianActorCritic_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    for p in pol.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    mlps = [M.GemmMLP(pol.actor, cdt or torch.float32), M.GemmMLP(pol.critic, cdt or torch.float32)]
    fused = O.FusedAdam(opt)
    assert fused.step()
    ref = O.StreamRefresher(fused, mlps)
    for m in mlps:
        m.refresh()
    assert ref.run()
    got = []
    for m in mlps:
        for name in ("_chain", "_bchain", "_f32"):
            st = getattr(m, name)
            if st is not None:
                assert name[1:] not in m._stale
                got.append((st, st.stream.clone(), st.bias.clone() if hasattr(st, "bias") else None))
    assert got
    for st, stream, bias in got:
        st.stream.zero_()
        st.refresh()
        assert torch.equal(st.stream, stream)
        if bias is not None and not getattr(st, "transposed", False):
            assert torch.equal(st.bias, bias)

    # once gathered, the optimizer step's own launch keeps every layout current (tg_adam_step_push: the inverse of the gather)
    for it in range(2):
        for p in pol.parameters():
            p.grad = torch.randn_like(p) * 1e-3
        assert fused.step(zero_grads=it == 0, refresher=ref) and fused.pushed
    for m in mlps:
        m.refresh()
        have = {n for n in ("chain", "bchain", "f32") if getattr(m, "_" + n) is not None}
        assert have and not (m._stale & have), m._stale                     # (the per-layer copies are not the gather's)
    for st, _, _ in got:
        pushed = (st.stream.clone(), st.bias.clone() if hasattr(st, "bias") else None)
        st.stream.zero_()
        st.refresh()
        assert torch.equal(st.stream, pushed[0])
        if pushed[1] is not None and not getattr(st, "transposed", False):
            assert torch.equal(st.bias, pushed[1])

    def learn(fused_flag):
        torch.manual_seed(5)
        pol2 = tg.GaussianActorCritic_NeuralNetwork(S, A, hidden, cov=0.3, device=dev)
        env = (lambda: tg.QuadPole(max_steps=16)) if S == 20 else (lambda: tg.CartPole(max_steps=16))
        mgr = tg.RolloutManager(env, pol2, num_workers=4, num_episodes_per_worker=32, seed=3, compute_dtype=cdt)
        buf = tg.Rollout_Buffer(mgr)
        buf.sample()
        opt2 = torch.optim.Adam(pol2.parameters(), lr=3e-4)
        if not fused_flag:
            opt2.register_step_post_hook(lambda *a, **k: None)       # (an optimizer with hooks keeps torch's own step())
        algo = tg.PPO(epsilon=0.2, policy=pol2, optimizer=opt2, ref_model=None,
                      updates_per_iter=3, gamma=0.99, batch_size=None, autocast_dtype=cdt)
        algo.learn(buf)
        assert algo._fused_adam.usable() == fused_flag
        torch.cuda.synchronize()
        return [p.detach().clone() for p in pol2.parameters()], algo.last_stats["total_loss"]

    (wa, la), (wb, lb) = learn(True), learn(False)
    assert la == lb and all(torch.equal(a, b) for a, b in zip(wa, wb))


# --------------------------------------------------------------------------------------------
# the fp32 chain learner (csrc/mlp_f32_chain.hip): the reference's own precision and net sizes
# --------------------------------------------------------------------------------------------
F32_SHAPES = [(5, 1, (128, 128)), (20, 4, (128,) * 4), (10, 2, (64,)), (5, 1, (64, 64, 64)), (32, 4, (128, 128, 128)), (3, 1, (128,)),
              (12, 3, (128, 128)), (20, 4, (128, 128)), (32, 2, (128, 128)), (7, 2, (128, 128)),       # (two 128-wide layers: the 8-wave weight-gradient job at every padded input width)
              # H = 256 (csrc/mlp_f32_wide.hip): the reference's QuadPole factory shape (quadpole_pipeline_ppo.py:54-58) and its edges
              (20, 4, (256,) * 5), (5, 1, (256, 256)), (10, 2, (256,)), (32, 3, (256, 256, 256))]


@pytest.mark.parametrize("dims", F32_SHAPES)
@pytest.mark.parametrize("rows", [1, 257, 70001, 300000])
def test_f32_chain_forward_matches_fp64(tg, dev, dims, rows):
    """tg_mlp_f32_forward (models/neural_network.py:67-77 in one launch, fp32 products on the matrix cores) against the same net
    in fp64: relative 1e-5 of the output scale.  rows > 65,536: several rounds per workgroup."""
    from trajopt_grpo_amd import mlp as M
    S, A, hidden = dims
    torch.manual_seed(rows + S)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    m = M.GemmMLP(net, torch.float32)
    assert m._f32 is not None and m.in_pad == (S + 7) // 8 * 8
    assert m._f32.res == (hidden in ((128,), (128, 128))) and m._f32.wide == (hidden[0] == 256)     # (which of the three kernels this is)
    X = torch.randn(rows, S, device=dev)
    xp = m.prepare_input(X)
    assert xp.shape == (rows, m.in_pad) and xp.dtype == torch.float32
    out = m.forward(xp, keep=False)
    pad = m.forward(xp, keep=False, padded=True)
    torch.cuda.synchronize()
    ref = copy.deepcopy(net).double()(X.double())
    scale = float(ref.abs().max()) + 1e-6
    assert out.shape == (rows, A) and float((out.double() - ref).abs().max()) <= 1e-5 * scale
    assert pad.shape == (rows, 4) and torch.equal(pad[:, :A].contiguous(), out) and torch.all(pad[:, A:] == 0)


@pytest.mark.parametrize("dims", [(5, 1, (128, 128)), (20, 4, (128, 128)), (9, 2, (128,))])
@pytest.mark.parametrize("rows", [1, 17, 193, 50001])
def test_f32_resident_kernel_is_interchangeable_with_the_chain_kernel(tg, dev, dims, rows, monkeypatch):
    """tg_mlp_f32r_forward_backward (16 rows per wave, weights resident in LDS) and tg_mlp_f32_forward_backward (32 rows per wave)
    on the same net and rows: loss sums, d loss / d output, stored activation / dZ to 1e-5 of their scale (the two sum a row's
    products in different orders), the top layer's mask words bit for bit wherever the pre-activation is not within rounding of 0
    (< 1e-4 of the bits may differ), and the gradients tg_mlp_f32_weight_grad forms from either to 2e-5.  Then the new entry
    points' argument checks (host side, nothing launched)."""
    from trajopt_grpo_amd import mlp as M
    Nn = tg._native
    S, A, hidden = dims
    torch.manual_seed(rows + S)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    X = torch.randn(rows, S, device=dev)
    act = torch.randn(rows, A, device=dev)
    lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0).contiguous()
    adv = torch.randn(rows, device=dev)
    var = torch.full((A,), 0.3)

    def run(resident):
        with monkeypatch.context() as mp:
            if not resident:
                mp.setattr(M, "f32_res_supported", lambda net_: 0)
            m = M.GemmMLP(net, torch.float32)
        assert m._f32.res == resident
        for p_ in net.parameters():
            p_.grad = torch.zeros_like(p_)
        xp = m.prepare_input(X)
        out = m.forward(xp, keep=False, padded=True).clone()
        sums = m.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, norm=[0.1, 1.3, 0.0, 1.0], var=var, epsilon=0.2, surr_coef=-1.0 / rows,
                              kl_coef=0.5 / rows).clone()
        keep = {"out": out, "sums": sums, "dout": m._dz_head.clone(), "tmask": None if m._tmask is None else m._tmask.clone(),
                "acts": [None if t is None else t.clone() for t in m._acts[1:]], "dz": [None if t is None else t.clone() for t in m._bits]}
        m.backward_fused()
        keep["grads"] = [p_.grad.clone() for p_ in net.parameters()]
        return keep

    r, c = run(True), run(False)
    close = lambda x, y, tol: float((x.double() - y.double()).abs().max()) <= tol * (float(y.double().abs().max()) + 1e-12)
    assert close(r["out"], c["out"], 1e-5) and close(r["sums"], c["sums"], 1e-6) and close(r["dout"], c["dout"], 1e-5)
    for x, y in zip(r["acts"] + r["dz"], c["acts"] + c["dz"]):
        assert (x is None) == (y is None)
        assert x is None or close(x, y, 1e-5)
    if r["tmask"] is not None:
        diff = (r["tmask"] ^ c["tmask"]).view(torch.uint8)
        flipped = sum(int(((diff >> b) & 1).sum()) for b in range(8))
        assert flipped <= max(1, int(1e-4 * rows * 128)), flipped
    for x, y in zip(r["grads"], c["grads"]):
        assert close(x, y, 2e-5 * max(1.0, (rows / 1000) ** 0.5))
    # ---- argument checks of the resident entry points ----
    lib, f = Nn.load(), M.GemmMLP(net, torch.float32)._f32
    xp = torch.zeros(16, f.in_pad, device=dev)
    o = torch.zeros(16, 4, device=dev)
    args = lambda H=128, nh=f.n_hidden, A_=A, pad=f.in_pad: (xp.data_ptr(), pad, f.stream.data_ptr(), f.w0.data_ptr(), f.table.data_ptr(), H, nh, A_, 16,
                                                              o.data_ptr(), Nn.stream_ptr(dev))
    assert lib.tg_mlp_f32r_forward(*args()) == 0
    for bad, what in ((args(H=64), b"hidden width"), (args(nh=3), b"hidden layers"), (args(A_=5), b"outputs"), (args(pad=12), b"padded input width")):
        assert lib.tg_mlp_f32r_forward(*bad) != 0 and what in lib.tg_last_error(), what
    assert lib.tg_mlp_f32r_supported(128, 2, 8) == 1 and lib.tg_mlp_f32r_supported(128, 3, 8) == 0 and lib.tg_mlp_f32r_supported(256, 2, 8) == 0
    assert lib.tg_mlp_f32r_grid(1) == 1 and lib.tg_mlp_f32r_grid(16 * 12 + 1) == 2 and lib.tg_mlp_f32r_grid(1 << 24) == lib.tg_mlp_f32_blocks()
    torch.cuda.synchronize()


@pytest.mark.parametrize("dims", F32_SHAPES)
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("rows", [1, 255, 4000, 70001])
def test_f32_chain_update_matches_fp64_autograd(tg, dev, dims, kind, rows, monkeypatch):
    """forward_loss() + backward_fused() of an fp32 net -- tg_mlp_f32_forward_backward (forward, clipped-surrogate / squared-error
    head as loss_kernels.hip, backward data) and tg_mlp_f32_weight_grad -- against torch autograd of the same loss in fp64
    (algorithms/ppo.py:159-183, grpo.py:122-145): loss sums to 1e-6, every stored activation / dZ to 1e-5 of its scale, every
    parameter gradient to 2e-5 (x sqrt(rows / 1000)) of its scale; gradients ACCUMULATE into .grad; bit-identical run to run.
    Both forms of the weight-gradient job are held to the same bar: operands read back from HBM, and (nets with >= 2 hidden
    layers, the default) the first activation / top dZ rebuilt on chip from the input row / d loss / d output + mask bits."""
    from trajopt_grpo_amd import mlp as M
    S, A, hidden = dims
    if kind == 1:
        A = 1
    torch.manual_seed(rows + S + kind)
    net = tg.NeuralNetwork(S, A, hidden, "ReLU").to(dev)
    X = torch.randn(rows, S, device=dev)
    act = torch.randn(rows, A, device=dev)
    lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0).contiguous()
    adv, ret = torch.randn(rows, device=dev), torch.randn(rows, device=dev)
    norm = [0.1, 1.3, -0.2, 0.7]
    var = torch.full((A,), 0.3)
    eps, sc, cc, kc = 0.2, -1.0 / rows, 0.5 / rows, 0.5 / rows

    def run(recompute):
        m = M.GemmMLP(net, torch.float32)
        m.f32_store_all = not recompute
        for i, p in enumerate(net.parameters()):
            p.grad = torch.full_like(p, 0.25 * (i + 1))                 # the kernels must ADD to what is there
        assert m.can_fuse_head()
        xp = m.prepare_input(X)
        if kind == 0:
            s = m.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, norm=norm[0:2], var=var, epsilon=eps, surr_coef=sc, kl_coef=kc)
        else:
            s = m.forward_loss(xp, 1, ret=ret, norm=norm[2:4], critic_coef=cc)
        cl = lambda t: None if t is None else t.clone()
        stored = [cl(t) for t in m._acts[1:]], [cl(t) for t in m._bits], m._dz_head.clone(), cl(m._tmask)
        m.backward_fused()
        torch.cuda.synchronize()
        return s.clone(), [p.grad.clone() for p in net.parameters()], stored

    s, got, (acts, dzs, dout, _) = run(False)
    s2, got2, _ = run(False)
    assert torch.equal(s, s2) and all(torch.equal(a, b) for a, b in zip(got, got2))
    assert all(t is not None for t in acts) and all(t is not None for t in dzs)
    sr, got_r, (acts_r, dzs_r, dout_r, tmask) = run(True)
    sr2, got_r2, _ = run(True)
    assert torch.equal(sr, s) and torch.equal(dout_r, dout)
    assert all(torch.equal(a, b) for a, b in zip(got_r, got_r2))
    n_hidden = len(hidden)
    wide = hidden[0] == 256                                             # (the H = 256 learner stores everything: its weight-gradient jobs rebuild nothing yet)
    if n_hidden >= 2 and not wide:
        # neither the first activation nor the top dZ was written; everything in between is the same bits as in the storing run.
        # The top layer's mask travels as 4 words per row: feature 32 mt + 8 q + 4 hh + low <-> word hh * (H / 64) + mt // 2,
        # bit low + 4 q + 16 (mt % 2)  (include/trajopt_grpo_hip.h, tg_mlp_f32_forward_backward)
        assert acts_r[0] is None and dzs_r[n_hidden - 1] is None and tmask is not None
        assert all(torch.equal(a, b) for a, b in zip(acts_r[1:], acts[1:])) and all(torch.equal(a, b) for a, b in zip(dzs_r[:-1], dzs[:-1]))
        Hh = hidden[-1]
        f = torch.arange(Hh, device=dev)
        mt, q, hh, low = f >> 5, (f & 31) >> 3, (f >> 2) & 1, f & 3
        word, bit = hh * (Hh // 64) + (mt >> 1), low + 4 * q + 16 * (mt & 1)
        got_bits = (tmask.to(torch.int64)[:, word] >> bit) & 1
        assert torch.equal(got_bits.bool(), acts[-1] > 0)
    else:
        assert tmask is None and all(torch.equal(a, b) for a, b in zip(got_r, got))
    # ---- the same update in fp64.  The ReLU masks are the KERNEL's (a pre-activation within fp32 rounding of zero may fall on
    # the other side of the ReLU in fp64; one such flip moves a weight gradient by a whole row's contribution): they are checked
    # against the fp64 pre-activations separately, everything else is then compared tightly ----
    lin = [mod for mod in copy.deepcopy(net).double().network if isinstance(mod, torch.nn.Linear)]
    W = [l.weight.detach() for l in lin]
    B = [l.bias.detach() for l in lin]
    masks = [(a > 0) for a in acts]
    hs, h = [], X.double()
    for l in range(len(hidden)):
        z = h @ W[l].t() + B[l]
        flips = (z > 0) != masks[l]
        assert float(flips.float().mean()) <= 1e-4 and (not bool(flips.any()) or float(z[flips].abs().max()) <= 1e-5 * (1.0 + float(z.abs().max()))), l
        h = z * masks[l]
        hs.append(h)
    out = (h @ W[-1].t() + B[-1]).requires_grad_()
    if kind == 0:
        logp = -0.5 * (((act.double() - out) ** 2) / var.double().to(dev)).sum(1) - 0.5 * A * np.log(2 * np.pi) - 0.5 * float(torch.log(var.double()).sum())
        rho = torch.exp(logp - lpo.double())
        an = (adv.double() - norm[0]) * norm[1]
        surr = torch.minimum(rho * an, torch.clamp(rho, 1 - eps, 1 + eps) * an)
        kl = torch.exp(lpo.double()) * (lpo.double() - logp)
        total = sc * surr.sum() + kc * kl.sum()
        want = {0: float(surr.sum()), 2: float(kl.sum()), 3: float(rows)}
    else:
        d = out[:, 0] - (ret.double() - norm[2]) * norm[3]
        total = cc * (d * d).sum()
        want = {1: float((d * d).sum()), 3: float(rows)}
    total.backward()
    g = out.grad
    for k, v in want.items():
        assert abs(float(s[k]) - v) <= 2e-6 * (abs(v) + 1.0), (k, float(s[k]), v)
    for i, (a, hh) in enumerate(zip(acts, hs)):
        assert float((a.double() - hh).abs().max()) <= 1e-5 * (float(hh.abs().max()) + 1e-6), i
    assert float((dout[:, :A].double() - g).abs().max()) <= 2e-5 * (float(g.abs().max()) + 1e-30) and torch.all(dout[:, A:] == 0)
    nh = len(hidden)
    ref_w, ref_b = [None] * (nh + 1), [None] * (nh + 1)
    ref_w[nh], ref_b[nh] = g.t() @ hs[-1], g.sum(0)
    dh = g @ W[nh]
    for l in range(nh - 1, -1, -1):
        dz = dh * masks[l]
        assert float((dzs[l].double() - dz).abs().max()) <= 2e-5 * (float(dz.abs().max()) + 1e-30), l
        ref_w[l], ref_b[l] = dz.t() @ (hs[l - 1] if l > 0 else X.double()), dz.sum(0)
        dh = dz @ W[l]
    ref = [t for pair in zip(ref_w, ref_b) for t in pair]                # parameters(): weight, bias per layer
    for form, grads in (("stored", got), ("rebuilt", got_r)):
        for i, (gg, r) in enumerate(zip(grads, ref)):
            base = 0.25 * (i + 1)
            scale = float(r.abs().max()) + 1e-12
            assert float((gg.double() - base - r).abs().max()) <= (2e-5 * max(1.0, (rows / 1000) ** 0.5)) * scale + 4e-7 * base, (form, i, rows)


@pytest.mark.parametrize("rows", [1, 300, 70001])
def test_f32_loss_sums_ride_on_the_weight_gradient_reduction(tg, dev, rows):
    """forward_loss(sums_out=...) + backward_fused() of an fp32 net: the weight-gradient reduction launch adds the chain kernel's
    per-workgroup loss sums into the caller's f64 [4] in a fixed order -- equal to the sums the plain call returns (f64 rounding),
    accumulating over calls, bit-reproducible; the gradients do not depend on which form ran."""
    from trajopt_grpo_amd import mlp as M
    torch.manual_seed(rows)
    net = tg.NeuralNetwork(5, 1, (128, 128), "ReLU").to(dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    m = M.GemmMLP(net, torch.float32)
    X, act = torch.randn(rows, 5, device=dev), torch.randn(rows, 1, device=dev)
    lpo, adv = (-0.5 * torch.rand(rows, device=dev) - 1.0).contiguous(), torch.randn(rows, device=dev)
    kw = dict(act=act, logp_old=lpo, adv=adv, var=torch.full((1,), 0.3), epsilon=0.2, surr_coef=-1.0 / rows, kl_coef=0.5 / rows)
    xp = m.prepare_input(X)
    s = m.forward_loss(xp, 0, **kw).clone()
    m.backward_fused()
    g_plain = [p.grad.clone() for p in net.parameters()]
    outs = []
    for rep in range(2):
        for p in net.parameters():
            p.grad.zero_()
        out = torch.zeros(4, dtype=torch.float64, device=dev)
        for n in (1, 2):
            assert m.forward_loss(xp, 0, sums_out=out, **kw) is None
            m.backward_fused()
            torch.cuda.synchronize()
            assert float((out - n * s).abs().max()) <= 1e-12 * n * float(s.abs().max() + 1.0), n
            assert float(out[3]) == n * rows
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    for p, g in zip(net.parameters(), g_plain):
        assert float((p.grad - 2 * g).abs().max()) <= 2e-6 * float(g.abs().max() + 1e-30)


def test_rollout_register_stream_is_rebuilt_by_the_step_launch(tg, dev):
    """The fused fp32 rollout's weight stream (mlp.RegisterStreamF32) rides on the launch that rebuilds the learner's layouts after
    every optimizer step (optim.StreamRefresher): after learn() it holds exactly the bytes its own refresh() builds, the next
    rollout skips that refresh, and any write to the policy through torch makes it stale again."""
    torch.manual_seed(5)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=64), pol, num_workers=8, num_episodes_per_worker=32, seed=3)
    assert mgr.engine.fused and mgr.engine._fused_f32
    buf = tg.Rollout_Buffer(mgr)
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=3)
    buf.sample()
    frag = mgr.engine._frag
    w_before = frag.stream.clone()
    algo.learn(buf)
    assert bool(algo._fused_adam) and algo._rollout_stream is frag and frag.is_fresh()
    s, t = frag.stream.clone(), frag.table.clone()
    assert not torch.equal(s, w_before)
    calls = []
    plain = frag.refresh
    frag.refresh = lambda: (calls.append(1), plain())[1]
    buf.sample()                                            # no refresh of its own
    assert not calls
    frag._refresh()
    assert torch.equal(frag.stream, s) and torch.equal(frag.table, t)
    with torch.no_grad():
        next(pol.actor.parameters()).mul_(1.5)              # a write through torch (version counter)
    assert not frag.is_fresh()
    buf.sample()                                            # the entry rebuild: the learner's gather launch, registered with the engine
    assert mgr.engine.entry_refresh is not None and not calls and frag.is_fresh()
    s2, t2 = frag.stream.clone(), frag.table.clone()
    frag._refresh()
    assert torch.equal(frag.stream, s2) and torch.equal(frag.table, t2) and not torch.equal(s2, s)
    with torch.no_grad():
        next(pol.actor.parameters()).data.mul_(2.0)         # ... and a write the version counters do not see
    assert frag.is_fresh()                                  # (the keys cannot know)
    buf.sample()
    s3 = frag.stream.clone()
    frag._refresh()
    assert torch.equal(frag.stream, s3) and not torch.equal(s3, s2), "the rollout ran on a stale weight stream after a .data write"
    mgr.engine.entry_refresh = None                         # without a learner's gather the stream refreshes itself at every entry
    buf.sample()
    assert calls == [1]


def test_c2_size_grpo_learn_matches_the_oracle(tg, dev):
    """BASELINE configs[1] at full size on the product path: CartPole GRPO, 4,096 envs (64 groups x 64), fp32 5-128-128-1 -- fused
    fp32 rollout, fp32 chain learner, fused optimizer step -- against the CPU oracle's GRPO step (algorithms/grpo.py:50-148 restated,
    pinned by the reference's goldens) on the SAME sampled buffer: objective values of both updates, post-step weights."""
    torch.manual_seed(21)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
    sd = {n: v.detach().cpu().clone() for n, v in pol.state_dict().items()}
    ora, old = L.OraclePolicy(5, 1, (128, 128), cov=0.5), L.OraclePolicy(5, 1, (128, 128), cov=0.5)
    ora.load_state_dict({n: v.clone() for n, v in sd.items()})
    old.load_state_dict({n: v.clone() for n, v in sd.items()})
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=128), pol, num_workers=64, num_episodes_per_worker=64, seed=9)
    assert mgr.engine.fused and mgr.engine._fused_f32
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=2)
    algo.learn(buf)
    m = algo._mlp(pol.actor)
    assert m._f32 is not None and bool(algo._fused_adam)
    n_valid = int(buf.device_traj.mask.sum())
    assert n_valid > 100000 and algo.last_stats["n_valid"] == n_valid
    torch.set_num_threads(8)
    Js = L.grpo_learn(ora, old, torch.optim.Adam(ora.parameters(), lr=3e-4), buf.group_observations, buf.group_actions, buf.group_rewards,
                      buf.group_masks, epsilon=0.15, gamma=0.5, updates_per_iter=2)
    for a, b in zip(algo.last_stats["J"], Js):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (a, b)
    for (n, p), q in zip(pol.actor.named_parameters(), ora.parameters()):
        d = p.detach().cpu() - q.detach()
        # Adam's normalised step: an entry whose gradient sits at rounding level may move by up to lr per step in either direction
        assert float(d.abs().max()) <= 2 * 2 * 3e-4 + 1e-6 and float(d.norm() / q.detach().norm()) < 3e-4, n


@pytest.mark.parametrize("hidden", [(128, 128), (64, 64, 64), (128,)])
def test_optimizer_step_riding_on_the_f32_reduction_is_bit_identical(tg, dev, hidden, monkeypatch):
    """GRPO on the fp32 chain learner, one rank, one chunk per update: `optimizer.step()` (algorithms/grpo.py:145) rides on the
    gradient-reduction launch (tg_mlp_f32_weight_grad_adam) from the second update on.  Against the same run with the step as
    its own launch (tg_adam_step_push): weights, gradients left in .grad, Adam moments and step counters, the derived weight
    layouts and the NEXT rollout's trajectory are bit-identical."""
    import trajopt_grpo_amd.algorithms as A
    import trajopt_grpo_amd.optim as O

    ask = A._GpuLearner._adam_rider

    def run(ride):
        # (the arm without the rider: the learner is never offered one -- what happens on several ranks or several chunks)
        monkeypatch.setattr(A._GpuLearner, "_adam_rider", ask if ride else (lambda self, *a, **k: None))
        rides = []
        orig = O.FusedAdam.rider

        def counting(self, *a, **k):
            r = orig(self, *a, **k)
            rides.append(r is not None)
            return r

        monkeypatch.setattr(O.FusedAdam, "rider", counting)
        torch.manual_seed(33)
        pol = tg.GaussianActor_NeuralNetwork(5, 1, hidden, cov=0.5, device=dev)
        mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=60), pol, num_workers=16, num_episodes_per_worker=32, seed=5)
        buf = tg.Rollout_Buffer(mgr)
        opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
        algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.9, policy=pol, optimizer=opt, updates_per_iter=3)
        for _ in range(2):
            buf.sample()
            algo.learn(buf)
        m = algo._mlp(pol.actor)
        assert m._f32 is not None
        buf.sample()
        tr = buf.device_traj
        out = {"obs": tr.obs.clone(), "act": tr.act.clone(), "mask": tr.mask.clone(), "stream": m._f32.stream.clone()}
        for n, p in pol.actor.named_parameters():
            st = opt.state[p]
            out["p." + n], out["g." + n], out["m." + n], out["v." + n] = p.detach().clone(), p.grad.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()
            out["t." + n] = st["step"].clone()
        monkeypatch.setattr(O.FusedAdam, "rider", orig)
        return out, rides, algo.last_stats

    (a, rides_a, sa), (b, rides_b, sb) = run(True), run(False)
    assert rides_a == [False] + [True] * 5, rides_a          # the first update gathers the layouts; every later step rides
    assert rides_b == []
    assert sa["J"] == sb["J"]
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert all(float(v) == 6.0 for k, v in a.items() if k.startswith("t."))


def test_optimizer_step_does_not_ride_when_the_optimizer_holds_more_than_the_net(tg, dev, monkeypatch):
    """The riding step is refused -- and the separate launch taken -- when the optimizer owns a tensor the backward pass's launch
    produces no gradient window for (it would silently miss its step): same result as with the rider switched off, and the
    C entry point itself refuses such a table."""
    import ctypes as C
    import trajopt_grpo_amd.algorithms as A
    import trajopt_grpo_amd.optim as O
    from trajopt_grpo_amd import _native as N

    ask = A._GpuLearner._adam_rider

    def run(ride):
        monkeypatch.setattr(A._GpuLearner, "_adam_rider", ask if ride else (lambda self, *a, **k: None))
        torch.manual_seed(3)
        pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
        extra = torch.nn.Parameter(torch.full((7,), 0.25, device=dev))
        extra.grad = torch.full((7,), 0.5, device=dev)
        mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=40), pol, num_workers=8, num_episodes_per_worker=16, seed=5)
        buf = tg.Rollout_Buffer(mgr)
        opt = torch.optim.Adam(list(pol.parameters()) + [extra], lr=1e-3)
        algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.9, policy=pol, optimizer=opt, updates_per_iter=2)
        for _ in range(2):
            buf.sample()
            algo.learn(buf)
        return [p.detach().clone() for p in pol.parameters()] + [extra.detach().clone()], algo

    calls = []
    orig = O.FusedAdam.rider
    monkeypatch.setattr(O.FusedAdam, "rider", lambda self, *a, **k: calls.append(1) or orig(self, *a, **k))
    (a, algo), (b, _) = run(True), run(False)
    assert calls == []                                       # the learner never even asks: the parameter sets differ
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[-1], torch.full((7,), 0.25, device=dev))      # the extra tensor was stepped by the separate launch
    # the C ABI's own check: a table with one tensor more than the launch has windows
    m = algo._mlp(algo.policy.actor)
    fa = algo._fused_adam
    tab, n, total = fa.table(0)
    r = N.AdamRider()
    r.h_table, r.n_tensors, r.zero_grads, r.total = fa._host_tables[0], n, 0, total
    r.lr, r.beta1, r.beta2, r.eps, r.step = 1e-3, 0.9, 0.999, 1e-8, 5
    xp = m.prepare_input(torch.randn(300, 5, device=dev))
    m.forward_loss(xp, 0, act=torch.randn(300, 1, device=dev), logp_old=torch.full((300,), -1.0, device=dev), adv=torch.randn(300, device=dev),
                   var=torch.full((1,), 0.5), epsilon=0.2, surr_coef=-1.0 / 300)
    before = [p.detach().clone() for p in algo.policy.parameters()]
    with pytest.raises(RuntimeError, match="gradient windows"):
        m.backward_fused(adam=r)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(algo.policy.parameters(), before))     # nothing was launched


def test_ppo_with_more_than_four_actions_keeps_one_prepared_input(tg, dev):
    """An fp32 actor with > 4 outputs is outside the fp32 chain learner while its 1-output critic is inside: PPO must put both on
    the per-layer path (they share one padded input) instead of feeding one of them a wrongly padded buffer."""
    torch.manual_seed(0)
    pol = tg.GaussianActorCritic_NeuralNetwork(12, 6, (64, 64), cov=0.3, device=dev)
    algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None, updates_per_iter=1,
                  batch_size=None)
    G, E, T = 2, 8, 6
    g = torch.Generator().manual_seed(1)
    buf = _Buf()
    buf.group_observations = torch.randn(G, E, T, 12, generator=g)
    buf.group_actions = torch.randn(G, E, T, 6, generator=g)
    buf.group_rewards = torch.randn(G, E, T, generator=g)
    buf.group_masks = torch.ones(G, E, T)
    buf.group_lengths = torch.full((G, E), float(T))
    before = [p.detach().clone() for p in pol.parameters()]
    algo.learn(buf)
    m_a, m_c = algo._mlp(pol.actor), algo._mlp(pol.critic)
    assert m_a._f32 is None and m_c._f32 is None and m_a.in_pad == m_c.in_pad == 32
    assert all(torch.isfinite(p).all() for p in pol.parameters()) and any(not torch.equal(p, b) for p, b in zip(pol.parameters(), before))


def test_f32_learn_is_independent_of_the_chunk_size(tg, dev):
    """The learner walks the valid rows in chunks (`chunk_rows`): on the fp32 chain learner several small chunks must give the
    one-chunk result up to fp32 summation order (loss sums, gradients accumulate across chunks)."""
    def run(chunk):
        torch.manual_seed(4)
        pol = tg.GaussianActorCritic_NeuralNetwork(5, 1, (128, 128, 128), cov=0.4, device=dev)
        mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=40), pol, num_workers=8, num_episodes_per_worker=32, seed=2)
        buf = tg.Rollout_Buffer(mgr)
        buf.sample()
        algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None, updates_per_iter=2,
                      gamma=0.99, batch_size=None, chunk_rows=chunk)
        algo.learn(buf)
        assert algo._mlp(pol.actor)._f32 is not None
        return [p.detach().clone() for p in pol.parameters()], algo.last_stats

    (wa, sa), (wb, sb) = run(None), run(1777)
    for k in ("total_loss", "actor_loss", "critic_loss"):
        np.testing.assert_allclose(sa[k], sb[k], rtol=1e-5, atol=1e-7)
    for a, b in zip(wa, wb):
        assert float((a - b).norm() / a.norm()) < 1e-5


# --------------------------------------------------------------------------------------------
# learn() at the shapes the hot learner kernels run, minibatch PPO, the configs' shard sizes
# --------------------------------------------------------------------------------------------
def _seeded_policy(tg, dev, g, kind, S, A):
    """The fixture's initial weights rebuilt from its seed (the constructors draw them from torch's CPU generator in the
    reference's order), checked against the fixture's samples, then moved to the GPU."""
    hidden = tuple(int(h) for h in g["hidden"])
    torch.manual_seed(int(g["seed"]))
    cls = tg.GaussianActorCritic_NeuralNetwork if kind == "ppo" else tg.GaussianActor_NeuralNetwork
    pol = cls(S, A, hidden, cov=float(g["cov"]), device="cpu")
    nets = ("actor", "critic") if kind == "ppo" else ("actor",)
    named = lambda: [(f"{n}.{k}", p) for n in nets for k, p in getattr(pol, n).named_parameters()]
    check_pinned(g, "init", named(), atol=0.0, sum_rtol=1e-9)
    return pol.to(dev), named


def _pinned_rel_errors(g, prefix, tensors):
    out = {}
    for k, t in tensors:
        stride = int(g[f"{prefix}_stride.{k}"])
        ref = g[f"{prefix}.{k}"].astype(np.float64)
        out[k] = float(np.linalg.norm(t.detach().double().reshape(-1).cpu().numpy()[::stride] - ref) / np.linalg.norm(ref))
    return out


@pytest.mark.parametrize("cdt", [None, torch.bfloat16])
@pytest.mark.parametrize("kind,tag,S,A", [("ppo", "h256", 20, 4), ("grpo", "h256", 20, 4), ("ppo", "h128", 5, 1), ("grpo", "h128", 5, 1)])
def test_learn_at_chain_kernel_shapes_matches_reference(tg, dev, kind, tag, S, A, cdt):
    """PPO.learn / GRPO.learn at 20-256x5 and 5-128x4 on ~4,000 rows, 2 updates, against the reference's loss scalars,
    gradients and post-step weights (pipelines/quadpole_pipeline_ppo.py:55-58, algorithms/grpo.py:106-148).
    The FIRST update's gradients are taken on the reference's own initial weights: one forward / backward pass, nothing else.
      fp32: h128 (5-128x4) runs on the fp32 chain learner (tg_mlp_f32_forward_backward / tg_mlp_f32_weight_grad -- asserted
        below: a gate that regresses must not silently re-test hipBLASLt), h256 on the H = 256 chain learner (tg_mlp_f32w_forward_backward; its weight gradients are
        are tg_mlp_f32_weight_grad jobs).  First gradients within 1e-2 in L2 per tensor and 1e-4 in the median -- torch autograd on the GPU sits at
        the same 3-5e-3 in three critic layers (a ReLU that flips for one row between the CPU's and the GPU's summation
        order), everything else at 1e-6; post-step weights <= 1e-5 (<= 0.5 % Adam-amplified outliers).  (The tight anchors of
        the fp32 chain kernels are test_f32_chain_update_matches_fp64_autograd and test_c2_size_grpo_learn_matches_the_oracle.)
      bf16 (tg_mlp_forward_chain / tg_mlp_backward_chain / tg_mlp_weight_grad): operands and stored activations are rounded
        to 8 significant bits, which costs torch's OWN bf16 autocast + autograd 0.3 % (head) to 10-12 % (first layer) of a
        first gradient in L2 at this depth.  The chain kernels must stay within 1.3 x that (measured in the same test with
        fused_mlp=False) and under 20 %; losses within 2e-3.  Later updates start from weights that already differ (Adam turns
        the sign of a noise-level gradient entry into a full +-lr step): second gradient and weight update within 35 % in L2."""
    g = load_golden(f"{kind}_step_{tag}.npz")

    def run(fused_mlp):
        pol, named = _seeded_policy(tg, dev, g, kind, S, A)
        init = {k: p.detach().clone() for k, p in named()}
        opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
        first = keep_first_step_gradients(opt, named)
        if kind == "ppo":
            algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, c1=0.5, kl_coeff=0.5,
                          gamma=float(g["gamma"]), lam=0.95, entropy=0.01, batch_size=None, autocast_dtype=cdt, fused_mlp=fused_mlp)
        else:
            algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=float(g["gamma"]), policy=pol, optimizer=opt, updates_per_iter=2,
                           autocast_dtype=cdt, fused_mlp=fused_mlp)
            gen = torch.Generator().manual_seed(int(g["seed"]))
            with torch.no_grad():
                for p_ in algo.old_policy.parameters():
                    p_.add_((float(g["old_policy_perturbation"]) * torch.randn(p_.shape, generator=gen)).to(dev))
        algo.learn(_buffer_from_golden(g))
        return pol, named, init, first, algo

    pol, named, init, first, algo = run(True)
    nets = [pol.actor] + ([pol.critic] if kind == "ppo" else [])
    if cdt is not None:                                           # the chain kernels really ran
        for net in nets:
            m = algo._mlp(net)
            assert m._chain is not None and m._bchain is not None and m._dw_ws is not None
    elif tag == "h128":                                           # ... and so did the fp32 chain learner
        for net in nets:
            m = algo._mlp(net)
            assert m._f32 is not None and m._dw_ws is not None, "5-128x4 fp32 fell off the fp32 chain learner"
    else:                                                         # 256 wide at the reference's precision: csrc/mlp_f32_wide.hip
        assert all(algo._mlp(net)._f32 is not None and algo._mlp(net)._f32.wide for net in nets), "20-256x5 fp32 fell off the H = 256 chain learner"
    st = algo.last_stats
    assert st["n_valid"] == int(g["n_valid"])
    lt = 2e-5 if cdt is None else 2e-3
    if kind == "ppo":
        np.testing.assert_allclose(st["total_loss"], g["total_loss"], rtol=lt, atol=lt)
        np.testing.assert_allclose(st["critic_loss"], g["critic_loss"], rtol=lt, atol=lt)
    else:
        np.testing.assert_allclose(st["J"], g["J"], rtol=10 * lt, atol=10 * lt)
    e1 = _pinned_rel_errors(g, "firstgrad", first.items())
    if cdt is None:
        assert max(e1.values()) < 1e-2 and float(np.median(list(e1.values()))) < 1e-4, e1
        check_pinned(g, "lastgrad", [(k, p.grad) for k, p in named()], norm_rel=5e-2, atol=1e-7)      # the same flips, one update on
        # (the three critic layers behind the flipping ReLU get a different second gradient: up to 10 % of their entries move)
        check_pinned(g, "final", named(), atol=1e-5, outlier_frac=0.10, outlier_atol=4 * float(g["lr"]), sum_rtol=1e-2)
    else:
        _, _, _, first_t, _ = run(False)                          # torch.autocast(bf16) + autograd on the same inputs
        et = _pinned_rel_errors(g, "firstgrad", first_t.items())
        for k in e1:
            assert e1[k] <= max(1.3 * et[k], 0.02) and e1[k] < 0.2, (k, e1[k], et[k])
        check_pinned(g, "lastgrad", [(k, p.grad) for k, p in named()], norm_rel=0.35, atol=1e-7)
        check_pinned(g, "delta", [(k, p.detach() - init[k]) for k, p in named()], norm_rel=0.35, atol=1e-7)


def test_ppo_minibatch_learn_matches_reference(tg, dev):
    """Minibatch PPO (algorithms/ppo.py:147-157) with the reference's own torch.randperm draws fed back as data.  The
    reference indexes its valid rows in (g, e, t) order, the device trajectory in (t, n) order: the recorded indices are
    mapped through the mask."""
    g = load_golden("ppo_minibatch.npz")
    pol = tg.GaussianActorCritic_NeuralNetwork(10, 2, (32, 32), cov=float(g["cov"]), device=dev)
    sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    pol.load_state_dict({"actor": {k[6:]: v for k, v in sd.items() if k.startswith("actor.")},
                         "critic": {k[7:]: v for k, v in sd.items() if k.startswith("critic.")}})
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    algo = tg.PPO(epsilon=float(g["epsilon"]), policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, c1=float(g["c1"]),
                  kl_coeff=float(g["kl_coeff"]), gamma=float(g["gamma"]), lam=0.95, entropy=float(g["entropy_coeff"]),
                  batch_size=int(g["batch_size"]))
    mask = torch.from_numpy(g["mask"]).bool()                    # (G, E, T)
    G, E, T = mask.shape
    ref_pos = torch.full((G * E, T), -1, dtype=torch.long)
    ref_pos[mask.reshape(G * E, T)] = torch.arange(int(mask.sum()))           # reference: env-major, time-minor
    ours = torch.full((int(mask.sum()),), -1, dtype=torch.long)
    tm = mask.reshape(G * E, T).t()                                           # device trajectory: time-major
    ours[ref_pos.t()[tm]] = torch.arange(int(mask.sum()))
    perms = iter([ours[torch.from_numpy(p)] for p in g["permutations"]])
    algo.permutation_fn = lambda n, device: next(perms).to(device)
    algo.learn(_buffer_from_golden(g))
    st = algo.last_stats
    assert len(st["total_loss"]) == len(g["total_loss"]) == 4
    np.testing.assert_allclose(st["total_loss"], g["total_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(st["critic_loss"], g["critic_loss"], rtol=2e-5, atol=2e-6)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-5)


def test_lazy_reference_view_copies_only_what_a_visualiser_reads(tg, dev):
    """with `limit_reference_view(max_episodes=k)` the group_* attributes are (G, k, T, .) slices cut on the
    device (visualize/dashboard.py:206-217 reads group_observations[i, ep < k, frame]); retrieve() / save_trajectory()
    still see everything."""
    T, G, Eps = 24, 3, 16
    torch.manual_seed(12)
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (32, 32), cov=0.5, device=dev)
    mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=T), pol, num_workers=G, num_episodes_per_worker=Eps, seed=3)
    buf = tg.Rollout_Buffer(mgr)
    buf.sample()
    full = [t.clone() for t in buf.retrieve()]
    buf.sample()                                                     # a new rollout drops the cached view
    buf.limit_reference_view(max_episodes=4)
    full = buf.device_traj.to_reference()
    assert buf.group_observations.shape == (G, 4, T, 5) and buf.group_lengths.shape == (G, 4)
    for name, ref in zip(tg.Rollout_Buffer._REF_FIELDS, full):
        assert torch.equal(getattr(buf, name), ref[:, :4])
    assert all(torch.equal(a, b) for a, b in zip(buf.retrieve(), full))
    assert buf.group_observations.shape == (G, Eps, T, 5)            # the full copy, once made, serves everyone
    sub = buf.device_traj.to_reference(max_groups=2, max_episodes=3)
    assert sub[0].shape == (2, 3, T, 5) and torch.equal(sub[2], full[2][:2, :3])
    # the pipeline scopes the slice to the visualiser's render(): inside the block the view is cut, outside it --
    # a reference-style CPU learner, a Publisher -- every episode is there again
    buf.limit_reference_view()                                       # (back to the default: everything)
    buf.sample()
    full = buf.device_traj.to_reference()

    class _Vis:
        max_episodes_per_render = 2
        seen = None

        def render(self):
            self.seen = buf.group_observations.clone()

    pipe = tg.Pipeline.__new__(tg.Pipeline)
    pipe.visualizer, pipe.buffer = _Vis(), buf
    pipe._render()
    assert pipe.visualizer.seen.shape == (G, 2, T, 5) and torch.equal(pipe.visualizer.seen, full[0][:, :2])
    assert buf.group_observations.shape == (G, Eps, T, 5) and torch.equal(buf.group_observations, full[0])


CONFIG_SHARDS = [
    # C4: QuadPole GRPO, 262,144 envs over 8 GPUs -> one rank: 128 restart groups x 256 episodes, bf16 (BASELINE configs[3])
    ("C4", "QuadPole", 1, 128, 256, True),
    # C5: swarm, 32,768 envs x 8 agents over 8 GPUs -> one rank: 4,096 envs x 8 bodies = 64 groups x 64 episodes x 8 (configs[4])
    ("C5", "QuadPoleSwarm", 8, 64, 64, True),
]


@pytest.mark.parametrize("cfg,env_name,agents,G,Eps,restart", CONFIG_SHARDS)
def test_grpo_configs_at_their_per_gpu_shard_size(tg, dev, cfg, env_name, agents, G, Eps, restart):
    """BASELINE.json configs[3] / configs[4] at ONE rank's shard size, end to end (rollout + GRPO.learn, bf16 policy):
    restart groups share their initial state, masks are prefixes of len, padding is zero, the group-relative advantages
    have zero mean / unit (unbiased) std per group over the valid steps, learn() is finite, moves the weights and is
    bit-reproducible from the same seeds."""
    T = 256
    mk = (lambda: tg.QuadPoleSwarm(n_agents=agents, max_steps=T)) if agents > 1 else (lambda: tg.QuadPole(max_steps=T))

    def run():
        torch.manual_seed(13)
        pol = tg.GaussianActor_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
        mgr = tg.RolloutManager(mk, pol, restart=restart, num_workers=G, num_episodes_per_worker=Eps, seed=41,
                                compute_dtype=torch.bfloat16)
        assert mgr.engine.fused
        buf = tg.Rollout_Buffer(mgr)
        buf.sample()
        algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.99, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4),
                       updates_per_iter=2, autocast_dtype=torch.bfloat16)
        algo.learn(buf)
        torch.cuda.synchronize()
        return pol, buf, algo

    pol, buf, algo = run()
    tr = buf.device_traj
    n = G * Eps * agents
    assert tr.n == n and tr.G == G and tr.E == Eps * agents and tr.T == T
    ln, mask = tr.len, tr.mask
    assert torch.equal(mask, (torch.arange(T, device=dev)[:, None] < ln[None, :]).to(torch.uint8))
    assert int(ln.min()) >= 1 and int(ln.max()) <= T and tr.env_steps() == int(ln.sum()) == int(mask.sum())
    m = mask.bool()
    assert torch.all(tr.rew[~m] == 0) and torch.all(tr.act[:, ~m] == 0) and torch.all(tr.obs[:, :T][:, ~m] == 0)
    first = tr.obs[:, 0, :].t().reshape(G, Eps * agents, 20)
    if agents == 1:
        assert torch.equal(first, first[:, :1].expand_as(first))                 # restart: one initial state per group
    else:
        per_env = ln.view(-1, agents)
        assert torch.equal(per_env, per_env[:, :1].expand_as(per_env))           # the bodies of an env stop together
    rtg = tg.hip_ops.rtg_scan(tr.rew, tr.mask, 0.99)
    adv = tg.hip_ops.group_normalize(rtg, tr.mask, tg.hip_ops.masked_moments(rtg, tr.mask, tr.E), 0, tr.E)
    a, mm = adv.t().reshape(G, -1), mask.t().reshape(G, -1).bool()
    for gi in range(0, G, max(1, G // 8)):
        v = a[gi][mm[gi]].double()
        assert abs(float(v.mean())) < 1e-3 and abs(float(v.std()) - 1) < 1e-3
    assert np.isfinite(algo.last_stats["J"]).all() and algo.last_stats["n_valid"] == tr.env_steps()
    assert all(torch.isfinite(p).all() for p in pol.parameters())
    pol2, buf2, algo2 = run()
    assert torch.equal(buf2.device_traj.obs, tr.obs) and algo2.last_stats["J"] == algo.last_stats["J"]
    for p, q in zip(pol.parameters(), pol2.parameters()):
        assert torch.equal(p, q)


def test_ppo_learn_at_c3_size_is_finite_and_deterministic(tg, dev):
    """BASELINE configs[2] at full size: 65,536 QuadPole envs x 256 steps, bf16 actor-critic 20-256x5-{4,1}, PPO.learn
    (4 updates here; the factory's 32 repeat the same step): finite losses and weights, n_valid = sum of the mask,
    bit-identical weights and statistics from two runs with the same seeds."""
    T, G, Eps = 256, 256, 256

    def run():
        torch.manual_seed(14)
        pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
        mgr = tg.RolloutManager(lambda: tg.QuadPole(max_steps=T), pol, num_workers=G, num_episodes_per_worker=Eps, seed=51,
                                compute_dtype=torch.bfloat16)
        buf = tg.Rollout_Buffer(mgr)
        buf.sample()
        algo = tg.PPO(epsilon=0.2, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), ref_model=None,
                      updates_per_iter=4, gamma=0.999, batch_size=None, autocast_dtype=torch.bfloat16)
        algo.learn(buf)
        torch.cuda.synchronize()
        return pol, buf, algo

    pol, buf, algo = run()
    st = algo.last_stats
    assert st["n_valid"] == float(buf.device_traj.mask.sum()) == buf.device_traj.env_steps()
    assert np.isfinite(st["total_loss"]).all() and np.isfinite(st["critic_loss"]).all() and len(st["total_loss"]) == 4
    assert abs(st["actor_loss"][0]) < 1e-3 and abs(st["kl_div"][0]) < 1e-6        # first pass: ratio == 1 (ppo.py:142-143)
    assert all(torch.isfinite(p).all() for p in pol.parameters())
    pol2, buf2, algo2 = run()
    assert algo2.last_stats["total_loss"] == st["total_loss"]
    for p, q in zip(pol.parameters(), pol2.parameters()):
        assert torch.equal(p, q)
