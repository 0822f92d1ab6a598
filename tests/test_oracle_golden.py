"""CPU: the oracle restatement against the golden vectors generated from the real
reference (oracle/tools/gen_goldens.py) and against the reference tests' known answers."""
import numpy as np
import pytest
import torch

from conftest import check_pinned, keep_first_step_gradients, load_golden
from oracle import envs as E
from oracle import learner as L

ENVS = ["CartPole", "QuadPole2D", "QuadPole"]


@pytest.mark.parametrize("name", ENVS)
def test_single_step_matches_reference(name):
    g = load_golden(f"env_step_{name.lower()}.npz")
    nxt, rew, trunc, steps, tb = E.ENV_SPECS[name]["step"](
        g["state"], g["action"], g["steps"], g["time_balanced"], max_steps=int(g["max_steps"]))
    np.testing.assert_allclose(nxt, g["next_state"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(trunc, g["truncated"])                      # bit-exact flags
    if name != "QuadPole":
        np.testing.assert_allclose(tb, g["time_balanced_after"], rtol=0, atol=1e-15)
    assert trunc.sum() > 8 and (~trunc).sum() > 100


@pytest.mark.parametrize("tag,name", [("cartpole_custom", "CartPole"), ("quadpole2d_custom", "QuadPole2D")])
def test_single_step_with_non_default_constructor_arguments(tag, name):
    g = load_golden(f"env_step_{tag}.npz")
    kw = {k[len("param_"):]: float(g[k]) for k in g if k.startswith("param_")}
    nxt, rew, trunc, _, _ = E.ENV_SPECS[name]["step"](
        g["state"], g["action"], g["steps"], g["time_balanced"], max_steps=int(g["max_steps"]), **kw)
    np.testing.assert_allclose(nxt, g["next_state"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(trunc, g["truncated"])


def test_cartpole_upright_known_answer():
    # reference tests/test_cartpole.py:91-104: upright, zero action -> 0 < r < 5 (closed form 2.8)
    nxt, rew, trunc, _, tb = E.cartpole_step(np.array([[0, 0, 0, 1, 0.0]]), np.zeros((1, 1), np.float32), [0], [0.0])
    assert 0 < rew[0] < 5
    assert rew[0] == pytest.approx(2.8, abs=1e-12)
    assert not trunc[0] and tb[0] == pytest.approx(0.02)


def test_cartpole_time_clause_is_inert_under_worker_cap():
    for ms in (10, 64, 100, 128, 256, 500):
        assert E.cartpole_time_trunc_step(ms) >= ms


def test_quadrotor_dynamics():
    g = load_golden("quadrotor_dynamics.npz")
    np.testing.assert_allclose(E.quadrotor_dynamics(g["state"], g["control"]), g["next_state"], rtol=1e-12, atol=1e-13)


def _policy_from_golden(g, prefix, S, A, hidden, cov, critic):
    pol = L.OraclePolicy(S, A, hidden, cov=[float(c) for c in cov], critic=critic)
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    if critic:
        pol.actor.load_state_dict({k[len("actor."):]: v for k, v in sd.items() if k.startswith("actor.")})
        pol.critic.load_state_dict({k[len("critic."):]: v for k, v in sd.items() if k.startswith("critic.")})
    else:
        pol.actor.load_state_dict(sd)
    return pol


@pytest.mark.parametrize("name,S,A,hidden,critic,T", [
    ("CartPole", 5, 1, (128, 128), False, 128),
    ("QuadPole2D", 10, 2, (32, 32), True, 128),
    ("QuadPole", 20, 4, (64, 64), True, 256)])
@pytest.mark.parametrize("tag", ["reset", "restart"])
def test_teacher_forced_rollout_matches_reference(name, S, A, hidden, critic, T, tag):
    g = load_golden(f"rollout_{name.lower()}.npz")
    obs, act, rew, ln, mask = (g[f"{tag}_{k}"] for k in ("obs", "act", "rew", "len", "mask"))
    G, Eps = ln.shape
    assert obs.shape == (G, Eps, T, S) and act.shape == (G, Eps, T, A)    # reference tests/test_rollout_manager.py:40-53
    assert rew.shape == (G, Eps, T) and mask.shape == (G, Eps, T)
    init = obs[:, :, 0, :].astype(np.float64)
    o2, a2, r2, l2, m2 = L.rollout(lambda: L.OracleEnv(name, max_steps=T), None, G, Eps,
                                   restart=(tag == "restart"), initial_states=init, forced_actions=act)
    assert np.array_equal(l2.numpy(), ln)            # bit-exact lengths
    assert np.array_equal(m2.numpy(), mask)          # bit-exact masks
    assert l2.dtype == torch.float32
    # initial obs went through a float32 round trip in the fixture -> trajectories agree to fp32 noise
    np.testing.assert_allclose(o2.numpy(), obs, rtol=0, atol=5e-4)
    np.testing.assert_allclose(r2.numpy(), rew, rtol=2e-4, atol=2e-4)
    assert np.array_equal(a2.numpy(), act)
    # zero padding beyond each episode
    assert np.all(obs[mask == 0] == 0) and np.all(o2.numpy()[m2.numpy() == 0] == 0)
    if tag == "restart":
        assert np.array_equal(obs[:, 1:, 0, :], np.broadcast_to(obs[:, :1, 0, :], obs[:, 1:, 0, :].shape))


@pytest.mark.parametrize("kind", ["actor", "actorcritic"])
def test_policy_closed_forms(kind):
    g = load_golden(f"policy_{kind}.npz")
    pol = _policy_from_golden(g, "policy.", 20, 4, (64, 64), g["cov"], kind == "actorcritic")
    obs = torch.from_numpy(g["obs"])
    mean = pol.actor(obs).detach()
    np.testing.assert_allclose(mean.numpy(), g["mean"], rtol=0, atol=1e-6)
    action = mean + torch.sqrt(pol.var) * torch.from_numpy(g["eps"])
    np.testing.assert_allclose(action.numpy(), g["action"], rtol=0, atol=1e-6)
    lp, ent = pol.log_prob(obs, torch.from_numpy(g["action"]))
    np.testing.assert_allclose(lp.detach().numpy(), g["logp_eval"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(lp.detach().numpy(), g["logp"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(ent.numpy(), g["entropy"], rtol=0, atol=1e-6)
    if kind == "actorcritic":
        np.testing.assert_allclose(pol.value(obs).detach().numpy(), g["value_squeezed"], rtol=0, atol=1e-6)
    # sampling identity: MultivariateNormal.sample() == mean + sqrt(var) * randn, same stream
    torch.manual_seed(9)
    a1, _, _ = pol(g["obs"][0])
    np.testing.assert_allclose(a1, g["action_row0"], rtol=0, atol=1e-6)


def test_rtg_known_answer_from_reference_test():
    # reference tests/test_rollout_buffer.py:76-92 (gamma 0.99, no masks)
    rew = np.array([[[1, 2, 3], [0, 1, 2]], [[3, 2, 1], [1, 0, 1]]], dtype=np.float32)
    exp = np.zeros_like(rew)
    for j in range(2, -1, -1):
        exp[:, :, j] = rew[:, :, j] + (0.99 * exp[:, :, j + 1] if j < 2 else 0)
    got = L.rtg_scan(torch.from_numpy(rew), torch.ones(2, 2, 3), 0.99).numpy()
    np.testing.assert_allclose(got, exp, atol=1e-5)


@pytest.mark.parametrize("gamma", [0.5, 0.99, 0.999])
def test_returns_and_advantages(gamma):
    g = load_golden("rtg_adv.npz")
    rew, mask = torch.from_numpy(g["rew"]), torch.from_numpy(g["mask"])
    tag = f"g{gamma}"
    rtg = L.rtg_scan(rew, mask, gamma)
    np.testing.assert_allclose(rtg.numpy(), g[f"{tag}_rtg"], rtol=0, atol=1e-5)
    for i, adv in enumerate(L.grpo_group_advantages(rtg, mask)):
        np.testing.assert_allclose(adv.numpy(), g[f"{tag}_grpo_adv_{i}"], rtol=1e-5, atol=1e-5)
    values = torch.from_numpy(g[f"{tag}_values"])
    for kind, mc in (("mc", True), ("gae", False)):
        adv, ret = L.ppo_advantages(rew, mask, values, gamma, 0.95, mc)
        np.testing.assert_allclose(adv.numpy(), g[f"{tag}_ppo_{kind}_adv"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret.numpy(), g[f"{tag}_ppo_{kind}_ret"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n_upd", [1, 2])
def test_grpo_step(n_upd):
    g = load_golden(f"grpo_step_u{n_upd}.npz")
    pol = _policy_from_golden(g, "init.", 5, 1, (32, 32), [float(g["cov"])], False)
    old = _policy_from_golden(g, "old_init.", 5, 1, (32, 32), [float(g["cov"])], False)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    t = lambda k: torch.from_numpy(g[k])
    Js = L.grpo_learn(pol, old, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=float(g["epsilon"]),
                      gamma=float(g["gamma"]), updates_per_iter=n_upd)
    np.testing.assert_allclose(Js, g["J"], rtol=1e-4, atol=2e-6)
    for k, p in pol.actor.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g[f"final.{k}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(p.grad.numpy(), g[f"lastgrad.{k}"], rtol=1e-3, atol=1e-5)
    # sign convention F6: the reference DEscends on J.  With lr>0 and old==policy on the first
    # pass (ratio == 1) the first-order change of J is -lr * |g|_adam <= 0.
    assert float(g["J"][0]) == pytest.approx(Js[0], abs=2e-6)


@pytest.mark.parametrize("n_upd", [1, 2])
def test_ppo_step(n_upd):
    g = load_golden(f"ppo_step_u{n_upd}.npz")
    pol = _policy_from_golden(g, "init.", 10, 2, (32, 32), [float(g["cov"])] * 2, True)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    t = lambda k: torch.from_numpy(g[k])
    logs = L.ppo_learn(pol, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=float(g["epsilon"]),
                       gamma=float(g["gamma"]), c1=float(g["c1"]), kl_coeff=float(g["kl_coeff"]),
                       entropy_coeff=float(g["entropy_coeff"]), updates_per_iter=n_upd)
    np.testing.assert_allclose([l["total"] for l in logs], g["total_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([l["actor"] for l in logs], g["actor_loss"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose([l["critic"] for l in logs], g["critic_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([l["kl"] for l in logs], g["kl_div"], rtol=1e-4, atol=1e-7)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.detach().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(p.grad.numpy(), g[f"lastgrad.{net}.{k}"], rtol=1e-3, atol=1e-6)


def test_ppo_step_with_gae_and_per_dimension_covariance():
    g = load_golden("ppo_gae_step_u2.npz")
    pol = _policy_from_golden(g, "init.", 10, 2, (32, 32), [float(c) for c in g["cov"]], True)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    t = lambda k: torch.from_numpy(g[k])
    logs = L.ppo_learn(pol, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=float(g["epsilon"]),
                       gamma=float(g["gamma"]), lam=float(g["lam"]), c1=float(g["c1"]), kl_coeff=float(g["kl_coeff"]),
                       entropy_coeff=float(g["entropy_coeff"]), updates_per_iter=2, monte_carlo=False)
    np.testing.assert_allclose([l["total"] for l in logs], g["total_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([l["critic"] for l in logs], g["critic_loss"], rtol=1e-5, atol=1e-6)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.detach().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("kind,tag,S,A", [("ppo", "h128", 5, 1), ("grpo", "h128", 5, 1), ("ppo", "h256", 20, 4), ("grpo", "h256", 20, 4)])
def test_learn_at_chain_kernel_shapes(kind, tag, S, A):
    """learn() at the net shapes the hot learner kernels run (128 x 4, 256 x 5; ~4,000 rows, 2 updates).  The initial
    weights are rebuilt from the fixture's seed (the constructors draw them from torch's CPU generator, as the
    reference's do) and checked against the fixture's samples before anything else."""
    g = load_golden(f"{kind}_step_{tag}.npz")
    hidden = tuple(int(h) for h in g["hidden"])
    torch.manual_seed(int(g["seed"]))
    pol = L.OraclePolicy(S, A, hidden, cov=float(g["cov"]), critic=kind == "ppo")
    nets = ("actor", "critic") if kind == "ppo" else ("actor",)
    named = lambda: [(f"{n}.{k}", p) for n in nets for k, p in getattr(pol, n).named_parameters()]
    check_pinned(g, "init", named(), atol=0.0, sum_rtol=1e-9)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    first = keep_first_step_gradients(opt, named)
    t = lambda k: torch.from_numpy(g[k])
    if kind == "ppo":
        logs = L.ppo_learn(pol, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=0.2, gamma=float(g["gamma"]), c1=0.5, kl_coeff=0.5,
                           entropy_coeff=0.01, updates_per_iter=2)
        np.testing.assert_allclose([l["total"] for l in logs], g["total_loss"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose([l["critic"] for l in logs], g["critic_loss"], rtol=1e-5, atol=1e-6)
    else:
        old = L.OraclePolicy(S, A, hidden, cov=float(g["cov"]), critic=False)
        old.load_state_dict(pol.state_dict())
        gen = torch.Generator().manual_seed(int(g["seed"]))
        with torch.no_grad():
            for p_ in old.parameters():
                p_.add_(float(g["old_policy_perturbation"]) * torch.randn(p_.shape, generator=gen))
        Js = L.grpo_learn(pol, old, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=0.15, gamma=float(g["gamma"]), updates_per_iter=2)
        np.testing.assert_allclose(Js, g["J"], rtol=2e-4, atol=1e-5)
    # <= 1e-5 (north_star: fp32 within 1e-5); <= 0.5 % of the entries may be Adam-amplified rounding noise (up to 2 lr steps)
    check_pinned(g, "final", named(), atol=1e-5, outlier_frac=0.005, outlier_atol=4 * float(g["lr"]), sum_rtol=1e-3)
    check_pinned(g, "firstgrad", list(first.items()), norm_rel=1e-4, atol=1e-8)
    check_pinned(g, "lastgrad", [(k, p.grad) for k, p in named()], norm_rel=1e-3, atol=1e-7)


def test_ppo_minibatch_step():
    """Minibatch PPO (algorithms/ppo.py:147-157), the reference's recorded torch.randperm draws fed back as data."""
    g = load_golden("ppo_minibatch.npz")
    pol = _policy_from_golden(g, "init.", 10, 2, (32, 32), [float(g["cov"])] * 2, True)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g["lr"]))
    t = lambda k: torch.from_numpy(g[k])
    logs = L.ppo_learn(pol, opt, t("obs"), t("act"), t("rew"), t("mask"), epsilon=float(g["epsilon"]), gamma=float(g["gamma"]),
                       c1=float(g["c1"]), kl_coeff=float(g["kl_coeff"]), entropy_coeff=float(g["entropy_coeff"]), updates_per_iter=2,
                       batch_size=int(g["batch_size"]), permutations=g["permutations"])
    assert len(logs) == len(g["total_loss"]) == 4                      # 2 updates x ceil(n_valid / 64) steps
    np.testing.assert_allclose([l["total"] for l in logs], g["total_loss"], rtol=1e-5, atol=1e-6)
    for net in ("actor", "critic"):
        for k, p in getattr(pol, net).named_parameters():
            np.testing.assert_allclose(p.detach().numpy(), g[f"final.{net}.{k}"], rtol=0, atol=2e-6)


# --------------------------------------------------------------------------------------------
# Pendulum: the env that can terminate (time_balanced > 5)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["default", "custom"])
def test_pendulum_single_step_matches_reference(tag):
    g = load_golden(f"env_step_pendulum_{tag}.npz")
    kw = {k[len("param_"):]: float(g[k]) for k in g if k.startswith("param_")}
    nxt, rew, trunc, _, tb = E.pendulum_step(g["state"], g["action"], g["steps"], g["time_balanced"],
                                             max_steps=int(g["max_steps"]), **kw)
    np.testing.assert_allclose(nxt, g["next_state"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(trunc, g["truncated"])
    np.testing.assert_allclose(tb, g["time_balanced_after"], rtol=0, atol=1e-15)
    assert np.array_equal(tb > E.PENDULUM_BALANCE_TIME, g["terminated"]) and g["terminated"].sum() > 4
    # the balanced-step count at which the accumulated time first exceeds 5 s
    dt = kw.get("timestep", 0.05)
    k = E.pendulum_balance_term_steps(dt)
    assert k == {0.05: 101, 0.02: 251}[dt]


@pytest.mark.parametrize("tag,T,ends", [("fall", 64, 64), ("hold", 140, 101)])
def test_pendulum_teacher_forced_rollout_matches_reference(tag, T, ends):
    g = load_golden("rollout_pendulum.npz")
    obs, act, rew, ln, mask = (g[f"{tag}_{k}"] for k in ("obs", "act", "rew", "len", "mask"))
    G, Eps = ln.shape
    assert int(g[f"{tag}_max_steps"]) == T and np.all(ln == ends)          # 'hold': terminated after 101 balanced steps
    init = obs[:, :, 0, :].astype(np.float64)
    o2, a2, r2, l2, m2 = L.rollout(lambda: L.OracleEnv("Pendulum", max_steps=T, gravity=float(g[f"{tag}_gravity"])), None, G, Eps,
                                   initial_states=init, forced_actions=act)
    assert np.array_equal(l2.numpy(), ln) and np.array_equal(m2.numpy(), mask)
    # the initial state went through a float32 round trip in the fixture -> trajectories agree to fp32 noise
    np.testing.assert_allclose(o2.numpy(), obs, rtol=0, atol=5e-4)
    np.testing.assert_allclose(r2.numpy(), rew, rtol=2e-4, atol=2e-4)
