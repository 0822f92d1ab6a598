#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the real reference.

Runs only in the build container (the reference checkout is read-only at
/root/reference and never travels).  The fixtures are data: inputs and the
reference's outputs.  No reference source text is stored.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 \
    PYTHONPATH=oracle/tools/gymnasium_standin:/root/reference \
    python oracle/tools/gen_goldens.py

`gymnasium` is absent from this image; `oracle/tools/gymnasium_standin` supplies
the tiny surface the reference touches.  Locals of `GRPO.learn` /
`PPO.learn` (reward-to-go, advantages, loss scalars) are not returned by the
reference, so they are read out of the running frames through the hooks below.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, os.environ.get("TRAJOPT_REFERENCE", "/root/reference"))

from environments import CartPole, QuadPole, QuadPole2D  # noqa: E402  (reference)
from environments.quadrotor_env import Quadrotor  # noqa: E402
from policies import GaussianActor_NeuralNetwork, GaussianActorCritic_NeuralNetwork  # noqa: E402
from rollout import RolloutManager  # noqa: E402
from algorithms import GRPO, PPO  # noqa: E402

torch.set_num_threads(1)


def sd_to_np(sd, prefix=""):
    out = {}
    for k, v in sd.items():
        if isinstance(v, dict):
            out.update(sd_to_np(v, prefix + k + "."))
        else:
            out[prefix + k] = v.detach().numpy().copy()
    return out


# ---------------------------------------------------------------------------
# 1. single-step maps
# ---------------------------------------------------------------------------
def set_env_state(env, name, st, steps, tb):
    if name == "CartPole":
        env.state_dict["cartpole"] = st.copy()
        env._time = 0
        for _ in range(int(steps)):
            env._time += env.timestep
    elif name == "QuadPole":
        env.reset()
        env.state_dict["quadrotor"] = st[:13].copy()
        env.state_dict["pendulum"] = st[13:].copy()
    else:
        env.reset()
        env.state_dict["quadrotor"] = st[:8].copy()
        env.state_dict["pendulum"] = st[8:].copy()
    env._steps = int(steps)
    env._time_balanced = float(tb)


def gen_env_steps(rng):
    MAXS = 100
    N = 256
    for name, cls, S, A in [("CartPole", CartPole, 5, 1), ("QuadPole2D", QuadPole2D, 10, 2),
                            ("QuadPole", QuadPole, 20, 4)]:
        env = cls(max_steps=MAXS)
        st = rng.normal(size=(N, S)) * 0.6
        act = (rng.normal(size=(N, A)) * 0.9).astype(np.float32)
        steps = rng.integers(0, MAXS - 1, size=N)
        tb = rng.choice([0.0, 0.02, 0.4], size=N)
        steps[:8] = MAXS - 1                                   # step == max_steps-1 -> truncation by count
        act[8:16] = np.where(rng.random((8, A)) < 0.5, -3.0, 2.5)  # |a| > 1 clipping
        act[16] = 1.0
        act[17] = -1.0
        act[18] = 0.0
        if name == "CartPole":
            ang = rng.uniform(-np.pi, np.pi, size=N)
            st[:, 2], st[:, 3] = np.sin(ang), np.cos(ang)
            st[:, 4] = rng.normal(size=N) * 4
            st[20:28, 4] = rng.choice([-14.0, 12.5, 10.0, -10.0], size=8)       # thetadot clamp
            st[28:40, 0] = rng.choice([0.99, 1.0, 1.001, -0.999, -1.002, 0.9995], size=12)
            st[28:40, 1] = rng.normal(size=12) * 0.3
            # balanced neighbourhood: |x|<0.1, cos>0.95, |thetadot|<0.1
            k = slice(40, 80)
            st[k, 0] = rng.uniform(-0.12, 0.12, size=40)
            st[k, 1] = rng.normal(size=40) * 0.05
            a2 = rng.uniform(-0.35, 0.35, size=40)
            st[k, 2], st[k, 3] = np.sin(a2), np.cos(a2)
            st[k, 4] = rng.uniform(-0.12, 0.12, size=40)
            act[k] *= 0.02
            st[80] = [0, 0, 0, 1, 0]                          # reference test_cartpole.py:91-104
            act[80] = 0.0
            steps[80], tb[80] = 0, 0.0
            st[81:90, 2:4] *= rng.uniform(0.5, 1.5, size=(9, 1))    # un-normalised (sin, cos)
        elif name == "QuadPole2D":
            for i0 in (4, 7):
                ang = rng.uniform(-np.pi, np.pi, size=N)
                st[:, i0], st[:, i0 + 1] = np.sin(ang), np.cos(ang)
            st[:, 0:2] = rng.uniform(-1.9, 1.9, size=(N, 2))
            st[20:32, 0] = rng.choice([1.99, 2.0, 2.01, -1.995, -2.02], size=12)     # x bounds
            st[32:44, 1] = rng.choice([1.99, 2.0, 2.01, -1.995, -2.02], size=12)     # z bounds
            st[20:44, 2:4] = rng.normal(size=(24, 2)) * 0.5
            k = slice(44, 90)                                 # balanced neighbourhood
            st[k, 0:2] = rng.uniform(-0.2, 0.2, size=(46, 2))
            st[k, 2:4] = rng.normal(size=(46, 2)) * 0.05
            a2 = np.pi + rng.uniform(-0.35, 0.35, size=46)
            st[k, 7], st[k, 8] = np.sin(a2), np.cos(a2)
            st[k, 9] = rng.uniform(-0.12, 0.12, size=46)
            act[k] *= 0.05
            st[90:100, 4:6] *= rng.uniform(0.5, 1.5, size=(10, 1))
        else:
            st[:, 0:3] = rng.uniform(-1.4, 1.4, size=(N, 3))
            for sl in (slice(6, 10), slice(13, 17)):
                q = rng.normal(size=(N, 4))
                st[:, sl] = q / np.linalg.norm(q, axis=1, keepdims=True)
            for ax in range(3):
                k = slice(20 + 10 * ax, 30 + 10 * ax)
                st[k, ax] = rng.choice([1.49, 1.5, 1.51, -1.495, -1.52], size=10)
                st[k, 3 + ax] = rng.normal(size=10) * 0.5
            st[50:60] = 0.0                                   # hover from the reset state
            st[50:60, 6] = 1.0
            al, be = rng.uniform(-1, 1, size=10), rng.uniform(-1, 1, size=10)
            st[50:60, 13] = np.cos(be / 2) * np.cos(al / 2)
            st[50:60, 14] = np.cos(be / 2) * np.sin(al / 2)
            st[50:60, 15] = np.sin(be / 2) * np.cos(al / 2)
            st[50:60, 16] = -np.sin(be / 2) * np.sin(al / 2)
            st[60:70, 6:10] *= rng.uniform(0.8, 1.2, size=(10, 1))  # un-normalised quaternion
        nxt = np.zeros((N, S))
        rew = np.zeros(N)
        trunc = np.zeros(N, dtype=bool)
        tb_after = np.zeros(N)
        info_tb = np.zeros(N)
        for i in range(N):
            set_env_state(env, name, st[i], steps[i], tb[i])
            o, r, term, tr, info = env.step(act[i])
            assert term is False
            nxt[i], rew[i], trunc[i] = np.asarray(o, dtype=np.float64), float(r), bool(tr)
            tb_after[i], info_tb[i] = env._time_balanced, info["time_balanced"]
        np.savez_compressed(os.path.join(OUT, f"env_step_{name.lower()}.npz"), state=st, action=act,
                            steps=steps, time_balanced=tb, max_steps=MAXS, next_state=nxt, reward=rew,
                            truncated=trunc, time_balanced_after=tb_after, info_time_balanced=info_tb)
        print(name, "single-step: truncated", int(trunc.sum()), "reward range", rew.min(), rew.max())

    # non-default constructor arguments (the physical parameters are live attributes in the reference)
    rng2 = np.random.default_rng(777)   # own stream: the fixtures generated after this keep their values
    envs = {
        "cartpole_custom": (CartPole(masscart=2.0, masspole=0.3, length=0.8, gravity=9.0, timestep=0.01, max_steps=MAXS),
                            "CartPole", 5, 1, dict(masscart=2.0, masspole=0.3, length=0.8, gravity=9.0, timestep=0.01)),
        "quadpole2d_custom": (QuadPole2D(max_steps=MAXS, timestep=0.01), "QuadPole2D", 10, 2, dict(timestep=0.01)),
    }
    for tag, (env, name, S, A, kw) in envs.items():
        n = 96
        st = rng2.normal(size=(n, S)) * 0.5
        for i0 in ((2,) if name == "CartPole" else (4, 7)):
            ang = rng2.uniform(-np.pi, np.pi, size=n)
            st[:, i0], st[:, i0 + 1] = np.sin(ang), np.cos(ang)
        act = (rng2.normal(size=(n, A)) * 0.9).astype(np.float32)
        steps = rng2.integers(0, MAXS - 1, size=n)
        tb = np.zeros(n)
        nxt, rew, trunc = np.zeros((n, S)), np.zeros(n), np.zeros(n, dtype=bool)
        for i in range(n):
            set_env_state(env, name, st[i], steps[i], tb[i])
            o, r, term, tr, info = env.step(act[i])
            nxt[i], rew[i], trunc[i] = np.asarray(o, dtype=np.float64), float(r), bool(tr)
        np.savez_compressed(os.path.join(OUT, f"env_step_{tag}.npz"), state=st, action=act, steps=steps, time_balanced=tb,
                            max_steps=MAXS, next_state=nxt, reward=rew, truncated=trunc,
                            **{f"param_{k}": v for k, v in kw.items()})

    # Quadrotor._dynamics pure function (class is a stub)
    q = Quadrotor()
    st = rng.normal(size=(64, 12)) * 0.5
    ctl = rng.uniform(0, 5, size=(64, 4))
    nxt = np.stack([q._dynamics(st[i], ctl[i]) for i in range(64)])
    np.savez_compressed(os.path.join(OUT, "quadrotor_dynamics.npz"), state=st, control=ctl, next_state=nxt)


# ---------------------------------------------------------------------------
# 2. in-process rollouts (the numerical oracle path)
# ---------------------------------------------------------------------------
def gen_rollouts():
    cfgs = [
        ("CartPole", lambda: CartPole(max_steps=128), 2, 2,
         lambda: GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5)),
        ("QuadPole2D", lambda: QuadPole2D(max_steps=128), 2, 3,
         lambda: GaussianActorCritic_NeuralNetwork(10, 2, (32, 32), cov=0.5)),
        ("QuadPole", lambda: QuadPole(max_steps=256), 2, 4,
         lambda: GaussianActorCritic_NeuralNetwork(20, 4, (64, 64), cov=0.3)),
    ]
    for name, env_fn, G, Eps, pol_fn in cfgs:
        out = {}
        for restart in (False, True):
            seed = 7 if not restart else 11
            np.random.seed(seed)
            torch.manual_seed(seed)
            policy = pol_fn()
            mgr = RolloutManager(env_fn=env_fn, policy=policy, restart=restart, num_workers=G,
                                 num_episodes_per_worker=Eps, use_multiprocessing=False)
            obs, act, rew, ln, mask = mgr.rollout()
            assert list(mgr.episodes_completed) == [Eps] * G
            tag = "restart" if restart else "reset"
            out.update({f"{tag}_obs": obs.numpy(), f"{tag}_act": act.numpy(), f"{tag}_rew": rew.numpy(),
                        f"{tag}_len": ln.numpy(), f"{tag}_mask": mask.numpy()})
            for k, v in sd_to_np(policy.state_dict()).items():
                out[f"{tag}_policy.{k}"] = v
            out[f"{tag}_cov"] = np.diag(policy.cov.numpy())
            print(name, tag, "lengths", ln.numpy().astype(int).tolist(), "return", float(rew.sum(2).mean()))
        np.savez_compressed(os.path.join(OUT, f"rollout_{name.lower()}.npz"), **out)


# ---------------------------------------------------------------------------
# 3. policy closed forms
# ---------------------------------------------------------------------------
def gen_policy():
    for kind, cls in (("actor", GaussianActor_NeuralNetwork), ("actorcritic", GaussianActorCritic_NeuralNetwork)):
        torch.manual_seed(3)
        pol = cls(20, 4, (64, 64), cov=[0.3, 0.2, 0.5, 0.1])
        obs = torch.randn(33, 20)
        torch.manual_seed(5)
        action, logp, value = pol(obs)
        torch.manual_seed(5)
        eps = torch.randn(33, 4)
        mu = pol.actor(obs).detach()
        assert np.array_equal(action, (mu + torch.sqrt(torch.diag(pol.cov)) * eps).numpy())
        logp2, ent = pol.log_prob(obs, torch.from_numpy(action))
        # single-row call, the shape the rollout worker uses
        torch.manual_seed(9)
        a1, lp1, v1 = pol(obs[0].numpy())
        torch.manual_seed(9)
        eps1 = torch.randn(4)
        out = dict(obs=obs.numpy(), eps=eps.numpy(), mean=mu.numpy(), action=action,
                   logp=logp.detach().numpy(), logp_eval=logp2.detach().numpy(), entropy=ent.detach().numpy(),
                   cov=np.diag(pol.cov.numpy()), action_row0=a1, logp_row0=lp1.detach().numpy(), eps_row0=eps1.numpy())
        if value is not None:
            out["value"] = value.detach().numpy()
            out["value_squeezed"] = pol.value(obs).detach().numpy()
        for k, v in sd_to_np(pol.state_dict()).items():
            out[f"policy.{k}"] = v
        np.savez_compressed(os.path.join(OUT, f"policy_{kind}.npz"), **out)


# ---------------------------------------------------------------------------
# 4/5. learner: returns, advantages, loss, one/two optimizer steps
# ---------------------------------------------------------------------------
class FakeBuffer:
    pass


def ragged_buffer(rng, G, Eps, T, S, A, reward_scale=1.0):
    buf = FakeBuffer()
    lengths = rng.integers(2, T + 1, size=(G, Eps))
    lengths[0, 0] = T
    lengths[-1, -1] = 2
    mask = (np.arange(T)[None, None, :] < lengths[..., None]).astype(np.float32)
    obs = (rng.normal(size=(G, Eps, T, S)) * mask[..., None]).astype(np.float32)
    act = (rng.normal(size=(G, Eps, T, A)) * 0.5 * mask[..., None]).astype(np.float32)
    rew = (rng.normal(size=(G, Eps, T)) * reward_scale * mask).astype(np.float32)
    buf.group_observations = torch.from_numpy(obs)
    buf.group_actions = torch.from_numpy(act)
    buf.group_rewards = torch.from_numpy(rew)
    buf.group_masks = torch.from_numpy(mask)
    buf.group_lengths = torch.from_numpy(lengths.astype(np.float32))
    return buf


def hook_frame_locals(obj, attr, names, sink, once_per_call=False):
    """Wrap obj.attr so every call first copies `names` out of the caller's frame."""
    orig = getattr(obj, attr)

    def wrapped(*a, **k):
        fr = sys._getframe(1)
        rec = {}
        for n in names:
            if n in fr.f_locals:
                v = fr.f_locals[n]
                rec[n] = v.detach().clone().numpy() if isinstance(v, torch.Tensor) else v
        sink.append(rec)
        return orig(*a, **k)

    setattr(obj, attr, wrapped)


def gen_learner(rng):
    # --- returns / advantages at three discounts -------------------------------------------
    out = {}
    G, Eps, T, S, A = 3, 4, 24, 5, 1
    buf = ragged_buffer(rng, G, Eps, T, S, A, reward_scale=2.0)
    out.update(rew=buf.group_rewards.numpy(), mask=buf.group_masks.numpy())
    for gamma in (0.5, 0.99, 0.999):
        torch.manual_seed(0)
        pol = GaussianActorCritic_NeuralNetwork(S, A, (16, 16), cov=0.5)
        opt = torch.optim.Adam(pol.parameters(), lr=0.0)
        grpo = GRPO(epsilon=0.15, beta=0.5, gamma=gamma, policy=pol, optimizer=opt, updates_per_iter=1)
        sink = []
        hook_frame_locals(grpo.old_policy, "log_prob", ["i", "A_i", "rtgs", "group_rtgs"], sink)
        grpo.learn(buf)
        tag = f"g{gamma}"
        out[f"{tag}_rtg"] = sink[0]["group_rtgs"].reshape(G, Eps, T)
        for rec in sink:
            out[f"{tag}_grpo_adv_{rec['i']}"] = rec["A_i"]
        for mc in (True, False):
            ppo = PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=1, gamma=gamma,
                      lam=0.95, batch_size=None, monte_carlo=mc)
            sink = []
            hook_frame_locals(pol, "log_prob", ["advantages", "rtgs", "group_rtgs", "group_advantages", "group_values"], sink)
            ppo.learn(buf)
            del pol.log_prob                                   # drop the instance-level hook
            kind = "mc" if mc else "gae"
            out[f"{tag}_ppo_{kind}_adv"] = sink[0]["advantages"]
            out[f"{tag}_ppo_{kind}_ret"] = sink[0]["rtgs"]
            out[f"{tag}_ppo_{kind}_group_rtgs"] = sink[0]["group_rtgs"]
            out[f"{tag}_ppo_{kind}_group_adv"] = sink[0]["group_advantages"]
            out[f"{tag}_values"] = sink[0]["group_values"]
    np.savez_compressed(os.path.join(OUT, "rtg_adv.npz"), **out)

    # --- GRPO optimizer steps ----------------------------------------------------------------
    for n_upd in (1, 2):
        G, Eps, T, S, A = 3, 4, 16, 5, 1
        rs = np.random.default_rng(100 + n_upd)
        buf = ragged_buffer(rs, G, Eps, T, S, A)
        torch.manual_seed(21)
        pol = GaussianActor_NeuralNetwork(S, A, (32, 32), cov=0.5)
        init = sd_to_np(pol.state_dict())
        opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
        grpo = GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=pol, optimizer=opt, updates_per_iter=n_upd)
        if n_upd == 2:
            # make old_policy != policy so the ratio is not identically 1 on the first pass
            with torch.no_grad():
                for p in grpo.old_policy.parameters():
                    p.add_(0.02 * torch.randn_like(p))
        old_init = sd_to_np(grpo.old_policy.state_dict())
        sink = []
        hook_frame_locals(opt, "zero_grad", ["J"], sink)
        grpo.learn(buf)
        out = dict(obs=buf.group_observations.numpy(), act=buf.group_actions.numpy(), rew=buf.group_rewards.numpy(),
                   mask=buf.group_masks.numpy(), J=np.array([float(r["J"]) for r in sink]), lr=3e-4, epsilon=0.15,
                   gamma=0.5, cov=0.5, updates_per_iter=n_upd)
        for k, v in init.items():
            out[f"init.{k}"] = v
        for k, v in old_init.items():
            out[f"old_init.{k}"] = v
        for k, v in sd_to_np(pol.state_dict()).items():
            out[f"final.{k}"] = v
        for (k, _), p in zip(pol.actor.named_parameters(), pol.actor.parameters()):
            out[f"lastgrad.{k}"] = p.grad.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"grpo_step_u{n_upd}.npz"), **out)
        print("GRPO J", out["J"])

    # --- PPO optimizer steps -----------------------------------------------------------------
    for n_upd in (1, 2):
        G, Eps, T, S, A = 3, 4, 16, 10, 2
        rs = np.random.default_rng(200 + n_upd)
        buf = ragged_buffer(rs, G, Eps, T, S, A)
        torch.manual_seed(22)
        pol = GaussianActorCritic_NeuralNetwork(S, A, (32, 32), cov=0.5)
        init = sd_to_np(pol.state_dict())
        opt = torch.optim.Adam(pol.parameters(), lr=2e-4)
        ppo = PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=n_upd, c1=0.5,
                  kl_coeff=0.5, gamma=0.99, lam=0.95, entropy=0.01, batch_size=None)
        sink = []
        hook_frame_locals(opt, "zero_grad", ["total_loss", "actor_loss", "critic_loss", "kl_div", "entropy_bonus"], sink)
        ppo.learn(buf)
        out = dict(obs=buf.group_observations.numpy(), act=buf.group_actions.numpy(), rew=buf.group_rewards.numpy(),
                   mask=buf.group_masks.numpy(), lr=2e-4, epsilon=0.2, gamma=0.99, cov=0.5, c1=0.5, kl_coeff=0.5,
                   entropy_coeff=0.01, updates_per_iter=n_upd)
        for name in ("total_loss", "actor_loss", "critic_loss", "kl_div", "entropy_bonus"):
            out[name] = np.array([float(r[name]) for r in sink])
        for k, v in init.items():
            out[f"init.{k}"] = v
        for k, v in sd_to_np(pol.state_dict()).items():
            out[f"final.{k}"] = v
        for net in ("actor", "critic"):
            for k, p in getattr(pol, net).named_parameters():
                out[f"lastgrad.{net}.{k}"] = p.grad.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"ppo_step_u{n_upd}.npz"), **out)
        print("PPO total", out["total_loss"])


def gen_ppo_gae_step():
    G, Eps, T, S, A = 3, 4, 16, 10, 2
    rs = np.random.default_rng(300)
    buf = ragged_buffer(rs, G, Eps, T, S, A)
    torch.manual_seed(23)
    pol = GaussianActorCritic_NeuralNetwork(S, A, (32, 32), cov=[0.5, 0.2])
    init = sd_to_np(pol.state_dict())
    opt = torch.optim.Adam(pol.parameters(), lr=2e-4)
    ppo = PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, c1=0.5, kl_coeff=0.5, gamma=0.97,
              lam=0.9, entropy=0.01, batch_size=None, monte_carlo=False)
    sink = []
    hook_frame_locals(opt, "zero_grad", ["total_loss", "actor_loss", "critic_loss", "kl_div"], sink)
    ppo.learn(buf)
    out = dict(obs=buf.group_observations.numpy(), act=buf.group_actions.numpy(), rew=buf.group_rewards.numpy(),
               mask=buf.group_masks.numpy(), lr=2e-4, epsilon=0.2, gamma=0.97, lam=0.9, cov=np.array([0.5, 0.2]), c1=0.5,
               kl_coeff=0.5, entropy_coeff=0.01)
    for name in ("total_loss", "actor_loss", "critic_loss", "kl_div"):
        out[name] = np.array([float(r[name]) for r in sink])
    for k, v in init.items():
        out[f"init.{k}"] = v
    for k, v in sd_to_np(pol.state_dict()).items():
        out[f"final.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "ppo_gae_step_u2.npz"), **out)
    print("PPO GAE total", out["total_loss"])


# ---------------------------------------------------------------------------
# 6. Pendulum: single steps incl. the balance-time termination, and in-process rollouts
# ---------------------------------------------------------------------------
def gen_pendulum():
    from environments.pendulum_env import Pendulum  # noqa: E402  (reference)
    rs = np.random.default_rng(424242)
    variants = {
        "default": dict(),
        "custom": dict(mass=2.0, length=0.7, gravity=3.0, timestep=0.02),
    }
    for tag, kw in variants.items():
        MAXS = 120
        env = Pendulum(max_steps=MAXS, **kw)
        dt = env.timestep
        n = 192
        ang = rs.uniform(-np.pi, np.pi, size=n)
        ang[:64] = np.pi + rs.uniform(-0.2, 0.2, size=64)          # around the balance threshold cos <= -0.99 (0.1415 rad)
        st = np.stack([np.sin(ang), np.cos(ang), rs.normal(size=n) * 3], axis=1)
        st[:64, 2] = rs.normal(size=64) * 0.3
        st[64:72, 2] = rs.choice([-14.0, 12.5, 10.0, -10.0], size=8)                # thetadot clamp
        st[72:80, :2] *= rs.uniform(0.5, 1.5, size=(8, 1))                        # un-normalised (sin, cos)
        act = (rs.normal(size=(n, 1)) * 0.9).astype(np.float32)
        act[80:88] = np.where(rs.random((8, 1)) < 0.5, -3.0, 2.5)                  # |a| > 1 clipping
        steps = rs.integers(0, MAXS - 1, size=n)
        # `_time > max_time` with float-accumulated time: first true after k_tr steps (k_tr is MAXS or MAXS + 1)
        t_acc, k_tr = 0, 0
        while not t_acc > env.max_time:
            t_acc += env.timestep
            k_tr += 1
        steps[88:96] = k_tr - 1 + rs.choice([-1, 0, 0, 1], size=8)              # the step lands on k_tr - 1, k_tr, k_tr + 1
        # time_balanced as the reference accumulates it: k consecutive balanced steps, k around the 5 s limit
        ks = rs.integers(0, 20, size=n)
        k_lim = int(round(5.0 / dt))
        ks[:64] = rs.choice([0, 1, k_lim - 2, k_lim - 1, k_lim, k_lim + 1], size=64)
        tb = np.zeros(n)
        for i in range(n):
            t = 0
            for _ in range(int(ks[i])):
                t = t + dt
            tb[i] = t
        nxt, rew = np.zeros((n, 3)), np.zeros(n)
        trunc, term = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
        tb_after, info_tb = np.zeros(n), np.zeros(n)
        for i in range(n):
            env.reset()
            env.state_dict["pendulum"] = st[i].copy()
            env._steps = int(steps[i])
            env._time = 0
            for _ in range(int(steps[i])):
                env._time += env.timestep
            env._time_balanced = tb[i] if ks[i] > 0 else 0
            o, r, tr, te, info = env.step(act[i])                    # NOTE the order: truncated before terminated
            nxt[i], rew[i], trunc[i], term[i] = np.asarray(o, dtype=np.float64), float(np.asarray(r).reshape(-1)[0]), bool(tr), bool(te)
            tb_after[i], info_tb[i] = env._time_balanced, info["time_balanced"]
        print("   sums", int(term.sum()), int(trunc.sum()), int((~term).sum()))
        assert term.sum() > 4 and trunc.sum() > 3 and (~term).sum() > 100
        np.savez_compressed(os.path.join(OUT, f"env_step_pendulum_{tag}.npz"), state=st, action=act, steps=steps,
                            time_balanced=tb, max_steps=MAXS, next_state=nxt, reward=rew, truncated=trunc, terminated=term,
                            time_balanced_after=tb_after, info_time_balanced=info_tb,
                            **{f"param_{k}": v for k, v in kw.items()})
        print("Pendulum", tag, "single-step: terminated", int(term.sum()), "truncated", int(trunc.sum()))

    # rollouts through the reference RolloutManager: 'fall' = default physics (episodes run to the horizon),
    # 'hold' = no gravity and a near-silent policy (the pendulum stays within the band: terminated at step 101)
    out = {}
    for tag, env_kw, cov, T in (("fall", dict(), 0.5, 64), ("hold", dict(gravity=0.0), 1e-4, 140)):
        np.random.seed(21)
        torch.manual_seed(21)
        policy = GaussianActor_NeuralNetwork(3, 1, (32, 32), cov=cov)
        if tag == "hold":
            with torch.no_grad():
                for prm in policy.actor.network[-1].parameters():
                    prm.mul_(1e-3)
        mgr = RolloutManager(env_fn=lambda: Pendulum(max_steps=T, **env_kw), policy=policy, restart=False, num_workers=2,
                             num_episodes_per_worker=3, use_multiprocessing=False)
        obs, act, rew, ln, mask = mgr.rollout()
        out.update({f"{tag}_obs": obs.numpy(), f"{tag}_act": act.numpy(), f"{tag}_rew": rew.numpy(), f"{tag}_len": ln.numpy(),
                    f"{tag}_mask": mask.numpy(), f"{tag}_cov": np.diag(policy.cov.numpy()), f"{tag}_max_steps": T,
                    f"{tag}_gravity": env_kw.get("gravity", 9.80665)})
        for k, v in sd_to_np(policy.state_dict()).items():
            out[f"{tag}_policy.{k}"] = v
        print("Pendulum rollout", tag, "lengths", ln.numpy().astype(int).tolist())
    np.savez_compressed(os.path.join(OUT, "rollout_pendulum.npz"), **out)



# ---------------------------------------------------------------------------
# 5. learn() at the shapes the chain kernels run (256 x 5, 128 x 4), minibatch PPO, trajectory export
# ---------------------------------------------------------------------------
def _sample(t, cap=2048):
    """Every stride-th element of a tensor (at most `cap` values): big nets are pinned by samples + whole-tensor sums,
    not by megabytes of weights.  The test rebuilds the initial weights from `seed` (the reference's constructors
    draw them from torch's CPU generator) and checks them against the same samples."""
    flat = t.detach().reshape(-1)
    stride = max(1, -(-flat.numel() // cap))
    return flat[::stride].numpy().copy(), stride


def _pin(out, prefix, named):
    for k, t in named:
        smp, stride = _sample(t)
        out[f"{prefix}.{k}"] = smp
        out[f"{prefix}_stride.{k}"] = stride
        out[f"{prefix}_sum.{k}"] = float(t.detach().double().sum())
        out[f"{prefix}_l2.{k}"] = float(t.detach().double().norm())


def gen_chain_shape_steps():
    cfgs = [("ppo", "h256", 20, 4, (256,) * 5, 0.3, 0.999, 3e-4, 400), ("grpo", "h256", 20, 4, (256,) * 5, 0.3, 0.5, 3e-4, 401),
            ("ppo", "h128", 5, 1, (128,) * 4, 0.5, 0.99, 2e-4, 402), ("grpo", "h128", 5, 1, (128,) * 4, 0.5, 0.5, 3e-4, 403)]
    for kind, tag, S, A, hidden, cov, gamma, lr, seed in cfgs:
        G, Eps, T = 8, 16, 64                                   # ~4,200 valid rows
        buf = ragged_buffer(np.random.default_rng(seed), G, Eps, T, S, A)
        torch.manual_seed(seed)
        if kind == "ppo":
            pol = GaussianActorCritic_NeuralNetwork(S, A, hidden, cov=cov)
            nets = ("actor", "critic")
        else:
            pol = GaussianActor_NeuralNetwork(S, A, hidden, cov=cov)
            nets = ("actor",)
        named = lambda: [(f"{n}.{k}", p) for n in nets for k, p in getattr(pol, n).named_parameters()]
        out = dict(obs=buf.group_observations.numpy(), act=buf.group_actions.numpy(), rew=buf.group_rewards.numpy(),
                   mask=buf.group_masks.numpy(), seed=seed, lr=lr, gamma=gamma, cov=cov, updates_per_iter=2,
                   hidden=np.array(hidden), n_valid=int(buf.group_masks.sum()))
        _pin(out, "init", named())
        init = {k: p.detach().clone() for k, p in named()}
        opt = torch.optim.Adam(pol.parameters(), lr=lr)
        # the gradients of the FIRST update (taken on the initial weights: they isolate the arithmetic of one
        # forward / backward pass from the path the training takes afterwards) are still in .grad when the second
        # update calls zero_grad()
        first = {}
        plain_zero = opt.zero_grad

        def zero_grad_keeping_first(*a, **k):
            if not first and all(p.grad is not None for _, p in named()):
                first.update({k_: p.grad.detach().clone() for k_, p in named()})
            return plain_zero(*a, **k)

        opt.zero_grad = zero_grad_keeping_first
        sink = []
        if kind == "ppo":
            algo = PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, c1=0.5, kl_coeff=0.5,
                       gamma=gamma, lam=0.95, entropy=0.01, batch_size=None)
            hook_frame_locals(opt, "zero_grad", ["total_loss", "actor_loss", "critic_loss", "kl_div"], sink)
        else:
            algo = GRPO(epsilon=0.15, beta=0.5, gamma=gamma, policy=pol, optimizer=opt, updates_per_iter=2)
            with torch.no_grad():                               # old_policy != policy: the ratio is not identically 1
                g = torch.Generator().manual_seed(seed)
                for p_ in algo.old_policy.parameters():
                    p_.add_(0.01 * torch.randn(p_.shape, generator=g))
            out["old_policy_perturbation"] = 0.01
            hook_frame_locals(opt, "zero_grad", ["J"], sink)
        algo.learn(buf)
        for name in sink[0]:
            out[name] = np.array([float(r[name]) for r in sink])
        _pin(out, "firstgrad", list(first.items()))
        _pin(out, "final", named())
        _pin(out, "delta", [(k, p.detach() - init[k]) for k, p in named()])
        _pin(out, "lastgrad", [(k, p.grad) for k, p in named()])
        np.savez_compressed(os.path.join(OUT, f"{kind}_step_{tag}.npz"), **out)
        print(kind, tag, "rows", out["n_valid"], {n: out[n] for n in sink[0]})


def gen_ppo_minibatch():
    """PPO with batch_size = 64 (algorithms/ppo.py:147-157): the permutations torch.randperm returned are recorded as
    data (indices into the reference's valid rows, (g, e, t) order)."""
    G, Eps, T, S, A = 3, 4, 16, 10, 2
    buf = ragged_buffer(np.random.default_rng(500), G, Eps, T, S, A)
    torch.manual_seed(24)
    pol = GaussianActorCritic_NeuralNetwork(S, A, (32, 32), cov=0.5)
    init = sd_to_np(pol.state_dict())
    opt = torch.optim.Adam(pol.parameters(), lr=2e-4)
    ppo = PPO(epsilon=0.2, policy=pol, optimizer=opt, ref_model=None, updates_per_iter=2, c1=0.5, kl_coeff=0.5, gamma=0.99,
              lam=0.95, entropy=0.01, batch_size=64)
    perms, sink = [], []
    orig = torch.randperm

    def recording_randperm(n, *a, **k):
        p_ = orig(n, *a, **k)
        perms.append(p_.numpy().copy())
        return p_

    torch.randperm = recording_randperm
    try:
        hook_frame_locals(opt, "zero_grad", ["total_loss", "actor_loss", "critic_loss", "kl_div"], sink)
        ppo.learn(buf)
    finally:
        torch.randperm = orig
    out = dict(obs=buf.group_observations.numpy(), act=buf.group_actions.numpy(), rew=buf.group_rewards.numpy(),
               mask=buf.group_masks.numpy(), lr=2e-4, epsilon=0.2, gamma=0.99, cov=0.5, c1=0.5, kl_coeff=0.5, entropy_coeff=0.01,
               batch_size=64, updates_per_iter=2, permutations=np.stack(perms))
    for name in ("total_loss", "actor_loss", "critic_loss", "kl_div"):
        out[name] = np.array([float(r[name]) for r in sink])
    for k, v in init.items():
        out[f"init.{k}"] = v
    for k, v in sd_to_np(pol.state_dict()).items():
        out[f"final.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "ppo_minibatch.npz"), **out)
    print("PPO minibatch: steps", len(sink), "perms", len(perms), "total", out["total_loss"])


def gen_trajectory_csv():
    """Rollout_Buffer.save_trajectory (buffers/rollout_buffer.py:72-102) on the golden CartPole rollout: the CSV the
    reference writes is the fixture (data: episode ids, observations, actions)."""
    import tempfile
    from buffers import Rollout_Buffer
    g = np.load(os.path.join(OUT, "rollout_cartpole.npz"))
    buf = Rollout_Buffer.__new__(Rollout_Buffer)
    buf.avg_reward = []
    buf.store(torch.from_numpy(g["reset_obs"]), torch.from_numpy(g["reset_act"]), torch.from_numpy(g["reset_rew"]),
              torch.from_numpy(g["reset_len"]), torch.from_numpy(g["reset_mask"]))
    with tempfile.TemporaryDirectory() as d:
        buf.save_trajectory(d)
        text = open(os.path.join(d, "trajectory.csv")).read()
    with open(os.path.join(OUT, "trajectory_cartpole_reset.csv"), "w") as f:
        f.write(text)
    print("trajectory.csv:", text.count("\n"), "lines")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20250613)
    gen_env_steps(rng)
    gen_rollouts()
    gen_policy()
    gen_learner(rng)
    gen_ppo_gae_step()
    gen_pendulum()
    gen_chain_shape_steps()
    gen_ppo_minibatch()
    gen_trajectory_csv()
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
