"""CPU oracle: torch-CPU fp32 restatement of the reference's policy, rollout loop
and GRPO/PPO learner arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations are `path:line`
in the reference checkout.  Third-party arithmetic the reference calls
(`torch.nn.Linear`, `torch.distributions.MultivariateNormal`, `torch.optim.Adam`,
`torch.std/mean/clamp/min/exp`) is restated through closed forms / the same
torch calls (torch 2.10.0 in this image; the reference pins no versions).
"""
from __future__ import annotations

import copy
import math

import numpy as np
import torch

from . import envs as E


# ---------------------------------------------------------------------------
# policy  (policies/actor_critic.py, models/neural_network.py)
# ---------------------------------------------------------------------------
class MLP(torch.nn.Module):
    """Sequential(Linear, act, ..., Linear).  models/neural_network.py:48-66.
    Parameter names match the reference (`network.{0,2,4,...}.{weight,bias}`)."""

    def __init__(self, input_dim, output_dim, hidden_dims, activation="ReLU"):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        layers = []
        for i in range(len(hidden_dims)):
            layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
            layers.append(getattr(torch.nn, activation)())
        layers.append(torch.nn.Linear(dims[-1], output_dim))
        self.network = torch.nn.Sequential(*layers)

    def forward(self, x):
        return self.network(x)


def gaussian_log_prob(mean, action, var):
    """log N(action; mean, diag(var)).  Closed form of
    MultivariateNormal(mean, cov).log_prob (actor_critic.py:131-136,159-160)
    for the reference's fixed diagonal covariance (:100-103)."""
    var = torch.as_tensor(var, dtype=mean.dtype)
    k = mean.shape[-1]
    quad = (((action - mean) ** 2) / var).sum(-1)
    return -0.5 * quad - 0.5 * k * math.log(2 * math.pi) - 0.5 * torch.log(var).sum()


def gaussian_entropy(var, k):
    var = torch.as_tensor(var, dtype=torch.float32)
    return 0.5 * k * (1.0 + math.log(2 * math.pi)) + 0.5 * torch.log(var).sum()


class OraclePolicy:
    """Gaussian actor(-critic) with fixed diagonal covariance.
    actor_critic.py:73-215 (actor only), :220-378 (actor-critic)."""

    def __init__(self, input_dim, output_dim, hidden_dims, activation="ReLU", cov=0.1, critic=False,
                 per_step_distribution=False):
        """per_step_distribution=True: __call__ builds a torch MultivariateNormal on every call, samples from it and evaluates
        its log-probability, under autograd -- what the reference's forward does on every env step (actor_critic.py:131-136,
        :279-284; the worker loop has no no_grad, rollout_worker.py:55).  Same draws and values as the closed form (SURVEY
        App. C); it is there so that the timed CPU baseline pays what the reference pays (bench.py cpu_baseline)."""
        self.input_dim, self.output_dim = input_dim, output_dim
        self.per_step_distribution = per_step_distribution
        self.var = torch.tensor(cov if isinstance(cov, list) else [cov] * output_dim, dtype=torch.float32)
        self.actor = MLP(input_dim, output_dim, hidden_dims, activation)
        self.critic = MLP(input_dim, 1, hidden_dims, activation) if critic else None

    # forward: sample, actor_critic.py:107-138 / :255-289
    def __call__(self, state):
        if isinstance(state, np.ndarray):
            state = torch.from_numpy(state).float()
        mean = self.actor(state)
        if self.per_step_distribution:
            dist = torch.distributions.MultivariateNormal(mean, torch.diag(self.var))
            action = dist.sample()
            value = self.critic(state) if self.critic is not None else None
            return action.detach().numpy(), dist.log_prob(action), value
        # MultivariateNormal.sample() == mean + sqrt(var) * randn (bit-exact)
        with torch.no_grad():
            action = mean + torch.sqrt(self.var) * torch.randn(mean.shape)
        logp = gaussian_log_prob(mean, action, self.var)
        value = self.critic(state) if self.critic is not None else None
        return action.detach().numpy(), logp, value

    def log_prob(self, obs, act):
        if isinstance(obs, np.ndarray):
            obs = torch.from_numpy(obs).float()
        if isinstance(act, np.ndarray):
            act = torch.from_numpy(act).float()
        mean = self.actor(obs)
        ent = gaussian_entropy(self.var, self.output_dim).expand(mean.shape[:-1])
        return gaussian_log_prob(mean, act, self.var), ent

    def value(self, obs):
        return self.critic(obs).squeeze()                     # actor_critic.py:323

    def parameters(self):
        ps = list(self.actor.parameters())
        if self.critic is not None:
            ps += list(self.critic.parameters())
        return ps

    def state_dict(self):
        if self.critic is None:
            return self.actor.state_dict()
        return {"actor": self.actor.state_dict(), "critic": self.critic.state_dict()}

    def load_state_dict(self, sd):
        if self.critic is None:
            self.actor.load_state_dict(sd)
        else:
            self.actor.load_state_dict(sd["actor"])
            self.critic.load_state_dict(sd["critic"])


# ---------------------------------------------------------------------------
# scalar env with the reference's reset/restart/step surface (for the port worker)
# ---------------------------------------------------------------------------
class OracleEnv:
    """One environment instance stepping through oracle.envs with N=1."""

    def __init__(self, env_name, max_steps=500, rng=None, **params):
        self.env_name = env_name
        self.spec = E.ENV_SPECS[env_name]
        self.max_steps = max_steps
        self.params = params                                  # constructor arguments of the reference env (masses, timestep, ...)
        self.obs_dim, self.act_dim = self.spec["obs_dim"], self.spec["act_dim"]
        self.rng = rng if rng is not None else np.random.default_rng()
        self.state = None
        self._initial = None
        self._steps = 0
        self._tb = 0.0

    def set_state(self, state):
        self.state = np.asarray(state, dtype=np.float64).reshape(1, -1).copy()
        self._initial = self.state.copy()
        self._steps, self._tb = 0, 0.0
        return self.state[0]

    def reset(self):
        return self.set_state(E.sample_initial_states(self.env_name, 1, self.rng)), {}

    def restart(self):
        self.state = self._initial.copy()
        self._steps, self._tb = 0, 0.0
        return self.state[0], {}

    def step(self, action):
        nxt, rew, trunc, steps, tb = self.spec["step"](
            self.state, np.asarray(action, dtype=np.float32).reshape(1, -1),
            np.array([self._steps]), np.array([self._tb]), max_steps=self.max_steps, **self.params)
        self.state, self._steps, self._tb = nxt, int(steps[0]), float(tb[0])
        limit = self.spec.get("balance_terminates")           # only Pendulum ever terminates (pendulum_env.py:151)
        return nxt[0], float(rew[0]), bool(limit is not None and self._tb > limit), bool(trunc[0]), {}


def run_episodes(env, policy, num_episodes, restart=False, initial_states=None,
                 forced_actions=None):
    """rollout/rollout_worker.py:19-84, restated.

    `initial_states` (E,S) / `forced_actions` (E,T,A) replace the RNG draws for
    teacher-forced parity runs (the reference's NumPy/Torch RNG streams are not
    reproducible on the GPU).  Returns float32 tensors shaped as the reference:
    obs (E,T,S), act (E,T,A), rew (E,T), len (E,) int32, mask (E,T)."""
    T, S, A = env.max_steps, env.obs_dim, env.act_dim
    obs_b = np.zeros((num_episodes, T, S))
    act_b = np.zeros((num_episodes, T, A))
    rew_b = np.zeros((num_episodes, T))
    len_b = np.zeros(num_episodes, dtype=int)
    mask_b = np.zeros((num_episodes, T))
    if initial_states is None:
        observation, _ = env.reset()                          # :31
    for ep in range(num_episodes):
        if initial_states is not None:
            # restart mode: every episode starts from the worker's first reset state
            observation = env.set_state(initial_states[0 if restart else ep])
        done, t = False, 0
        while not done and t < T:                             # :51
            obs_b[ep, t] = observation                        # :53 obs before action
            if forced_actions is not None:
                action = forced_actions[ep, t]
            else:
                action, _, _ = policy(observation)            # :55
            observation, reward, term, trunc, _ = env.step(action)  # :56
            act_b[ep, t] = action
            rew_b[ep, t] = reward
            done = term or trunc
            t += 1
        len_b[ep] = t
        mask_b[ep, :t] = 1
        if initial_states is None:
            if restart:
                observation, _ = env.restart()                # :70-71
            else:
                observation, _ = env.reset()                  # :72-73 (also after the last episode)
    f = lambda a: torch.from_numpy(a).float()
    return f(obs_b), f(act_b), f(rew_b), torch.from_numpy(len_b).int(), f(mask_b)


def rollout(env_fn, policy, num_workers, num_episodes, restart=False,
            initial_states=None, forced_actions=None):
    """In-process branch of RolloutManager.rollout (rollout/rollout_manager.py:85-91,113-125).
    Returns (G,E,T,S), (G,E,T,A), (G,E,T), (G,E) float32, (G,E,T)."""
    outs = []
    for g in range(num_workers):
        env = env_fn()
        outs.append(run_episodes(
            env, policy, num_episodes, restart,
            None if initial_states is None else initial_states[g],
            None if forced_actions is None else forced_actions[g]))
    obs, act, rew, ln, mask = (torch.stack([o[i] for o in outs]) for i in range(5))
    return obs, act, rew, ln.float(), mask                    # lengths float32, :89


# ---------------------------------------------------------------------------
# returns and advantages
# ---------------------------------------------------------------------------
def rtg_scan(rewards, masks, gamma):
    """Reward-to-go reverse scan.  algorithms/grpo.py:66-74 == algorithms/ppo.py:100-111.
    R[T-1] = r[T-1] m[T-1];  R[t] = r[t] m[t] + gamma R[t+1] m[t+1]."""
    rewards = torch.as_tensor(rewards, dtype=torch.float32)
    m = torch.as_tensor(masks, dtype=torch.float32)
    T = rewards.shape[-1]
    out = torch.zeros_like(rewards)
    for i in reversed(range(T)):
        if i < T - 1:
            out[..., i] = rewards[..., i] * m[..., i] + gamma * out[..., i + 1] * m[..., i + 1]
        else:
            out[..., i] = rewards[..., i] * m[..., i]
    return out


def grpo_group_advantages(rtgs, masks):
    """Per group g over all valid steps of its E episodes:
    A = (R - mean R) / std(R + 1e-8), unbiased std.  algorithms/grpo.py:110-115.
    Returns a list of 1-D tensors (ragged) in (e, t) row-major order."""
    G = rtgs.shape[0]
    r = rtgs.reshape(G, -1)
    m = torch.as_tensor(masks).reshape(G, -1).bool()
    out = []
    for g in range(G):
        v = r[g][m[g]]
        out.append((v - torch.mean(v)) / torch.std(v + 1e-8))
    return out


def gae_scan(rewards, values, masks, gamma, lam):
    """GAE branch, algorithms/ppo.py:112-124 (delta uses the unmasked r[t])."""
    m = torch.as_tensor(masks, dtype=torch.float32)
    T = rewards.shape[-1]
    adv = torch.zeros_like(rewards)
    for i in reversed(range(T)):
        if i < T - 1:
            nv = values[..., i + 1] * m[..., i + 1]
            delta = rewards[..., i] + gamma * nv - values[..., i]
            adv[..., i] = delta + gamma * lam * adv[..., i + 1] * m[..., i + 1]
        else:
            adv[..., i] = rewards[..., i] - values[..., i]
    return adv, values + adv


def ppo_advantages(rewards, masks, values, gamma, lam=0.95, monte_carlo=True):
    """algorithms/ppo.py:93-139: returns (advantages, returns) over the valid
    batch, each normalised by its own mean / (unbiased std + 1e-8)."""
    if monte_carlo:
        rtg = rtg_scan(rewards, masks, gamma)
        adv = rtg - values
    else:
        adv, rtg = gae_scan(rewards, values, masks, gamma, lam)
    valid = torch.as_tensor(masks).reshape(-1).bool()
    adv = adv.reshape(-1).detach()[valid]
    rtg = rtg.reshape(-1).detach()[valid]
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    rtg = (rtg - rtg.mean()) / (rtg.std() + 1e-8)
    return adv, rtg


# ---------------------------------------------------------------------------
# GRPO / PPO learn
# ---------------------------------------------------------------------------
def grpo_objective(policy, old_policy, obs, act, rtgs, masks, epsilon):
    """J as accumulated at algorithms/grpo.py:106-140 (ref_model is None)."""
    G = obs.shape[0]
    o = obs.reshape(G, -1, obs.shape[-1])
    a = act.reshape(G, -1, act.shape[-1])
    r = rtgs.reshape(G, -1)
    m = masks.reshape(G, -1).bool()
    J = 0
    for g in range(G):
        og, ag, rg = o[g][m[g]], a[g][m[g]], r[g][m[g]]
        A = (rg - torch.mean(rg)) / torch.std(rg + 1e-8)
        with torch.no_grad():
            old_lp, _ = old_policy.log_prob(og, ag)
        lp, _ = policy.log_prob(og, ag)
        ratio = torch.exp(lp - old_lp)
        J = J + torch.min(ratio * A, torch.clamp(ratio, 1 - epsilon, 1 + epsilon) * A).sum()
    return J / G


def grpo_learn(policy, old_policy, optimizer, obs, act, rew, masks, *, epsilon, gamma,
               updates_per_iter):
    """algorithms/grpo.py:50-148.  Gradient DESCENT on J (no sign flip), as written.
    Returns the list of J values (one per update)."""
    rtgs = rtg_scan(rew, masks, gamma)
    Js = []
    for _ in range(updates_per_iter):
        J = grpo_objective(policy, old_policy, obs, act, rtgs, masks.float(), epsilon)
        optimizer.zero_grad()
        J.backward()
        optimizer.step()
        Js.append(float(J.detach()))
    old_policy.load_state_dict(copy.deepcopy(policy.state_dict()))
    return Js


def ppo_loss(policy, obs, act, adv, ret, old_lp, *, epsilon, c1, kl_coeff, entropy_coeff):
    """algorithms/ppo.py:159-179 on one (mini)batch."""
    lp, ent = policy.log_prob(obs, act)
    ratio = torch.exp(lp - old_lp)
    actor = -torch.min(ratio * adv, torch.clamp(ratio, 1 - epsilon, 1 + epsilon) * adv).mean()
    critic = torch.nn.functional.mse_loss(policy.value(obs), ret)
    ent_bonus = entropy_coeff * ent.mean()
    kl = (torch.exp(old_lp) * (old_lp - lp)).mean()
    total = actor + c1 * critic - ent_bonus + kl_coeff * kl
    return total, dict(actor=float(actor.detach()), critic=float(critic.detach()),
                       kl=float(kl.detach()), entropy=float(ent.mean()))


def ppo_learn(policy, optimizer, obs, act, rew, masks, *, epsilon, gamma, lam=0.95, c1=0.5,
              kl_coeff=0.5, entropy_coeff=0.01, updates_per_iter=1, monte_carlo=True, batch_size=None,
              permutations=None):
    """algorithms/ppo.py:64-186.  batch_size=None: one full-batch step per update (every shipped factory);
    otherwise the valid rows are permuted once per update and walked in steps of batch_size (:147-157).
    `permutations` (one index vector per update) replaces torch.randperm (fixtures carry the reference's draws)."""
    S, A = obs.shape[-1], act.shape[-1]
    with torch.no_grad():
        values = policy.value(obs.reshape(-1, S)).reshape(rew.shape)
    adv, ret = ppo_advantages(rew, masks, values, gamma, lam, monte_carlo)
    valid = masks.reshape(-1).bool()
    o = obs.reshape(-1, S)[valid]
    a = act.reshape(-1, A)[valid]
    with torch.no_grad():
        old_lp, _ = policy.log_prob(o, a)                     # ppo.py:142-143 (current policy)
    logs = []
    n = o.shape[0]
    for u in range(updates_per_iter):
        if batch_size is None:
            batches = [slice(None)]
        else:
            perm = torch.randperm(n) if permutations is None else torch.as_tensor(permutations[u], dtype=torch.long)
            batches = [perm[lo:lo + batch_size] for lo in range(0, n, batch_size)]
        for b in batches:
            total, parts = ppo_loss(policy, o[b], a[b], adv[b], ret[b], old_lp[b], epsilon=epsilon, c1=c1,
                                    kl_coeff=kl_coeff, entropy_coeff=entropy_coeff)
            optimizer.zero_grad()
            total.backward()
            optimizer.step()
            parts["total"] = float(total.detach())
            logs.append(parts)
    return logs
