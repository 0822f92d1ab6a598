"""Environments with the reference's `Env` surface, stepped by HIP kernels.

Mirrors environments/env.py:10-68, environments/cartpole_env.py:6-182 and
environments/quadrotor_env.py:353-713, :867-1223 of the reference: same constructor
arguments, attributes (`max_steps`, `observation_space`, `action_space`, `env_name`,
`timestep`, `_is_3d`, `state_dict`, `_initial_state`) and `reset/restart/step` return
shapes.  A single instance is a drop-in for the reference's scalar env (every call is one
kernel launch on one env: correct, not fast); the throughput path is
`rollout.DeviceRollout`, which steps `num_envs x horizon` from the same parameters.
`render` is out of scope (matplotlib) and raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as N


class Box:
    """The slice of gymnasium.spaces.Box the reference touches (env.py, rollout_worker.py:34-35)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else tuple(np.shape(low))
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box(shape={self.shape}, dtype={self.dtype})"


class Env:
    """Base class (environments/env.py:10-68): reset / restart / step / render."""

    ENV_ID = None
    _state_split = None    # ((key, size), ...) of state_dict, as the reference stores it

    def __init__(self, env_name: str, device=None, dtype=torch.float64):
        self.env_name = env_name
        self._device = torch.device(device) if device is not None else torch.device("cuda", 0)
        self._dtype = dtype
        self._state = None           # device [S][1]
        self._initial = None
        self._steps_t = None
        self._tb_t = None
        self._seed = int(np.random.randint(0, 2 ** 31 - 1))
        self._reset_count = 0
        self._time_balanced = 0
        self._steps = 0
        self._time = 0
        self._initial_state = None

    # -- parameters -------------------------------------------------------
    def native_params(self) -> N.EnvParams:
        """tg_env_params built from the instance attributes (users may edit them, like the reference's)."""
        p = N.default_params(self.ENV_ID, self.max_steps)
        p.timestep = float(self.timestep)
        self._fill_params(p)
        N.check(N.load().tg_env_finalize_params(C.byref(p)), "tg_env_finalize_params")
        return p

    def _fill_params(self, p):
        raise NotImplementedError

    @property
    def obs_dim(self):
        return self.observation_space.shape[0]

    @property
    def act_dim(self):
        return self.action_space.shape[0]

    # -- scalar drop-in API ----------------------------------------------
    def _alloc(self):
        if self._state is None:
            S = self.obs_dim
            self._state = torch.zeros(S, 1, dtype=self._dtype, device=self._device)
            self._steps_t = torch.zeros(1, dtype=torch.int32, device=self._device)
            self._tb_t = torch.zeros(1, dtype=self._dtype, device=self._device)
            self._rew_t = torch.zeros(1, dtype=self._dtype, device=self._device)
            self._trunc_t = torch.zeros(1, dtype=torch.uint8, device=self._device)
            self._act_t = torch.zeros(self.act_dim, 1, dtype=torch.float32, device=self._device)

    def _obs_np(self):
        return self._state[:, 0].double().cpu().numpy()

    def _sync_state_dict(self):
        o = self._obs_np()
        off = 0
        for key, size in self._state_split:
            self.state_dict[key] = o[off:off + size].copy()
            off += size

    def _get_info(self):
        return {"time_balanced": self._time_balanced}

    def _get_obs(self):
        """The observation = the full state, as the reference's `_get_obs` (cartpole_env.py:130-131, quadrotor_env.py:600-604)."""
        return np.hstack([np.asarray(self.state_dict[k]).ravel() for k, _ in self._state_split])[:self.obs_dim]

    # -- the reference's per-step helpers as callable members (environments/cartpole_env.py:48-100, quadrotor_env.py:409-417,
    #    :578-585, :928, :1024-1042; tests/test_cartpole.py:36-40 calls `env._dynamics`) ---------------------------------------
    _WRAP = None           # (offset attribute or None, scale attribute or constant): u = offset + scale * clip(a, -1, 1)

    def _wrap_scalars(self):
        off, scale = self._WRAP
        return (getattr(self, off) if isinstance(off, str) else off), (getattr(self, scale) if isinstance(scale, str) else scale)

    def _wrap_action(self, action):
        """The control the dynamics receive: float32 in, float32 out under NumPy's weak-scalar promotion, like the reference's line."""
        off, scale = self._wrap_scalars()
        u = scale * np.clip(action, -1, 1)
        return u if off is None else off + u

    def _unwrap_control(self, control) -> np.ndarray:
        """A float32 action `a` with `_wrap_action(a) == control` bit for bit when `control` is a value the wrap can produce (every
        control step() ever passes on), else the nearest one (<= 1 float32 ulp off, and clipped to the wrap's range)."""
        off, scale = self._wrap_scalars()
        u = np.asarray(np.atleast_1d(control), dtype=np.float32).reshape(self.act_dim)
        t = u if off is None else u - np.float32(off)
        a = np.clip((t.astype(np.float64) / float(np.float32(scale))).astype(np.float32), -1, 1).astype(np.float32)
        for k in range(a.size):                                  # a few float32 neighbours: the two roundings of the wrap
            best, cand = None, a[k]
            lo = hi = cand
            cands = [cand]
            for _ in range(4):
                lo, hi = np.nextafter(lo, np.float32(-2)), np.nextafter(hi, np.float32(2))
                cands += [lo, hi]
            for c in cands:
                c = np.float32(min(max(c, np.float32(-1)), np.float32(1)))
                w = np.asarray(self._wrap_action(np.array([c], dtype=np.float32)), dtype=np.float32)[0]
                err = abs(float(w) - float(u[k]))
                if best is None or err < best[0]:
                    best = (err, c)
                if err == 0.0:
                    break
            a[k] = best[1]
        return a

    def _dynamics(self, state, control):
        """next state = f(state, wrapped control): the reference's pure one-step map (cartpole_env.py:52-92, quadrotor_env.py:417-528,
        :1044-1130), evaluated by one tg_env_step on temporaries in this env's dtype (float64 by default); the env's own state does
        not move.  `control` is what `_wrap_action` returns (a scalar is taken as a 1-vector)."""
        a = self._unwrap_control(control)
        S = self.obs_dim
        st = torch.as_tensor(np.asarray(state, dtype=np.float64).reshape(S, 1)).to(self._dtype).to(self._device)
        act = torch.as_tensor(a.reshape(self.act_dim, 1)).to(self._device)
        steps = torch.zeros(1, dtype=torch.int32, device=self._device)
        tb, rew = torch.zeros(1, dtype=self._dtype, device=self._device), torch.zeros(1, dtype=self._dtype, device=self._device)
        trunc = torch.zeros(1, dtype=torch.uint8, device=self._device)
        p = self.native_params()
        with torch.cuda.device(self._device):
            N.check(N.load().tg_env_step(C.byref(p), N.dtype_code(self._dtype), st.data_ptr(), 1, act.data_ptr(), 1, st.data_ptr(), 1,
                                         steps.data_ptr(), tb.data_ptr(), rew.data_ptr(), trunc.data_ptr(), 1,
                                         N.stream_ptr(self._device)), "tg_env_step")
        return st[:, 0].double().cpu().numpy()

    def _propagate_state(self, control):
        """state_dict <- _dynamics(state, control): the reference's `_propegate*` helpers (no reward, no step counting)."""
        nxt = self._dynamics(self._get_obs(), control)
        self._alloc()
        self._state.copy_(torch.as_tensor(nxt, dtype=self._dtype).reshape(-1, 1))
        self._sync_state_dict()

    def reset(self):
        self._alloc()
        p = self.native_params()
        self._reset_count += 1
        with torch.cuda.device(self._device):
            N.check(N.load().tg_env_reset(C.byref(p), N.dtype_code(self._dtype), self._state.data_ptr(), 1, 1,
                                          self._seed, self._reset_count, 0, 1, N.stream_ptr(self._device)), "tg_env_reset")
        self._initial = self._state.clone()
        self._steps_t.zero_()
        self._tb_t.zero_()
        self._steps, self._time, self._time_balanced = 0, 0, 0
        self._sync_state_dict()
        self._initial_state = {k: v.copy() for k, v in self.state_dict.items()}
        return self._obs_np(), self._get_info()

    def set_state(self, state):
        """Place the env in `state` (S,) and make it the restart point (used by parity tests)."""
        self._alloc()
        self._state.copy_(torch.as_tensor(np.asarray(state), dtype=self._dtype).reshape(-1, 1))
        self._initial = self._state.clone()
        self._steps_t.zero_()
        self._tb_t.zero_()
        self._steps, self._time, self._time_balanced = 0, 0, 0
        self._sync_state_dict()
        self._initial_state = {k: v.copy() for k, v in self.state_dict.items()}
        return self._obs_np()

    def restart(self):
        if self._initial is None:
            raise RuntimeError("restart() before reset()")
        self._state.copy_(self._initial)
        self._steps_t.zero_()
        self._tb_t.zero_()
        self._steps, self._time, self._time_balanced = 0, 0, 0
        self._sync_state_dict()
        return self._obs_np(), self._get_info()

    def step(self, action):
        info = self._get_info()     # the reference builds `info` BEFORE updating time_balanced
        reward, truncated = self._native_step(action)
        return self._obs_np(), reward, False, truncated, info

    def _native_step(self, action):
        """Advance the scalar env by one tg_env_step; returns (reward, truncated) and updates the bookkeeping."""
        if self._state is None:
            raise RuntimeError("step() before reset()")
        p = self.native_params()
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(self.act_dim, 1), device=self._device)
        self._act_t.copy_(a)
        with torch.cuda.device(self._device):
            N.check(N.load().tg_env_step(C.byref(p), N.dtype_code(self._dtype), self._state.data_ptr(), 1,
                                         self._act_t.data_ptr(), 1, self._state.data_ptr(), 1, self._steps_t.data_ptr(),
                                         self._tb_t.data_ptr(), self._rew_t.data_ptr(), self._trunc_t.data_ptr(), 1,
                                         N.stream_ptr(self._device)), "tg_env_step")
        self._steps += 1
        self._time += self.timestep
        self._time_balanced = float(self._tb_t.item())
        self._sync_state_dict()
        return float(self._rew_t.item()), bool(self._trunc_t.item())

    def render(self, *a, **k):
        raise NotImplementedError("render() is matplotlib plotting in the reference and is out of scope here")


class CartPole(Env):
    """Swing-up cart-pole.  environments/cartpole_env.py:6-182."""
    ENV_ID = N.TG_ENV_CARTPOLE
    _state_split = (("cartpole", 5),)

    def __init__(self, env_name: str = "CartPole", masscart: float = 1.0, masspole: float = 1.0, length: float = 0.5,
                 gravity: float = 9.80665, timestep: float = 0.02, max_steps: int = 500, device=None,
                 dtype=torch.float64):
        super().__init__(env_name, device, dtype)
        self.masscart, self.masspole, self.length, self.gravity = masscart, masspole, length, gravity
        self.timestep, self.max_steps = timestep, max_steps
        self.max_time = max_steps * timestep
        self.state_dict = {"cartpole": np.zeros(5)}
        self._is_3d = False
        self.observation_space = Box(low=-1, high=1, shape=(5,), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(1,), dtype=np.float32)

    def _fill_params(self, p):
        p.p[0], p.p[1], p.p[2], p.p[3] = self.masscart, self.masspole, self.length, self.gravity

    _WRAP = (None, 5)                                                          # cartpole_env.py:48-49

    def _propegate_cartpole(self, state, control):                             # cartpole_env.py:94-100 (`state` is the env's own)
        self._propagate_state(control)


class Pendulum(Env):
    """Torque-driven pendulum, upright at theta = pi.  environments/pendulum_env.py:7-158.

    The one env whose episode TERMINATES: once `time_balanced` (consecutive time with cos(theta) <= -0.99) exceeds 5 s.
    `step` returns `(observation, reward, truncated, terminated, info)` -- truncated BEFORE terminated, as the reference
    does (:158); its rollout worker unpacks them the other way round and only uses their disjunction."""
    ENV_ID = N.TG_ENV_PENDULUM
    _state_split = (("pendulum", 3),)
    BALANCE_TIME = 5.0                                                        # :151

    def __init__(self, env_name: str = "Pendulum", swingup: bool = False, mass: float = 1.0, length: float = 0.5,
                 gravity: float = 9.80665, timestep: float = 0.05, max_steps: int = 200, device=None, dtype=torch.float64):
        super().__init__(env_name, device, dtype)
        self.swingup, self.mass, self.length, self.gravity = swingup, mass, length, gravity
        self.timestep, self.max_steps = timestep, max_steps
        self.max_time = max_steps * timestep
        self.state_dict = {"pendulum": np.zeros(3)}
        self.observation_space = Box(low=-1, high=1, shape=(3,), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(1,), dtype=np.float32)

    def _fill_params(self, p):
        p.p[0], p.p[1], p.p[2], p.p[3] = self.mass, self.length, self.gravity, 1.0 if self.swingup else 0.0

    _WRAP = (None, 1)                                                          # pendulum_env.py:45-46

    def _propegate_pendulum(self, state, control):                             # pendulum_env.py:77-83
        self._propagate_state(control)

    def step(self, action):
        reward, truncated = self._native_step(action)
        info = self._get_info()                       # built AFTER the time_balanced update here (:135-139)
        return self._obs_np(), reward, truncated, self._time_balanced > self.BALANCE_TIME, info


class QuadPole2D(Env):
    """Planar quadrotor + pendulum payload.  environments/quadrotor_env.py:867-1223."""
    ENV_ID = N.TG_ENV_QUADPOLE2D
    _state_split = (("quadrotor", 8), ("pendulum", 2))     # as _propogate leaves it (:1041-1042)

    def __init__(self, env_name="QuadPole2D", max_steps=500, timestep=0.02, device=None, dtype=torch.float64):
        super().__init__(env_name, device, dtype)
        self.mq, self.mp, self.I, self.Lq, self.Lp = 1.5, 0.5, 4e-1, 0.5, 0.75
        self.gravity, self.timestep, self.max_steps = 9.80665, timestep, max_steps
        self.spatial_bounds = ((-2.0, 2.0), (-2.0, 2.0))
        self.balance_radius = 0.25
        self._is_3d = False
        self._xbounds, self._zbounds = self.spatial_bounds
        self.hover_force = (self.mq + self.mp) * self.gravity / 2
        self.state_dict = {"quadrotor": np.zeros(8), "pendulum": np.zeros(4)}
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(10,), dtype=np.float32)
        self.action_space = Box(low=0.0, high=20.0, shape=(2,), dtype=np.float32)

    def _fill_params(self, p):
        b = self.spatial_bounds
        if not (b[0][1] == b[1][1] == -b[0][0] == -b[1][0]):
            raise ValueError("the HIP kernel supports symmetric, equal spatial bounds only")
        for i, v in enumerate((self.mq, self.mp, self.I, self.Lq, self.Lp, self.gravity, b[0][1], self.balance_radius)):
            p.p[i] = v

    _WRAP = ("hover_force", "hover_force")                                     # quadrotor_env.py:928

    def out_of_bounds(self):
        """quadrotor_env.py:1009-1022: x or z outside the bounds (reads `state_dict`, like the reference)."""
        x, z = self.state_dict["quadrotor"][0:2]
        return bool(x < self._xbounds[0] or x > self._xbounds[1] or z < self._zbounds[0] or z > self._zbounds[1])

    def _propogate(self, action):                                              # quadrotor_env.py:1024-1042 (sic)
        self._propagate_state(action)


class QuadPole(Env):
    """3-D quadrotor (quaternion attitude) + tethered payload.  environments/quadrotor_env.py:353-713."""
    ENV_ID = N.TG_ENV_QUADPOLE
    _state_split = (("quadrotor", 13), ("pendulum", 7))

    def __init__(self, env_name="QuadPole", max_steps=500, device=None, dtype=torch.float64):
        super().__init__(env_name, device, dtype)
        self.max_steps = max_steps
        self.mass, self.load_mass, self.gravity, self.tether_length = 1.5, 0.5, 9.80665, 0.5
        self.Ixx, self.Iyy, self.Izz = 4e-1, 4e-1, 2.5e-1
        self.torque_constant, self.arm_length, self.timestep = 0.1, 0.5, 0.02
        self.hover_force = (self.mass + self.load_mass) * self.gravity / 4
        self.spatial_bounds = ((-1.5, 1.5), (-1.5, 1.5), (-1.5, 1.5))
        self._xbounds, self._ybounds, self._zbounds = self.spatial_bounds
        self.state_dict = {"quadrotor": np.zeros(13), "pendulum": np.zeros(7)}
        self._is_3d = True
        self.detailed_rendering = False
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(20,), dtype=np.float32)
        self.action_space = Box(low=0.0, high=20.0, shape=(4,), dtype=np.float32)

    def _fill_params(self, p):
        b = self.spatial_bounds
        hi = b[0][1]
        if not all(x[1] == hi and x[0] == -hi for x in b):
            raise ValueError("the HIP kernel supports symmetric, equal spatial bounds only")
        for i, v in enumerate((self.mass, self.load_mass, self.gravity, self.tether_length, self.Ixx, self.Iyy,
                               self.Izz, self.torque_constant, self.arm_length, hi)):
            p.p[i] = v

    _WRAP = ("hover_force", "hover_force")                                     # quadrotor_env.py:409-413

    def _out_of_bounds(self):
        """quadrotor_env.py:613-622: any position coordinate outside its bounds (reads `state_dict`, like the reference)."""
        x, y, z = self.state_dict["quadrotor"][0:3]
        return bool(x < self._xbounds[0] or x > self._xbounds[1] or y < self._ybounds[0] or y > self._ybounds[1]
                    or z < self._zbounds[0] or z > self._zbounds[1])

    def _propegate(self, action):                                              # quadrotor_env.py:578-585
        self._propagate_state(action)


class QuadPoleSwarm(QuadPole):
    """N-body swarm for BASELINE config 5 (build-defined: the reference's `QuadrotorSwarm` is an empty subclass,
    quadrotor_env.py:185-186, so there is nothing to mirror -- SURVEY F3 / 8f.3).

    One environment = `n_agents` independent QuadPole bodies driven by ONE shared policy (each body observes its
    own 20-dim state and receives its own 4 rotor commands and its own reward).  The bodies are coupled only
    through termination: the env truncates for everybody when any body leaves the bounds (or at max_steps).
    In a rollout the bodies of an env occupy `n_agents` consecutive env slots, so the returned tensors are
    `(G, E*n_agents, T, .)` and a GRPO group spans the `E*n_agents` bodies of its E episodes (group-relative
    advantage across the swarm).  `n_agents = 1` is exactly `QuadPole`.  The scalar reset/step API steps a single
    body (no coupling)."""

    def __init__(self, env_name="QuadPoleSwarm", n_agents: int = 8, max_steps=500, device=None, dtype=torch.float64):
        super().__init__(env_name, max_steps, device, dtype)
        if n_agents < 1 or n_agents > 32 or (n_agents & (n_agents - 1)):
            raise ValueError("n_agents must be a power of two <= 32")
        self.n_agents = n_agents

    def _fill_params(self, p):
        super()._fill_params(p)
        p.agents = self.n_agents


class Quadrotor:
    """The reference's `Quadrotor` is a stub whose only usable member is `_dynamics`
    (environments/quadrotor_env.py:6-182); this mirrors that pure function, batched."""

    def __init__(self, mass=1.0, arm_length=0.2, Ixx=0.005, Iyy=0.005, Izz=0.006, torque_constant=0.017,
                 gravity=9.80665, timestep=0.05, max_steps=200, device=None):
        self.mass, self.arm_length, self.Ixx, self.Iyy, self.Izz = mass, arm_length, Ixx, Iyy, Izz
        self.torque_constant, self.gravity, self.timestep, self.max_steps = torque_constant, gravity, timestep, max_steps
        self._device = torch.device(device) if device is not None else torch.device("cuda", 0)

    def _dynamics(self, state, control):
        """state (12,) or (n,12), control (4,) or (n,4) -> next state, same shape (NumPy float64)."""
        st = np.atleast_2d(np.asarray(state, dtype=np.float64))
        ct = np.atleast_2d(np.asarray(control, dtype=np.float64))
        n = st.shape[0]
        p = N.default_params(N.TG_ENV_QUADROTOR12, self.max_steps)
        p.timestep = float(self.timestep)
        for i, v in enumerate((self.mass, self.arm_length, self.Ixx, self.Iyy, self.Izz, self.torque_constant, self.gravity)):
            p.p[i] = v
        s_d = torch.as_tensor(st.T.copy(), device=self._device)
        c_d = torch.as_tensor(ct.T.copy(), device=self._device)
        o_d = torch.empty_like(s_d)
        N.check(N.load().tg_quadrotor12_dynamics(C.byref(p), N.TG_F64, s_d.data_ptr(), n, c_d.data_ptr(), n,
                                                 o_d.data_ptr(), n, n, N.stream_ptr(self._device)), "tg_quadrotor12_dynamics")
        out = o_d.cpu().numpy().T
        return out[0] if np.ndim(state) == 1 else out

    # The rest of the reference's class is scaffolding (quadrotor_env.py:68-111, :172-182): `_pack_state` / `_unpack_state` /
    # `_spawn_obstacles` / `_spawn_goal` do nothing, `_spwan_quadrotor` (sic) draws a position, and reset / restart / step / render
    # forward to a `self.env` that is never set (AttributeError there, and here).
    spatial_bounds = ((-5, 5), (-5, 5), (-5, 5))

    def _pack_state(self):
        return None

    def _unpack_state(self, state):
        pass

    def _spwan_quadrotor(self):
        b = self.spatial_bounds
        self.quadrotor = np.array([np.random.uniform(b[0][0], b[0][1]), np.random.uniform(b[1][0], b[1][1]),
                                   np.random.uniform(b[2][0], b[2][1]), 0, 0, 0, 0, 0, 0, 0, 0, 0])
        return self.quadrotor

    def _spawn_obstacles(self):
        pass

    def _spawn_goal(self):
        pass

    def _no_env(self, *a, **k):
        raise AttributeError(f"'{type(self).__name__}' object has no attribute 'env' (the reference's Quadrotor forwards reset / restart / "
                             "step / render to a member it never sets: quadrotor_env.py:172-182; only `_dynamics` is usable)")

    reset = restart = step = render = _no_env


class QuadrotorSwarm(Quadrotor):
    """`class QuadrotorSwarm(Quadrotor): pass` in the reference (quadrotor_env.py:185-186)."""
    pass


ENV_CLASSES = {"CartPole": CartPole, "QuadPole2D": QuadPole2D, "QuadPole": QuadPole, "QuadPoleSwarm": QuadPoleSwarm,
               "Pendulum": Pendulum}
