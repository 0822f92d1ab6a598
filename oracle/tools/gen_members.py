#!/usr/bin/env python3
"""Write tests/golden/reference_public_members.json: for every class of the reference that the drop-in mirrors, the names of its
members (methods, properties, class attributes defined in the reference's own modules; dunder names left out), plus dynamics-call
fixtures (`_dynamics` / `_wrap_action` / `out_of_bounds` inputs and outputs) for the member tests.  Names and numbers only -- no
reference source text.  Runs in the build container, like gen_goldens.py:

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/tools/gen_members.py
"""
import inspect
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, os.environ.get("TRAJOPT_REFERENCE", "/root/reference"))

import environments as E  # noqa: E402  (reference)
from environments.quadrotor_env import Quadrotor, QuadrotorSwarm  # noqa: E402
from environments.pendulum_env import Pendulum  # noqa: E402
import policies as P  # noqa: E402
import rollout as R  # noqa: E402
import buffers as B  # noqa: E402
from buffers.buffer import Buffer  # noqa: E402
import algorithms as A  # noqa: E402
from algorithms.algorithm import Algorithm  # noqa: E402
from models.neural_network import NeuralNetwork  # noqa: E402
from pipelines.pipeline import Pipeline  # noqa: E402

CLASSES = {"Env": E.Env, "CartPole": E.CartPole, "QuadPole": E.QuadPole, "QuadPole2D": E.QuadPole2D, "Pendulum": Pendulum,
           "Quadrotor": Quadrotor, "QuadrotorSwarm": QuadrotorSwarm,
           "GaussianActor_NeuralNetwork": P.GaussianActor_NeuralNetwork,
           "GaussianActorCritic_NeuralNetwork": P.GaussianActorCritic_NeuralNetwork,
           "RolloutWorker": R.RolloutWorker, "RolloutManager": R.RolloutManager,
           "Buffer": Buffer, "Rollout_Buffer": B.Rollout_Buffer,
           "Algorithm": Algorithm, "GRPO": A.GRPO, "PPO": A.PPO, "NeuralNetwork": NeuralNetwork, "Pipeline": Pipeline}


def own_members(cls):
    """Names defined by the reference's own classes in `cls`'s MRO (not by gymnasium / torch / abc / object)."""
    names = set()
    for k in cls.__mro__:
        mod = getattr(k, "__module__", "")
        if mod.split(".")[0] in ("builtins", "abc", "torch", "gymnasium", "typing"):
            continue
        for n, v in vars(k).items():
            if n.startswith("__") or n in ("_abc_impl",):
                continue
            names.add(n)
    return sorted(names)


out = {"classes": {n: own_members(c) for n, c in CLASSES.items()}, "calls": {}}

rng = np.random.default_rng(7)
calls = out["calls"]
# CartPole: _wrap_action, _dynamics on the wrapped (float32) control -- cartpole_env.py:48-92
env = E.CartPole()
cs = []
for _ in range(6):
    st = np.array([rng.uniform(-1, 1), rng.uniform(-2, 2), 0, 0, rng.uniform(-12, 12)])
    th = rng.uniform(-np.pi, np.pi)
    st[2], st[3] = np.sin(th), np.cos(th)
    a = np.array([rng.uniform(-1.5, 1.5)], dtype=np.float32)
    u = env._wrap_action(a)
    cs.append({"state": st.tolist(), "action": a.tolist(), "wrapped": np.asarray(u, dtype=np.float64).tolist(),
               "wrapped_dtype": str(np.asarray(u).dtype), "next": env._dynamics(st, u).tolist()})
calls["CartPole"] = cs
for name, cls, S, Aa in (("QuadPole", E.QuadPole, 20, 4), ("QuadPole2D", E.QuadPole2D, 10, 2)):
    env = cls()
    env.reset()
    cs = []
    for _ in range(6):
        env.reset()
        st = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in env.state_dict.values()])[:S]
        st = st + rng.normal(0, 0.2, S)
        if name == "QuadPole":
            st[6:10] /= np.linalg.norm(st[6:10])
            st[13:17] /= np.linalg.norm(st[13:17])
        a = rng.uniform(-1.5, 1.5, Aa).astype(np.float32)
        u = env._wrap_action(a)
        cs.append({"state": st.tolist(), "action": a.tolist(), "wrapped": np.asarray(u, dtype=np.float64).tolist(),
                   "wrapped_dtype": str(np.asarray(u).dtype), "next": np.asarray(env._dynamics(st, u), dtype=np.float64).tolist()})
    calls[name] = cs
# out-of-bounds predicates (quadrotor_env.py:613-622, :1009-1022)
oob = []
env3, env2 = E.QuadPole(), E.QuadPole2D()
env3.reset(); env2.reset()
for pos in ([0, 0, 0], [1.6, 0, 0], [0, -1.51, 0], [0, 0, 1.5], [1.4, 1.4, -1.6], [-2.1, 0, 0], [0, 0, 2.0], [1.9, 0, -2.01]):
    # (fresh float arrays: reset() leaves INTEGER arrays in state_dict, into which 1.6 would be stored as 1)
    env3.state_dict["quadrotor"] = np.array(list(pos) + [0.0] * 10, dtype=np.float64)
    env2.state_dict["quadrotor"] = np.array([pos[0], pos[2]] + [0.0] * 6, dtype=np.float64)
    oob.append({"pos": pos, "QuadPole": bool(env3._out_of_bounds()), "QuadPole2D": bool(env2.out_of_bounds())})
calls["out_of_bounds"] = oob

path = os.path.join(REPO, "tests", "golden", "reference_public_members.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", path)
for n, m in out["classes"].items():
    print(n, m)
