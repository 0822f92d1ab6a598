"""Host-side mirror of the reference surface that needs no GPU: trajectory export, the lazy reference view's bookkeeping."""
import os
import types

import numpy as np
import torch

import trajopt_grpo_amd as tg
from conftest import GOLDEN, load_golden


def test_save_trajectory_writes_the_reference_csv_byte_for_byte(tmp_path):
    """Rollout_Buffer.save_trajectory (buffers/rollout_buffer.py:72-102) on the golden CartPole rollout against the CSV the
    reference itself wrote for the same tensors (tests/golden/trajectory_cartpole_reset.csv, oracle/tools/gen_goldens.py)."""
    g = load_golden("rollout_cartpole.npz")
    buf = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    buf.store(*(torch.from_numpy(g[f"reset_{k}"]) for k in ("obs", "act", "rew", "len", "mask")))
    buf.limit_reference_view(max_episodes=1)                   # must not affect the export
    buf.save_trajectory(str(tmp_path))
    got = open(os.path.join(tmp_path, "trajectory.csv")).read()
    want = open(os.path.join(GOLDEN, "trajectory_cartpole_reset.csv")).read()
    assert got == want
    lines = got.splitlines()
    assert lines[0] == "episode_id," + ",".join(f"observation_{i}" for i in range(5)) + ",action_0"
    assert len(lines) - 1 == int(g["reset_len"].sum())
    assert float(buf.avg_reward[-1]) == float(torch.from_numpy(g["reset_rew"]).sum(2).mean())


def test_retrieve_returns_the_five_stored_tensors():
    g = load_golden("rollout_cartpole.npz")
    buf = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    t = tuple(torch.from_numpy(g[f"reset_{k}"]) for k in ("obs", "act", "rew", "len", "mask"))
    buf.store(*t)
    assert all(a is b for a, b in zip(buf.retrieve(), t)) and buf.group_observations is t[0]
