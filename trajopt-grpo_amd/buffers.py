"""Rollout_Buffer with the reference's surface, holding the trajectory on the device.

Mirrors buffers/buffer.py and buffers/rollout_buffer.py:10-126.  `sample()` keeps the
`DeviceTrajectory` the rollout produced; `group_observations / group_actions / group_rewards /
group_lengths / group_masks` are the reference's CPU float32 `(G,E,T,.)` tensors, materialised
lazily the first time a legacy consumer (Dashboard, Publisher, a reference learner) reads them.
`avg_reward` is the same metric: mean over (G,E) of the undiscounted episode return (:70),
reduced over ranks from (sum, count) under torch.distributed.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch

from . import distributed as D


class Buffer:
    """buffers/buffer.py."""

    def __init__(self):
        pass

    def visualize(self):                 # buffers/buffer.py:7-9: an abstract no-op
        pass


class Rollout_Buffer(Buffer):
    _REF_FIELDS = ("group_observations", "group_actions", "group_rewards", "group_lengths", "group_masks")

    def __init__(self, rollout_manager, rtg: bool = True):
        self.rollout_manager = rollout_manager
        self.env = rollout_manager.env_fn()
        self.rtg = rtg                       # stored, never used -- as in the reference (:14,19)
        self.device_traj = None              # DeviceTrajectory of the last sample()
        self._ref = None                     # cached reference-layout CPU tensors
        self._ref_limits = (None, None)      # (max_groups, max_episodes) of the lazy reference view; None = everything
        self._ref_is_full = False
        self._avg_reward = []
        self._pending = None                 # (pinned f64 [3] = reward sum, env count, local valid rows; event) of the last sample()
        self._stats_host = None
        self.fig = None
        self.axs = None

    # ---- lazy reference-layout attributes ------------------------------------------
    def limit_reference_view(self, max_groups=None, max_episodes=None):
        """Restrict what the lazy `group_*` attributes copy from the device to the first `max_groups` groups and the
        first `max_episodes` episodes of each group: a visualiser that renders `max_episodes_per_render`
        episodes per group (visualize/dashboard.py:206-217) then costs a few MB per render instead of the whole
        trajectory (1.7 GB at C3).  `retrieve()` and `save_trajectory()` always use the full trajectory."""
        self._ref_limits = (max_groups, max_episodes)
        if not self._ref_is_full:
            self._ref = None

    @contextlib.contextmanager
    def limited_view(self, max_groups=None, max_episodes=None):
        """`limit_reference_view` for the duration of a `with` block only: Pipeline wraps a visualiser's render() in it, so the
        slice is what the visualiser sees while every other consumer of `buffer.group_*` (a reference-style CPU learner, a
        Publisher) still gets the whole trajectory."""
        saved = self._ref_limits
        self.limit_reference_view(max_groups, max_episodes)
        try:
            yield self
        finally:
            self._ref_limits = saved
            if not self._ref_is_full:
                self._ref = None              # a sliced cache must not outlive the block

    def _materialise(self, full: bool = False):
        if self._ref is not None and (self._ref_is_full or not full):
            return self._ref
        if self.device_traj is None:
            return self._ref                  # tensors handed to store(): always complete
        limits = (None, None) if full else self._ref_limits
        self._ref = dict(zip(self._REF_FIELDS, self.device_traj.to_reference(*limits)))
        self._ref_is_full = limits == (None, None)
        return self._ref

    def __getattr__(self, name):
        if name in Rollout_Buffer._REF_FIELDS:
            ref = self._materialise()
            return None if ref is None else ref[name]
        raise AttributeError(name)

    # ---- reference surface ---------------------------------------------------------------
    def load(self, path: str):
        self.avg_reward = np.atleast_1d(np.loadtxt(os.path.join(path, "reward.csv"), delimiter=",")).tolist()
        return len(self.avg_reward)

    # ---- avg_reward: the reference's list of per-iteration averages (rollout_buffer.py:70), resolved lazily ----------------
    @property
    def avg_reward(self):
        self._resolve()
        return self._avg_reward

    @avg_reward.setter
    def avg_reward(self, value):
        self._pending = None
        self._avg_reward = value

    def _resolve(self):
        """Wait for the last sample()'s statistics (an asynchronous device -> pinned-host copy) and append its average reward.
        sample() itself does not wait: the learner enqueues its return-to-go / advantage kernels first and asks for the number of
        valid rows (`valid_rows()`) only when it needs a shape -- by then the copy has landed behind the rollout."""
        if self._pending is not None:
            host, ev = self._pending
            self._pending = None
            ev.synchronize()
            s, c, v = host.tolist()
            self._stats_host = (s, c, int(v))
            self._avg_reward.append(np.asarray(s / c, dtype=np.float32))
        return self._stats_host

    def valid_rows(self) -> int:
        """Valid (mask = 1) env-steps of the last sample() on this rank."""
        return self._resolve()[2]

    def sample(self):
        mgr = self.rollout_manager
        if hasattr(mgr, "rollout_device"):
            self._resolve()                                   # (the previous iteration's, long since complete)
            traj = mgr.rollout_device()
            self.device_traj, self._ref, self._ref_is_full = traj, None, False
            dev = traj.rew.device
            if getattr(traj, "stats_fresh", False):
                # {sum of rewards, n, this rank's valid rows} came out of the rollout's own last launch (tg_rollout_finish_stats)
                stats = traj.stats
                if D.rank_world(getattr(mgr, "process_group", None))[1] > 1 or D._ALWAYS:
                    stats = stats.clone()                     # (the all-reduce must not turn the trajectory's own record global)
            else:
                stats = torch.empty(3, dtype=torch.float64, device=dev)
                stats[0:1] = traj.rew.sum(dtype=torch.float64)
                stats[1].fill_(float(traj.n))
                stats[2:3].copy_(traj.counters[0:1])          # this rank's valid rows (tg_rollout_finish)
            D.allreduce_sum_(stats[0:2], getattr(mgr, "process_group", None), "avg_reward")
            if getattr(self, "_pinned", None) is None:
                self._pinned = torch.empty(3, dtype=torch.float64).pin_memory()
            self._pinned.copy_(stats, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._pending = (self._pinned, ev)
            traj.host_valid_rows = self.valid_rows            # (DeviceTrajectory.env_steps() / the learner ask through this)
        else:
            self.store(*mgr.rollout())

    def store(self, group_observations, group_actions, group_rewards, group_lengths, group_masks):
        """rollout_buffer.py:55-70 (CPU tensors from a legacy manager)."""
        self.device_traj = None
        self._ref = dict(zip(self._REF_FIELDS, (group_observations, group_actions, group_rewards, group_lengths,
                                                group_masks)))
        self._ref_is_full = True
        self._resolve()
        self._avg_reward.append(group_rewards.sum(2).mean().detach().numpy())

    def retrieve(self):
        """The reference reads attributes that are never set (:107-108); return the five stored tensors."""
        r = self._materialise(full=True)
        return tuple(r[k] for k in self._REF_FIELDS)

    def save_trajectory(self, path: str):
        """trajectory.csv: episode_id, observation_i..., action_j...  (rollout_buffer.py:72-102)."""
        import pandas as pd
        ref = self._materialise(full=True)
        obs = ref["group_observations"].numpy()
        act = ref["group_actions"].numpy()
        lens = ref["group_lengths"].numpy().astype(int)
        rows_o, rows_a, eid = [], [], []
        for i in range(lens.shape[0]):
            for j in range(lens.shape[1]):
                L = lens[i, j]
                rows_o.append(obs[i, j, :L])
                rows_a.append(act[i, j, :L])
                eid.extend([j + i * lens.shape[1]] * L)
        header = ["episode_id"] + [f"observation_{i}" for i in range(obs.shape[3])] + \
                 [f"action_{i}" for i in range(act.shape[3])]
        data = np.hstack([np.array(eid).reshape(-1, 1), np.vstack(rows_o), np.vstack(rows_a)])
        df = pd.DataFrame(data, columns=header)
        df["episode_id"] = df["episode_id"].astype(int)
        df.to_csv(os.path.join(path, "trajectory.csv"), index=False)

    def metadata(self):
        return {"avg_reward": float(self.avg_reward[-1]) if len(self.avg_reward) > 0 else None}

    def save(self, path: str):
        with open(os.path.join(path, "reward.csv"), "w") as f:
            for reward in self.avg_reward:
                f.write(f"{reward}\n")
