"""GPU-resident rollout: replaces the reference's RolloutManager / RolloutWorker process pool.

Reference: rollout/rollout_manager.py:8-133 and rollout/rollout_worker.py:5-84.  There a
pool of CPU workers each steps ONE NumPy env and calls the policy on ONE observation per
step.  Here all `N = num_workers x num_episodes_per_worker` episodes of a rollout run in
lock-step on the GPU: per time step one batched actor forward (PyTorch-ROCm GEMMs) and one
fused HIP kernel (sample action -> Env.step -> record -> terminate).  The trajectory never
leaves HBM; the reference's CPU float32 `(G,E,T,.)` 5-tuple is materialised only when a
legacy caller asks for it (`rollout()` / buffer attributes).

Device layout (env index fastest, see include/trajopt_grpo_hip.h):
    obs [S][T+1][N]   act [A][T][N] f32   rew [T][N]   mask [T][N] u8   len [N] i32
Flat env index n = g*E + e (buffers/rollout_buffer.py:85-89).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as N
from . import distributed as D
from . import mlp as M


class DeviceTrajectory:
    """The tensors of one rollout, on the device, plus views the learner consumes."""

    def __init__(self, S, A, T, n, G, E, dtype, device):
        self.S, self.A, self.T, self.n, self.G, self.E = S, A, T, n, G, E
        self.dtype, self.device = dtype, device
        self.obs = torch.zeros(S, T + 1, n, dtype=dtype, device=device)
        self.act = torch.zeros(A, T, n, dtype=torch.float32, device=device)
        self.rew = torch.zeros(T, n, dtype=dtype, device=device)
        self.mask = torch.zeros(T, n, dtype=torch.uint8, device=device)
        self.len = torch.zeros(n, dtype=torch.int32, device=device)
        self.counters = torch.zeros(4, dtype=torch.int64, device=device)
        # {sum of rewards, n, valid env-steps} of the last rollout (tg_rollout_finish_stats) and its scratch
        self.stats = torch.zeros(3, dtype=torch.float64, device=device)
        self._stats_work = torch.empty(256, dtype=torch.float64, device=device)
        self.stats_fresh = False             # `stats` belongs to the rollout these tensors hold (a caller that edits them says so by clearing it)
        self.host_valid_rows = None          # callable -> this rollout's valid rows, already on the host (Rollout_Buffer.sample)

    def native(self) -> N.Traj:
        t = N.Traj()
        t.d_obs, t.d_act, t.d_rew = self.obs.data_ptr(), self.act.data_ptr(), self.rew.data_ptr()
        t.d_mask, t.d_len, t.d_counters = self.mask.data_ptr(), self.len.data_ptr(), self.counters.data_ptr()
        t.n, t.horizon, t.dtype = self.n, self.T, N.dtype_code(self.dtype)
        return t

    # ---- learner views (no copies) ----------------------------------------
    def obs_rows(self) -> torch.Tensor:
        """[T*N][S] strided view: row i = t*N + n is the observation before action t of env n."""
        return self.obs[:, :self.T, :].reshape(self.S, self.T * self.n).t()

    def act_rows(self) -> torch.Tensor:
        return self.act.reshape(self.A, self.T * self.n).t()

    def env_steps(self) -> int:
        if self.host_valid_rows is not None:
            return int(self.host_valid_rows())
        return int(self.counters[0].item())

    # ---- reference layout (CPU float32), rollout_manager.py:86-90 -------------------
    def to_reference(self, max_groups=None, max_episodes=None):
        """The reference's (G, E, T, .) CPU float32 5-tuple.  max_groups / max_episodes: only the first groups and the
        first episodes of each group are sliced out ON THE DEVICE and copied -- what the reference's Dashboard reads
        (`group_observations[i, ep, frame]` for ep < max_episodes_per_render, visualize/dashboard.py:206-217,
        visualize/visualizer.py:120-140) is a few MB, the whole C3 trajectory 1.7 GB."""
        G, E, T = self.G, self.E, self.T
        g = G if max_groups is None else max(0, min(G, int(max_groups)))
        e = E if max_episodes is None else max(0, min(E, int(max_episodes)))

        def cut(x, lead):                        # x [lead...][n] -> [lead...][g][e]
            return x.reshape(*lead, G, E)[..., :g, :e]

        # (the permutation runs on the device: a 1.7 GB strided gather is milliseconds there, seconds on the host)
        obs = cut(self.obs[:, :T, :], (self.S, T)).permute(2, 3, 1, 0).float().contiguous().cpu()
        act = cut(self.act, (self.A, T)).permute(2, 3, 1, 0).float().contiguous().cpu()
        rew = cut(self.rew, (T,)).permute(1, 2, 0).float().contiguous().cpu()
        mask = cut(self.mask, (T,)).permute(1, 2, 0).float().contiguous().cpu()
        ln = cut(self.len, ()).float().contiguous().cpu()          # float32, like the manager's torch.zeros (:89)
        return obs, act, rew, ln, mask


class DeviceRollout:
    """Runs `num_groups x episodes_per_group` episodes of `env` under `policy` on one GPU.

    restart=False: every episode draws its own initial state (rollout_worker.py:72-73).
    restart=True : the E episodes of a group share the group's initial state (:70-71), the
                   GRPO "same prompt, different samples" arrangement.
    """

    def __init__(self, env, policy, num_groups: int, episodes_per_group: int, restart: bool = False,
                 dtype=torch.float32, device=None, seed: int = 0, group_offset: int = 0,
                 compute_dtype: Optional[torch.dtype] = None, use_graph: Optional[bool] = False,
                 fused: Optional[bool] = None):
        self.lib = N.load()
        self.env, self.policy = env, policy
        # a swarm env contributes n_agents bodies per episode, laid out as consecutive env slots of the group
        self.agents = int(getattr(env, "n_agents", 1))
        self.G, self.E = int(num_groups), int(episodes_per_group) * self.agents
        self.n = self.G * self.E
        self.restart = bool(restart)
        self.device = torch.device(device) if device is not None else policy.device
        if self.device.type != "cuda":
            raise N.NativeLibraryError("DeviceRollout needs an MI355X (cuda) device; there is no CPU fallback")
        self.S, self.A, self.T = env.obs_dim, env.act_dim, int(env.max_steps)
        self.dtype = dtype
        self.compute_dtype = compute_dtype
        self.group_offset = int(group_offset)
        self.traj = DeviceTrajectory(self.S, self.A, self.T, self.n, self.G, self.E, dtype, self.device)
        self.rng = torch.tensor([int(seed), 0], dtype=torch.int64, device=self.device)
        self.params = env.native_params()
        self._sigma = (C.c_float * self.A)(*[float(v) for v in torch.sqrt(policy.var)])
        self._linears = [m for m in policy.actor.network if isinstance(m, torch.nn.Linear)]
        self._lowp = None
        self._mlp = None
        if M.supports(policy.actor):
            # hand-scheduled GEMM chain (bias+ReLU in the GEMM epilogue, padded K / head): mlp.py
            self._mlp = M.GemmMLP(policy.actor, compute_dtype or torch.float32)
            self._xp = torch.zeros(self.n, self._mlp.in_pad, dtype=self._mlp.cd, device=self.device)
        elif compute_dtype is not None and compute_dtype != torch.float32:
            self._lowp = [(torch.empty_like(l.weight, dtype=compute_dtype), torch.empty_like(l.bias, dtype=compute_dtype))
                          for l in self._linears]
        self.use_graph = use_graph
        self._graph = None
        self._graph_baked = None
        # fused persistent rollout kernel (csrc/fused_rollout.hip): whole T-step loop in one launch, actor MLP on
        # the matrix cores.  Auto-selected for bf16 policies whose shape it supports; `fused=True` insists.
        H = M.fused_rollout_supported(policy.actor, self.S, self.A)
        can_fuse = bool(H) and dtype == torch.float32 and compute_dtype == torch.bfloat16
        # ... and its float32 sibling (csrc/fused_rollout_f32.hip) for fp32 policies of the reference's sizes (64 / 128
        # wide, up to 4 hidden layers): fp32 products, weights register-resident, one workgroup per 32 envs.
        H32 = M.fused_rollout_f32_supported(policy.actor, self.S, self.A)
        self._fused_f32 = (not can_fuse) and bool(H32) and dtype == torch.float32 and compute_dtype in (None, torch.float32)
        if self._fused_f32:
            can_fuse, H = True, H32
        if fused and not can_fuse:
            raise ValueError("fused rollout needs a float32 trajectory and an actor Linear(S,H) ReLU [Linear(H,H) ReLU]* "
                             "Linear(H,A), S<=32, A<=4: H in {128,256} with compute_dtype=bfloat16, or H in {64,128} and "
                             "1..4 hidden layers in float32")
        self.fused = can_fuse if fused is None else bool(fused)
        self._fused_H = H
        self._frag = None
        if use_graph is None:          # auto: the per-step launch path is launch-bound, replay it as one hipGraph
            self.use_graph = not self.fused
        # when set to a list, every tg_rollout_step launch is bracketed by HIP events on the launch
        # stream (bench.py reads them back for the dynamics kernel's roofline)
        self.step_events = None
        # set by a learner whose fused optimizer step keeps this engine's weight stream current: callable -> True when it has just
        # rebuilt the stream (one gather launch covering every layout of the policy); run() calls it at every entry
        self.entry_refresh = None
        # True: a teacher-forced replay runs as T launches of the golden-pinned tg_rollout_step instead of one tg_rollout_forced
        # launch (the parity tests replay both ways and compare bits)
        self.forced_per_step = False
        # 16 / 32: envs per workgroup of the fp32 fused rollout, fixed before the first run() (tests hold both kernels to the same
        # bar); None: tg_fused_rollout_f32_block_envs decides
        self.f32_block_envs = None
        self._f32_block_envs = 32

    # ---- policy mean for time step t -------------------------------------------------
    def _refresh_weights(self, entry: bool = False):
        """entry: called at a rollout's entry -- whoever wrote the weights since the last rollout may have done so through `.data`,
        which moves no version key: every derived layout counts as stale (TG_TRUST_VERSION_KEYS=1: the keys decide)."""
        if self._mlp is not None:
            self._mlp.refresh(force=entry and not N.TRUST_KEYS)
        if self._lowp is not None:
            for (w, b), lin in zip(self._lowp, self._linears):
                w.copy_(lin.weight)
                b.copy_(lin.bias)

    def _actor_mean(self, t: int) -> torch.Tensor:
        """Policy mean for slot t: fp32 [N][>=A] with unit column stride (row stride = .stride(0))."""
        x = self.traj.obs[:, t, :].t()                        # [N][S] view of the SoA slot
        if self._mlp is not None:
            self._xp[:, :self.S].copy_(x)
            return self._mlp.forward(self._xp, keep=False, padded=True)
        if self._lowp is None:
            h = x if x.dtype == torch.float32 else x.float()
            return self.policy.actor(h).contiguous()
        h = x.to(self.compute_dtype)
        li = 0
        for mod in self.policy.actor.network:
            if isinstance(mod, torch.nn.Linear):
                w, b = self._lowp[li]
                li += 1
                h = F.linear(h, w, b)
            else:
                h = mod(h)
        return h.float().contiguous()

    # ---- one rollout -------------------------------------------------------------------
    def _reset(self, tr, st):
        # initial states go to slot 0 of obs; the draw is keyed by the global env (or group) index and
        # by the rollout counter rng[1] (read on the host: it only selects the Philox stream)
        seed, stream_id = int(self._seed_host), int(self._stream_host)
        N.check(self.lib.tg_env_reset(C.byref(self.params), tr.dtype, tr.d_obs, (self.T + 1) * self.n, self.n,
                                      seed, stream_id, self.group_offset * self.E, self.E if self.restart else 1, st),
                "tg_env_reset")

    @torch.no_grad()
    def run(self, initial_states=None, forced_actions=None) -> DeviceTrajectory:
        """One rollout.  `initial_states` (N,S) and `forced_actions` (N,T,A) or (G,E,T,A)
        replace the RNG draws (teacher-forced parity runs)."""
        self.params = self.env.native_params()
        self.traj.host_valid_rows = None                 # (set again by Rollout_Buffer.sample for THIS rollout)
        self.traj.stats_fresh = False
        # the policy's covariance is read fresh every rollout (the reference reads self.cov in every forward,
        # actor_critic.py:131-136; the learner reads policy.var in every learn())
        self._sigma = (C.c_float * self.A)(*[float(v) for v in torch.sqrt(self.policy.var)])
        if not hasattr(self, "_stream_host"):
            self._seed_host, self._stream_host = int(self.rng[0].item()), 0
        sample = forced_actions is None
        if self.fused and sample:
            with torch.cuda.device(self.device):
                self._enqueue_prepare(initial_states)
                self._enqueue_fused(0, self.T)
        elif self.use_graph and sample and initial_states is None:
            self._run_graph()
        else:
            with torch.cuda.device(self.device):
                if forced_actions is not None:
                    fa = torch.as_tensor(np.asarray(forced_actions), dtype=torch.float32).reshape(self.n, self.T, self.A)
                    self._enqueue_prepare(initial_states)
                    self.traj.act.copy_(fa.permute(2, 1, 0).to(self.device))
                    self._enqueue_steps(sample=False)
                else:
                    self._enqueue_prepare(initial_states)
                    self._enqueue_steps(sample=True, entry=True)
        self._stream_host += 1
        return self.traj

    def _enqueue_prepare(self, initial_states):
        lib, tr, st = self.lib, self.traj.native(), N.stream_ptr(self.device)
        N.check(lib.tg_rollout_begin(C.byref(tr), self.S, self.A, st), "tg_rollout_begin")
        if initial_states is None:
            self._reset(tr, st)
        else:
            init = torch.as_tensor(np.asarray(initial_states), dtype=self.dtype).reshape(self.n, self.S)
            self.traj.obs[:, 0, :].copy_(init.t().to(self.device))

    def _enqueue_fused(self, t_begin: int, t_end: int):
        """All steps [t_begin, t_end) in one persistent launch (tg_fused_rollout)."""
        lib, tr, st = self.lib, self.traj.native(), N.stream_ptr(self.device)
        if self._frag is None:
            if self._fused_f32:
                # 16 envs per workgroup while every workgroup still gets a CU of its own (C2's 4,096 envs), else 32
                self._f32_block_envs = self.f32_block_envs or lib.tg_fused_rollout_f32_block_envs(self.n, int(self.params.agents))
                self._frag = M.RegisterStreamF32(self.policy.actor, self._fused_H, self._f32_block_envs)
            else:
                self._frag = M.FragmentStream(self.policy.actor, self._fused_H)
        elif N.TRUST_KEYS and getattr(self._frag, "is_fresh", lambda: False)():
            pass                                                  # (the learner's last launch rebuilt this stream, and the keys are trusted)
        else:
            # weights change every learn(); a write through `.data` leaves no trace in any key, so the stream is rebuilt at every
            # entry: by the learner's one gather launch when it has registered one (entry_refresh), else by the stream's own refresh
            hook = getattr(self, "entry_refresh", None)
            if hook is None or not hasattr(self._frag, "segments") or not hook():
                self._frag.refresh()
        n_hidden = len(self._linears) - 1
        ev = None
        if self.step_events is not None:
            ev = N.event_pair()
            ev[0].record()
        if self._fused_f32:
            N.check(lib.tg_fused_rollout_f32(C.byref(self.params), C.byref(tr), self._frag.stream.data_ptr(),
                                             self._frag.table.data_ptr(), self._fused_H, n_hidden, self._f32_block_envs, self._sigma,
                                             self.rng.data_ptr(), self.group_offset * self.E, t_begin, t_end, st),
                    "tg_fused_rollout_f32")
        else:
            N.check(lib.tg_fused_rollout(C.byref(self.params), C.byref(tr), self._frag.stream.data_ptr(),
                                         self._frag.bias.data_ptr(), self._fused_H, n_hidden, self._sigma, self.rng.data_ptr(),
                                         self.group_offset * self.E, t_begin, t_end, st), "tg_fused_rollout")
        if ev is not None:
            ev[1].record()
            self.step_events.append((None, ev[0], ev[1]))
        self._enqueue_finish(tr, st)

    def _enqueue_finish(self, tr, st):
        """Episode counters, the statistics sample() reads, and the RNG stream's advance: two launches (tg_rollout_finish_stats)."""
        N.check(self.lib.tg_rollout_finish_stats(C.byref(tr), self.rng.data_ptr(), self.traj.stats.data_ptr(), self.traj._stats_work.data_ptr(), st),
                "tg_rollout_finish_stats")
        self.traj.stats_fresh = True

    def _enqueue_steps(self, sample: bool, entry: bool = False):
        lib, tr, st = self.lib, self.traj.native(), N.stream_ptr(self.device)
        p = C.byref(self.params)
        if sample:
            self._refresh_weights(entry)
        elif not self.forced_per_step:
            # teacher-forced replay: every time step in ONE launch, the state in registers between steps (tg_rollout_forced;
            # bit-identical to T launches of tg_rollout_step)
            ev = None
            if self.step_events is not None:
                ev = N.event_pair()
                ev[0].record()
            N.check(lib.tg_rollout_forced(p, C.byref(tr), 0, self.T, st), "tg_rollout_forced")
            if ev is not None:
                ev[1].record()
                self.step_events.append((None, ev[0], ev[1]))
            self._enqueue_finish(tr, st)
            return
        env_offset = self.group_offset * self.E
        for t in range(self.T):
            ev = None
            if sample:
                mean = self._actor_mean(t)
                if self.step_events is not None:
                    ev = N.event_pair()
                    ev[0].record()
                N.check(lib.tg_rollout_step(p, C.byref(tr), t, mean.data_ptr(), mean.stride(0), self._sigma,
                                            self.rng.data_ptr(), env_offset, st), "tg_rollout_step")
            else:
                if self.step_events is not None:
                    ev = N.event_pair()
                    ev[0].record()
                N.check(lib.tg_rollout_step(p, C.byref(tr), t, None, 0, None, None, env_offset, st), "tg_rollout_step")
            if ev is not None:
                ev[1].record()
                self.step_events.append((t, ev[0], ev[1]))
        self._enqueue_finish(tr, st)

    # ---- hipGraph replay of the whole T-step loop ---------------------------------------------
    def _run_graph(self):
        with torch.cuda.device(self.device):
            # the reset draw depends on the host-side stream id, so it stays outside the graph
            lib, tr, st = self.lib, self.traj.native(), N.stream_ptr(self.device)
            N.check(lib.tg_rollout_begin(C.byref(tr), self.S, self.A, st), "tg_rollout_begin")
            self._reset(tr, st)
            # tg_rollout_step takes sigma (and the env parameters) BY VALUE in its kernel arguments: a captured graph replays the
            # values of its capture.  A covariance annealed or restored since then (the learner reads policy.var afresh in every
            # learn(), actor_critic.py:131-136 reads self.cov in every forward) means a new capture.
            baked = (tuple(self._sigma), bytes(self.params))
            if self._graph is not None and baked != self._graph_baked:
                self._graph = None
            if self._graph is None:
                side = torch.cuda.Stream(self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):      # warm-up outside capture (library init, autotune)
                    self._refresh_weights()
                    self._actor_mean(0)
                torch.cuda.current_stream(self.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue_steps(sample=True)
                self._graph = g
                self._graph_baked = baked
            # The weights change with every learn().  The capture recorded no rebuild of the operands the forward launches read
            # (its warm-up had just built them: nothing was stale), so they are rebuilt here, eagerly, in front of every replay;
            # the graph reads them through the same buffers.  (The low-precision copies of a non-MLP actor ARE captured: their
            # copies are unconditional.)
            if self._mlp is not None:
                self._mlp.fresh_forward(force=not N.TRUST_KEYS)
            self._graph.replay()
            self.traj.stats_fresh = True                 # (the captured tg_rollout_finish_stats has just re-written them)


# ---------------------------------------------------------------------------
# drop-in classes
# ---------------------------------------------------------------------------
class RolloutWorker:
    """rollout/rollout_worker.py:5-84.  `run_episodes` runs the E episodes in parallel on the GPU
    and returns the reference's CPU tensors: obs (E,T,S), act (E,T,A), rew (E,T), len (E,) int32, mask (E,T)."""

    def __init__(self, worker_id: int, env, policy, episodes_completed, **rollout_kwargs):
        self.worker_id, self.env, self.policy = worker_id, env, policy
        self.episodes_completed = episodes_completed
        self._kw = rollout_kwargs
        self._engines = {}

    def _engine(self, num_episodes, restart):
        key = (num_episodes, bool(restart))
        if key not in self._engines:
            kw = dict(self._kw)
            kw.setdefault("seed", 0x9E3779B9 * (self.worker_id + 1) & 0x7FFFFFFF)
            self._engines[key] = DeviceRollout(self.env, self.policy, 1, num_episodes, restart, **kw)
        return self._engines[key]

    def run_episodes(self, num_episodes: int = 5, restart: bool = False, initial_states=None, forced_actions=None):
        traj = self._engine(num_episodes, restart).run(initial_states, forced_actions)
        obs, act, rew, ln, mask = traj.to_reference()
        self.episodes_completed[self.worker_id] = num_episodes
        return obs[0], act[0], rew[0], ln[0].int(), mask[0]


class RolloutManager:
    """rollout/rollout_manager.py:21-133, same constructor.  `use_multiprocessing` and
    `worker_class` are accepted for signature compatibility; the GPU pipeline replaces both.

    Under torch.distributed (one process per GPU) the `num_workers` groups are split into
    contiguous whole-group ranges per rank: this object then holds rank-local
    groups and `rollout()` returns the local shard.
    """

    def __init__(self, env_fn, policy, worker_class=RolloutWorker, restart=False, num_workers: int = 4,
                 num_episodes_per_worker: int = 5, use_multiprocessing: bool = True, *, dtype=torch.float32,
                 device=None, seed: int = 0, compute_dtype=None, use_graph: Optional[bool] = None, process_group=None,
                 fused=None):
        self.env_fn, self.worker_class, self.policy = env_fn, worker_class, policy
        self.restart = restart
        self.num_workers = num_workers
        self.num_episodes_per_worker = num_episodes_per_worker
        self.use_multiprocessing = use_multiprocessing
        self.env = env_fn()
        self.obs_dim = self.env.observation_space.shape[0]
        self.act_dim = self.env.action_space.shape[0]
        self.max_steps = self.env.max_steps
        self.process_group = process_group
        self.rank, self.world_size = D.rank_world(process_group)
        self.group_lo, self.group_hi = D.shard_groups(num_workers, self.rank, self.world_size)
        self.local_groups = self.group_hi - self.group_lo
        self.episodes_completed = [0 for _ in range(num_workers)]
        self._engine_kw = dict(dtype=dtype, device=device, seed=seed, compute_dtype=compute_dtype, use_graph=use_graph,
                               fused=fused)
        self._engine = None

    @property
    def engine(self) -> DeviceRollout:
        if self._engine is None:
            self._engine = DeviceRollout(self.env, self.policy, self.local_groups, self.num_episodes_per_worker,
                                         self.restart, group_offset=self.group_lo, **self._engine_kw)
        return self._engine

    def rollout_device(self, initial_states=None, forced_actions=None) -> DeviceTrajectory:
        traj = self.engine.run(initial_states, forced_actions)
        for g in range(self.group_lo, self.group_hi):
            self.episodes_completed[g] = self.num_episodes_per_worker
        return traj

    def rollout(self, initial_states=None, forced_actions=None):
        """-> (obs (G,E,T,S), act (G,E,T,A), rew (G,E,T), len (G,E), mask (G,E,T)) float32 CPU tensors,
        zero beyond each episode's length (rollout_manager.py:85-125)."""
        return self.rollout_device(initial_states, forced_actions).to_reference()

    def print_progress(self):      # cosmetic ANSI bars in the reference (:63-83); nothing to draw here
        pass

    def shutdown(self):
        self._engine = None
