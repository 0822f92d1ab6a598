"""GRPO and PPO with the reference's constructor / `learn(buffer)` surface, running on the GPU.

Mirrors algorithms/algorithm.py:3-35, algorithms/grpo.py:12-169 and algorithms/ppo.py:8-225.
`learn` reproduces the reference arithmetic as written, including:
  * reward-to-go with the next-step mask inside the recurrence (grpo.py:69-72);
  * GRPO group statistics over ALL valid steps of a group's E episodes, unbiased std, the
    epsilon inside std() (i.e. none) (grpo.py:115);
  * GRPO performs gradient DESCENT on J (grpo.py:137-145) -- `maximize=True`
    is the explicit, non-default divergence;
  * PPO `old_log_probs` come from the CURRENT policy (ppo.py:142-143); the critic regresses on
    batch-normalised returns while A = R_raw - V (ppo.py:111,139,169); the entropy of the
    fixed-covariance Gaussian is a constant (zero gradient).
What changes is where it runs: RTG / moments / normalisation / log-prob / the loss head are HIP
kernels over the device trajectory; the MLP forward/backward run through mlp.GemmMLP (forward chain
kernel, backward-data chain kernel, weight-gradient kernel); gradients of all ranks are summed
with ONE flat all-reduce per optimizer step.
"""
from __future__ import annotations

import copy
import math
import os
from abc import ABC, abstractmethod

import torch

from . import distributed as D
from . import hip_ops as K
from . import mlp as M
from . import optim as O
from .rollout import DeviceTrajectory


# 0: always a no-grad pass of the old policy for the old log-probabilities (for callers that write weights through `.data` between
# learn() calls and do not want the fold's bitwise check to stop them)
_FOLD_OLD_LOGP = os.environ.get("TG_FOLD_OLD_LOGP", "1") == "1"
_SMALL_N_RETURNS = 16384                                           # envs up to which tg_returns_moments replaces tg_rtg_scan + tg_masked_moments


class Algorithm(ABC):
    """algorithms/algorithm.py:3-35."""

    def __init__(self):
        pass

    @abstractmethod
    def learn(self, buffer):
        pass

    @abstractmethod
    def metadata(self):
        return {}

    @abstractmethod
    def save(self, path: str):
        pass

    @abstractmethod
    def load(self, path: str):
        pass


def device_trajectory(buffer, device) -> DeviceTrajectory:
    """The buffer's device trajectory; reference-layout CPU tensors (a legacy manager) are uploaded."""
    traj = getattr(buffer, "device_traj", None)
    if traj is not None:
        return traj
    obs, act = buffer.group_observations, buffer.group_actions
    rew, mask = buffer.group_rewards, buffer.group_masks
    G, E, T, S = obs.shape
    A = act.shape[-1]
    n = G * E
    tr = DeviceTrajectory(S, A, T, n, G, E, torch.float32, device)
    tr.obs[:, :T, :].copy_(obs.reshape(n, T, S).permute(2, 1, 0))
    tr.act.copy_(act.reshape(n, T, A).permute(2, 1, 0))
    tr.rew.copy_(rew.reshape(n, T).t())
    tr.mask.copy_(mask.reshape(n, T).t().to(torch.uint8))
    tr.len.copy_(mask.reshape(n, T).sum(1).to(torch.int32))
    return tr


class _GpuLearner(Algorithm):
    # rows per forward/backward chunk (~8 KiB of activations, masks and dZ per row and net at 256x5 bf16).  One chunk for
    # everything was measured up to 3 % faster and, when the row count grows from one iteration to the next, up to 2x
    # slower (GB-sized blocks outgrow the caching allocator every time); a fixed first chunk keeps the big blocks reusable.
    chunk_rows = 1 << 22

    def _setup(self, policy, optimizer, chunk_rows, autocast_dtype, process_group, fused_mlp=True):
        self.policy, self.optimizer = policy, optimizer
        if chunk_rows is not None:
            self.chunk_rows = int(chunk_rows)
        elif os.environ.get("TG_CHUNK_ROWS"):
            self.chunk_rows = int(os.environ["TG_CHUNK_ROWS"])
        self.autocast_dtype = autocast_dtype
        self.process_group = process_group
        self.fused_mlp = fused_mlp
        self._bucket = None
        self._fused_adam = None
        self._adam_covers_bucket = False
        self._refresher = None
        self._mlps = {}
        self._ws = M._Workspace()       # per-iteration tensors whose size follows the number of valid rows
        self._stats, self._stats_pending = {}, None

    def learn(self, buffer) -> None:
        """One training iteration on `buffer` (algorithms/grpo.py:50-148, ppo.py:64-186).  The native kernels are
        launched through ctypes on the policy's device: make it the current one for the duration."""
        with torch.cuda.device(self.policy.device):
            # the fused fp32 rollout's weight stream of THIS policy's actor, if the buffer's manager has one: rebuilt by the launch
            # that follows every optimizer step, so the next rollout starts without a refresh of its own
            frag = getattr(getattr(getattr(buffer, "rollout_manager", None), "engine", None), "_frag", None)
            ok = hasattr(frag, "segments") and getattr(frag, "lin", None) == [m for m in self.policy.actor.network if isinstance(m, torch.nn.Linear)]
            self._rollout_stream = frag if ok else None
            # every per-row workspace (activations, dZ, mask bits, gathered rows) is allocated ONCE for the largest chunk an
            # iteration of this buffer can bring: 288 GB of HBM are there to be used, and growth by re-allocation stalls the queue
            traj = getattr(buffer, "device_traj", None)
            if traj is not None:
                cap = min(self.chunk_rows, traj.T * traj.n)
                self._ws.default_cap = max(self._ws.default_cap, cap)
                for m in self._mlps.values():
                    if m is not None:
                        m._ws.default_cap = max(m._ws.default_cap, cap)
            self._rollout_engine = getattr(getattr(buffer, "rollout_manager", None), "engine", None) if ok else None
            self._check_deferred()
            self._learn(buffer)

    def _entry_refresh(self, *nets):
        """The derived weight layouts of `nets` at the entry of learn(): rebuilt whatever the version keys say -- a weight written
        through `.data` since the last learn() leaves no trace in them -- by the one gather launch when there is one (the fused
        optimizer step's StreamRefresher), else by marking everything stale.  TG_TRUST_VERSION_KEYS=1: the keys decide, as they do
        between the updates of a learn()."""
        if M.N.TRUST_KEYS:
            return self._refresh(*nets)
        ref = self._refresher
        if ref is not None and ref[0][:len(nets)] == tuple(id(n) for n in nets) and ref[1].run():
            for net in nets:
                m = self._mlp(net)
                if m is not None:
                    m.refresh()
                    m._stale.update(("w", "dx"))           # (what the gather does not cover: rebuilt lazily, only if a path reads it)
            return
        for net in nets:
            m = self._mlp(net)
            if m is not None:
                m.refresh(force=True)

    def _check_deferred(self):
        """Checks whose answers are copied to the host asynchronously: the mask's own row count against the rollout's statistic, the
        bitwise comparison behind the folded old-policy pass.  Read at the END of the learn() that enqueued them (both were written
        before its first update ran, so the host does not wait) and again at the next entry / statistics read for a learn() that
        raised in between.  The optimizer steps of that learn() have been applied by then: the error says the weights are suspect, it
        does not roll them back."""
        self._check_row_count()
        pend = getattr(self, "_fold_pending", None)
        if pend is not None:
            self._fold_pending = None
            host, ev = pend
            ev.synchronize()
            if int(host[0]):
                raise RuntimeError("policy.actor and old_policy.actor held different weights although nothing had written either through "
                                   "torch since old_policy <- policy: they were modified through `.data` (or raw pointers).  The last "
                                   "learn() took its old log-probabilities from the current policy; set TG_FOLD_OLD_LOGP=0, or follow "
                                   "such a write with an in-place torch operation (p.add_(0)).")

    def _verify_old_is_current(self):
        """Enqueue the bitwise comparison of policy.actor with old_policy.actor (one launch; the flag is read at the next learn() entry
        or when the statistics are): the fold of the old-policy pass relies on version keys, which `.data` writes do not move."""
        a, b = list(self.policy.actor.parameters()), list(self.old_policy.actor.parameters())
        sig = tuple(p.data_ptr() for p in a + b)
        tab = getattr(self, "_differ_table", None)
        if tab is None or tab[0] != sig:
            rows, first = [], 0
            for p, q in zip(a, b):
                assert p.shape == q.shape and p.is_contiguous() and q.is_contiguous() and p.dtype == q.dtype == torch.float32
                rows.append([p.data_ptr(), q.data_ptr(), 0, 0, first])
                first += p.numel()
            dev = a[0].device
            self._differ_table = tab = (sig, torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), first,
                                        torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32).pin_memory())
        _, table, n, total, flag, host = tab
        flag.zero_()
        K.N.check(K.N.load().tg_params_differ(table.data_ptr(), n, total, flag.data_ptr(), K.N.stream_ptr(flag.device)), "tg_params_differ")
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flag.device))
        self._fold_pending = (host, ev)

    @property
    def last_stats(self) -> dict:
        """Loss statistics of the last learn() as Python numbers.  They are read from the device when first asked for, not at the
        end of learn(): the host goes on to enqueue the next rollout while the last updates still run."""
        if self._stats_pending is not None:
            self._stats, self._stats_pending = self._stats_pending(), None
            self._check_deferred()
        return self._stats

    @last_stats.setter
    def last_stats(self, value) -> None:
        self._stats, self._stats_pending = value, None

    @torch.no_grad()
    def _copy_policy_to_old(self) -> None:
        """old_policy <- policy (grpo.py:148, ppo.py:186) as ONE multi-tensor copy when the two state dicts line up (they are
        deep copies of each other), else through load_state_dict."""
        def leaves(d):                                                      # (the actor-critic's state dict nests one per net)
            out = []
            for v in d.values():
                out += leaves(v) if isinstance(v, dict) else [v]
            return out

        src, dst = leaves(self.policy.state_dict()), leaves(self.old_policy.state_dict())
        if len(src) == len(dst) and all(torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype
                                        and a.device == b.device for a, b in zip(src, dst)):
            # ... and with them the weight streams already built from these weights (the launch after the last optimizer step):
            # the old policy's forward pass of the next learn() then needs no rebuild of its own
            marks = []
            for name in ("actor", "critic"):
                new, old = getattr(self.policy, name, None), getattr(self.old_policy, name, None)
                mn, mo = (self._mlps.get(id(new)), self._mlps.get(id(old))) if new is not None and old is not None else (None, None)
                if mn is None or mo is None:
                    continue
                k = mn._key()
                for what, attr, fields in (("f32", "_f32", ("stream",)), ("chain", "_chain", ("stream", "bias"))):
                    sn, so = getattr(mn, attr, None), getattr(mo, attr, None)
                    if sn is None or so is None or what in mn._stale or mn._built.get(what) != k:
                        continue
                    ts = [(getattr(sn, f_), getattr(so, f_)) for f_ in fields]
                    if all(a.shape == b.shape and a.dtype == b.dtype for a, b in ts):
                        src += [a for a, _ in ts]
                        dst += [b for _, b in ts]
                        marks.append((mo, what))
            torch._foreach_copy_(dst, src)
            for mo, what in marks:
                mo.mark_built(what)                                         # (keyed on the old net's weights AFTER the copy)
        else:
            self.old_policy.load_state_dict(self.policy.state_dict())
        self._old_synced = self._actor_keys()                               # old_policy.actor == policy.actor as long as both keys stand

    def _actor_keys(self):
        """(key of policy.actor's parameters, key of old_policy.actor's): storage, torch version counters, raw-write count."""
        def key(net):
            return (tuple((p.data_ptr(), p._version) for p in net.parameters()), M.N.RAW_PARAM_WRITES[0])
        return key(self.policy.actor), key(self.old_policy.actor)

    def _old_actor_is_current(self) -> bool:
        """Nothing has written either actor since old_policy <- policy (grpo.py:148): their weights are the same bits, and so are
        their log-probabilities -- the first update's own forward pass can stand in for the old policy's."""
        return getattr(self, "_old_synced", None) is not None and self._old_synced == self._actor_keys()

    def sync_old_policy(self) -> None:
        """old_policy <- policy.  The constructors deep-copy the policy BEFORE a checkpoint is loaded into it
        (pipelines/pipeline.py:93-100 loads after construction), so a resume must re-synchronise the copy -- GRPO's
        first learn() would otherwise form its ratios against the random-init weights."""
        self.old_policy.load_state_dict(self.policy.state_dict())
        self._old_synced = self._actor_keys()

    @property
    def bucket(self) -> D.GradBucket:
        if self._bucket is None:
            self._bucket = D.GradBucket(list(self.policy.parameters()))
        return self._bucket

    # ---- MLP execution: hand-scheduled GEMM path (mlp.py) when the net is a ReLU MLP, else autograd ----
    def _mlp(self, net):
        key = id(net)
        if key not in self._mlps:
            ok = self.fused_mlp and M.supports(net) and next(net.parameters()).is_cuda
            self._mlps[key] = M.GemmMLP(net, self.autocast_dtype or torch.float32) if ok else None
            if self._mlps[key] is not None:
                self._mlps[key]._ws.default_cap = self._ws.default_cap
        return self._mlps[key]

    def _refresh(self, *nets):
        for net in nets:
            m = self._mlp(net)
            if m is not None:
                m.refresh()

    def _zero_grads(self):
        """`optimizer.zero_grad()` (grpo.py:143, ppo.py:181) on the flat bucket -- skipped when the previous update's Adam launch
        already left every gradient zero (FusedAdam.step(zero_grads=True))."""
        fa = self._fused_adam
        if fa and fa.grads_zeroed:
            fa.grads_zeroed = False
            return
        self.bucket.zero_()

    def _optimizer_step(self, *nets, last=True):
        """`optimizer.step()` (grpo.py:145, ppo.py:183) and the refresh of every weight layout derived from `nets`.  A plain default
        torch.optim.Adam takes ONE launch on its own state tensors (optim.FusedAdam: bit-identical to torch's ~8) and one gather
        rebuilds all layouts; anything else -- hooks, a patched step, another optimizer -- runs as written, layouts refreshed lazily.
        last=False: another update of this learn() follows -- the Adam launch also zeroes the gradients it consumed (the final
        update's gradients stay in .grad, as after the reference's learn())."""
        refresher = self._optimizer_setup(*nets)
        stepped = bool(self._fused_adam) and self._fused_adam.step(zero_grads=not last and self._adam_covers_bucket, refresher=refresher)
        if not stepped:
            self.optimizer.step()
        self._refresh(*nets)
        if stepped and not self._fused_adam.pushed:
            refresher.run()

    def _optimizer_setup(self, *nets):
        """The fused optimizer step and the refresher of `nets`' derived layouts (None without a fused step), created on first use."""
        if self._fused_adam is None:
            self._fused_adam = O.FusedAdam(self.optimizer)
            if self._fused_adam:
                owned = {id(p) for g in self.optimizer.param_groups for p in g["params"]}
                self._adam_covers_bucket = all(id(p) in owned for p in self.bucket.params)
        if not self._fused_adam:
            return None
        extra = [x for x in (getattr(self, "_rollout_stream", None),) if x is not None]
        key = tuple(id(n) for n in nets) + tuple(id(x) for x in extra)
        if self._refresher is None or self._refresher[0] != key:
            self._refresher = (key, O.StreamRefresher(self._fused_adam, [self._mlp(n) for n in nets], extra))
            eng = getattr(self, "_rollout_engine", None)
            if extra and eng is not None:
                eng.entry_refresh = self._refresher[1].run          # the rollout's own entry rebuild: this one gather
        return self._refresher[1]

    def _adam_rider(self, net, last, whole_update):
        """`optimizer.step()` (grpo.py:145) as a rider of the backward pass's last launch, where nothing stands between the gradients
        and the step: the fp32 chain learner, one rank (no all-reduce), the update's rows in ONE chunk, and an optimizer that holds
        exactly this net's parameters.  None otherwise -- the caller then all-reduces and calls _optimizer_step() as before."""
        m = self._mlp(net)
        if not (whole_update and m is not None and m._f32 is not None) or D.rank_world(self.process_group)[1] != 1 or D._ALWAYS:
            return None                     # (TG_COLLECTIVES_AT_WORLD_1=1: the gradient all-reduce is wanted even at one rank)
        refresher = self._optimizer_setup(net)
        if refresher is None or not self._adam_covers_bucket:
            return None
        owned = {id(p) for g in self.optimizer.param_groups for p in g["params"]}
        if owned != {id(p) for p in net.parameters()}:
            return None
        return self._fused_adam.rider(zero_grads=not last, refresher=refresher)

    def _prep(self, net, X, cap_rows=0):
        m = self._mlp(net)
        if m is None:
            return X
        return m.prepare_input(X, out=self._ws.get("xin", X.shape[0], m.in_pad, m.cd, X.device, cap_rows))

    def _forward(self, net, x, train=False, view=False):
        """fp32 output [rows][out].  train=True keeps what backward needs (activations or the autograd graph).
        view=True: may return a unit-column-stride view of the padded output (row stride > out) instead of a copy."""
        m = self._mlp(net)
        if m is not None:
            if view:
                return m.forward(x, keep=train, padded=True)[:, :m.out_dim]
            return m.forward(x, keep=train)
        with torch.set_grad_enabled(train):
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    y = net(x)
                return y.float()
            return net(x)

    def _backward(self, net, out, grad):
        m = self._mlp(net)
        if m is not None:
            m.backward(grad)
        else:
            out.backward(grad)

    def _gather_valid(self, traj):
        """Indices of valid (t, n) rows (time-major) and the gathered observations / actions."""
        flat = traj.mask.reshape(-1)
        if traj.host_valid_rows is not None and hasattr(torch, "nonzero_static"):
            # the count is on the host already (it rode on the rollout's statistics): no host-device round trip for the shape
            idx = torch.nonzero_static(flat, size=int(traj.host_valid_rows())).squeeze(1)
        else:
            idx = flat.nonzero().squeeze(1)
        rows_all, cap = traj.obs_rows(), traj.T * traj.n
        if rows_all.dtype == torch.float32:
            X = torch.index_select(rows_all, 0, idx, out=self._ws.get("X", idx.numel(), traj.S, torch.float32, idx.device, cap))
        else:
            X = rows_all.index_select(0, idx).float().contiguous()
        acts_all = traj.act_rows()
        act = torch.index_select(acts_all, 0, idx, out=self._ws.get("act", idx.numel(), traj.A, acts_all.dtype, idx.device, cap))
        return idx, X, act

    # ---- the prologue as four launches (csrc/learn_kernels.hip) ------------------------------------------------------
    def _check_row_count(self):
        """The flag of the previous learn()'s tg_learn_count, read long after it was written: the valid rows the mask held were not
        the number the rollout's statistic gave the host (a mask edited after sample(), a hand-built trajectory)."""
        pend = getattr(self, "_count_pending", None)
        if pend is None:
            return
        host, ev, expected = pend
        self._count_pending = None
        ev.synchronize()
        total = int(host[0])
        if total != expected:
            raise RuntimeError(f"the trajectory's mask held {total} valid rows, the rollout's statistic said {expected}: the last learn() "
                               "ran on truncated / padded rows (was the mask edited after sample()?)")

    def _prepare(self, traj, m, src0=None, moments=None, norm_mode=0, group_size=0, src1=None):
        """What `obs[mask]`, `act[mask]`, `adv[mask]` (algorithms/ppo.py:126-135, grpo.py:76-112) and GemmMLP.prepare_input() produce,
        straight from the device trajectory into the learner's workspaces: (idx int64 [rows], xin [rows][in_pad] compute dtype with the
        ones column, act [rows][A], src0's valid entries (normalised with `moments` when given), src1's valid entries).  None when
        this net has no GemmMLP or the trajectory's dtype is not f32 / f64 (the torch path then does it).
        = _prepare_finish(_prepare_enqueue(...)): the first half only enqueues, the second half is where the host waits for the row
        count -- whatever host work does not need the count belongs between the two."""
        return self._prepare_finish(self._prepare_enqueue(traj, m, src0, moments, norm_mode, group_size, src1))

    def _prepare_enqueue(self, traj, m, src0=None, moments=None, norm_mode=0, group_size=0, src1=None):
        if m is None or traj.obs.dtype not in (torch.float32, torch.float64) or m.in_pad > 64 or traj.S > m.in_pad:
            return None
        if m.in_pad % (8 if m.cd == torch.bfloat16 else 4) or m.cd not in (torch.bfloat16, torch.float32):
            return None
        dev, cap = traj.mask.device, traj.T * traj.n
        work = self._ws.get("cnt_work", (K.learn_count_workspace(cap) + 3) // 4, 1, torch.int32, dev)
        total = torch.empty(2, dtype=torch.int64, device=dev)
        # Everything is enqueued BEFORE the host asks for the row count: the kernels take the buffers' capacity, not the count, so
        # they queue up behind the rollout while the host is still waiting for the rollout's statistic -- and run while it prepares
        # the first forward launch (asked first, the count cost the GPU ~30 us of idle time per step at 4,096 envs).
        K.learn_count(traj.mask, -1, work, total)
        idx_c = self._ws.get("idx", cap, 1, torch.int64, dev, cap).view(-1)
        xin_c = self._ws.get("xin", cap, m.in_pad, m.cd, dev, cap)
        act_c = self._ws.get("act", cap, traj.A, torch.float32, dev, cap)
        d0_c = self._ws.get("row0", cap, 1, torch.float32, dev, cap).view(-1) if src0 is not None else None
        d1_c = self._ws.get("row1", cap, 1, torch.float32, dev, cap).view(-1) if src1 is not None else None
        ones = 31 if (m.in_pad == 32 and m.in_dim < 32 and m._f32 is None) else -1
        K.learn_compact(traj, work, cap, xin_c, ones, act_c, idx_c, src0, d0_c, src1, d1_c, moments, norm_mode, group_size)
        return traj, total, idx_c, xin_c, act_c, d0_c, d1_c, ones

    def _prepare_finish(self, handle):
        if handle is None:
            return None
        traj, total, idx_c, xin_c, act_c, d0_c, d1_c, ones = handle
        dev = total.device
        if traj.host_valid_rows is not None:
            # the rollout's own statistic (on the host without a round trip); the count of the mask itself follows asynchronously and is
            # compared with it at the next learn() entry
            self._check_row_count()
            if getattr(self, "_count_pinned", None) is None:
                self._count_pinned = torch.empty(2, dtype=torch.int64).pin_memory()
            self._count_pinned.copy_(total, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            rows = int(traj.host_valid_rows())
            self._count_pending = (self._count_pinned, ev, rows)
        else:
            rows = int(total[0].item())
        idx, xin, act = idx_c[:rows], xin_c[:rows], act_c[:rows]
        d0 = d0_c[:rows] if d0_c is not None else None
        d1 = d1_c[:rows] if d1_c is not None else None
        M.set_ones_column(xin, ones >= 0)
        return idx, xin, act, d0, d1

    def _small(self, name, numel, dtype, device):
        """A cached device buffer whose size does not follow the row count (the row-sized workspace rounds up to a whole chunk)."""
        cache = self.__dict__.setdefault("_small_bufs", {})
        t = cache.get(name)
        if t is None or t.numel() != numel or t.dtype != dtype or t.device != device:
            cache[name] = t = torch.empty(numel, dtype=dtype, device=device)
        return t

    def _logp_nograd(self, actor, xin, act, var):
        out = torch.empty(xin.shape[0], dtype=torch.float32, device=xin.device)
        for lo in range(0, xin.shape[0], self.chunk_rows):
            hi = min(lo + self.chunk_rows, xin.shape[0])
            mean = self._forward(actor, xin[lo:hi], view=True)
            K.gaussian_logp(mean, act[lo:hi], var, out=out[lo:hi])
        return out


class GRPO(_GpuLearner):
    """Group Relative Policy Optimization.  algorithms/grpo.py:12-169."""

    def __init__(self, epsilon: float, beta: float, gamma: float, policy, optimizer, ref_model=None,
                 updates_per_iter: int = 10, *, maximize: bool = False, chunk_rows=None, autocast_dtype=None,
                 process_group=None, fused_mlp: bool = True):
        self.epsilon, self.beta, self.gamma = epsilon, beta, gamma
        self.ref_model = ref_model
        self.updates_per_iter = updates_per_iter
        self.maximize = maximize
        self._setup(policy, optimizer, chunk_rows, autocast_dtype, process_group, fused_mlp)
        self.old_policy = copy.deepcopy(self.policy)                        # grpo.py:48
        self._old_synced = self._actor_keys()

    def _learn(self, buffer) -> None:
        if self.ref_model is not None:
            raise NotImplementedError("the reference's ref_model branch mis-unpacks a 3-tuple (grpo.py:129-132) "
                                      "and never runs in shipped code; it is not reproduced")
        traj = device_trajectory(buffer, self.policy.device)
        var = self.policy.var
        rew = traj.rew if traj.rew.dtype == torch.float32 else traj.rew.float()
        if traj.n <= _SMALL_N_RETURNS and traj.T <= K.returns_moments_max_horizon():       # grpo.py:66-74; per group, :110-115
            rtg, moments = K.returns_moments(rew, traj.mask, self.gamma, traj.E)
        else:
            rtg = K.rtg_scan(rew, traj.mask, self.gamma)
            moments = K.masked_moments(rtg, traj.mask, traj.E)
        actor = self.policy.actor
        m_actor = self._mlp(actor)
        # the valid rows: index, padded input row, action and group-relative advantage (grpo.py:76-115) in one pass over the mask --
        # enqueued here; the host asks for the row count (and waits for the rollout's statistic) only after everything that does not
        # depend on it has been enqueued too
        handle = self._prepare_enqueue(traj, m_actor, src0=rtg, moments=moments, norm_mode=0, group_size=traj.E)
        _, world = D.rank_world(self.process_group)
        G_global = traj.G * world
        coef = (-1.0 if self.maximize else 1.0) / G_global                  # J /= group_size, descent on J
        self._entry_refresh(actor)
        self._refresh(self.old_policy.actor)
        _ = self.bucket                                                     # (the gradient windows exist before can_fuse_head() asks)
        # grpo.py:118-119.  When old_policy still IS the policy (the usual case: grpo.py:148 copied it at the end of the last learn()
        # and nothing has touched either since), its log-probabilities are the ones the first update's forward pass computes
        # anyway: that pass writes them (ratio exactly 1 there, as in the reference) and the no-grad pass is not run.
        fold_old = (_FOLD_OLD_LOGP and self.updates_per_iter > 0 and m_actor is not None and m_actor.can_write_old_logp()
                    and self._old_actor_is_current())
        if fold_old and not M.N.TRUST_KEYS:
            self._verify_old_is_current()
        all_sums = torch.zeros(max(self.updates_per_iter, 1), 4, dtype=torch.float64, device=traj.mask.device)
        if self.updates_per_iter > 0:
            self._zero_grads()                                              # (the first update's, ahead of the wait below)
        prepared = self._prepare_finish(handle)
        if prepared is not None:
            idx, xin, act, adv, _ = prepared
        else:
            adv_full = K.group_normalize(rtg, traj.mask, moments, 0, traj.E)
            idx, X, act = self._gather_valid(traj)
            adv = adv_full.reshape(-1).index_select(0, idx)
            xin = self._prep(actor, X, traj.T * traj.n)
        X = xin                                                             # (the loops below only ask for its row count and device)
        old_logp = (self._ws.get("old_logp", X.shape[0], 1, torch.float32, X.device, traj.T * traj.n).view(-1) if fold_old else
                    self._logp_nograd(self.old_policy.actor, xin, act, var))
        for u in range(self.updates_per_iter):
            if u > 0:
                self._zero_grads()
            sums = all_sums[u]
            fuse = m_actor is not None and m_actor.can_fuse_head()
            last = u == self.updates_per_iter - 1
            rider = None
            for lo in range(0, X.shape[0], self.chunk_rows):
                hi = min(lo + self.chunk_rows, X.shape[0])
                if fuse:        # loss head + head gradient inside the forward chain (tg_mlp_forward_chain_loss)
                    m_actor.forward_loss(xin[lo:hi], 0, act=act[lo:hi], logp_old=old_logp[lo:hi], adv=adv[lo:hi], var=var,
                                         epsilon=self.epsilon, surr_coef=coef, sums_out=sums,
                                         logp_old_out=old_logp[lo:hi] if (fold_old and u == 0) else None)
                    # (asked for right before the launch it rides on: it marks the weight layouts as current)
                    rider = self._adam_rider(actor, last, whole_update=lo == 0 and hi == X.shape[0])
                    m_actor.backward_fused(adam=rider)
                    continue
                else:
                    mean = self._forward(actor, xin[lo:hi], train=True, view=True)     # the loss kernel takes a row stride
                    _, s, g_mean, _ = K.surrogate_loss(mean.detach(), None, act[lo:hi], old_logp[lo:hi], adv[lo:hi], None,
                                                       None, None, var, self.epsilon, coef, 0.0, 0.0, want_total=False)
                    self._backward(actor, mean, g_mean)
                sums += s
            if rider is not None:                                            # (one rank: no all-reduce; the step rode on the reduction)
                self._refresh(actor)                                         # (what the rider does not write -- "w", "dx" -- is stale now)
                continue
            self.bucket.allreduce(self.process_group)                        # one RCCL all-reduce / step
            self._optimizer_step(actor, last=last)
        self._check_deferred()                                              # (this learn()'s own row count / fold flag: landed long ago)
        self._copy_policy_to_old()                                          # grpo.py:148
        if self.updates_per_iter > 0:
            allJ = all_sums
            D.allreduce_sum_(allJ, self.process_group, "loss_stats")
            self._stats_pending = lambda: {"J": (allJ[:, 0] / G_global).tolist(), "n_valid": allJ[0, 3].item()}

    def save(self, path: str) -> None:
        torch.save(self.optimizer.state_dict(), os.path.join(path, "optimizer.pth"))   # grpo.py:154

    def load(self, path: str) -> None:
        self.optimizer.load_state_dict(torch.load(os.path.join(path, "optimizer.pth"), weights_only=True))

    def metadata(self):
        return {"algorithm": "GRPO", "epsilon": self.epsilon, "beta": self.beta,
                "updates_per_iter": self.updates_per_iter}


class PPO(_GpuLearner):
    """Proximal Policy Optimization.  algorithms/ppo.py:8-225."""

    def __init__(self, epsilon: float, policy, optimizer, ref_model, updates_per_iter: int, c1: float = 0.5,
                 kl_coeff: float = 0.5, gamma: float = 0.99, lam: float = 0.95, entropy: float = 0.01,
                 batch_size: int = 64, monte_carlo: bool = True, *, chunk_rows=None, autocast_dtype=None,
                 process_group=None, seed: int = 0, fused_mlp: bool = True):
        self.epsilon, self.c1, self.ref_model = epsilon, c1, ref_model
        self.updates_per_iter = updates_per_iter
        self.gamma, self.lam, self.entropy = gamma, lam, entropy
        self.batch_size, self.kl_coeff, self.monte_carlo = batch_size, kl_coeff, monte_carlo
        self._setup(policy, optimizer, chunk_rows, autocast_dtype, process_group, fused_mlp)
        self.old_policy = copy.deepcopy(self.policy)                        # ppo.py:62 (never read in learn)
        self._seed = seed
        self._gen = None
        # minibatch mode: callable (n_rows, device) -> int64 permutation of this rank's valid rows (time-major order);
        # None = torch.randperm on the device from `seed` (the reference draws torch.randperm on the CPU, ppo.py:148)
        self.permutation_fn = None

    def _step(self, xin, act, adv, ret, old_logp, norm8, var, sums_out, host=None, last=True, write_old=False):
        """One optimizer step on the given rows (all local rows, or one minibatch).  norm8: the device f32 [8] of tg_ppo_norm -- the
        normalisation constants of ppo.py:138-139 and the 1 / n of :165-179, read by the loss heads on the device.  host: minibatch
        mode only -- (the four normalisation constants as a list, this step's global row count): the 1 / n of a minibatch is its own.
        write_old: this is the first step of a full-batch learn() on a chain learner -- its forward pass WRITES `old_logp` (ppo.py:142-143
        takes the old log-probabilities from the current policy: the same numbers) instead of reading it."""
        actor, critic = self.policy.actor, self.policy.critic
        self._zero_grads()
        # [actor | critic] loss sums: a row of the learn()'s pre-zeroed table when there is one (full batch: one fill per learn(), not per update)
        both = self._sum_rows.pop() if getattr(self, "_sum_rows", None) else torch.zeros(2, 4, dtype=torch.float64, device=xin.device)
        sums = both[0]
        m_a, m_c = self._mlp(actor), self._mlp(critic)
        fuse = m_a is not None and m_c is not None and m_a.can_fuse_head() and m_c.can_fuse_head()
        if host is None:
            dev8, nh, coefs = norm8, (None,) * 4, (0.0, 0.0, 0.0)
        else:
            dev8, nh, n_global = None, host[0], host[1]
            coefs = (-1.0 / n_global, self.c1 / n_global, self.kl_coeff / n_global)
        for lo in range(0, xin.shape[0], self.chunk_rows):
            hi = min(lo + self.chunk_rows, xin.shape[0])
            if fuse:            # both loss heads + head gradients inside the forward chains (tg_mlp_forward_chain_loss)
                m_a.forward_loss(xin[lo:hi], 0, act=act[lo:hi], logp_old=old_logp[lo:hi], adv=adv[lo:hi], norm=nh[0:2] if host else None,
                                 var=var, epsilon=self.epsilon, surr_coef=coefs[0], kl_coef=coefs[2], sums_out=both[0],
                                 logp_old_out=old_logp[lo:hi] if write_old else None, norm8=dev8)
                m_a.backward_fused()
                m_c.forward_loss(xin[lo:hi], 1, ret=ret[lo:hi], norm=nh[2:4] if host else None, critic_coef=coefs[1], sums_out=both[1],
                                 norm8=dev8)
                m_c.backward_fused()
                continue
            mean = self._forward(actor, xin[lo:hi], train=True, view=True)         # the loss kernel takes a row stride
            vout = self._forward(critic, xin[lo:hi], train=True)
            value = vout.reshape(-1).contiguous()
            _, s, g_mean, g_val = K.surrogate_loss(mean.detach(), value.detach(), act[lo:hi], old_logp[lo:hi], adv[lo:hi],
                                                   ret[lo:hi], None, norm8[:4], var, self.epsilon, coefs[0], coefs[1], coefs[2],
                                                   want_total=False, coef=norm8[4:7] if host is None else None)
            self._backward(actor, mean, g_mean)
            self._backward(critic, vout, g_val.view_as(vout))
            sums += s
        self.bucket.allreduce(self.process_group)                            # one RCCL all-reduce / step
        self._optimizer_step(actor, critic, last=last)
        sums_out.append(both)

    def _learn(self, buffer) -> None:
        traj = device_trajectory(buffer, self.policy.device)
        var = self.policy.var
        T, n = traj.T, traj.n
        cap = T * n
        rew = traj.rew if traj.rew.dtype == torch.float32 else traj.rew.float()
        actor, critic = self.policy.actor, self.policy.critic
        m_a, m_c = self._mlp(actor), self._mlp(critic)
        if m_a is not None and m_c is not None and m_a.in_pad != m_c.in_pad:
            m_a.disable_f32_chain()                       # (only one of the two fits the fp32 chain learner: both take the
            m_c.disable_f32_chain()                       #  per-layer path, so that they keep sharing ONE prepared input)
        # the valid rows (ppo.py:126-135): index, padded input row (actor and critic share input width / compute dtype), action --
        # enqueued on the buffers' capacity; the host asks for the row count (and waits for the rollout) only after everything that
        # does not depend on it has been enqueued too
        shared = m_c is not None and m_a is not None and m_c.in_pad == m_a.in_pad and m_c.cd == m_a.cd
        handle = self._prepare_enqueue(traj, m_a) if shared else None
        self._entry_refresh(actor, critic)
        _ = self.bucket                                                     # (the gradient windows exist before can_fuse_head() asks)
        dev = traj.mask.device
        V = self._ws.get("V", cap, 1, torch.float32, dev, cap).view(T, n)
        V.zero_()                                                           # padded entries: V = 0 (they never reach a result)
        adv_full = self._ws.get("adv_full", cap, 1, torch.float32, dev, cap).view(T, n)
        ret_full = self._ws.get("ret_full", cap, 1, torch.float32, dev, cap).view(T, n)
        work = self._small("ppo_work", 6 * n, torch.float64, dev)
        prepared = self._prepare_finish(handle)
        if prepared is not None:
            idx, xin, act, _, _ = prepared
        else:
            idx, X, act = self._gather_valid(traj)
            xin = self._prep(actor, X, cap)
        n_rows = xin.shape[0]
        # ppo.py:93: V of the valid rows (padded rows are masked in both scans), scattered onto the [T][n] grid by the launch that
        # follows each chunk's no-grad pass
        for lo in range(0, n_rows, self.chunk_rows):
            hi = min(lo + self.chunk_rows, n_rows)
            out = m_c.forward(xin[lo:hi], keep=False, padded=True) if m_c is not None else self._forward(critic, xin[lo:hi])
            K.scatter_rows(out, idx[lo:hi], V)
        # ppo.py:100-124 + the masked moments of :138-139 in two launches; the ranks' sums in one all-reduce; the normalisation
        # constants and 1 / n on the device (tg_ppo_norm): nothing of this visits the host
        moments = K.ppo_returns(rew, V, traj.mask, self.gamma, self.lam, self.monte_carlo, adv_full, ret_full, work)
        D.allreduce_sum_(moments, self.process_group, "ppo_moments")
        norm8 = K.ppo_norm(moments, self.c1, self.kl_coeff, out=self._small("norm8", 8, torch.float32, dev))
        self.norm8 = norm8                                                  # (diagnostics: this learn()'s constants, on the device)
        adv = self._ws.get("row0", n_rows, 1, torch.float32, dev, cap).view(-1)
        ret = self._ws.get("row1", n_rows, 1, torch.float32, dev, cap).view(-1)
        K.gather_rows2(idx, adv_full, adv, ret_full, ret)
        # ppo.py:142-143: the old log-probabilities come from the CURRENT policy -- on a chain learner the first full-batch
        # update's own forward pass writes them (ratio exactly 1 there, as in the reference), no no-grad pass
        fold_old = (_FOLD_OLD_LOGP and self.batch_size is None and self.updates_per_iter > 0 and m_a is not None and m_c is not None
                    and m_a.can_write_old_logp() and m_c.can_fuse_head())
        old_logp = (self._ws.get("old_logp", n_rows, 1, torch.float32, dev, cap).view(-1) if fold_old else
                    self._logp_nograd(actor, xin, act, var))
        all_sums = []
        # full batch: the updates' [actor | critic] loss sums are the rows of ONE pre-zeroed table (update u <- row u): one fill, one
        # all-reduce, and the arithmetic that turns them into losses waits until somebody asks (last_stats)
        table = (torch.zeros(self.updates_per_iter, 2, 4, dtype=torch.float64, device=dev)
                 if self.batch_size is None and self.updates_per_iter > 0 else None)
        self._sum_rows = list(table.unbind(0))[::-1] if table is not None else None
        norm_host = None
        for u in range(self.updates_per_iter):
            final = u == self.updates_per_iter - 1
            if self.batch_size is None:
                # full batch: the reference permutes and takes one "minibatch" of everything (ppo.py:147-150)
                self._step(xin, act, adv, ret, old_logp, norm8, var, all_sums, last=final, write_old=fold_old and u == 0)
            else:
                if self.permutation_fn is not None:
                    perm = self.permutation_fn(n_rows, dev)
                else:
                    if self._gen is None:
                        self._gen = torch.Generator(device=dev)
                        self._gen.manual_seed(self._seed)
                    perm = torch.randperm(n_rows, device=dev, generator=self._gen)
                # every rank takes the same number of optimizer steps (each one is a collective); a rank that has run
                # out of rows joins the remaining ones with an empty slice
                local_bs, n_steps, sizes = D.minibatch_schedule(n_rows, self.batch_size, self.process_group, dev)
                if norm_host is None:
                    norm_host = norm8[:4].tolist()           # (a minibatch's 1 / n is its own: host numbers; one read per learn())
                for k in range(n_steps):
                    b = perm[k * local_bs:(k + 1) * local_bs]
                    # (the minibatch's rows are copies: they keep the prepared input's ones column, and say so)
                    self._step(M.inherit_ones_column(xin.index_select(0, b), xin), act.index_select(0, b), adv.index_select(0, b),
                               ret.index_select(0, b), old_logp.index_select(0, b), norm8, var, all_sums,
                               host=(norm_host, float(sizes[k])), last=final and k == n_steps - 1)
        self._check_deferred()                                              # (this learn()'s own row count: landed long ago)
        self._copy_policy_to_old()                                          # ppo.py:186
        if all_sums:
            S2 = table if table is not None else torch.stack(all_sums)      # [steps][actor | critic][4]
            D.allreduce_sum_(S2, self.process_group, "loss_stats")
            ent = 0.5 * act.shape[1] * (1.0 + math.log(2 * math.pi)) + 0.5 * float(torch.log(var).sum())
            c1, ent_c, kl_c = self.c1, self.entropy, self.kl_coeff
            n_dev = moments[0, 0].clone()                                   # (the buffers above are re-used by the next learn())

            def stats():
                S = S2[:, 0].clone()
                S[:, 1] += S2[:, 1, 1]                                      # the critic's squared error
                nn = S[:, 3]
                a_loss, c_loss, kl = -S[:, 0] / nn, S[:, 1] / nn, S[:, 2] / nn
                total = a_loss + c1 * c_loss - ent_c * ent + kl_c * kl
                return {"actor_loss": a_loss.tolist(), "critic_loss": c_loss.tolist(), "kl_div": kl.tolist(),
                        "total_loss": total.tolist(), "entropy": ent, "n_valid": float(n_dev)}
            self._stats_pending = stats

    def metadata(self) -> dict:
        return {"algorithm": "PPO", "epsilon": self.epsilon, "c1": self.c1, "kl_coeff": self.kl_coeff,
                "gamma": self.gamma, "lam": self.lam, "entropy": self.entropy, "batch_size": self.batch_size,
                "updates_per_iter": self.updates_per_iter}

    def save(self, path: str) -> None:
        torch.save(self.optimizer.state_dict(), os.path.join(path, "optimizer.pt"))    # ppo.py:214

    def load(self, path: str) -> None:
        self.optimizer.load_state_dict(torch.load(os.path.join(path, "optimizer.pt"), weights_only=True))
