"""One process per GPU: shard whole groups across ranks, one flattened gradient all-reduce per
optimizer step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code; its only parallelism is the rollout
process pool (rollout/rollout_manager.py:44-56).  Envs are independent, so the rollout needs
no collective; the learner needs (a) the summed gradient, (b) PPO's global advantage / return
moments (algorithms/ppo.py:138-139) and (c) the avg_reward metric (buffers/rollout_buffer.py:70).
All three are reductions of SUMS, never of local means, so results do not depend on the number
of ranks beyond floating-point summation order.
"""
from __future__ import annotations

import os
from typing import Iterable, Tuple

import torch
import torch.distributed as dist


def rank_world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_groups(num_groups: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous whole-group range [lo, hi) owned by `rank` (GRPO statistics stay rank-local)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if num_groups % world != 0:
        raise ValueError(f"num_workers={num_groups} must be divisible by the number of ranks ({world})")
    per = num_groups // world
    return rank * per, (rank + 1) * per


# TG_COLLECTIVES_AT_WORLD_1=1: a one-rank process group still issues every collective (rehearses the RCCL path -- communicator
# set-up, dtypes, stream ordering -- on a single-GPU box; `bench.py` under torchrun with --nproc-per-node 1)
_ALWAYS = os.environ.get("TG_COLLECTIVES_AT_WORLD_1", "0") == "1"


# When set to a list, every collective of the path appends (tag, bytes, start event, end event) -- HIP events on the stream the
# collective is ordered on (bench.py: `collectives` in the JSON line, so that a scaling curve can be read against the time and
# the bytes its all-reduces took).  Tags: "grad" (the flat gradient bucket, one per optimizer step), "ppo_moments", "avg_reward",
# "loss_stats", "minibatch_rows".
COLLECTIVE_LOG = None


def _collective(t: torch.Tensor, group, tag: str) -> None:
    log = COLLECTIVE_LOG
    if log is None or not t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream(t.device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    b.record(torch.cuda.current_stream(t.device))
    log.append((tag, t.numel() * t.element_size(), a, b))


def allreduce_sum_(t: torch.Tensor, group=None, tag: str = "other") -> torch.Tensor:
    _, world = rank_world(group)
    if world > 1 or (_ALWAYS and dist.is_available() and dist.is_initialized()):
        _collective(t, group, tag)
    return t


def minibatch_schedule(m_local: int, batch_size: int, group=None, device=None):
    """Minibatch PPO over sharded rows (algorithms/ppo.py:147-157 permutes the global batch and walks it in steps of
    `batch_size`): every rank walks its own permuted rows in steps of ceil(batch_size / world).  Ranks hold different
    numbers of valid rows (episodes end at different times), but every optimizer step issues collectives, so ALL
    ranks must take the same number of steps: the count is derived from the all-gathered row counts, and a rank that
    has run out of rows takes the remaining steps with an empty slice (zero gradient, still in every all-reduce).
    -> (local_bs, n_steps, global_sizes): global_sizes[k] = rows of ALL ranks in step k (the loss normaliser)."""
    _, world = rank_world(group)
    local_bs = max(1, -(-int(batch_size) // world))
    counts = [int(m_local)]
    if world > 1 or (_ALWAYS and dist.is_available() and dist.is_initialized()):
        t = torch.zeros(world, dtype=torch.int64, device=device)
        t[dist.get_rank(group)] = int(m_local)
        _collective(t, group, "minibatch_rows")                        # an all-gather of one integer per rank
        counts = [int(v) for v in t.tolist()]
    n_steps = max(-(-c // local_bs) for c in counts)
    sizes = [sum(min(max(c - k * local_bs, 0), local_bs) for c in counts) for k in range(n_steps)]
    return local_bs, n_steps, sizes


class GradBucket:
    """All parameter gradients as views into ONE flat buffer, so an optimizer step costs exactly one
    all-reduce (0.2-2.2 MB for the reference's policies: latency-bound, so one bucket, not many)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()
        # optimizers may have replaced .grad (set_to_none); re-attach the views
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def allreduce(self, group=None):
        allreduce_sum_(self.flat, group, "grad")


def unbiased_moments(count: float, s1: float, s2: float) -> Tuple[float, float]:
    """(mean, unbiased std) from (count, sum, sum of squares) -- torch.std's default correction."""
    mean = s1 / count
    var = (s2 - s1 * mean) / (count - 1.0) if count > 1 else float("nan")
    return mean, (max(var, 0.0) ** 0.5 if var == var else var)
