"""The optimizer step and the derived weight layouts as two HIP launches (csrc/optim_kernels.hip).

The reference builds `torch.optim.Adam(policy.parameters(), lr=...)` in its factories (pipelines/*_pipeline_*.py) and calls
`optimizer.step()` once per update (algorithms/grpo.py:145, ppo.py:183).  On the GPU that call is ~8 multi-tensor launches, and
every weight layout the kernels consume (bf16 fragment streams, fp32 chain stream) costs a cat + gather (+ convert) each: ~20 small
launches per update -- a fifth of C2's step.  `FusedAdam.step()` performs the SAME update on the optimizer's OWN state tensors
(so `optimizer.state_dict()` / `optimizer.pt` stay what torch writes) with bit-identical results; `StreamRefresher.run()`
rebuilds all layouts of all nets in one gather.

Nothing here changes semantics: anything other than a plain default `torch.optim.Adam` (hooks, a patched `step`, amsgrad,
weight decay, ...) keeps torch's own `optimizer.step()` and the per-stream refresh.
"""
from __future__ import annotations

import torch

from . import _native as N


class FusedAdam:
    """tg_adam_step on the state of an existing torch.optim.Adam.  `usable()` is re-checked at every step."""

    def __init__(self, optimizer):
        self.opt = optimizer
        self._sig = None
        self._tables = None          # per param group: (device int64 [T][5] table, n_tensors, total elements)
        self.tensor_ids = {}         # id(param) -> (group index, tensor index in the group's table)
        self.grads_zeroed = False    # the last step() left every .grad zero (step(zero_grads=True)): the caller may skip its zero_grad

    @staticmethod
    def _plain_adam(opt) -> bool:
        if type(opt) is not torch.optim.Adam or "step" in opt.__dict__:                  # a patched instance (tests, LR schedulers)
            return False
        if opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks:
            return False
        from torch.optim import optimizer as _o
        if getattr(_o, "_global_optimizer_pre_hooks", None) or getattr(_o, "_global_optimizer_post_hooks", None):
            return False
        return True

    def usable(self) -> bool:
        opt = self.opt
        if not self._plain_adam(opt):
            return False
        for g in opt.param_groups:
            if g.get("amsgrad") or g.get("weight_decay", 0) != 0 or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return False
            if g.get("fused") or g.get("foreach") is False or g.get("decoupled_weight_decay"):
                return False
            if not isinstance(g["lr"], float) or not all(isinstance(b, float) for b in g["betas"]) or 1.0 - g["betas"][0] >= 0.5:
                return False
            if not g["params"] or len(g["params"]) > 64:
                return False
            for p in g["params"]:
                if (p.grad is None or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32
                        or not p.grad.is_contiguous() or p.grad.is_sparse or p.numel() >= 1 << 24):
                    return False
        return True

    def _init_state(self):
        """State as torch.optim.Adam._init_group creates it on the first step (CPU float32 step counter, zero moments)."""
        for g in self.opt.param_groups:
            for p in g["params"]:
                st = self.opt.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _build(self):
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.opt.state[p]["exp_avg"].data_ptr(), self.opt.state[p]["exp_avg_sq"].data_ptr())
                    for g in self.opt.param_groups for p in g["params"])
        if sig == self._sig:
            return
        self._sig, self._tables, self.tensor_ids, self._host_tables = sig, [], {}, []
        for gi, g in enumerate(self.opt.param_groups):
            rows, first = [], 0
            for ti, p in enumerate(g["params"]):
                st = self.opt.state[p]
                assert st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous() and st["exp_avg"].dtype == torch.float32
                rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), first])
                self.tensor_ids[id(p)] = (gi, ti)
                first += p.numel()
            dev = g["params"][0].device
            self._tables.append((torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), first))
            host = (N.AdamTensor * len(rows))()
            for slot, r in zip(host, rows):
                slot.p, slot.g, slot.m, slot.v, slot.first = r
            self._host_tables.append(host)

    def table(self, group: int = 0):
        return self._tables[group]

    def _steps_uniform(self) -> bool:
        for g in self.opt.param_groups:
            steps = {float(self.opt.state[p]["step"]) for p in g["params"]}
            if len(steps) != 1 or not isinstance(self.opt.state[g["params"][0]]["step"], torch.Tensor) or self.opt.state[g["params"][0]]["step"].is_cuda:
                return False
        return True

    @torch.no_grad()
    def rider(self, zero_grads: bool, refresher: "StreamRefresher") -> "N.AdamRider | None":
        """This optimizer step as a RIDER of the fp32 learner's gradient-reduction launch (tg_mlp_f32_weight_grad_adam: the thread that
        completes a gradient element steps its parameter and writes the derived layouts) -- no launch of its own.  Returns the
        filled struct; NOTHING on the host has changed yet: the caller passes it to the launch that follows immediately and, once that
        launch has been accepted, calls `r.commit()` -- step()'s host-side bookkeeping (step counters, freshness marks).  A launch
        that is refused leaves the optimizer where it was.  None when this form does not apply (more than one parameter
        group, layouts not yet gathered once, anything step() itself would refuse): the caller then calls step() after its backward
        pass as before."""
        if len(self.opt.param_groups) != 1 or refresher is None or not self.usable():
            return None
        self._init_state()
        if not self._steps_uniform():
            return None
        self._build()
        push = refresher.push_tables()
        if push is None:
            return None
        g = self.opt.param_groups[0]
        tab, n, total = self._tables[0]
        seg, n_seg, inv_start, inv_dst = push
        r = N.AdamRider()
        r.h_table, r.n_tensors, r.zero_grads, r.total = self._host_tables[0], n, 1 if zero_grads else 0, total
        r.lr, r.beta1, r.beta2, r.eps = g["lr"], g["betas"][0], g["betas"][1], g["eps"]
        r.step = int(self.opt.state[g["params"][0]]["step"]) + 1
        r.d_segments, r.n_segments, r.d_inv_start, r.d_inv_dst = seg.data_ptr(), n_seg, inv_start.data_ptr(), inv_dst.data_ptr()
        r._keep = (self._host_tables[0], seg, inv_start, inv_dst)

        def commit():
            for p in g["params"]:
                self.opt.state[p]["step"] += 1
            N.RAW_PARAM_WRITES[0] += 1
            self.grads_zeroed = bool(zero_grads)
            refresher.mark_all()
            self.pushed = True
        r.commit = commit
        return r

    @torch.no_grad()
    def step(self, zero_grads: bool = False, refresher: "StreamRefresher" = None) -> bool:
        """One optimizer step; False (nothing done) when the fused form does not apply -- the caller then runs optimizer.step().
        zero_grads: the launch also zeroes every .grad it has consumed (the next update's optimizer.zero_grad() -- grpo.py:143,
        ppo.py:181 -- folded in); `grads_zeroed` then tells the caller that its own zeroing launch can be skipped.
        refresher: the derived weight layouts of this optimizer's nets.  Once they have been built by a gather (StreamRefresher.run),
        the step's own launch keeps them current (tg_adam_step_push: the thread that updates a weight writes it into every layout
        position derived from it) and marks them fresh -- `self.pushed` says so; otherwise the caller runs refresher.run()."""
        self.grads_zeroed = False
        self.pushed = False
        if not self.usable():
            return False
        self._init_state()
        if not self._steps_uniform():
            return False
        self._build()
        lib = N.load()
        push = refresher.push_tables() if refresher is not None else None
        for gi, ((tab, n, total), g) in enumerate(zip(self._tables, self.opt.param_groups)):
            for p in g["params"]:
                self.opt.state[p]["step"] += 1
            step = int(self.opt.state[g["params"][0]]["step"])
            dev = g["params"][0].device
            with torch.cuda.device(dev):
                if push is not None and gi == 0:
                    seg, n_seg, inv_start, inv_dst = push
                    N.check(lib.tg_adam_step_push(tab.data_ptr(), n, total, g["lr"], g["betas"][0], g["betas"][1], g["eps"], step,
                                                  1 if zero_grads else 0, seg.data_ptr(), n_seg, inv_start.data_ptr(), inv_dst.data_ptr(),
                                                  N.stream_ptr(dev)), "tg_adam_step_push")
                else:
                    N.check(lib.tg_adam_step(tab.data_ptr(), n, total, g["lr"], g["betas"][0], g["betas"][1], g["eps"], step,
                                             1 if zero_grads else 0, N.stream_ptr(dev)), "tg_adam_step")
        N.RAW_PARAM_WRITES[0] += 1
        self.grads_zeroed = bool(zero_grads)
        if push is not None:
            refresher.mark_all()
            self.pushed = True
        return True


class StreamRefresher:
    """All derived weight layouts of a set of GemmMLPs rebuilt from the fp32 masters in ONE launch (tg_gather_streams).  Built
    against a FusedAdam's tensor table (the masters' pointers); rebuilt when that table changes."""

    def __init__(self, adam: FusedAdam, mlps, extra_streams=()):
        """extra_streams: objects with segments(tensor_ids) -> [(dst, int32 codes, is_bf16)] and mark_fresh() (the fused fp32
        rollout's register stream: the next rollout then starts without a refresh of its own)."""
        self.adam, self.mlps = adam, [m for m in mlps if m is not None]
        self.extra = [x for x in extra_streams if x is not None]
        self._sig = None
        self._seg = None
        self._push = None            # (segment table, n segments, inv_start, inv_dst) once the layouts have been gathered under this signature
        self._gathered_sig = None

    def push_tables(self):
        """The inverse of the gather's codes, for tg_adam_step_push -- or None until the layouts have been built once by run() under
        the current signature (padding positions are only ever written by the gather), or when this launch cannot cover them."""
        try:
            if not self._build() or self._gathered_sig != self._layout_sig():
                return None
        except (KeyError, AssertionError):
            self._seg = None
            return None
        if self._push is None:
            seg, n, total, keep, dev = self._seg
            tab, _, n_elem = self.adam.table(0)
            firsts = tab[:, 4].cpu().tolist()
            if any(c.numel() >= 1 << 26 for c in keep) or n > 32:
                return None
            elem, dst = [], []
            first_of = torch.tensor(firsts, dtype=torch.int64, device=dev)
            for si, code in enumerate(keep):
                c = code.to(torch.int64)
                ok = c >= 0
                j = torch.arange(c.numel(), device=dev, dtype=torch.int64)[ok]
                cc = c[ok]
                elem.append(first_of[cc >> 24] + (cc & 0xFFFFFF))
                dst.append((si << 26) | j)
            elem, dst = torch.cat(elem), torch.cat(dst)
            order = torch.argsort(elem, stable=True)
            counts = torch.bincount(elem, minlength=n_elem)
            inv_start = torch.zeros(n_elem + 1, dtype=torch.int64, device=dev)
            inv_start[1:] = torch.cumsum(counts, 0)
            self._push = (seg, n, inv_start.to(torch.int32).contiguous(), dst[order].to(torch.int32).contiguous())
        return self._push

    def _layout_sig(self):
        """What "the layouts have been gathered" is a statement about: the master tensors and the destination buffers (not the
        gradient or moment tensors, which the optimizer's own table signature also carries)."""
        masters = tuple(t[0] for t in (self.adam._sig or ()))
        return (masters, self._sig[1] if self._sig else None)

    def mark_all(self):
        """Every layout this refresher covers has just been brought up to date (by run(), or by the optimizer step's own launch)."""
        for m, what in self._marks:
            m.mark_built(what)
        for x in self.extra:
            x.mark_fresh()

    def _codes(self, stream_obj, group):
        """int32 code per element of `stream_obj._idx` (FragmentStream / F32ChainStream): master tensor << 24 | offset, -1 = zero."""
        lin = stream_obj.lin
        parts = []
        for l in lin:
            gi, ti = self.adam.tensor_ids[id(l.weight)]
            assert gi == group
            parts.append((ti << 24) + torch.arange(l.weight.numel(), dtype=torch.int64))
        for l in lin:
            gi, ti = self.adam.tensor_ids[id(l.bias)]
            assert gi == group
            parts.append((ti << 24) + torch.arange(l.bias.numel(), dtype=torch.int64))
        parts.append(torch.tensor([-1], dtype=torch.int64))
        src = torch.cat(parts).to(stream_obj._idx.device)
        return src[stream_obj._idx].to(torch.int32)

    def _signature(self):
        """What the cached segment table is a function of: the optimizer's tensor table (the masters' pointers) AND every
        destination buffer -- a stream that was dropped or re-allocated since (GemmMLP.disable_f32_chain, a re-created rollout
        engine) must not be written through its old address."""
        dst = []
        for m in self.mlps:
            for s_ in (m._chain, m._bchain, m._f32):
                dst.append(None if s_ is None else (s_.stream.data_ptr(), getattr(s_, "bias", s_.stream).data_ptr()))
        for x in self.extra:
            dst.append((x.stream.data_ptr(), x.table.data_ptr()))
        return (self.adam._sig, tuple(dst))

    def _build(self):
        sig = self._signature()
        if sig == self._sig:
            return self._seg is not None                  # (an unsupported layout is not re-attempted at every step)
        self._sig, self._seg, self._marks, self._push = sig, None, [], None
        segs, keep, first = [], [], 0
        dev = None
        for m in self.mlps:
            streams = []
            if m._chain is not None:
                c = self._codes(m._chain, 0)
                ns = m._chain._n_stream
                streams += [(m._chain.stream, c[:ns].contiguous(), 1), (m._chain.bias.view(-1), c[ns:].contiguous(), 0)]
                self._marks.append((m, "chain"))
            if m._bchain is not None:
                streams += [(m._bchain.stream, self._codes(m._bchain, 0).contiguous(), 1)]
                self._marks.append((m, "bchain"))
            if m._f32 is not None:
                streams += [(m._f32.stream, self._codes(m._f32, 0).contiguous(), 0)]
                self._marks.append((m, "f32"))
            for dst, code, is_bf16 in streams:
                assert code.numel() == dst.numel() and dst.is_contiguous()
                segs.append([dst.data_ptr(), code.data_ptr(), first, is_bf16])      # (is_bf16 | pad) share one int64: little endian
                keep.append(code)
                first += dst.numel()
                dev = dst.device
        for x in self.extra:
            for dst, code, is_bf16 in x.segments({k: v for k, v in self.adam.tensor_ids.items()}):
                assert code.numel() == dst.numel() and dst.is_contiguous() and code.dtype == torch.int32
                segs.append([dst.data_ptr(), code.data_ptr(), first, is_bf16])
                keep.append(code)
                first += dst.numel()
                dev = dst.device
        if not segs or len(segs) > 32:
            return False
        self._seg = (torch.tensor(segs, dtype=torch.int64).to(dev), len(segs), first, keep, dev)
        return True

    def run(self) -> bool:
        """Rebuild the layouts (and mark them fresh in their GemmMLP); False when there is nothing this launch covers."""
        try:
            if not self._build():
                return False
        except (KeyError, AssertionError):
            self._seg = None                              # a net whose parameters are not in the optimizer's first group:
            return False                                  # remembered under this signature, not rebuilt at every step
        seg, n, total, _, dev = self._seg
        tab, _, _ = self.adam.table(0)
        with torch.cuda.device(dev):
            N.check(N.load().tg_gather_streams(seg.data_ptr(), n, total, tab.data_ptr(), N.stream_ptr(dev)), "tg_gather_streams")
        self._gathered_sig = self._layout_sig()
        self.mark_all()
        return True
