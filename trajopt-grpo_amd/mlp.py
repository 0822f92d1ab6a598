"""Hand-scheduled forward/backward of the reference's ReLU MLP on the MI355X matrix cores.

`models/neural_network.py:48-66` is `Sequential(Linear, ReLU, ..., Linear)`; under torch autograd (the reference,
algorithms/*.py `loss.backward()`) that costs ~12 launches per layer and, at the shapes of this path
(10^6..10^7 rows x 256 features), every pass is bound by HBM traffic (137 GFLOP per GB).  `GemmMLP` therefore
minimises bytes per row:
  * forward: ONE persistent chain kernel (tg_mlp_forward_chain) for bf16 nets Linear(S<=32, H) ReLU [Linear(H, H)
    ReLU]* Linear(H, A<=16), H in {128, 256}: activations stay on chip between layers and are written once
    (2.7 instead of 5.2 KB/row at 20-256x5-4).  Other shapes: per-layer GEMMs with bias + ReLU in the epilogue
    (`torch._addmm_activation`), K = 20 and N = 1 padded to 32 / 8 (slow hipBLASLt paths otherwise);
  * backward data: tg_dx_relu_bias fuses `dA = dZ W` with the ReLU backward and bias gradient of the layer below
    (1.5 instead of 2.5 KB/row) for square bf16 layers of width 64 / 128 / 256; the head's rank-A product is
    formed inside the top layer's pass (tg_head_bwd_relu_bias); otherwise a GEMM + tg_relu_bwd_bias;
  * backward data, chain shapes (bf16, H in {128, 256}, 3..6 hidden layers, <= 8 outputs): tg_mlp_backward_chain, the dZ of
    all hidden layers in one launch (2.7 KB/row at 5 x 256);
  * weight gradients, chain shapes: tg_mlp_weight_grad -- every weight and hidden bias gradient of the net in ONE persistent
    launch (each dZ and activation read once, the first hidden activation recomputed from the input row instead of stored);
    other shapes: dW = dZ^T A as a batched GEMM over row blocks with fp32 partials plus tg_dw_finish (a plain GEMM has 16
    output tiles on 256 CUs: 1.6-1.9 ms per 2^20 rows).
Gradients are accumulated in fp32 straight into `param.grad` (the learner's flat all-reduce bucket).

Only ReLU hidden activations take this path; anything else stays on torch autograd.
"""
from __future__ import annotations

import ctypes as C
import logging

import torch

from . import _native as N

_ROW_BLOCK = 8192


_LOG = logging.getLogger("trajopt_grpo_amd")
_LOGGED_SHAPES = set()

# "This input has a column of ones" is a property of the BYTES some GemmMLP.prepare_input() wrote, shared by the nets that read them
# (the learner's actor and critic take the same prepared input).  It is carried on the tensor's storage object -- torch keeps ONE
# Python object per live storage, so the mark is seen through every slice / view and dies with the allocation (a block the caching
# allocator hands out again is a new storage: no stale mark) -- as the byte range that was prepared.
def _mark_ones_column(xp: torch.Tensor) -> None:
    lo = xp.storage_offset() * xp.element_size()
    xp.untyped_storage()._tg_ones = (lo, lo + xp.numel() * xp.element_size(), xp.shape[1] * xp.element_size())


def set_ones_column(xp: torch.Tensor, has: bool) -> None:
    """`xp` was just written by a prepare kernel other than prepare_input() (tg_learn_compact): record whether it carries the ones column."""
    if has:
        _mark_ones_column(xp)
    elif getattr(xp.untyped_storage(), "_tg_ones", None) is not None:
        del xp.untyped_storage()._tg_ones


def has_ones_column(xin: torch.Tensor) -> bool:
    """`xin` is a contiguous run of whole rows of an input some prepare_input() gave the ones column."""
    if xin.dim() != 2 or not xin.is_contiguous():
        return False
    r = getattr(xin.untyped_storage(), "_tg_ones", None)
    lo = xin.storage_offset() * xin.element_size()
    return (r is not None and xin.shape[1] * xin.element_size() == r[2] and r[0] <= lo and lo + xin.numel() * xin.element_size() <= r[1]
            and (lo - r[0]) % r[2] == 0)


def inherit_ones_column(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """`dst` holds whole rows copied out of `src` (index_select of a minibatch): it has the ones column iff `src` has."""
    if dst.dim() == 2 and dst.is_contiguous() and dst.shape[1:] == src.shape[1:] and dst.dtype == src.dtype and has_ones_column(src):
        _mark_ones_column(dst)
    return dst


def lin_ok(l) -> bool:
    return l.weight.grad is not None and l.weight.grad.dtype == torch.float32 and l.bias.grad is not None
_SPLIT_BATCHES = 128


def weight_grad(hidden: int, jobs, rows: int, workspace: torch.Tensor, w0frag: torch.Tensor = None, b0: torch.Tensor = None,
                whfrag: torch.Tensor = None, extra=(), loss_rider=None):
    """tg_mlp_weight_grad: every job's `wgrad += P^T Q` (and `bgrad += column sums of P`) in one persistent launch + one
    fixed-order reduction.  jobs = [(kind, P, Q, wgrad, bgrad or None[, aux])]; P / Q bf16 row-major [rows][*], wgrad a 2-D
    fp32 view with unit column stride (e.g. a window of the learner's flat gradient bucket).  Kind HR rebuilds its Q (the first
    hidden activation) from the net input with `w0frag` / `b0`; kind RH rebuilds its P (the top layer's dZ) from d loss / d head
    output, the layer's ReLU mask bits (`aux`) and `whfrag` (the backward chain's weight stream)."""
    arr = (N.DwJob * len(jobs))()
    for slot, job in zip(arr, jobs):
        kind, p, q, wgrad, bgrad = job[:5]
        aux = job[5] if len(job) > 5 else None
        N.require_cuda(p, q, wgrad, bgrad, aux)
        assert (kind == N.TG_DW_RH) == (aux is not None)
        slot.d_aux = N.ptr(aux)
        assert p.dtype == torch.bfloat16 and q.dtype == torch.bfloat16 and p.is_contiguous() and q.is_contiguous()
        assert p.shape[0] >= rows and q.shape[0] >= rows
        assert wgrad.dtype == torch.float32 and wgrad.dim() == 2 and wgrad.stride(1) == 1
        assert bgrad is None or (bgrad.dtype == torch.float32 and bgrad.is_contiguous() and bgrad.numel() == wgrad.shape[0])
        slot.d_p, slot.d_q, slot.d_wgrad, slot.d_bgrad = p.data_ptr(), q.data_ptr(), wgrad.data_ptr(), N.ptr(bgrad)
        slot.wgrad_ld, slot.kind, slot.m_out, slot.n_out = wgrad.stride(0), kind, wgrad.shape[0], wgrad.shape[1]
    # extra = [(slab tensor, element offset, slab_stride, n_slabs, row_pitch, grad, m_out, n_out)]: the chain kernels' own partial
    # gradients, reduced by the same second launch (tg_mlp_weight_grad_ex); loss_rider = (f64 work [rows][4], n rows, f64 sums [4])
    ex = (N.SlabSum * max(len(extra), 1))()
    for slot, (slab, off, stride, n_slabs, pitch, grad, m_out, n_out) in zip(ex, extra):
        N.require_cuda(slab, grad)
        assert slab.dtype == grad.dtype == torch.float32 and slab.is_contiguous() and (grad.dim() == 1 or grad.stride(-1) == 1)
        assert off + (n_slabs - 1) * stride + (m_out - 1) * pitch + n_out <= slab.numel() or n_slabs == 0
        slot.d_slab, slot.slab_stride, slot.n_slabs, slot.row_pitch = slab.data_ptr() + 4 * off, stride, n_slabs, pitch
        slot.d_grad, slot.grad_ld, slot.m_out, slot.n_out = grad.data_ptr(), (grad.stride(0) if grad.dim() == 2 else 1), m_out, n_out
    lw, ln, ls = (loss_rider[0].data_ptr(), loss_rider[1], loss_rider[2].data_ptr()) if loss_rider else (None, 0, None)
    N.check(N.load().tg_mlp_weight_grad_ex(hidden, arr, len(jobs), rows, N.ptr(w0frag), N.ptr(b0), N.ptr(whfrag), workspace.data_ptr(),
                                           workspace.numel() * workspace.element_size(), ex, len(extra), lw, ln, ls,
                                           N.stream_ptr(workspace.device)), "tg_mlp_weight_grad_ex")


def weight_grad_workspace(hidden: int, device) -> torch.Tensor:
    return torch.empty(N.load().tg_mlp_weight_grad_workspace(hidden) // 4, dtype=torch.float32, device=device)


def supports(net) -> bool:
    mods = list(net.network)
    if len(mods) < 3 or len(mods) % 2 == 0:
        return False
    for i, m in enumerate(mods):
        if i % 2 == 0 and not isinstance(m, torch.nn.Linear):
            return False
        if i % 2 == 1 and not isinstance(m, torch.nn.ReLU):
            return False
    return all(l.out_features % 8 == 0 for l in mods[0:-1:2])


def _round_up(x, m):
    return (x + m - 1) // m * m


class _Workspace:
    """Grow-only named device buffers for the per-chunk tensors of a forward/backward (activations, mask bits, dZ).
    Their size follows the number of valid rows, which changes every iteration; handed to the caching allocator, a
    GB-sized request that outgrows what is cached costs a hipMalloc (the update was measured up to 2x slower when the
    row count kept growing).  A buffer here is allocated once per name at the largest size seen and handed out as a view;
    with the learner's fixed first chunk that size is reached in the first update."""

    def __init__(self):
        self._buf = {}

    default_cap = 0     # rows every buffer is sized for when it is (re)allocated (the learner sets it to its chunk size: the
                        # row count of an iteration grows while a policy learns to survive, and a GB-sized buffer that is re-allocated
                        # a little larger every iteration costs tens of milliseconds of hipMalloc / hipFree with the GPU busy)

    def get(self, name: str, rows: int, cols: int, dtype, device, cap_rows: int = 0) -> torch.Tensor:
        """[rows][cols] view of buffer `name`; a (re)allocation sizes it for max(rows, cap_rows, default_cap) rows."""
        need = rows * cols
        b = self._buf.get(name)
        if b is None or b.numel() < need or b.dtype != dtype or b.device != device:
            self._buf[name] = b = torch.empty(max(rows, cap_rows, self.default_cap) * cols, dtype=dtype, device=device)
        return b[:need].view(rows, cols)


class GemmMLP:
    def __init__(self, net, compute_dtype=torch.bfloat16):
        assert supports(net)
        self.net = net
        self.cd = compute_dtype
        self.linears = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        dev = self.linears[0].weight.device
        self.in_dim, self.out_dim = self.linears[0].in_features, self.linears[-1].out_features
        # fp32 nets of the reference's own shapes: the fp32 chain learner (tg_mlp_f32_forward / _forward_backward / _weight_grad)
        self._f32 = None
        # (parity tests: every hidden activation and dZ of the fp32 chain learner written to HBM, so that each can be compared with
        # fp64; the product rebuilds the first activation and the top dZ on chip instead)
        self.f32_store_all = False
        if compute_dtype == torch.float32 and f32_res_supported(net):
            self._f32 = F32ResStream(net)                       # H = 128, <= 2 hidden layers: the resident 16-row kernel (C2's shape)
        elif compute_dtype == torch.float32 and f32_chain_supported(net):
            self._f32 = F32ChainStream(net, f32_chain_supported(net))
        elif compute_dtype == torch.float32 and f32_wide_supported(net):
            self._f32 = F32WideStream(net)                      # H = 256: csrc/mlp_f32_wide.hip
        self.in_pad = _round_up(self.in_dim, 32) if self._f32 is None else self._f32.in_pad
        self.out_pad = _round_up(self.out_dim, 8)
        self.w, self.b = [], []
        for i, l in enumerate(self.linears):
            o = self.out_pad if i == len(self.linears) - 1 else l.out_features
            k = self.in_pad if i == 0 else l.in_features
            self.w.append(torch.zeros(o, k, dtype=compute_dtype, device=dev))
            self.b.append(torch.zeros(o, dtype=compute_dtype, device=dev))
        self._acts = None
        self._bits = None
        self._ws = _Workspace()
        self._partial = None
        # hidden layers whose backward-data product runs fused with the ReLU backward below it (tg_dx_relu_bias)
        self._dxfrag = [None] * len(self.linears)
        if compute_dtype == torch.bfloat16:
            lib = N.load()
            for i in range(1, len(self.linears) - 1):
                o, k = self.w[i].shape
                if lib.tg_dx_relu_bias_supported(o, k):
                    self._dxfrag[i] = torch.empty(o * k, dtype=torch.bfloat16, device=dev)
        self._dx_partial = None
        self._head_partial = None
        self._dw_ws = None
        self._w0_slabs = None
        self._head_ws = None
        self._dz_head = None
        self._tmask = None
        # like dx_events, for every tg_mlp_weight_grad launch / every training (keep=True) tg_mlp_forward_chain launch
        self.dw_events = None
        self.fwd_events = None
        # when set to a list, every backward-data launch (tg_mlp_backward_chain or tg_dx_relu_bias) is bracketed by HIP
        # events on the launch stream and (start, end, rows, algorithmic bytes per row, kernel name) is appended
        # (bench.py reads them back for that kernel's roofline)
        self.dx_events = None
        # all layers of the forward pass in one launch (tg_mlp_forward_chain) when the shape allows
        self._chain = None
        H = fused_rollout_supported(net, min(self.in_dim, 32), 1) if compute_dtype == torch.bfloat16 else 0
        self._bchain = None
        if H and self.in_dim <= 32 and self.out_pad <= 16 and len(self.linears) - 1 <= 8:
            self._chain = FragmentStream(net, H, layout="chain")
            # ... and the backward-data pass as one launch too (tg_mlp_backward_chain): dZ stays on chip from the head down
            if H in (128, 256) and self.out_pad == 8 and 3 <= len(self.linears) - 1 <= 6:     # 6: the kernel's LDS budget
                self._bchain = FragmentStream(net, H, layout="chain", transposed=True)
        self.bias_out_f32 = torch.zeros(self.out_pad, dtype=torch.float32, device=dev)
        self._log_path(net)
        self._built = {}            # derived operand -> the key (_key()) of the weights it was last built from
        self.refresh()

    def disable_f32_chain(self):
        """Back to the per-layer path with the 32-wide padded input (a learner whose actor and critic would otherwise expect
        differently padded inputs: e.g. > 4 actions beside a 1-output critic)."""
        if self._f32 is None:
            return
        self._f32 = None
        self.in_pad = _round_up(self.in_dim, 32)
        w0 = self.w[0]
        self.w[0] = torch.zeros(w0.shape[0], self.in_pad, dtype=w0.dtype, device=w0.device)
        self._built.clear()                                         # (a re-allocated operand: nothing built counts any more)
        self.refresh()

    def _log_path(self, net):
        """One INFO line per net shape (logger `trajopt_grpo_amd`) saying which kernels run it, and a WARNING when a net falls off the
        hand-written chain kernels onto library GEMMs + per-layer glue (that used to be silent)."""
        lin = self.linears
        shape = f"{self.in_dim}-" + "-".join(str(l.out_features) for l in lin[:-1]) + f"-{self.out_dim} {str(self.cd).replace('torch.', '')}"
        if shape in _LOGGED_SHAPES:
            return
        _LOGGED_SHAPES.add(shape)
        if self._f32 is not None and self._f32.wide:
            _LOG.info("%s: fp32 chain learner at H = 256 (tg_mlp_f32w_forward / _forward_backward / tg_mlp_f32_weight_grad)", shape)
        elif self._f32 is not None:
            _LOG.info("%s: fp32 chain learner (tg_mlp_f32_forward / _forward_backward / _weight_grad)", shape)
        elif self._chain is not None and self._bchain is not None:
            _LOG.info("%s: bf16 chain kernels (tg_mlp_forward_chain[_loss] / tg_mlp_backward_chain / tg_mlp_weight_grad)", shape)
        else:
            why = ("fp32 chain learners: hidden width 64 / 128 with 1-4 equal hidden layers or 256 with 1-5, <= 32 inputs, <= 4 outputs"
                   if self.cd == torch.float32 else
                   "bf16 chain kernels: hidden width 128 / 256, 3-6 equal hidden layers, <= 32 inputs, <= 8 outputs"
                   + ("; the forward chain alone covers this net" if self._chain is not None else ""))
            log = _LOG.warning if sum(l.weight.numel() for l in lin) >= 4096 else _LOG.info       # (scaffolding-sized nets: not worth a warning)
            log("%s is outside the hand-written learner kernels' shapes (%s): its update runs on hipBLASLt GEMMs + per-layer "
                         "HIP kernels, several times slower per row", shape, why)

    def refresh(self, force: bool = False):
        """The fp32 master weights changed: every derived operand (padded compute-dtype copies, chain streams, packed
        backward-data fragments) is rebuilt the next time the path that reads it runs -- with the chain kernels active
        the per-layer copies are never touched (17 of 36 tiny launches per update and net).  force=True: whatever the keys say
        (an entry point that cannot know who wrote the weights since its last call)."""
        k = self._key()
        self._stale = {w for w in ("w", "chain", "bchain", "dx", "f32") if force or N.ALWAYS_REBUILD or self._built.get(w) != k}

    def fresh_forward(self, force: bool = False):
        """Rebuild NOW what the no-grad forward() reads.  A captured hipGraph holds the forward launches only (a rebuild enqueued
        during the capture's warm-up leaves nothing stale for the capture itself to record), so the per-step rollout calls this
        eagerly in front of every replay: the graph then reads the rebuilt operands through the same buffers."""
        self.refresh(force)
        self._fresh("f32" if self._f32 is not None else ("chain" if self._chain is not None else "w"))

    def _key(self):
        """What every derived operand is a function of: the master tensors' storage and torch version counters, and the count of
        raw-pointer parameter writes (the fused Adam launch bypasses the version counters).  refresh() marks stale only what was
        built from a different key, so a learn() that starts with the weights its last optimizer step left rebuilds nothing."""
        return (tuple((p.data_ptr(), p._version) for l in self.linears for p in (l.weight, l.bias)), N.RAW_PARAM_WRITES[0])

    def mark_built(self, what: str) -> None:
        """`what` has just been rebuilt from the current weights by someone else (optim.StreamRefresher, a copy of an identical
        net's stream)."""
        self._stale.discard(what)
        self._built[what] = self._key()

    def _fresh(self, what: str):
        if what not in self._stale:
            return
        self.mark_built(what)
        with torch.no_grad():
            if what == "w":
                for w, b, l in zip(self.w, self.b, self.linears):
                    w[:l.out_features, :l.in_features].copy_(l.weight)
                    b[:l.out_features].copy_(l.bias)
                self.bias_out_f32[:self.out_dim].copy_(self.linears[-1].bias)
            elif what == "chain":
                self._chain.refresh()
            elif what == "bchain":
                self._bchain.refresh()
            elif what == "f32":
                self._f32.refresh()
            elif what == "dx":
                self._fresh("w")
                for w, frag in zip(self.w, self._dxfrag):
                    if frag is not None:
                        N.check(N.load().tg_dx_pack_weights(w.data_ptr(), frag.data_ptr(), w.shape[0], w.shape[1],
                                                            N.stream_ptr(w.device)), "tg_dx_pack_weights")

    def prepare_input(self, X: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """[M][in_dim] (any float dtype, any strides) -> contiguous [M][in_pad] compute dtype, zero padded
        (into `out` when given: a [M][in_pad] buffer of the compute dtype)."""
        xp = torch.zeros(X.shape[0], self.in_pad, dtype=self.cd, device=X.device) if out is None else out.zero_()
        xp[:, :self.in_dim].copy_(X)
        if self.in_pad == 32 and self.in_dim < 32 and self._f32 is None:
            # a padding column of ones: the first layer's weights are zero there (the forward pass does not see it), and
            # tg_mlp_backward_chain_w0 delivers the first layer's bias gradient as that column of dW0.  The prepared bytes are
            # marked on their storage: only inputs prepared HERE (slices of them, row copies registered with
            # inherit_ones_column) take that path -- a caller that pads its own input with zeros gets the HX job of
            # tg_mlp_weight_grad instead of a silently zero bias gradient.
            xp[:, 31] = 1.0
            _mark_ones_column(xp)
        elif getattr(xp.untyped_storage(), "_tg_ones", None) is not None:
            del xp.untyped_storage()._tg_ones                 # (a buffer re-prepared by a net that writes no ones column)
        return xp

    def _has_ones_column(self, xin: torch.Tensor) -> bool:
        return has_ones_column(xin)

    def _chain_backward_ok(self) -> bool:
        """backward() can take the chain path (tg_mlp_backward_chain + tg_mlp_weight_grad): fp32 gradients with unit column
        stride on every layer.  forward(keep=True) decides with the SAME predicate whether the first activation may be left out."""
        return self._bchain is not None and all(
            l.weight.grad is not None and l.bias.grad is not None and l.weight.grad.dtype == torch.float32
            and l.weight.grad.stride(1) == 1 and l.bias.grad.dtype == torch.float32 and l.bias.grad.is_contiguous()
            for l in self.linears)

    @torch.no_grad()
    def forward(self, xp: torch.Tensor, keep: bool = True, padded: bool = False) -> torch.Tensor:
        """-> fp32 [rows][out_dim] (contiguous), or the padded output buffer itself with padded=True
        (row stride >= out_dim -- read it from .stride(0); columns >= out_dim are zero).  keep=True stores the activations for backward().
        `xp`: [rows][in_pad] compute dtype, zero padded -- normally from prepare_input(), whose ones column (31) lets the
        backward chain form the first layer's bias gradient; a caller-padded input works too (kind HX job instead)."""
        L = len(self.linears)
        if self._f32 is not None and not keep and xp.shape[0] > 0:
            # the no-grad pass of an fp32 net (old log-probs, values, the per-step rollout's policy mean): one launch
            self._fresh("f32")
            f = self._f32
            assert xp.dtype == torch.float32 and xp.is_contiguous() and xp.shape[1] == f.in_pad
            out = torch.empty(xp.shape[0], 4, dtype=torch.float32, device=xp.device)
            if f.res:
                N.check(N.load().tg_mlp_f32r_forward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), f.w0.data_ptr(), f.table.data_ptr(), f.H, f.n_hidden,
                                                     f.out_dim, xp.shape[0], out.data_ptr(), N.stream_ptr(xp.device)), "tg_mlp_f32r_forward")
            elif f.wide:
                N.check(N.load().tg_mlp_f32w_forward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), f.table.data_ptr(), f.n_hidden, xp.shape[0],
                                                     out.data_ptr(), N.stream_ptr(xp.device)), "tg_mlp_f32w_forward")
            else:
                N.check(N.load().tg_mlp_f32_forward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), f.H, f.n_hidden, xp.shape[0],
                                                    out.data_ptr(), N.stream_ptr(xp.device)), "tg_mlp_f32_forward")
            self._acts = self._bits = None
            return out if padded else out[:, :self.out_dim].contiguous()
        if self._chain is not None and xp.shape[0] > 0:
            self._fresh("chain")
            rows, H = xp.shape[0], self._chain.H
            # with the backward chain active the first hidden activation is never read again: tg_mlp_weight_grad
            # recomputes it from the input row (kind HR), so it is not written either (its mask bits still are)
            skip_a0 = keep and L - 1 >= 2 and self._chain_backward_ok()
            hid = [None if (i == 0 and skip_a0) else self._ws.get(f"a{i}", rows, H, self.cd, xp.device)
                   for i in range(L - 1)] if keep else []
            # 1 bit per stored activation (its ReLU mask): all the backward-data kernels need of it
            bits = [self._ws.get(f"m{i}", rows, H // 32, torch.int32, xp.device) for i in range(L - 1)] if keep else []
            # (<= 4 outputs: a 16-B output row -- tg_rollout_step then reads the policy mean without padding)
            oc = 4 if self.out_dim <= 4 else self.out_pad
            out = torch.empty(rows, oc, dtype=torch.float32, device=xp.device)
            ptrs = (N.C.c_void_p * (L - 1))(*[N.ptr(t) or None for t in hid]) if keep else None
            mptrs = (N.C.c_void_p * (L - 1))(*[t.data_ptr() for t in bits]) if keep else None
            ev = None
            if keep and self.fwd_events is not None:
                ev = N.event_pair()
                ev[0].record()
            N.check(N.load().tg_mlp_forward_chain(xp.data_ptr(), self._chain.stream.data_ptr(), self._chain.bias.data_ptr(), H,
                                                  L - 1, rows, ptrs, mptrs, out.data_ptr(), oc,
                                                  N.stream_ptr(xp.device)), "tg_mlp_forward_chain")
            if ev is not None:
                ev[1].record()
                stored = sum(1 for t in hid if t is not None)
                self.fwd_events.append((ev[0], ev[1], rows, 2 * self.in_pad + stored * 2 * H + (L - 1) * (H // 8) + 4 * oc,
                                        f"tg::mlp_fwd_chain_kernel<{H},8,true,4,{'true' if stored == L - 1 else 'false'}>"))
            self._acts = [xp] + hid if keep else None
            self._bits = [None] + bits if keep else None
            return out if padded else out[:, :self.out_dim].contiguous()
        self._fresh("w")
        acts = [xp]
        h = xp
        for i in range(L - 1):
            h = torch._addmm_activation(self.b[i], h, self.w[i].t())       # bias + ReLU in the GEMM epilogue
            acts.append(h)
        if self.cd == torch.float32:
            out = torch.addmm(self.b[-1], h, self.w[-1].t())
        else:                                                              # fp32 accumulate AND fp32 output for the head
            out = torch.mm(h, self.w[-1].t(), out_dtype=torch.float32)
            out += self.bias_out_f32
        self._acts = acts if keep else None
        self._bits = None
        return out if padded else out[:, :self.out_dim].contiguous()

    def can_fuse_head(self) -> bool:
        """The loss head + the head's weight gradient inside the forward chain (tg_mlp_forward_chain_loss): chain shapes with the
        backward chain active, at most 4 outputs, fp32 gradient windows."""
        if self._f32 is not None:
            return all(lin_ok(l) and l.weight.grad.stride(1) == 1 and l.bias.grad.is_contiguous() for l in self.linears)
        return (self._chain is not None and self._bchain is not None and self.cd == torch.bfloat16
                and len(self.linears) - 1 >= 3 and self.out_dim <= 4 and all(lin_ok(l) for l in self.linears)
                and self.linears[-1].weight.grad.stride(1) == 1)

    @torch.no_grad()
    def forward_loss(self, xp: torch.Tensor, kind: int, *, act=None, logp_old=None, adv=None, ret=None, norm=None, var=None,
                     epsilon=0.0, surr_coef=0.0, critic_coef=0.0, kl_coef=0.0, sums_out=None, logp_old_out=None, norm8=None) -> torch.Tensor:
        """Training forward pass with the loss head inside it (kind 0: actor, clipped surrogate; kind 1: critic, squared error).
        Stores what backward_fused() needs, adds the head's weight / bias gradient into their windows and returns the f64 sums
        [surrogate, squared error, KL, count] of these rows -- or, given `sums_out` (f64 [4] on the device), ADDS them there and
        returns None; on the fp32 chain learner that addition rides on backward_fused()'s reduction launch (no launches of its own),
        so `sums_out` is complete once backward_fused() has been enqueued.
        norm8: device f32 [8] (hip_ops.ppo_norm): the normalisation pair and the three coefficients are read from it on the device;
        `norm` and the `*_coef` arguments are then ignored."""
        lib = N.load()
        if self._f32 is not None:
            return self._forward_loss_f32(xp, kind, act, logp_old, adv, ret, norm, var, epsilon, surr_coef, critic_coef, kl_coef, sums_out,
                                          logp_old_out, norm8)
        self._flush_riders()                 # (a forward_loss() that was never followed by backward_fused(): its head gradient is due)
        self._fresh("chain")
        L = len(self.linears)
        rows, H, dev = xp.shape[0], self._chain.H, xp.device
        hid = [None if i in (0, L - 2) else self._ws.get(f"a{i}", rows, H, self.cd, dev) for i in range(L - 1)]
        bits = [self._ws.get(f"m{i}", rows, H // 32, torch.int32, dev) for i in range(L - 1)]
        dz_head = self._ws.get("z_head", rows, self.out_pad, self.cd, dev)
        nblk = lib.tg_mlp_forward_chain_blocks()
        if self._head_ws is None:
            self._head_ws = (torch.empty(nblk * 4 * 16 * H, dtype=torch.float32, device=dev),
                             torch.empty(nblk * 4, dtype=torch.float64, device=dev),
                             torch.empty(nblk * 4, dtype=torch.float32, device=dev))
        slabs, work, bpart = self._head_ws
        # (norm = host pair (mean, 1 / (std + eps)) of the advantage / return, or None; logp_old_out: the old policy is the current
        # one -- this pass writes the old log-probabilities instead of reading them)
        a = self._loss_args(kind, rows, act, logp_old, adv, ret, norm, var, epsilon, surr_coef, critic_coef, kl_coef, logp_old_out, norm8)
        a.d_dout8, a.d_head_slabs, a.d_work, a.d_bias_partial = dz_head.data_ptr(), slabs.data_ptr(), work.data_ptr(), bpart.data_ptr()
        ptrs = (N.C.c_void_p * (L - 1))(*[N.ptr(t) or None for t in hid])
        mptrs = (N.C.c_void_p * (L - 1))(*[t.data_ptr() for t in bits])
        ev = None
        if self.fwd_events is not None:
            ev = N.event_pair()
            ev[0].record()
        N.check(lib.tg_mlp_forward_chain_loss(xp.data_ptr(), self._chain.stream.data_ptr(), self._chain.bias.data_ptr(), H, L - 1,
                                              rows, ptrs, mptrs, N.C.byref(a), N.stream_ptr(dev)), "tg_mlp_forward_chain_loss")
        if ev is not None:
            ev[1].record()
            stored = sum(1 for t in hid if t is not None)
            per_row = 2 * self.in_pad + stored * 2 * H + (L - 1) * (H // 8) + 16 + (4 * self.out_dim + 8 if kind == 0 else 4)
            self.fwd_events.append((ev[0], ev[1], rows, per_row, f"tg::mlp_fwd_chain_kernel<{H},8,true,4,false,true>"))
        grid = min(nblk, -(-rows // 256))
        lin = self.linears[-1]
        # the head's partial weight / bias gradients (rows >= 4 of a slab are never written) and -- with sums_out -- the loss sums are
        # added by backward_fused()'s weight-gradient reduction launch, in a fixed order (three torch launches each before)
        self._riders = ([(slabs, 0, 16 * H, grid * 4, H, lin.weight.grad, self.out_dim, H),
                         (bpart, 0, 4, grid, 4, lin.bias.grad, 1, self.out_dim)],
                        (work, grid, sums_out) if sums_out is not None else None)
        if sums_out is not None:
            assert sums_out.dtype == torch.float64 and sums_out.is_cuda and sums_out.numel() == 4 and sums_out.is_contiguous()
        self._acts = [xp] + hid
        self._bits = [None] + bits
        self._dz_head = dz_head
        return None if sums_out is not None else work[:grid * 4].view(grid, 4).sum(0)

    def _flush_riders(self):
        """The partial gradients / loss sums a forward_loss() left for backward_fused()'s reduction launch, added here by torch instead
        (same sums, torch's order): only when that backward_fused() never came."""
        riders, loss_rider = getattr(self, "_riders", None) or ([], None)
        self._riders = None
        for slab, off, stride, n_slabs, pitch, grad, m_out, n_out in riders:
            if n_slabs > 0:
                v = torch.as_strided(slab, (n_slabs, m_out, n_out), (stride, pitch, 1), off).sum(0)
                grad.add_(v.view_as(grad))
        if loss_rider is not None:
            work, n, sums = loss_rider
            sums += work[:n * 4].view(n, 4).sum(0)

    def _loss_args(self, kind, rows, act, logp_old, adv, ret, norm, var, epsilon, surr_coef, critic_coef, kl_coef, logp_old_out=None,
                   norm8=None) -> "N.ChainLoss":
        a = N.ChainLoss()
        if norm8 is not None:
            N.require_cuda(norm8)
            assert norm8.dtype == torch.float32 and norm8.is_contiguous() and norm8.numel() == 8
            a.d_norm8 = norm8.data_ptr()
        a.kind, a.act_dim = kind, self.out_dim
        if kind == 0:
            N.require_cuda(act, logp_old, adv, logp_old_out)
            assert act.dtype == torch.float32 and adv.dtype == torch.float32
            assert adv.is_contiguous() and act.shape == (rows, self.out_dim) and act.is_contiguous()
            a.d_act, a.act_row_stride, a.act_col_stride = act.data_ptr(), act.stride(0), act.stride(1)
            if logp_old_out is not None:                 # the old policy is the current one: this pass WRITES the old log-probabilities
                assert logp_old_out.dtype == torch.float32 and logp_old_out.is_contiguous() and logp_old_out.numel() == rows
                a.d_logp_old_out = logp_old_out.data_ptr()
            else:
                assert logp_old.dtype == torch.float32 and logp_old.is_contiguous()
                a.d_logp_old = logp_old.data_ptr()
            a.d_adv = adv.data_ptr()
            va = [float(v) for v in (var.tolist() if isinstance(var, torch.Tensor) else var)]
            for i in range(self.out_dim):
                a.var[i] = va[i]
        else:
            N.require_cuda(ret)
            assert ret.dtype == torch.float32 and ret.is_contiguous() and self.out_dim == 1
            a.d_ret = ret.data_ptr()
            a.var[0] = 1.0
        a.norm_mean, a.norm_inv = (0.0, 1.0) if norm is None else (float(norm[0]), float(norm[1]))
        a.epsilon, a.surr_coef, a.critic_coef, a.kl_coef = float(epsilon), float(surr_coef), float(critic_coef), float(kl_coef)
        return a

    def can_write_old_logp(self) -> bool:
        """forward_loss(logp_old_out=...) is available: the loss head runs inside the forward chain (bf16 or fp32)."""
        return self.can_fuse_head()

    def _forward_loss_f32(self, xp, kind, act, logp_old, adv, ret, norm, var, epsilon, surr_coef, critic_coef, kl_coef, sums_out=None,
                          logp_old_out=None, norm8=None):
        """forward_loss() of an fp32 net: forward + loss head + backward-data pass in ONE launch (tg_mlp_f32_forward_backward);
        every hidden layer's activation and dZ is written for backward_fused() (tg_mlp_f32_weight_grad)."""
        lib = N.load()
        self._fresh("f32")
        f = self._f32
        rows, H, dev, nh = xp.shape[0], f.H, xp.device, f.n_hidden
        assert xp.dtype == torch.float32 and xp.is_contiguous() and xp.shape[1] == f.in_pad
        # with >= 2 hidden layers the first activation and the top layer's dZ are neither written nor read: the weight-gradient job
        # that needs them rebuilds them from the input row / from d loss / d output + the top layer's mask bits (16 B per row)
        # (the H = 256 learner's weight-gradient jobs read every operand back: nothing is rebuilt there yet)
        rec = nh >= 2 and not self.f32_store_all and not f.wide
        acts = [None if (rec and i == 0) else self._ws.get(f"fa{i}", rows, H, torch.float32, dev) for i in range(nh)]
        dzs = [None if (rec and i == nh - 1) else self._ws.get(f"fz{i}", rows, H, torch.float32, dev) for i in range(nh)]
        tmask = self._ws.get("fmask", rows, 4, torch.int32, dev) if rec else None
        dout = self._ws.get("fz_head", rows, 4, torch.float32, dev)
        nblk = lib.tg_mlp_f32w_blocks() if f.wide else lib.tg_mlp_f32_blocks()
        if self._head_ws is None:
            self._head_ws = torch.empty(nblk * 4, dtype=torch.float64, device=dev)
        a = self._loss_args(kind, rows, act, logp_old, adv, ret, norm, var, epsilon, surr_coef, critic_coef, kl_coef, logp_old_out, norm8)
        a.d_dout8, a.d_work = dout.data_ptr(), self._head_ws.data_ptr()
        ptrs = (N.C.c_void_p * nh)(*[N.ptr(t) for t in acts])
        zptrs = (N.C.c_void_p * nh)(*[N.ptr(t) for t in dzs])
        ev = None
        if self.fwd_events is not None:
            ev = N.event_pair()
            ev[0].record()
        if f.res:
            N.check(lib.tg_mlp_f32r_forward_backward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), f.w0.data_ptr(), f.table.data_ptr(), H, nh, rows,
                                                     ptrs, zptrs, N.ptr(tmask), N.C.byref(a), N.stream_ptr(dev)), "tg_mlp_f32r_forward_backward")
        elif f.wide:
            N.check(lib.tg_mlp_f32w_forward_backward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), f.table.data_ptr(), nh, rows, ptrs, zptrs,
                                                     N.C.byref(a), N.stream_ptr(dev)), "tg_mlp_f32w_forward_backward")
        else:
            N.check(lib.tg_mlp_f32_forward_backward(xp.data_ptr(), f.in_pad, f.stream.data_ptr(), H, nh, rows, ptrs, zptrs, N.ptr(tmask),
                                                    N.C.byref(a), N.stream_ptr(dev)), "tg_mlp_f32_forward_backward")
        if ev is not None:
            ev[1].record()
            # matrix-core flops per row: first layer + forward and backward products of the H x H layers (head: vector unit)
            # algorithmic flops per row (un-padded): forward first layer + H x H layers + head, backward head + H x H layers
            self.fwd_events.append((ev[0], ev[1], rows, 2 * H * self.in_dim + 4 * (nh - 1) * H * H + 4 * H * self.out_dim,
                                    "tg::mlp_f32_wide_kernel<true>" if f.wide else
                                    (f"tg::mlp_f32_res_kernel<{H},{f.in_pad // 4},true>" if f.res else f"tg::mlp_f32_chain_kernel<{H},true>")))
        grid = lib.tg_mlp_f32r_grid(rows) if f.res else min(nblk, -(-rows // (64 if f.wide else 256)))      # (the launchers' own grids)
        self._acts, self._bits, self._dz_head, self._tmask = [xp] + acts, dzs, dout, tmask
        assert getattr(self, "_loss_rider", None) is None, "forward_loss(sums_out=...) must be followed by backward_fused()"
        if sums_out is not None:
            assert sums_out.dtype == torch.float64 and sums_out.is_cuda and sums_out.numel() == 4 and sums_out.is_contiguous()
            self._loss_rider = (grid, sums_out)             # added by tg_mlp_f32_weight_grad's reduction launch
            return None
        return self._head_ws[:grid * 4].view(grid, 4).sum(0)

    def _backward_fused_f32(self, adam=None):
        f = self._f32
        xp, acts, dzs, dout = self._acts[0], self._acts[1:], self._bits, self._dz_head
        rows, H, nh, lin = xp.shape[0], f.H, f.n_hidden, self.linears
        if nh == 1:
            assert acts[0] is not None and dzs[0] is not None
        if self._dw_ws is None:
            self._dw_ws = torch.empty(N.load().tg_mlp_f32_weight_grad_workspace(H) // 4, dtype=torch.float32, device=xp.device)
        # (kind, P, Q, columns of Q, weight window, bias window, rows x columns of the window, rebuild bits)
        specs = []
        fused = False
        for i in range(nh - 1, 0, -1):
            rp, rq = dzs[i] is None, acts[i - 1] is None           # top layer's dZ / first activation rebuilt on chip
            fused = fused or rp or rq
            specs.append((N.TG_F32DW_MM, dout if rp else dzs[i], xp if rq else acts[i - 1], H, lin[i].weight.grad, lin[i].bias.grad, H, H,
                          (2 if rp else 0) | (1 if rq else 0)))
        if not fused:                                               # (the rebuilding jobs carry these two as riders)
            specs.append((N.TG_F32DW_MM, dzs[0], xp, f.in_pad, lin[0].weight.grad, lin[0].bias.grad, H, self.in_dim, 0))
            specs.append((N.TG_F32DW_HEAD, dout, acts[nh - 1], H, lin[nh].weight.grad, lin[nh].bias.grad, self.out_dim, H, 0))
        arr = (N.F32DwJob * len(specs))()
        for slot, (kind, p, q, ncols, wg, bg, m_out, n_out, rebuild) in zip(arr, specs):
            assert wg.dtype == torch.float32 and wg.stride(1) == 1 and bg.dtype == torch.float32 and bg.is_contiguous()
            slot.d_p, slot.d_q, slot.d_wgrad, slot.d_bgrad = p.data_ptr(), q.data_ptr(), wg.data_ptr(), bg.data_ptr()
            slot.wgrad_ld, slot.kind, slot.n_cols, slot.m_out, slot.n_out = wg.stride(0), kind, ncols, m_out, n_out
            slot.recompute, slot.in_pad, slot.in_dim, slot.act_dim = rebuild, f.in_pad, self.in_dim, self.out_dim
            if rebuild & 1:                                         # rider: the first layer's gradient
                w0, b0 = lin[0].weight, lin[0].bias
                assert w0.is_contiguous() and w0.dtype == torch.float32 and w0.grad.stride(1) == 1 and b0.grad.is_contiguous()
                slot.d_w0, slot.d_b0, slot.d_dz0 = w0.data_ptr(), b0.data_ptr(), dzs[0].data_ptr()
                slot.d_w0grad, slot.d_b0grad, slot.w0grad_ld = w0.grad.data_ptr(), b0.grad.data_ptr(), w0.grad.stride(0)
            if rebuild & 2:                                         # rider: the head's gradient
                wh, bh = lin[nh].weight, lin[nh].bias
                assert wh.is_contiguous() and wh.dtype == torch.float32 and wh.grad.stride(1) == 1 and bh.grad.is_contiguous()
                slot.d_wh, slot.d_maskbits, slot.d_a_top = wh.data_ptr(), self._tmask.data_ptr(), acts[nh - 1].data_ptr()
                slot.d_whgrad, slot.d_bhgrad, slot.whgrad_ld = wh.grad.data_ptr(), bh.grad.data_ptr(), wh.grad.stride(0)
        ev = None
        if self.dw_events is not None:
            ev = N.event_pair()
            ev[0].record()
        rider = getattr(self, "_loss_rider", None)
        self._loss_rider = None
        N.check(N.load().tg_mlp_f32_weight_grad_adam(H, arr, len(specs), rows, self._dw_ws.data_ptr(), self._dw_ws.numel() * 4,
                                                     self._head_ws.data_ptr() if rider else None, rider[0] if rider else 0,
                                                     rider[1].data_ptr() if rider else None, C.byref(adam) if adam is not None else None,
                                                     N.stream_ptr(xp.device)), "tg_mlp_f32_weight_grad")
        if adam is not None:
            adam.commit()                # (the launch was accepted: now the optimizer's step counters and the layouts' marks move)
        if ev is not None:
            ev[1].record()
            # algorithmic flops per row (un-padded): every layer's dZ^T A
            one_job = H == 128 and nh == 2
            self.dw_events.append((ev[0], ev[1], rows, 2 * (nh - 1) * H * H + 2 * H * self.in_dim + 2 * H * self.out_dim,
                                   "tg::mlp_f32_wide_dw_kernel + finish" if f.wide else
                                   (f"tg::mlp_f32_dw_fused8_kernel<{H},...> + finish" if one_job else f"tg::mlp_f32_dw_kernel<{H}> + finish")))
        self._acts = self._bits = self._dz_head = self._tmask = None

    @torch.no_grad()
    def backward_fused(self, adam=None):
        """The rest of the backward pass after forward_loss(): the backward chain and the weight gradients (no head job).
        adam (fp32 chain learner only): an optim.FusedAdam.rider() -- the optimizer step rides on the gradient reduction."""
        if self._f32 is not None:
            assert self._acts is not None and self._dz_head is not None, "backward_fused() needs forward_loss()"
            return self._backward_fused_f32(adam)
        assert adam is None, "the optimizer step rides only on the fp32 chain learner's reduction"
        acts, bits = self._acts, self._bits
        assert acts is not None and self._dz_head is not None, "backward_fused() needs forward_loss()"
        self._backward_chain(self._dz_head, acts, bits, acts[0].shape[0], acts[0].device)
        self._dz_head = None

    def _dw(self, dz: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        """dz^T a in fp32 via a batched GEMM over row blocks (split-K with fp32 partials)."""
        rows = dz.shape[0]
        if rows >= _SPLIT_BATCHES * 4096:
            # big batches: a fixed number of row blocks (2 output tiles each: every CU busy) keeps the fp32 partials --
            # written once, read once by the reduction -- at 128 x 256 KB whatever the row count (3 % at 4 M rows)
            nb, bs = _SPLIT_BATCHES, rows // _SPLIT_BATCHES            # any block length: the tail is < 128 rows
        else:
            nb, bs = rows // _ROW_BLOCK, _ROW_BLOCK
        out = None
        main = nb * bs
        if nb > 0:
            p = torch.bmm(dz[:main].view(nb, bs, dz.shape[1]).transpose(1, 2), a[:main].view(nb, bs, a.shape[1]),
                          out_dtype=torch.float32)
            out = p.sum(0)
        if rows - main > 0:
            tail = torch.mm(dz[main:].t(), a[main:], out_dtype=torch.float32) if self.cd != torch.float32 else dz[main:].t() @ a[main:]
            out = tail if out is None else out + tail
        return out

    def _bias_into(self, grads, partial: torch.Tensor):
        """grads[v] += partial[:, v, :].sum(0) for all v in ONE launch (tg_colsum_finish); partial f32 [blocks][len(grads)][width]
        or [blocks][width] for a single gradient."""
        width = partial.shape[-1]
        if len(grads) > 8 or any(g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != width for g in grads):
            p3 = partial.view(partial.shape[0], len(grads), width).sum(0)
            for v, g in enumerate(grads):
                g.add_(p3[v])
            return
        ptrs = (N.C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        N.check(N.load().tg_colsum_finish(partial.data_ptr(), partial.shape[0], len(grads), width, ptrs,
                                          N.stream_ptr(partial.device)), "tg_colsum_finish")

    def _dw_into(self, grad: torch.Tensor, dz: torch.Tensor, a: torch.Tensor):
        """grad += (dz^T a)[:grad.shape[0], :grad.shape[1]].  Big batches: the split-K batched GEMM, then ONE launch of
        tg_dw_finish (partial sums + the < 128-row tail product + the accumulation) instead of a reduction, a tail
        GEMM and two additions."""
        rows = dz.shape[0]
        if (rows < _SPLIT_BATCHES * 4096 or grad.stride(1) != 1 or grad.dtype != torch.float32
                or self.cd not in (torch.bfloat16, torch.float32)):
            grad.add_(self._dw(dz, a)[:grad.shape[0], :grad.shape[1]])
            return
        nb, bs = _SPLIT_BATCHES, rows // _SPLIT_BATCHES
        main = nb * bs
        M, K = dz.shape[1], a.shape[1]
        p = torch.bmm(dz[:main].view(nb, bs, M).transpose(1, 2), a[:main].view(nb, bs, K), out_dtype=torch.float32) \
            if self.cd != torch.float32 else torch.bmm(dz[:main].view(nb, bs, M).transpose(1, 2), a[:main].view(nb, bs, K))
        tail = rows - main
        N.check(N.load().tg_dw_finish(p.data_ptr(), nb, M, K, dz[main:].data_ptr() if tail else None,
                                      a[main:].data_ptr() if tail else None, tail, 1 if self.cd == torch.bfloat16 else 0,
                                      grad.data_ptr(), grad.stride(0), grad.shape[0], grad.shape[1], N.stream_ptr(dz.device)),
                "tg_dw_finish")

    def _backward_chain(self, dz_head, acts, bits, rows, device):
        """All hidden layers' dZ in one launch (tg_mlp_backward_chain), then every weight and hidden bias gradient in one
        more (tg_mlp_weight_grad: each dZ and activation is read once; the bias sums ride in the same contraction)."""
        lib = N.load()
        self._fresh("bchain")
        L = len(self.linears)
        nh = L - 1                                             # hidden layers; chain order = top (i = L-2) down to i = 0
        H = self._bchain.H
        # the top layer's dZ is neither written nor read: tg_mlp_weight_grad rebuilds it from dz_head and the mask bits (kind RH)
        dzs = [self._ws.get(f"z{j}", rows, H, self.cd, device) if j > 0 else None for j in range(nh)]
        dz_ptrs = (N.C.c_void_p * nh)(*[N.ptr(t) for t in dzs])
        m_ptrs = (N.C.c_void_p * nh)(*[bits[L - 1 - j].data_ptr() for j in range(nh)])     # bits[i + 1] masks hidden layer i
        # the first layer's weight (and, through the ones column of the input, bias) gradient is formed inside the backward chain
        # from the bottom dZ it holds in registers: that dZ is neither written nor read (an input without the ones column -- padded by
        # the caller, not by prepare_input() / tg_learn_compact -- takes the stored form, kind HX)
        xin = acts[0]
        fuse0 = (xin is not None and xin.shape[1] == 32 and self.in_dim < 32 and lin_ok(self.linears[0])
                 and self._has_ones_column(xin))
        if fuse0:
            dzs[nh - 1] = None
            dz_ptrs = (N.C.c_void_p * nh)(*[N.ptr(t) for t in dzs])
            if self._w0_slabs is None:
                self._w0_slabs = torch.empty(2 * lib.tg_mlp_backward_chain_blocks() * H * 32, dtype=torch.float32, device=device)
        ev = None
        if self.dx_events is not None:
            ev = N.event_pair()
            ev[0].record()
        if fuse0:
            n_slabs = N.C.c_int32(0)
            N.check(lib.tg_mlp_backward_chain_w0(dz_head.data_ptr(), self._bchain.stream.data_ptr(), H, nh, rows, dz_ptrs, m_ptrs,
                                                 xin.data_ptr(), self._w0_slabs.data_ptr(), self._w0_slabs.numel(),
                                                 N.C.byref(n_slabs), N.stream_ptr(device)), "tg_mlp_backward_chain_w0")
        else:
            N.check(lib.tg_mlp_backward_chain(dz_head.data_ptr(), self._bchain.stream.data_ptr(), H, nh, rows, dz_ptrs, m_ptrs,
                                              None, N.stream_ptr(device)), "tg_mlp_backward_chain")
        if ev is not None:
            ev[1].record()
            n_stored = nh - 1 - (1 if fuse0 else 0)
            self.dx_events.append((ev[0], ev[1], rows, 16 + nh * (H // 8) + n_stored * 2 * H + (64 if fuse0 else 0),
                                   f"tg::mlp_bwd_chain_kernel<{H},8,{'true' if fuse0 else 'false'}>"))
        riders, loss_rider = getattr(self, "_riders", None) or ([], None)
        self._riders = None
        if fuse0 and rows > 0:
            # the first layer's partial gradients ([H][32] per slab; column 31 = the bias): added by the weight-gradient reduction launch
            l0 = self.linears[0]
            riders = riders + [(self._w0_slabs, 0, H * 32, n_slabs.value, 32, l0.weight.grad, H, self.in_dim),
                               (self._w0_slabs, 31, H * 32, n_slabs.value, 32, l0.bias.grad, H, 1)]
        if self._dw_ws is None:
            self._dw_ws = weight_grad_workspace(H, device)
        lin = self.linears
        # head (bias: tg_head_prep) -- unless forward_loss() already formed its gradient and did not store the top activation
        jobs = [(N.TG_DW_DH, dz_head, acts[L - 1], lin[L - 1].weight.grad, None)] if acts[L - 1] is not None else []
        for j in range(nh - 1):
            i = L - 2 - j                                                                      # hidden-to-hidden layer i
            if j == 0:                                                                         # top layer: dZ not stored
                assert acts[i] is not None
                jobs.append((N.TG_DW_RH, dz_head, acts[i], lin[i].weight.grad, lin[i].bias.grad, bits[L - 1]))
            elif acts[i] is None:                                                              # i == 1: input not stored
                jobs.append((N.TG_DW_HR, dzs[j], acts[0], lin[i].weight.grad, lin[i].bias.grad))
            else:
                jobs.append((N.TG_DW_HH, dzs[j], acts[i], lin[i].weight.grad, lin[i].bias.grad))
        if not fuse0:
            jobs.append((N.TG_DW_HX, dzs[nh - 1], acts[0], lin[0].weight.grad, lin[0].bias.grad))
        ev = None
        if self.dw_events is not None:
            ev = N.event_pair()
            ev[0].record()
        weight_grad(H, jobs, rows, self._dw_ws, self._chain.stream, self._chain.bias[0], self._bchain.stream, riders, loss_rider)
        if ev is not None:
            ev[1].record()
            per_row = sum({N.TG_DW_HH: 4 * H, N.TG_DW_HX: 2 * H + 64, N.TG_DW_HR: 2 * H + 64, N.TG_DW_DH: 2 * H + 16,
                           N.TG_DW_RH: 2 * H + 16 + H // 8}[jb[0]] for jb in jobs)
            self.dw_events.append((ev[0], ev[1], rows, per_row, f"tg::dw_kernel<{H}>"))
        self._acts = self._bits = None

    @torch.no_grad()
    def backward(self, dout: torch.Tensor):
        """dout fp32 [rows][out_dim] = d loss / d output.  Accumulates into weight.grad / bias.grad (fp32)."""
        acts = self._acts
        assert acts is not None, "backward() needs forward(keep=True)"
        bits = self._bits if self._bits is not None else [None] * len(acts)      # ReLU mask bits of acts[i] (chain forward)
        lib = N.load()
        rows = dout.shape[0]
        L = len(self.linears)
        dout = dout.contiguous()
        dz = self._ws.get("z_head", rows, self.out_pad, self.cd, dout.device)
        lin = self.linears[-1]
        if (self.out_dim <= 8 and self.out_pad <= 16 and self.cd in (torch.bfloat16, torch.float32) and dout.dtype == torch.float32
                and lin.bias.grad.dtype == torch.float32 and lin.bias.grad.is_contiguous()):
            # padded compute-dtype copy of dout + the head's bias gradient: two launches instead of four
            if self._head_partial is None:
                self._head_partial = torch.empty(lib.tg_head_prep_blocks(), self.out_dim, dtype=torch.float32, device=dout.device)
            N.check(lib.tg_head_prep(dout.data_ptr(), rows, self.out_dim, self.out_pad, 1 if self.cd == torch.bfloat16 else 0,
                                     dz.data_ptr(), self._head_partial.data_ptr(), N.stream_ptr(dout.device)), "tg_head_prep")
            self._bias_into([lin.bias.grad], self._head_partial)
        else:
            dz.zero_()
            dz[:, :self.out_dim].copy_(dout)
            lin.bias.grad.add_(dout.sum(0))
        if self._bits is not None and self._chain_backward_ok():
            self._backward_chain(dz, acts, bits, rows, dout.device)
            return
        assert all(a is not None for a in acts), "the gradient buffers changed between forward(keep=True) and backward()"
        self._dw_into(lin.weight.grad, dz, acts[L - 1])
        self._fresh("dx")
        is_bf16 = 1 if self.cd == torch.bfloat16 else 0
        nblk = lib.tg_relu_bwd_bias_blocks()
        fuse_head = self.out_dim <= 8
        st = N.stream_ptr(dout.device)
        for i in range(L - 2, -1, -1):
            a = acts[i + 1]                                    # post-ReLU output of hidden layer i
            cols = a.shape[1]
            frag = self._dxfrag[i + 1] if i < L - 2 else None
            if frag is not None:
                # dZ_i = (dZ_{i+1} W_{i+1}) * (a > 0) and its column sums in one pass on the matrix cores
                if self._dx_partial is None or self._dx_partial.shape[1] != cols:
                    self._dx_partial = torch.empty(lib.tg_dx_relu_bias_blocks(), cols, dtype=torch.float32, device=dout.device)
                partial = self._dx_partial
                dz_below = torch.empty_like(a)
                ev = None
                if self.dx_events is not None:
                    ev = N.event_pair()
                    ev[0].record()
                mb = bits[i + 1]
                N.check(lib.tg_dx_relu_bias(dz.data_ptr(), frag.data_ptr(), a.data_ptr(), None if mb is None else mb.data_ptr(),
                                            dz_below.data_ptr(), rows, dz.shape[1], cols, partial.data_ptr(), st),
                        "tg_dx_relu_bias")
                if ev is not None:
                    ev[1].record()
                    self.dx_events.append((ev[0], ev[1], rows, 2 * (dz.shape[1] + cols) + (cols // 8 if mb is not None else 2 * cols),
                                           "tg::dx_relu_bias_kernel<%d,%d>" % (cols, dz.shape[1])))
                dz = dz_below
            else:
                if self._partial is None or self._partial.shape[1] != cols:
                    self._partial = torch.empty(nblk, cols, dtype=torch.float32, device=dout.device)
                partial = self._partial
                if i == L - 2 and fuse_head:
                    # top hidden layer: dA = dout . W_head is a rank-A product, formed inside the ReLU-backward kernel
                    da = torch.empty_like(a)
                    mb = bits[i + 1]
                    N.check(lib.tg_head_bwd_relu_bias(dout.data_ptr(), self.out_dim, self.linears[-1].weight.data_ptr(),
                                                      a.data_ptr(), None if mb is None else mb.data_ptr(), da.data_ptr(), rows,
                                                      cols, is_bf16, partial.data_ptr(), st), "tg_head_bwd_relu_bias")
                else:
                    da = dz @ self.w[i + 1]
                    N.check(lib.tg_relu_bwd_bias(da.data_ptr(), a.data_ptr(), rows, cols, is_bf16, partial.data_ptr(), st),
                            "tg_relu_bwd_bias")
                dz = da
            lin = self.linears[i]
            self._bias_into([lin.bias.grad], partial)
            self._dw_into(lin.weight.grad, dz, acts[i])
        self._acts = self._bits = None


# ---------------------------------------------------------------------------------------------
# weight stream of the fused rollout kernel (csrc/fused_rollout.hip)
# ---------------------------------------------------------------------------------------------
def fused_rollout_supported(net, obs_dim: int, act_dim: int) -> int:
    """Hidden width H if `net` is Linear(S,H) ReLU [Linear(H,H) ReLU]* Linear(H,A) with H in {128,256}, else 0."""
    if not supports(net) or obs_dim > 32 or act_dim > 4:
        return 0
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    if H not in (128, 256) or any(l.out_features != H for l in lin[:-1]) or any(l.in_features != H for l in lin[1:]):
        return 0
    return H


def _fragment_index(k_pad: int, device):
    """[k_pad/16][64 lanes][8]: column of W that element j of lane (r, h) holds at k-step ks:
    16*ks + 8*(j>>2) + 4*h + (j&3)  (the order in which an MFMA accumulator tile hands its rows over)."""
    ks = torch.arange(k_pad // 16, device=device).view(-1, 1, 1)
    lane = torch.arange(64, device=device).view(1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, -1)
    return 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)


def _chain_fragment_index(k_pad: int, device):
    """[k_pad/32 k-steps][2 feature halves f][64 lanes][8] -> (row within the 32-row block, column) of W for the chain kernels'
    `v_mfma_f32_16x16x32_bf16` A fragments.  Lane (i = lane & 15, g = lane >> 4) of half f holds, for k-step ks, the 8 weights
    W[rowmap(f, i)][32 ks + 8 g + j]: the k order is natural (the B operand of the next layer is the packed accumulator of this
    one, whose lane (col, g) holds the 8 CONSECUTIVE features 8 g .. 8 g + 7 of a block -- that is what the row map arranges:
    accumulator register r of lane group g in half f is output feature 8 g + 4 f + r)."""
    ks = torch.arange(k_pad // 32, device=device).view(-1, 1, 1, 1)
    f = torch.arange(2, device=device).view(1, -1, 1, 1)
    lane = torch.arange(64, device=device).view(1, 1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, 1, -1)
    i = lane & 15
    chain_row = (8 * (i >> 2) + 4 * f + (i & 3)).expand(k_pad // 32, 2, 64, 8)      # outputs that feed another layer / are stored
    natural_row = (16 * f + i).expand(k_pad // 32, 2, 64, 8)                          # the forward head: outputs 0..15 in half 0
    col = (32 * ks + 8 * (lane >> 4) + j).expand(k_pad // 32, 2, 64, 8)
    return chain_row, natural_row, col


class FragmentStream:
    """bf16 weight stream + f32 bias table in the layout tg_fused_rollout (layout="rollout", 32x32x16 MFMA) or the chain
    kernels (layout="chain", 16x16x32 MFMA: tg_mlp_forward_chain, tg_mlp_backward_chain, tg_mlp_weight_grad's recompute)
    consume, refreshed from the fp32 master weights with one gather (the permutation is built once).
    rollout: blocks = the first layer's output tiles (2 k-steps each), then one block per 32-row output tile of every later
      layer; a block = its k-steps x 64 lanes x 8 bf16 (1 KiB per k-step); lane (m, h) holds 8 weights of output row 32*tile + m.
    chain: every block is H/16 pieces of 1 KiB.  A matrix with K <= 32 (the first layer; the backward stream's W_head^T) is ONE
      block [output block mt][half f]; a matrix with K = H is one block per 32 output features, [k-step ks][half f]
      (_chain_fragment_index documents the lane map); the forward head keeps its <= 16 outputs in half 0, natural order."""

    def __init__(self, net, H: int, layout: str = "rollout", transposed: bool = False):
        """transposed=True: the stream of the backward chain (tg_mlp_backward_chain): the head first, then the
        hidden-to-hidden layers from the top down, every matrix transposed ([in][out]); no bias table."""
        assert layout in ("rollout", "chain")
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        self.transposed = transposed
        self.lin = lin[:0:-1] if transposed else lin            # head, L-2, ..., 1  |  0, 1, ..., head
        dev = lin[0].weight.device
        self.H = H
        m = torch.arange(64, device=dev) & 31
        # One gather builds the stream: source = the master weights laid end to end (row-major, unpadded) + the biases +
        # one zero that every padded position points at.
        flat_idx, off = [], 0
        n_src = sum(l.weight.numel() for l in self.lin)
        n_bias = sum(l.bias.numel() for l in self.lin)
        zero_at = n_src + n_bias
        for li, l in enumerate(self.lin):
            rows_, cols_ = (l.in_features, l.out_features) if transposed else (l.out_features, l.in_features)
            m_pad, k_pad = _round_up(rows_, 32), _round_up(cols_, 32)
            if layout == "rollout":
                kidx = _fragment_index(k_pad, dev)
                rmap = m.view(1, -1, 1).expand_as(kidx)
            else:
                chain_row, natural_row, kidx = _chain_fragment_index(k_pad, dev)
                rmap = natural_row if (not transposed and li == len(self.lin) - 1) else chain_row
            for mo in range(m_pad // 32):
                r = 32 * mo + rmap                                             # row / column of the (transposed) padded matrix
                c = kidx
                src = off + (c * l.in_features + r if transposed else r * l.in_features + c)   # weight is [out][in]
                flat_idx.append(torch.where((r < rows_) & (c < cols_), src, torch.full_like(src, zero_at)).reshape(-1))
            off += l.weight.numel()
        n_stream = sum(t.numel() for t in flat_idx)
        if not transposed:                                                     # bias table [layers][H], zero padded
            boff = n_src
            for l in self.lin:
                j = torch.arange(H, device=dev)
                flat_idx.append(torch.where(j < l.out_features, boff + j, torch.full_like(j, zero_at)))
                boff += l.bias.numel()
        self._idx = torch.cat(flat_idx)
        self._n_stream = n_stream
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)
        self._gath = torch.empty(self._idx.numel(), dtype=torch.float32, device=dev)
        self.stream = torch.empty(n_stream, dtype=torch.bfloat16, device=dev)
        self.bias = torch.zeros(len(self.lin), H, dtype=torch.float32, device=dev)
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Three launches: concatenate the master tensors, gather, convert to bf16 (+ one copy for the bias table)."""
        src = torch.cat([l.weight.reshape(-1) for l in self.lin] + [l.bias for l in self.lin] + [self._zero])
        torch.index_select(src.float() if src.dtype != torch.float32 else src, 0, self._idx, out=self._gath)
        self.stream.copy_(self._gath[:self._n_stream])
        if not self.transposed:
            self.bias.view(-1).copy_(self._gath[self._n_stream:])


def fragment_stream(net, H: int):
    """(bf16 weight stream, f32 bias table) in the layout tg_fused_rollout consumes (one-shot form of FragmentStream)."""
    fs = FragmentStream(net, H)
    return fs.stream, fs.bias


# ---------------------------------------------------------------------------------------------
# register-resident fp32 weights of the fp32 fused rollout kernel (csrc/fused_rollout_f32.hip)
# ---------------------------------------------------------------------------------------------
def fused_rollout_f32_supported(net, obs_dim: int, act_dim: int) -> int:
    """Hidden width H if `net` is Linear(S,H) ReLU [Linear(H,H) ReLU]* Linear(H,A) with H in {64,128} and 1..4 hidden
    layers (what tg_fused_rollout_f32 has kernels for), else 0."""
    if not supports(net) or obs_dim > 32 or act_dim > 4:
        return 0
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    if H not in (64, 128) or not (1 <= len(lin) - 1 <= 4):
        return 0
    if any(l.out_features != H for l in lin[:-1]) or any(l.in_features != H for l in lin[1:]):
        return 0
    return H


class RegisterStreamF32:
    """fp32 weights in the order tg_fused_rollout_f32 loads them into registers, refreshed from the master weights
    with one gather: [H/32 waves][K1/2 + n_hh*H/2 registers][64 lanes] (first layer: K1 = in_features rounded up to 8, zero beyond).
    block_envs 32: register 4q + j of lane (m, kh) of wave w holds W[32w + m][8q + 4kh + j];
    block_envs 16: register tt * (k / 4) + s of lane (i, g) holds W[32w + 16 tt + i][first layer: 4 s + g; H x H: 16 (s >> 2) + 4 g + (s & 3)].
    `table` = [n_hidden*H hidden biases][4*H head weights][4 head biases]."""

    def __init__(self, net, H: int, block_envs: int = 32):
        assert block_envs in (16, 32)
        self.block_envs = block_envs
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        self.lin, self.H = lin, H
        dev = lin[0].weight.device
        n_hidden = len(lin) - 1
        k1 = _round_up(lin[0].in_features, 8)
        self._k = [k1] + [H] * (n_hidden - 1)
        lane = torch.arange(64, device=dev).view(1, 1, -1)
        wave = torch.arange(H // 32, device=dev).view(-1, 1, 1)
        idx, off = [], 0
        for li, k in enumerate(self._k):
            r = torch.arange(k // 2, device=dev).view(1, -1, 1)
            if block_envs == 32:
                row = 32 * wave + (lane & 31)
                colk = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)
            else:
                tt, s_ = r // (k // 4), r % (k // 4)
                row = 32 * wave + 16 * tt + (lane & 15)
                colk = 4 * s_ + (lane >> 4) if li == 0 else 16 * (s_ >> 2) + 4 * (lane >> 4) + (s_ & 3)
            idx.append(off + row * k + colk)                   # [waves][k/2][64]
            off += H * k
        self._idx = torch.cat(idx, dim=1).reshape(-1)
        self._wflat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.stream = torch.empty(self._idx.numel(), dtype=torch.float32, device=dev)
        self.table = torch.zeros(n_hidden * H + 4 * H + 4, dtype=torch.float32, device=dev)
        self._fresh_key = None
        self.refresh()

    def _key(self):
        """What the stream is a function of: the parameters' storage and torch version counters, and the count of raw-pointer
        parameter writes (the fused Adam launch bypasses the version counters)."""
        return (tuple((p.data_ptr(), p._version) for l in self.lin for p in (l.weight, l.bias)), N.RAW_PARAM_WRITES[0])

    def mark_fresh(self) -> None:
        self._fresh_key = self._key()

    def is_fresh(self) -> bool:
        return not N.ALWAYS_REBUILD and self._fresh_key == self._key()

    def segments(self, tensor_ids):
        """The stream and the table as tg_gather_streams segments: [(dst, int32 code per element, is_bf16)], code = master tensor
        << 24 | offset, -1 = zero (optim.StreamRefresher: rebuilt in the launch that follows an optimizer step)."""
        H, lin = self.H, self.lin
        n_hidden = len(lin) - 1
        flat = torch.full((self._wflat.numel(),), -1, dtype=torch.int64)
        off = 0
        for l, k in zip(lin[:-1], self._k):
            gi, ti = tensor_ids[id(l.weight)]
            assert gi == 0
            r, c = torch.arange(H).view(H, 1), torch.arange(k).view(1, k)
            flat[off:off + H * k] = torch.where(c < l.in_features, (ti << 24) + r * l.in_features + c, -1).reshape(-1)
            off += H * k
        dev = self.stream.device
        s_codes = flat[self._idx.cpu()].to(torch.int32).to(dev)
        t = torch.full((self.table.numel(),), -1, dtype=torch.int64)
        for li, l in enumerate(lin[:-1]):
            gi, ti = tensor_ids[id(l.bias)]
            assert gi == 0
            t[li * H:(li + 1) * H] = (ti << 24) + torch.arange(H)
        head = lin[-1]
        gi, tw = tensor_ids[id(head.weight)]
        gi2, tb = tensor_ids[id(head.bias)]
        assert gi == 0 and gi2 == 0
        for a in range(head.out_features):
            t[n_hidden * H + a * H:n_hidden * H + (a + 1) * H] = (tw << 24) + a * H + torch.arange(H)
            t[n_hidden * H + 4 * H + a] = (tb << 24) + a
        return [(self.stream, s_codes, 0), (self.table, t.to(torch.int32).to(dev), 0)]

    def refresh(self):
        self._refresh()
        self.mark_fresh()

    @torch.no_grad()
    def _refresh(self):
        H, off = self.H, 0
        n_hidden = len(self.lin) - 1
        for l, k in zip(self.lin[:-1], self._k):
            self._wflat[off:off + H * k].view(H, k)[:, :l.in_features].copy_(l.weight)
            off += H * k
        torch.index_select(self._wflat, 0, self._idx, out=self.stream)
        for li, l in enumerate(self.lin[:-1]):
            self.table[li * H:(li + 1) * H].copy_(l.bias)
        head = self.lin[-1]
        hw = self.table[n_hidden * H:n_hidden * H + 4 * H].view(4, H)
        hw[:head.out_features].copy_(head.weight)
        self.table[n_hidden * H + 4 * H:n_hidden * H + 4 * H + head.out_features].copy_(head.bias)


# ---------------------------------------------------------------------------------------------
# fp32 chain learner (csrc/mlp_f32_chain.hip): weight stream of tg_mlp_f32_forward / _forward_backward
# ---------------------------------------------------------------------------------------------
def f32_chain_supported(net) -> int:
    """Hidden width H if `net` is Linear(S<=32, H) ReLU [Linear(H, H) ReLU]{0..3} Linear(H, A<=4) with H in {64, 128} -- the
    reference's own policy shapes (pipelines/cartpole_pipeline_grpo.py:54-76, cartpole_pipeline_ppo.py:54-79) -- else 0."""
    if not supports(net):
        return 0
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    if H not in (64, 128) or not (1 <= len(lin) - 1 <= 4) or lin[0].in_features > 32 or lin[-1].out_features > 4:
        return 0
    if any(l.out_features != H for l in lin[:-1]) or any(l.in_features != H for l in lin[1:]):
        return 0
    return H


def f32_res_supported(net) -> int:
    """128 if `net` is Linear(S<=32, 128) ReLU [Linear(128, 128) ReLU]{0..1} Linear(128, A<=4) -- BASELINE configs[1]'s policy
    (5-128-128-1) and its like: the whole weight stream fits the LDS beside the tables -- else 0."""
    if not supports(net):
        return 0
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    if H != 128 or not (1 <= len(lin) - 1 <= 2) or lin[0].in_features > 32 or lin[-1].out_features > 4:
        return 0
    if any(l.out_features != H for l in lin[:-1]) or any(l.in_features != H for l in lin[1:]):
        return 0
    return H if N.load().tg_mlp_f32r_supported(H, len(lin) - 1, _round_up(lin[0].in_features, 8)) else 0


class F32ResStream:
    """Weight stream, first-layer table and bias / head table of the resident 16-row kernel (csrc/mlp_f32_wide.hip, mlp_f32_res_kernel),
    ONE buffer refreshed with one gather: lane = (i = lane & 15, g = lane >> 4)
      stream: forward blocks [8 mo][8 t][64 lanes][4]: W_1[16 mo + i][16 t + 4 g + e]; backward blocks [8 ko][8 t][64][4]:
              W_1[16 t + 4 g + e][16 ko + i]            (nothing with one hidden layer)
      w0:     [8 mo][in_pad / 4 steps s][64 lanes]: W0[16 mo + i][4 s + g]              (zero beyond the inputs)
      table:  [2][128] hidden biases | [4][128] head weights (rows >= A zero) | [4] head bias | 12 zeros"""
    wide, res = False, True

    def __init__(self, net):
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        self.lin, self.H = lin, 128
        H, NT = 128, 8
        dev = lin[0].weight.device
        nh = len(lin) - 1
        self.n_hidden, self.in_dim, self.out_dim = nh, lin[0].in_features, lin[-1].out_features
        self.in_pad = _round_up(self.in_dim, 8)
        K4 = self.in_pad // 4
        woff, off = [], 0
        for l in lin:
            woff.append(off)
            off += l.weight.numel()
        boff = []
        for l in lin:
            boff.append(off)
            off += l.bias.numel()
        zero_at = off
        lane = torch.arange(64, device=dev).view(1, 64, 1)
        i, g = lane & 15, lane >> 4
        e = torch.arange(4, device=dev).view(1, 1, 4)
        t = torch.arange(NT, device=dev).view(NT, 1, 1)
        k = (16 * t + 4 * g + e).expand(NT, 64, 4)
        idx = []
        for l in range(1, nh):                                                       # forward: W_l[16 mo + i][k]
            for mo in range(NT):
                idx.append((woff[l] + (16 * mo + i) * H + k).reshape(-1))
        for l in range(nh - 1, 0, -1):                                               # backward: W_l[k][16 ko + i]
            for ko in range(NT):
                idx.append((woff[l] + k * H + (16 * ko + i)).reshape(-1))
        n_stream = sum(x.numel() for x in idx)
        s4 = torch.arange(K4, device=dev).view(K4, 1)
        ln = torch.arange(64, device=dev).view(1, 64)
        col = (4 * s4 + (ln >> 4)).expand(K4, 64)
        for mo in range(NT):
            src = woff[0] + (16 * mo + (ln & 15)).expand(K4, 64) * self.in_dim + col
            idx.append(torch.where(col < self.in_dim, src, torch.full_like(src, zero_at)).reshape(-1))
        n_w0 = NT * K4 * 64
        for l in range(2):
            idx.append(boff[l] + torch.arange(H, device=dev) if l < nh else torch.full((H,), zero_at, device=dev))
        for a in range(4):
            idx.append(woff[nh] + a * H + torch.arange(H, device=dev) if a < self.out_dim else torch.full((H,), zero_at, device=dev))
        idx.append(torch.tensor([boff[nh] + a if a < self.out_dim else zero_at for a in range(4)] + [zero_at] * 12, device=dev))
        self._idx = torch.cat([x.reshape(-1) for x in idx])
        lib = N.load()
        assert n_stream == lib.tg_mlp_f32r_stream_floats(H, nh) and n_w0 == lib.tg_mlp_f32r_w0_floats(H, self.in_pad)
        assert self._idx.numel() - n_stream - n_w0 == lib.tg_mlp_f32r_table_floats(H)
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)
        self.stream = torch.empty(self._idx.numel(), dtype=torch.float32, device=dev)          # blocks, first-layer table, tables
        self.w0 = self.stream[n_stream:n_stream + n_w0]
        self.table = self.stream[n_stream + n_w0:]
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Two launches: concatenate the master tensors, gather."""
        src = torch.cat([l.weight.reshape(-1) for l in self.lin] + [l.bias for l in self.lin] + [self._zero])
        torch.index_select(src if src.dtype == torch.float32 else src.float(), 0, self._idx, out=self.stream)


def f32_wide_supported(net) -> int:
    """256 if `net` is Linear(S<=32, 256) ReLU [Linear(256, 256) ReLU]{0..4} Linear(256, A<=4) -- the reference's QuadPole factory
    (pipelines/quadpole_pipeline_ppo.py:54-58: 20-256x5-{4,1}) -- else 0."""
    if not supports(net):
        return 0
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    if H != 256 or not (1 <= len(lin) - 1 <= 5) or lin[0].in_features > 32 or lin[-1].out_features > 4:
        return 0
    if any(l.out_features != H for l in lin[:-1]) or any(l.in_features != H for l in lin[1:]):
        return 0
    return H


class F32WideStream:
    """The fp32 weight stream + tables of the H = 256 chain learner (csrc/mlp_f32_wide.hip), refreshed from the master weights with ONE
    gather.  16-KiB blocks of 16 pieces x 64 lanes x 16 B, lane = (i = lane & 15, g = lane >> 4); a lane's 16 B are four consecutive
    floats of a weight row (forward) or of a weight column (backward):
      first layer, 2 blocks: block b, piece 2 tt + q:     W0[16 (8 b + tt) + i][8 g + 4 q + e]      (zero beyond the inputs)
      forward, layer l = 1 .. nh - 1, block mo, piece t:  W_l[16 mo + i][16 t + 4 g + e]
      backward, layer l = nh - 1 .. 1, block ko, piece t: W_l[16 t + 4 g + e][16 ko + i]
    then the tables: [5][256] hidden biases | [4][256] head weights (rows >= A zero) | [4] head bias | 12 zeros."""
    wide, res = True, False

    def __init__(self, net):
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        self.lin, self.H = lin, 256
        H = 256
        dev = lin[0].weight.device
        nh = len(lin) - 1
        self.n_hidden, self.in_dim, self.out_dim = nh, lin[0].in_features, lin[-1].out_features
        self.in_pad = _round_up(self.in_dim, 8)
        woff, off = [], 0
        for l in lin:
            woff.append(off)
            off += l.weight.numel()
        boff = []
        for l in lin:
            boff.append(off)
            off += l.bias.numel()
        zero_at = off
        lane = torch.arange(64, device=dev).view(1, 64, 1)
        i, g = lane & 15, lane >> 4
        e = torch.arange(4, device=dev).view(1, 1, 4)
        idx = []
        # first layer: 16 tiles x 2 pieces
        q = torch.arange(2, device=dev).view(2, 1, 1)
        for mo in range(16):
            col = (8 * g + 4 * q + e).expand(2, 64, 4)
            src = woff[0] + (16 * mo + i) * self.in_dim + col
            idx.append(torch.where(col < self.in_dim, src, torch.full_like(src, zero_at)).reshape(-1))
        t = torch.arange(16, device=dev).view(16, 1, 1)
        k = (16 * t + 4 * g + e).expand(16, 64, 4)                                   # the contraction index of piece t, lane group g
        for l in range(1, nh):                                                       # forward: W_l[16 mo + i][k]
            for mo in range(16):
                idx.append((woff[l] + (16 * mo + i) * H + k).reshape(-1))
        for l in range(nh - 1, 0, -1):                                               # backward: W_l[k][16 ko + i]
            for ko in range(16):
                idx.append((woff[l] + k * H + (16 * ko + i)).reshape(-1))
        n_stream = sum(x.numel() for x in idx)
        for l in range(5):
            idx.append(boff[l] + torch.arange(H, device=dev) if l < nh else torch.full((H,), zero_at, device=dev))
        for a in range(4):
            idx.append(woff[nh] + a * H + torch.arange(H, device=dev) if a < self.out_dim else torch.full((H,), zero_at, device=dev))
        idx.append(torch.tensor([boff[nh] + a if a < self.out_dim else zero_at for a in range(4)] + [zero_at] * 12, device=dev))
        self._idx = torch.cat([x.reshape(-1) for x in idx])
        lib = N.load()
        assert n_stream == lib.tg_mlp_f32w_stream_floats(nh) and self._idx.numel() - n_stream == lib.tg_mlp_f32w_table_floats()
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)
        self.stream = torch.empty(self._idx.numel(), dtype=torch.float32, device=dev)          # blocks, then the tables
        self.table = self.stream[n_stream:]
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Two launches: concatenate the master tensors, gather."""
        src = torch.cat([l.weight.reshape(-1) for l in self.lin] + [l.bias for l in self.lin] + [self._zero])
        torch.index_select(src if src.dtype == torch.float32 else src.float(), 0, self._idx, out=self.stream)


class F32ChainStream:
    wide, res = False, False

    """The fp32 weight stream of the chain learner, refreshed from the master weights with ONE gather:
      [first layer, MFMA fragment order: H/32 tiles x k2/4 groups x 64 lanes x 4]  lane (i, kk), step s = 4 g + e of tile mo holds
                                                                                 W0[32 mo + i][s + kk k2]   (k2 = in_pad / 2)
      [hidden biases: n_hidden x H]  [head weights: 4 x H, rows >= A zero]  [head bias: 4]          (natural order)
      [forward blocks]   layer l = 1..n_hh, output tile mo: H/8 groups x 64 lanes x 4; step s = 16 mt + t holds
                         W_l[32 mo + i][32 mt + F(t, kk)],  F(t, kk) = (t & 3) + 8 (t >> 2) + 4 kk  -- register t of the two lane
                         halves of the previous layer's accumulator tile mt (v_mfma_f32_32x32x2_f32: see the kernel's header)
      [backward blocks]  layer l = n_hh..1 (top first), output tile ko over the layer's INPUT features:
                         W_l[32 mt + F(t, kk)][32 ko + i]"""

    def __init__(self, net, H: int):
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        self.lin, self.H = lin, H
        dev = lin[0].weight.device
        nh = len(lin) - 1
        self.n_hidden, self.in_dim, self.out_dim = nh, lin[0].in_features, lin[-1].out_features
        self.in_pad = _round_up(self.in_dim, 8)
        K2, MT = self.in_pad // 2, H // 32
        woff, off = [], 0
        for l in lin:
            woff.append(off)
            off += l.weight.numel()
        boff = []
        for l in lin:
            boff.append(off)
            off += l.bias.numel()
        zero_at = off
        lane = torch.arange(64, device=dev).view(1, 64, 1)
        i, kk = lane & 31, lane >> 5
        e = torch.arange(4, device=dev).view(1, 1, 4)
        idx = []
        # first layer
        g = torch.arange(K2 // 4, device=dev).view(-1, 1, 1)
        col = (4 * g + e) + kk * K2
        for mo in range(MT):
            row = (32 * mo + i).expand(K2 // 4, 64, 4)
            src = woff[0] + row * self.in_dim + col
            idx.append(torch.where(col.expand_as(src) < self.in_dim, src, torch.full_like(src, zero_at)).reshape(-1))
        # hidden biases, head weights (4 rows), head bias (4)
        for l in range(nh):
            idx.append(boff[l] + torch.arange(H, device=dev))
        for a in range(4):
            idx.append(woff[nh] + a * H + torch.arange(H, device=dev) if a < self.out_dim else torch.full((H,), zero_at, device=dev))
        idx.append(torch.tensor([boff[nh] + a if a < self.out_dim else zero_at for a in range(4)], device=dev))
        # H x H blocks
        g = torch.arange(H // 8, device=dev).view(-1, 1, 1)
        s = 4 * g + e
        mt, t = s // 16, s % 16
        feat = (32 * mt + (t & 3) + 8 * (t >> 2) + 4 * kk).expand(H // 8, 64, 4)      # the contraction index of step s, lane half kk
        for l in range(1, nh):                                                          # forward: W_l[32 mo + i][feat]
            for mo in range(MT):
                idx.append((woff[l] + (32 * mo + i) * H + feat).reshape(-1))
        for l in range(nh - 1, 0, -1):                                                  # backward: W_l[feat][32 ko + i]
            for ko in range(MT):
                idx.append((woff[l] + feat * H + (32 * ko + i)).reshape(-1))
        self._idx = torch.cat([x.reshape(-1) for x in idx])
        n = N.load().tg_mlp_f32_stream_floats(H, nh, self.in_pad)
        assert self._idx.numel() == n, (self._idx.numel(), n)
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)
        self.stream = torch.empty(n, dtype=torch.float32, device=dev)
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Two launches: concatenate the master tensors, gather."""
        src = torch.cat([l.weight.reshape(-1) for l in self.lin] + [l.bias for l in self.lin] + [self._zero])
        torch.index_select(src if src.dtype == torch.float32 else src.float(), 0, self._idx, out=self.stream)
