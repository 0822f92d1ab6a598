#!/usr/bin/env python3
"""The fp32 learner at the reference's own QuadPole factory shape (20-256x5-{4,1}, pipelines/quadpole_pipeline_ppo.py:54-58):
this tree's H = 256 chain kernel (csrc/mlp_f32_wide.hip: forward + loss head + backward data in one launch) against the per-layer
hipBLASLt path on the same rows, same box.  One JSON line per row count:

    python tools/f32_h256_probe.py [--rows 1048576] [--iters 10] [--out-dim 4]

  lib_forward_ms         per-layer path: forward(keep=True)      (6 GEMMs with bias + ReLU epilogues)
  lib_backward_ms        per-layer path: backward()              (backward-data GEMMs + ReLU-backward / bias kernels + split-K dW GEMMs)
  lib_dw_ms              per-layer path: the weight-gradient part of backward() alone (the same split-K GEMMs on stored operands)
  chain_ms               tg_mlp_f32w_forward_backward            (forward + loss head + backward data)
  wide_dw_ms             tg_mlp_f32w_weight_grad when built      (every weight / bias gradient)
The comparison that decides the item: chain_ms against lib_forward_ms + (lib_backward_ms - lib_dw_ms) + the loss-head launch."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd import mlp as M  # noqa: E402
from trajopt_grpo_amd import hip_ops as K  # noqa: E402


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1 << 20])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out-dim", type=int, default=4)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--wide-only", action="store_true", help="only this tree's kernels (PMC passes)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    S, A, H, L = 20, a.out_dim, 256, a.layers
    net = tg.NeuralNetwork(S, A, (H,) * L, "ReLU").to(dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    wide = M.GemmMLP(net, torch.float32)
    assert wide._f32 is not None and wide._f32.wide
    lib = M.GemmMLP(net, torch.float32)
    lib.disable_f32_chain()
    assert lib._f32 is None
    n_w = sum(l.weight.numel() for l in net.network if isinstance(l, torch.nn.Linear))
    hh = (L - 1) * H * H
    fl_fwd = 2.0 * n_w
    fl_bwd = 2.0 * (hh + H * A)
    fl_dw = 2.0 * n_w
    for rows in a.rows:
        X = torch.randn(rows, S, device=dev)
        act = torch.randn(rows, A, device=dev)
        lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0).contiguous()
        adv, ret = torch.randn(rows, device=dev), torch.randn(rows, device=dev)
        var = torch.full((A,), 0.3)
        xw, xl = wide.prepare_input(X), lib.prepare_input(X)
        g = torch.randn(rows, A, device=dev) / rows
        out = {"rows": rows, "shape": f"{S}-{H}x{L}-{A}", "iters": a.iters}

        def chain():
            if A == 1:
                wide.forward_loss(xw, 1, ret=ret, norm=[0.0, 1.0], critic_coef=0.5 / rows)
            else:
                wide.forward_loss(xw, 0, act=act, logp_old=lpo, adv=adv, norm=[0.0, 1.0], var=var, epsilon=0.2, surr_coef=-1.0 / rows, kl_coef=0.5 / rows)

        out["chain_ms"] = timed(chain, a.iters)
        chain()
        acts_w, dzs_w, dout_w = [t for t in wide._acts], [t for t in wide._bits], wide._dz_head
        out["chain_plus_dw_ms"] = timed(lambda: (chain(), wide.backward_fused()), a.iters)
        out["wide_dw_ms"] = out["chain_plus_dw_ms"] - out["chain_ms"]
        out["nograd_ms"] = timed(lambda: wide.forward(xw, keep=False, padded=True), a.iters)
        if a.wide_only:
            print(json.dumps(out), flush=True)
            continue

        out["lib_forward_ms"] = timed(lambda: lib.forward(xl, keep=True), a.iters)
        out["lib_nograd_ms"] = timed(lambda: lib.forward(xl, keep=False), a.iters)

        def lib_fb():
            y = lib.forward(xl, keep=True)
            lib.backward(g)

        out["lib_forward_backward_ms"] = timed(lib_fb, a.iters)
        out["lib_backward_ms"] = out["lib_forward_backward_ms"] - out["lib_forward_ms"]
        # the weight-gradient GEMMs alone, on the operands the chain kernel stored
        lins = wide.linears

        def dw_only():
            for i in range(L):
                lib._dw_into(lins[i].weight.grad, dzs_w[i], acts_w[0] if i == 0 else acts_w[i])
                lins[i].bias.grad.add_(dzs_w[i].sum(0))
            lib._dw_into(lins[L].weight.grad, dout_w, acts_w[L])

        out["lib_dw_ms"] = timed(dw_only, a.iters)
        mean = torch.randn(rows, A, device=dev)
        if A > 1:
            out["lib_loss_head_ms"] = timed(lambda: K.surrogate_loss(mean, None, act, lpo, adv, None, None, None, var, 0.2, -1.0 / rows, 0.0, 0.5 / rows,
                                                                      want_total=False), a.iters)
        else:
            out["lib_loss_head_ms"] = 0.0
        lib_chain = out["lib_forward_ms"] + out["lib_backward_ms"] - out["lib_dw_ms"] + out["lib_loss_head_ms"]
        out["lib_chain_equivalent_ms"] = lib_chain
        out["chain_speedup_vs_lib"] = lib_chain / out["chain_ms"]
        out["chain_TFLOPs"] = (fl_fwd + fl_bwd) * rows / out["chain_ms"] / 1e9
        out["lib_chain_equivalent_TFLOPs"] = (fl_fwd + fl_bwd) * rows / lib_chain / 1e9
        out["lib_dw_TFLOPs"] = fl_dw * rows / out["lib_dw_ms"] / 1e9
        out["lib_update_ms"] = out["lib_forward_backward_ms"] + out["lib_loss_head_ms"]
        out["lib_update_TFLOPs"] = (fl_fwd + fl_bwd + fl_dw) * rows / out["lib_update_ms"] / 1e9
        out["wide_update_ms"] = out["chain_plus_dw_ms"]
        out["wide_update_TFLOPs"] = (fl_fwd + fl_bwd + fl_dw) * rows / out["wide_update_ms"] / 1e9
        out["wide_dw_TFLOPs"] = fl_dw * rows / out["wide_dw_ms"] / 1e9
        out["update_speedup_vs_lib"] = out["lib_update_ms"] / out["wide_update_ms"]
        print(json.dumps(out), flush=True)
        del X, act, xw, xl


if __name__ == "__main__":
    main()
