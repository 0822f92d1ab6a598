#!/usr/bin/env python3
"""Times the fp32 chain learner's launches (tg_mlp_f32_forward, tg_mlp_f32_forward_backward, tg_mlp_f32_weight_grad) against the
per-layer hipBLASLt path of the same GemmMLP, at fixed row counts.  Prints one JSON line per (shape, rows).
usage: python tools/f32_chain_probe.py [--rows N ...] [--iters K] [--no-gemm]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd import mlp as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[176584, 1 << 20])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-gemm", action="store_true")
ap.add_argument("--shapes", default="5:1:128x2,5:1:128x4,20:4:128x3,5:1:64x2")
ap.add_argument("--no-res", action="store_true", help="H = 128 nets with <= 2 hidden layers through the 32-row chain kernel (rounds 2-4) instead of the resident 16-row kernel")
a = ap.parse_args()
if a.no_res:
    M.f32_res_supported = lambda net: 0
dev = torch.device("cuda", 0)


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


for spec in a.shapes.split(","):
    S, A, hw = spec.split(":")
    S, A = int(S), int(A)
    H, nh = (int(v) for v in hw.split("x"))
    torch.manual_seed(0)
    net = tg.NeuralNetwork(S, A, (H,) * nh, "ReLU").to(dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    m = M.GemmMLP(net, torch.float32)
    assert m._f32 is not None
    for rows in a.rows:
        X = torch.randn(rows, S, device=dev)
        xp = m.prepare_input(X)
        act = torch.randn(rows, A, device=dev)
        lpo = -0.5 * torch.rand(rows, device=dev) - 1.0
        adv = torch.randn(rows, device=dev)
        var = torch.full((A,), 0.3)
        fl = lambda: m.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=var, epsilon=0.2, surr_coef=-1.0 / rows, kl_coef=0.5 / rows)
        t_fwd = timeit(lambda: m.forward(xp, keep=False, padded=True), a.iters)
        t_fb = timeit(fl, a.iters)
        fl()
        saved = (m._acts, m._bits, m._dz_head, m._tmask)

        def dw():
            m._acts, m._bits, m._dz_head, m._tmask = saved
            m._backward_fused_f32()
        t_dw = timeit(dw, a.iters)
        flop_fwd = 2.0 * (H * m.in_pad + (nh - 1) * H * H)
        flop_fb = 2.0 * H * m.in_pad + 4.0 * (nh - 1) * H * H
        flop_dw = 2.0 * (nh - 1) * H * H + 2.0 * H * 32
        out = {"shape": spec, "rows": rows, "forward_us": t_fwd, "forward_backward_us": t_fb, "weight_grad_us": t_dw,
               "forward_TFLOPs": flop_fwd * rows / t_fwd / 1e6, "forward_backward_TFLOPs": flop_fb * rows / t_fb / 1e6,
               "weight_grad_TFLOPs": flop_dw * rows / t_dw / 1e6,
               "note": "matrix-core flops only; fp32 matrix peak 157.3 TFLOP/s; host-side slab sums included in forward_backward"}
        if not a.no_gemm:
            g = M.GemmMLP(net, torch.float32)
            g._f32 = None                       # per-layer hipBLASLt GEMMs + glue kernels (what fp32 nets ran through round 2)
            g.in_pad = M._round_up(S, 32)
            g.w[0] = torch.zeros(H, g.in_pad, device=dev)
            xg = g.prepare_input(X)
            gout = torch.randn(rows, A, device=dev)

            def gemm_update():
                g.forward(xg, keep=True)
                g.backward(gout)
            out["gemm_forward_us"] = timeit(lambda: g.forward(xg, keep=False), a.iters)
            out["gemm_forward_backward_weight_grad_us"] = timeit(gemm_update, a.iters)
        print(json.dumps(out), flush=True)
