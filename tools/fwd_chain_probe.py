#!/usr/bin/env python3
"""Time tg_mlp_forward_chain (all layers, one launch) against the per-layer GEMM chain on the bench's actor shape."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd.mlp import GemmMLP  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1 << 20, 1 << 22])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--fused-head", action="store_true",
                    help="time tg_mlp_forward_chain_loss (the learner's training pass: loss head + head gradient inside, no top activation store) only")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    net = tg.NeuralNetwork(20, 4, (a.hidden,) * a.layers, "ReLU").to(dev)
    mlp = GemmMLP(net, torch.bfloat16)
    H, L = a.hidden, a.layers
    flop_row = 2.0 * (32 * H + (L - 1) * H * H + H * 32)
    res = []
    for rows in a.rows:
        xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
        out = {"rows": rows}
        if a.fused_head:
            for p in net.parameters():
                p.grad = torch.zeros_like(p)
            act = torch.randn(rows, 4, device=dev)
            lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0)
            adv = torch.randn(rows, device=dev)
            var = torch.full((4,), 0.3)
            t_c = timed(lambda: mlp.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=var, epsilon=0.2, surr_coef=-1.0 / rows,
                                                 kl_coef=0.5 / rows), a.iters)
            bpr = 64 + (L - 2) * 2 * H + L * (H // 8) + 16 + 24
            out.update(chain_keep_us=t_c, chain_nokeep_us=0.0, bytes_per_row=bpr, fused_head=True, chain_keep_GBps=rows * bpr / t_c / 1e3,
                       chain_keep_frac_of_8TBps=rows * bpr / t_c / 1e3 / 8000, chain_keep_TFLOPs=flop_row * rows / t_c / 1e6,
                       note="includes the host-side sums of the head slabs (three small torch kernels per call)")
            res.append(out)
            continue
        for keep in (True, False):
            chain = mlp._chain
            t_c = timed(lambda: mlp.forward(xp, keep=keep, padded=True), a.iters)
            stored = sum(1 for t in (mlp._acts or [None])[1:] if t is not None)
            mlp._chain = None
            t_l = timed(lambda: mlp.forward(xp, keep=keep, padded=True), a.iters)
            mlp._chain = chain
            k = "keep" if keep else "nokeep"
            out[f"chain_{k}_us"], out[f"layers_{k}_us"] = t_c, t_l
            out[f"chain_{k}_TFLOPs"] = flop_row * rows / t_c / 1e6
            if keep:
                # as the learner runs it: the first activation is not stored (tg_mlp_weight_grad recomputes it)
                bpr = 64 + stored * 2 * H + L * (H // 8) + 32
                out["bytes_per_row"], out["stored_activations"] = bpr, stored
                out["chain_keep_GBps"] = rows * bpr / t_c / 1e3
                out["chain_keep_frac_of_8TBps"] = rows * bpr / t_c / 1e3 / 8000
        res.append(out)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
