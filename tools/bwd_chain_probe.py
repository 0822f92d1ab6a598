#!/usr/bin/env python3
"""Time the learner's backward pass on the bench's actor shape: tg_mlp_backward_chain (all hidden layers' dZ in one
launch), as the learner runs it: no bias column sums (tg_mlp_weight_grad forms them); --bias times the variant with them."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd.mlp import GemmMLP  # noqa: E402

N = tg._native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1 << 20, 1 << 22])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bias", action="store_true")
    ap.add_argument("--no-fuse-w0", action="store_true", help="write the bottom layer's dZ instead of forming dW0 inside the chain")
    ap.add_argument("--store-top", action="store_true", help="also write the top layer's dZ (the learner leaves it out: kind RH of tg_mlp_weight_grad rebuilds it)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = N.load()
    net = tg.NeuralNetwork(20, 4, (256,) * 5, "ReLU").to(dev)
    mlp = GemmMLP(net, torch.bfloat16)
    nh, H = 5, 256
    res = []
    for rows in a.rows:
        xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
        mlp.forward(xp, keep=True)
        bits = mlp._bits
        dzh = torch.zeros(rows, 8, dtype=torch.bfloat16, device=dev)
        dzh[:, :4] = (torch.randn(rows, 4, device=dev) * 1e-3).bfloat16()
        dzs = [torch.empty(rows, H, dtype=torch.bfloat16, device=dev) for _ in range(nh)]
        part = torch.empty(lib.tg_mlp_backward_chain_blocks(), nh, H, dtype=torch.float32, device=dev)
        top = a.store_top or a.bias
        dz_ptrs = (N.C.c_void_p * nh)(*[(t.data_ptr() if (j > 0 or top) else None) for j, t in enumerate(dzs)])
        m_ptrs = (N.C.c_void_p * nh)(*[bits[nh - j].data_ptr() for j in range(nh)])
        st = N.stream_ptr(dev)
        fuse0 = not (a.no_fuse_w0 or a.bias)
        if fuse0:
            slabs = torch.empty(2 * lib.tg_mlp_backward_chain_blocks() * H * 32, dtype=torch.float32, device=dev)
            nsl = N.C.c_int32(0)
            dz_ptrs = (N.C.c_void_p * nh)(*[(t.data_ptr() if ((j > 0 or top) and j < nh - 1) else None) for j, t in enumerate(dzs)])
            run = lambda: N.check(lib.tg_mlp_backward_chain_w0(dzh.data_ptr(), mlp._bchain.stream.data_ptr(), H, nh, rows, dz_ptrs, m_ptrs,
                                                               xp.data_ptr(), slabs.data_ptr(), slabs.numel(), N.C.byref(nsl), st))
        else:
            run = lambda: N.check(lib.tg_mlp_backward_chain(dzh.data_ptr(), mlp._bchain.stream.data_ptr(), H, nh, rows, dz_ptrs, m_ptrs,
                                                            part.data_ptr() if a.bias else None, st))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        bpr = 16 + nh * (H // 8) + ((nh if top else nh - 1) - (1 if fuse0 else 0)) * 2 * H + (64 if fuse0 else 0)
        res.append({"rows": rows, "bias_sums": bool(a.bias), "chain_us": us, "bytes_per_row": bpr, "GBps": bpr * rows / us / 1e3,
                    "frac_of_8TBps": bpr * rows / us / 1e3 / 8000, "TFLOPs": 2.0 * rows * (32 * H + (nh - 1) * H * H) / us / 1e6})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
