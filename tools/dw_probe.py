"""Times tg_mlp_weight_grad (all weight gradients of one net in one launch) at the C3 shape against the split-K batched
GEMM path it replaces.  Prints one JSON line.
usage: python tools/dw_probe.py [--rows N] [--hidden H] [--layers L] [--no-recompute] [--iters K] [--no-gemm]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import _native as N
from trajopt_grpo_amd import mlp as M

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 22)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--layers", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--no-recompute", action="store_true", help="read the stored first activation (kind HH) instead of recomputing it (HR)")
ap.add_argument("--no-rh", action="store_true", help="read a stored top-layer dZ (kind HH) instead of rebuilding it from the head gradient (RH)")
ap.add_argument("--dh", action="store_true", help="include the head's job (kind DH); the learner forms that gradient inside the forward chain")
ap.add_argument("--hx", action="store_true", help="include the first layer's job (kind HX); the learner forms that gradient inside the backward chain")
ap.add_argument("--no-gemm", action="store_true", help="skip the split-K GEMM comparison (profiling runs)")
a_ = ap.parse_args()
rows, H, nh, recompute = a_.rows, a_.hidden, a_.layers, 0 if a_.no_recompute else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = tg.NeuralNetwork(20, 4, (H,) * nh, "ReLU").to(dev)
for p in net.parameters():
    p.grad = torch.zeros_like(p)
mlp = M.GemmMLP(net, torch.bfloat16)
xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
keep_chain, mlp._bchain = mlp._bchain, None       # store every activation (the GEMM path reads the first one)
mlp.forward(xp, keep=True)
mlp._bchain = keep_chain
acts = mlp._acts
dzs = [(torch.randn(rows, H, device=dev) * (torch.rand(rows, H, device=dev) > 0.4)).to(torch.bfloat16) for _ in range(nh)]
dh = torch.zeros(rows, 8, device=dev, dtype=torch.bfloat16)
dh[:, :4] = torch.randn(rows, 4, device=dev)
lin = mlp.linears
ws = M.weight_grad_workspace(H, dev)
jobs = [(N.TG_DW_DH, dh, acts[nh], lin[nh].weight.grad, None)] if a_.dh else []
for i in range(nh - 1, 0, -1):
    if i == nh - 1 and not a_.no_rh and nh >= 3:
        mlp.forward(xp, keep=True)                # (chain mode: mask bits; the activations are the same)
        jobs.append((N.TG_DW_RH, dh, acts[i], lin[i].weight.grad, lin[i].bias.grad, mlp._bits[nh]))
    elif i == 1 and recompute:
        jobs.append((N.TG_DW_HR, dzs[i], xp, lin[i].weight.grad, lin[i].bias.grad))
    else:
        jobs.append((N.TG_DW_HH, dzs[i], acts[i], lin[i].weight.grad, lin[i].bias.grad))
if a_.hx:
    jobs.append((N.TG_DW_HX, dzs[0], xp, lin[0].weight.grad, lin[0].bias.grad))
bytes_per_row = sum({N.TG_DW_HH: 4 * H, N.TG_DW_HX: 2 * H + 64, N.TG_DW_HR: 2 * H + 64, N.TG_DW_DH: 2 * H + 16,
                     N.TG_DW_RH: 2 * H + 16 + H // 8}[j[0]] for j in jobs)


def ours():
    M.weight_grad(H, jobs, rows, ws, mlp._chain.stream, mlp._chain.bias[0], mlp._bchain.stream)


def gemms():
    if a_.dh:
        mlp._dw_into(lin[nh].weight.grad, dh, acts[nh])
    for i in range(nh - 1, -1 if a_.hx else 0, -1):
        mlp._dw_into(lin[i].weight.grad, dzs[i], acts[i])


def timeit(fn, n=a_.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# interleaved rounds in one process (two rounds each)
t = {"tg_mlp_weight_grad": [], "split_k_gemms": []}
for _ in range(2):
    t["tg_mlp_weight_grad"].append(timeit(ours))
    if not a_.no_gemm:
        t["split_k_gemms"].append(timeit(gemms))
# correctness at this size: one layer against an fp32 GEMM of the same operands
for p in net.parameters():
    p.grad.zero_()
ours()
ref = dzs[2].float().t() @ acts[2].float()
err = float((lin[2].weight.grad - ref).norm() / ref.norm())
ms = min(t["tg_mlp_weight_grad"])
print(json.dumps({"rows": rows, "H": H, "hidden_layers": nh, "recompute_first_activation": bool(recompute), "bytes_per_row": bytes_per_row,
                  "ms": t, "GBps": bytes_per_row * rows / ms / 1e6, "frac_of_8TBps": bytes_per_row * rows / ms / 1e6 / 8000.0,
                  "gemm_GBps": (bytes_per_row + (448 if recompute else 0)) * rows / min(t["split_k_gemms"]) / 1e6 if t["split_k_gemms"] else None,
                  "rel_err_layer2_vs_fp32_gemm": err}))
