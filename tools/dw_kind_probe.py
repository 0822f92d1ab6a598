#!/usr/bin/env python3
"""Per-kind cost of tg_mlp_weight_grad: one launch with a SINGLE job of each kind over 2^22 rows (all CUs on that job) ->
microseconds per 32-row stage and CU, and the job's byte rate.  Shows which kinds are bound by bytes and which by their own
instruction stream (the recomputing kinds HR / RH).

    python3 tools/dw_kind_probe.py [--rows N] [--hidden 256]
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import mlp as M, _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 22)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only", default=None, help="one kind (HH, HX, DH, HR, RH): for counter runs")
a = ap.parse_args()
rows, H, nh = a.rows, a.hidden, 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = tg.NeuralNetwork(20, 4, (H,) * nh, "ReLU").to(dev)
mlp = M.GemmMLP(net, torch.bfloat16)
xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
keep, mlp._bchain = mlp._bchain, None
mlp.forward(xp, keep=True)
acts = list(mlp._acts)
mlp._bchain = keep
mlp.forward(xp, keep=True)
bits = mlp._bits
mlp._fresh("bchain")
dz = (torch.randn(rows, H, device=dev) * (torch.rand(rows, H, device=dev) > 0.4)).to(torch.bfloat16)
dh = torch.zeros(rows, 8, device=dev, dtype=torch.bfloat16)
dh[:, :4] = torch.randn(rows, 4, device=dev)
w = torch.zeros(H, H, device=dev); b = torch.zeros(H, device=dev)
wx = torch.zeros(H, 32, device=dev); wd = torch.zeros(8, H, device=dev)
ws = M.weight_grad_workspace(H, dev)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
kinds = {"HH": ([(N.TG_DW_HH, dz, acts[2], w, b)], 4 * H), "HX": ([(N.TG_DW_HX, dz, xp, wx, b)], 2 * H + 64),
         "DH": ([(N.TG_DW_DH, dh, acts[3], wd, None)], 2 * H + 16), "HR": ([(N.TG_DW_HR, dz, xp, w, b)], 2 * H + 64),
         "RH": ([(N.TG_DW_RH, dh, acts[2], w, b, bits[3])], 2 * H + 16 + H // 8)}
out = {}
for name, (jobs, bpr) in kinds.items():
    if a.only and name != a.only:
        continue
    run = lambda: M.weight_grad(H, jobs, rows, ws, mlp._chain.stream, mlp._chain.bias[0], mlp._bchain.stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    out[name] = {"ms": round(ms, 3), "us_per_stage_and_cu": round(ms * 1e3 * cus / (rows / 32), 3), "TBps": round(bpr * rows / ms / 1e9, 2)}
print(json.dumps(out))
