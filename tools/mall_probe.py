#!/usr/bin/env python3
"""Cache-residency probe of the update's three kernel families (stage A).

The learner runs forward chain -> backward chain -> weight gradients over 2^22-row chunks: every stored activation / dZ makes a
round trip through HBM (6 GB per chunk >> the 256 MiB Infinity Cache).  This probe asks what the SAME launches cost when that
round trip is served on-die: it runs the three launches over sub-chunks of `--rows` rows (65,536 rows = 98 MB of activations +
98 MB of dZ live) in two arms inside one process, interleaved:
  warm: every iteration reuses ONE workspace (what a producer wrote is still in the Infinity Cache when its consumer reads it);
  cold: the iterations rotate through `--rotate` workspaces (> 1 GB in all: every read comes from HBM).
Same launch size on both arms, so the difference is HBM-vs-cache alone.  Per family: ns per row from per-launch HIP events.
Run once with the product library (non-temporal activation stores) and once with TG_NATIVE_LIB pointing at a
-DTG_ACT_STORE_NT=0 build (default-policy stores).  Prints one JSON line."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd import mlp as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[32768, 65536, 131072, 1 << 20])
ap.add_argument("--rotate", type=int, default=0, help="workspaces of the cold arm (0: enough for 1.5 GB)")
ap.add_argument("--iters", type=int, default=48)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--layers", type=int, default=5)
ap.add_argument("--arm", default="both", choices=["both", "warm", "cold"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
H, L = a.hidden, a.layers
net = tg.NeuralNetwork(20, 4, (H,) * L, "ReLU").to(dev)
for p in net.parameters():
    p.grad = torch.zeros_like(p)
mlp = M.GemmMLP(net, torch.bfloat16)
assert mlp.can_fuse_head()
var = torch.full((4,), 0.3)
out = {"lib": os.environ.get("TG_NATIVE_LIB", "product"), "hidden": H, "layers": L, "sizes": []}


def make_inputs(rows):
    xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
    return (xp, torch.randn(rows, 4, device=dev), -0.5 * torch.rand(rows, device=dev) - 1.0, torch.randn(rows, device=dev))


def one(inp, ws):
    mlp._ws = ws
    xp, act, lpo, adv = inp
    mlp.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=var, epsilon=0.2, surr_coef=-1.0 / xp.shape[0], kl_coef=0.5 / xp.shape[0])
    mlp.backward_fused()


def family_ns(events, rows):
    return sum(e0.elapsed_time(e1) for e0, e1, *_ in events) * 1e6 / (len(events) * rows)


for rows in a.rows:
    live_bytes = rows * (2 * (L - 2) * 2 * H + L * (H // 8) + 64 + 16 + 24)      # activations + dZ + mask bits + per-row inputs
    K = a.rotate or max(2, -(-(3 << 29) // live_bytes))
    sets = [(make_inputs(rows), M._Workspace()) for _ in range(K)]
    iters = max(K, a.iters) if rows <= 1 << 18 else max(4, a.iters // 8)
    res = {"rows": rows, "live_MB": live_bytes / 1e6, "rotate": K, "iters": iters, "warm": [], "cold": []}
    for inp, ws in sets:                                                            # allocate every workspace
        one(inp, ws)
    torch.cuda.synchronize()
    for rnd in range(a.rounds):
        for arm in ("warm", "cold"):
            if a.arm not in ("both", arm):
                continue
            pick = (lambda i: sets[0]) if arm == "warm" else (lambda i: sets[i % K])
            for i in range(4):
                one(*pick(i))
            mlp.fwd_events, mlp.dx_events, mlp.dw_events = [], [], []
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                one(*pick(i))
            e1.record()
            torch.cuda.synchronize()
            res[arm].append({"fwd_ns_row": family_ns(mlp.fwd_events, rows), "bwd_ns_row": family_ns(mlp.dx_events, rows),
                             "dw_ns_row": family_ns(mlp.dw_events, rows), "wall_ns_row": e0.elapsed_time(e1) * 1e6 / (iters * rows)})
            mlp.fwd_events = mlp.dx_events = mlp.dw_events = None
    for arm in ("warm", "cold"):
        if res[arm]:
            res[arm + "_median"] = {k: sorted(r[k] for r in res[arm])[len(res[arm]) // 2] for k in res[arm][0]}
    if res["warm"] and res["cold"]:
        res["warm_over_cold"] = {k: res["warm_median"][k] / res["cold_median"][k] for k in res["warm_median"]}
    out["sizes"].append(res)
    del sets
    torch.cuda.empty_cache()
print(json.dumps(out))
