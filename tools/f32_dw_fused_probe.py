#!/usr/bin/env python3
"""Times tg_mlp_f32_weight_grad's fused job (rebuilt operands + riders) of a 5-128-128-1 net at a fixed row count; with STAMPS=1 and
TG_NATIVE_LIB=scratch/libtg_dwstamps.so (tools/build_probe_libs.sh) prints the per-phase cycle counts of the stage loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import mlp as M
dev = torch.device("cuda", 0)
rows = int(os.environ.get("ROWS", 176584))
shape = os.environ.get("SHAPE", "5:1:128x2")
S, A, hw = shape.split(":"); S, A = int(S), int(A); H, nh = (int(v) for v in hw.split("x"))
torch.manual_seed(0)
net = tg.NeuralNetwork(S, A, (H,) * nh, "ReLU").to(dev)
for p in net.parameters():
    p.grad = torch.zeros_like(p)
m = M.GemmMLP(net, torch.float32)
X = torch.randn(rows, S, device=dev); xp = m.prepare_input(X)
act = torch.randn(rows, A, device=dev); lpo = -0.5 * torch.rand(rows, device=dev) - 1.0; adv = torch.randn(rows, device=dev)
m.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=torch.full((A,), 0.3), epsilon=0.2, surr_coef=-1.0 / rows, kl_coef=0.5 / rows)
saved = (m._acts, m._bits, m._dz_head, m._tmask)
def dw():
    m._acts, m._bits, m._dz_head, m._tmask = saved
    m._backward_fused_f32()
for _ in range(3): dw()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dw()
e1.record(); torch.cuda.synchronize()
print(shape, rows, "weight_grad %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
if os.environ.get("STAMPS"):
    import ctypes, numpy as np
    raw = ctypes.CDLL(os.environ["TG_NATIVE_LIB"])
    buf = (ctypes.c_ulonglong * (4096 * 12))()
    assert raw.tg_debug_f32_stamps3(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 12).astype(np.float64)
    a = a[a[:, 6] > 0]
    per = a[:, :6] / a[:, 6:7]
    print("waves", len(a), "stages/wave %.1f" % a[:, 6].mean())
    print("cycles per stage: wait+barrier %.0f  issue %.0f  phase1 %.0f  barrier %.0f  operand reads %.0f  products %.0f  | sum %.0f" % (*per.mean(0), per.mean(0).sum()))
    nw = int(os.environ.get("WAVES", 8))                        # waves per workgroup of the kernel that ran (8: the 8-wave fused job)
    for w in range(nw):
        print(" wave", w, np.round(per[w::nw].mean(0)))
    t0 = a[:, 7].min()
    print("memtime: entry spread %.0f, entry->loop %.0f, loop %.0f, loop end->exit %.0f, last exit - first entry %.0f cycles" % (
        a[:, 7].max() - t0, (a[:, 8] - a[:, 7]).mean(), (a[:, 9] - a[:, 8]).mean(), (a[:, 10] - a[:, 9]).mean(), a[:, 10].max() - t0))
    clk = ((a[:, 10] - a[:, 7]) / (a[:, 11] * 10.0))
    print("per-wave kernel time (100 MHz realtime) mean %.1f us max %.1f us; in-kernel clock %.3f GHz" % (a[:, 11].mean() / 100.0, a[:, 11].max() / 100.0, clk.mean()))
