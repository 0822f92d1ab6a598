import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import _native as N
dev = torch.device("cuda", 0)
lib = N.load()
H, rows = int(os.environ.get("H", 128)), int(os.environ.get("ROWS", 1 << 20))       # H=256: the 8-wave job kernel of mlp_f32_wide.hip
XC = 8 if H != 256 else 24
ws = torch.empty(lib.tg_mlp_f32_weight_grad_workspace(H) // 4, device=dev)
dz = torch.randn(rows, H, device=dev); a = torch.randn(rows, H, device=dev); x = torch.randn(rows, XC, device=dev); g = torch.randn(rows, 4, device=dev)
wg = torch.zeros(H, H, device=dev); bg = torch.zeros(H, device=dev); w0 = torch.zeros(H, 5 if H != 256 else 20, device=dev); wh = torch.zeros(1, H, device=dev); bh = torch.zeros(1, device=dev)
def job(kind, p, q, ncols, w, b, m_out, n_out):
    j = N.F32DwJob(); j.d_p, j.d_q, j.d_wgrad, j.d_bgrad = p.data_ptr(), q.data_ptr(), w.data_ptr(), b.data_ptr()
    j.wgrad_ld, j.kind, j.n_cols, j.m_out, j.n_out = w.stride(0), kind, ncols, m_out, n_out
    return j
J = {"mm": job(0, dz, a, H, wg, bg, H, H), "x": job(0, dz, x, XC, w0, bg, H, 5 if H != 256 else 20), "head": job(1, g, a, H, wh, bh, 1, H)}
def run(names):
    arr = (N.F32DwJob * len(names))(*[J[n] for n in names])
    def f():
        N.check(lib.tg_mlp_f32_weight_grad(H, arr, len(names), rows, ws.data_ptr(), ws.numel() * 4, None, 0, None, N.stream_ptr(dev)))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
sel = os.environ.get("JOBS")
for names in ([sel.split(",")] if sel else (["mm"], ["x"], ["head"], ["mm", "mm"], ["x", "head"], ["mm", "x", "head"], ["mm"] * 4, ["mm"] * 4 + ["x", "head"])):
    print(names, "%.0f us" % run(names), flush=True)

if os.environ.get("STAMPS"):
    import ctypes, numpy as np
    raw = ctypes.CDLL(os.environ["TG_NATIVE_LIB"])
    buf = (ctypes.c_ulonglong * (4096 * 4))()
    torch.cuda.synchronize()
    assert raw.tg_debug_f32_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
    a = a[a[:, 3] > 0]
    per = a[:, :3] / a[:, 3:4]
    print("waves", len(a), "stages/wave", a[:, 3].mean(), "cycles per stage: wait+bias %.0f  arrive %.0f  products+reads %.0f" % tuple(per.mean(0)))
    for w in range(4):
        print(" wave", w, per[w::4].mean(0))

    buf2 = (ctypes.c_ulonglong * (4096 * 6))()
    assert raw.tg_debug_f32_stamps2(buf2) == 0
    b = np.frombuffer(buf2, dtype=np.uint64).reshape(-1, 6).astype(np.float64)
    b = b[b[:, 3] > 0]
    t0 = b[:, 0].min(); r0 = b[:, 4].min()
    print("waves", len(b), "memtime: entry spread %.0f, loop start-entry %.0f, loop %.0f, exit-loop end %.0f, last exit - first entry %.0f cycles" % (
        b[:, 0].max() - t0, (b[:, 1] - b[:, 0]).mean(), (b[:, 2] - b[:, 1]).mean(), (b[:, 3] - b[:, 2]).mean(), b[:, 3].max() - t0))
    print("realtime (100 MHz): last exit - first entry %.1f us; clock = %.3f GHz" % ((b[:, 5].max() - r0) / 100.0, (b[:, 3].max() - t0) / ((b[:, 5].max() - r0) * 10.0)))
