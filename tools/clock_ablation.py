#!/usr/bin/env python3
"""Cycles, not microseconds: every update kernel (and the forward chain with and without its activation stores) run back to back
for ~1 s each on the bench's net (20-256x5-4, 2^22 rows), with its in-kernel clock (tg_clock_probe_attach), the socket power
and the duration per launch.  cycles per launch = GHz x us: two builds / variants that differ in time but not in cycles differ
by the clock the package grants them (energy); variants that differ in cycles differ by their schedule.
    python tools/clock_ablation.py [--rows 4194304] [--seconds 1.0]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import trajopt_grpo_amd as tg  # noqa: E402
from trajopt_grpo_amd import mlp as M  # noqa: E402
from bench import PowerSampler, clock_probe_result, sustained_mfma_peak  # noqa: E402

N_ = tg._native


def run(name, fn, family, seconds, sampler, dev, rows, flop_row):
    lib = N_.load()
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    one = a.elapsed_time(b)
    n_warm, n = max(3, int(500.0 * seconds / one)), max(3, int(500.0 * seconds / one))
    for _ in range(n_warm):
        fn()
    torch.cuda.synchronize()
    probe = torch.zeros(N_.TG_CLOCK_PROBE_U64, dtype=torch.int64, device=dev)
    if family is not None:
        N_.check(lib.tg_clock_probe_attach(family, probe.data_ptr()))
    t0 = time.perf_counter()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if family is not None:
        N_.check(lib.tg_clock_probe_attach(family, None))
    us = a.elapsed_time(b) * 1e3 / n
    ghz, wgs = clock_probe_result(probe) if family is not None else (None, 0)
    pw = sampler.window(t0, t1)
    out = {"what": name, "us_per_launch": us, "clock_GHz": ghz, "Mcycles_per_launch": ghz * us * 1e-3 if ghz else None,
           "cycles_per_row": ghz * us * 1e3 / rows if ghz else None, "TFLOPs": flop_row * rows / us / 1e6 if flop_row else None,
           "power_W_mean": pw["mean"] if pw else None, "power_W_max": pw["max"] if pw else None, "launches": n}
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 22)
    ap.add_argument("--seconds", type=float, default=1.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rows, H, L = a.rows, 256, 5
    net = tg.NeuralNetwork(20, 4, (H,) * L, "ReLU").to(dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    mlp = M.GemmMLP(net, torch.bfloat16)
    xp = mlp.prepare_input(torch.randn(rows, 20, device=dev))
    act = torch.randn(rows, 4, device=dev)
    lpo = (-0.5 * torch.rand(rows, device=dev) - 1.0)
    adv = torch.randn(rows, device=dev)
    var = torch.full((4,), 0.3)
    sampler = PowerSampler(torch, 0).start()
    w_all = sum(l.weight.numel() for l in mlp.linears)
    w_first = mlp.linears[0].weight.numel()
    res = []
    res.append(run("forward chain, no stores (no-grad pass)", lambda: mlp.forward(xp, keep=False, padded=True), N_.TG_PROBE_FWD_CHAIN_PLAIN,
                   a.seconds, sampler, dev, rows, 2.0 * w_all))
    res.append(run("forward chain, activations + mask bits stored (plain training pass)", lambda: mlp.forward(xp, keep=True, padded=True),
                   N_.TG_PROBE_FWD_CHAIN_PLAIN, a.seconds, sampler, dev, rows, 2.0 * w_all))

    def fwd_loss():
        mlp.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=var, epsilon=0.2, surr_coef=-1.0 / rows, kl_coef=0.5 / rows)
    res.append(run("forward chain + loss head (the learner's pass)", fwd_loss, N_.TG_PROBE_FWD_CHAIN, a.seconds, sampler, dev, rows, 2.0 * w_all))
    # backward chain + weight gradients, as the learner runs them (one forward_loss in front of each: its outputs are their inputs)
    fwd_loss()
    acts, bits, dzh = mlp._acts, mlp._bits, mlp._dz_head

    def bwd_dw():
        mlp._acts, mlp._bits, mlp._dz_head = acts, bits, dzh
        mlp.backward_fused()
    # (the two launches share a stream: each family's probe sees its own kernel only)
    res.append(run("backward chain (+ weight gradients behind it)", bwd_dw, N_.TG_PROBE_BWD_CHAIN, a.seconds, sampler, dev, rows, None))
    res.append(run("weight gradients (+ backward chain in front)", bwd_dw, N_.TG_PROBE_WEIGHT_GRAD, a.seconds, sampler, dev, rows, None))
    s = sustained_mfma_peak(tg, torch, dev, 0, sampler)
    print(json.dumps({"what": "bare MFMA + LDS loop", **{k: v for k, v in s.items() if k != "what"}}), flush=True)
    sampler.stop()


if __name__ == "__main__":
    main()
