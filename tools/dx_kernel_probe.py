#!/usr/bin/env python3
"""Time tg_dx_relu_bias against the two passes it replaces (hipBLASLt dA = dZ @ W, then tg_relu_bwd_bias)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg  # noqa: E402

N = tg._native


def pack_mask_bits(act):
    """[rows][H] activations -> [rows][H/32] int32 mask words in tg_mlp_forward_chain's layout (include/trajopt_grpo_hip.h)."""
    rows, H = act.shape
    f = torch.arange(H, device=act.device)
    mt, h, r = f >> 5, (f >> 4) & 1, f & 15
    word = h * (H // 64) + (mt >> 1)
    bit = (mt & 1) * 8 + (r >> 1) + 16 * (r & 1)
    out = torch.zeros(rows, H // 32, dtype=torch.int64, device=act.device)
    out.index_add_(1, word, (act > 0).to(torch.int64) << bit)
    return torch.where(out >= 2 ** 31, out - 2 ** 32, out).to(torch.int32)          # bit 31 = the sign of the int32 word


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1 << 20, 1 << 22])
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bits", action="store_true", help="ReLU masks as 1 bit per activation (tg_mlp_forward_chain layout)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = N.load()
    H = a.width
    out = []
    for rows in a.rows:
        dz = (torch.randn(rows, H, device=dev) * 0.5).bfloat16()
        W = (torch.randn(H, H, device=dev) / H ** 0.5).bfloat16()
        act = torch.relu(torch.randn(rows, H, device=dev)).bfloat16()
        frag = torch.empty(H * H, dtype=torch.bfloat16, device=dev)
        N.check(lib.tg_dx_pack_weights(W.data_ptr(), frag.data_ptr(), H, H, N.stream_ptr(dev)))
        dzo = torch.empty_like(act)
        part = torch.empty(lib.tg_dx_relu_bias_blocks(), H, dtype=torch.float32, device=dev)
        part2 = torch.empty(lib.tg_relu_bwd_bias_blocks(), H, dtype=torch.float32, device=dev)
        st = N.stream_ptr(dev)

        bits_ptr = None
        if a.bits:
            bits_ptr = pack_mask_bits(act).data_ptr()

        def fused():
            N.check(lib.tg_dx_relu_bias(dz.data_ptr(), frag.data_ptr(), act.data_ptr(), bits_ptr, dzo.data_ptr(), rows, H, H,
                                        part.data_ptr(), st))

        da = torch.empty_like(act)

        def two_pass():
            torch.mm(dz, W, out=da)
            N.check(lib.tg_relu_bwd_bias(da.data_ptr(), act.data_ptr(), rows, H, 1, part2.data_ptr(), st))

        tf, t2 = timed(fused, a.iters), timed(two_pass, a.iters)
        bytes_alg = rows * H * 2 * 3 if not a.bits else rows * (H * 2 * 2 + H // 8)
        out.append({"rows": rows, "width": H, "fused_us": tf, "two_pass_us": t2, "fused_GBps": bytes_alg / tf / 1e3,
                    "fused_frac_of_8TBps": bytes_alg / tf / 1e3 / 8000, "fused_TFLOPs": 2.0 * rows * H * H / tf / 1e6,
                    "speedup": t2 / tf})
        del dz, act, dzo, da
    print(json.dumps(out))


if __name__ == "__main__":
    main()
