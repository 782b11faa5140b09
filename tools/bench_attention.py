#!/usr/bin/env python3
"""Times the attention gate (unet_attention_f32 / unet_attention_backward_f32) on one GPU at the size of the
full-resolution gate of an Attention U-Net (x = 16 x 64 x 256 x 256, g = 16 x 128 x 128 x 128, 32 intermediate
channels) and prints one JSON line with the achieved share of the HBM roofline (8 TB/s) from the algorithmic
bytes: forward reads x and g and writes int_1, int_2 and the output; backward reads grad_out, x, g, int_1 and
writes grad_x, grad_g and the ReLU-input gradient (plus the read-modify-write of grad_x at the even pixels)."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unet_amd import capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cx", type=int, default=64)
    ap.add_argument("--cg", type=int, default=128)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    n, cx, cg, ci, h = a.batch, a.cx, a.cg, a.cx // 2, a.size
    dev = "cuda:0"
    capi.unet_set_device(0)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((n, cx, h, h), device=dev, generator=g)
    gg = torch.randn((n, cg, h // 2, h // 2), device=dev, generator=g)
    params = torch.randn(((cx + cg + 2) * ci + 1,), device=dev, generator=g) * 0.1
    dy = torch.randn_like(x)
    int1 = torch.empty((n, ci, h // 2, h // 2), device=dev)
    alpha = torch.empty((n, h // 2, h // 2), device=dev)
    out, dx, dg = torch.empty_like(x), torch.empty_like(x), torch.empty_like(gg)
    g1, g2, dp = torch.empty_like(int1), torch.empty_like(alpha), torch.empty_like(params)
    p = lambda t: t.data_ptr()

    def fwd():
        capi.unet_attention_f32(p(int1), p(alpha), p(out), p(params), p(x), p(gg), n, h, h, cx, cg, ci, None)

    def bwd():
        capi.unet_attention_backward_f32(p(dy), p(int1), p(alpha), p(x), p(gg), p(params), p(g1), p(g2), p(dx), p(dg),
                                         p(dp), n, h, h, cx, cg, ci, None)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    ex, eg, ei, cells = x.numel(), gg.numel(), int1.numel(), alpha.numel()
    fwd_bytes = 4.0 * (2 * ex + eg + ei + cells)
    bwd_bytes = 4.0 * (3.5 * ex + 2 * eg + 2 * ei + 3 * cells)
    t_f, t_b = timed(fwd), timed(bwd)
    print(json.dumps({"op": "attention_gate_f32", "shape": {"N": n, "Cx": cx, "Cg": cg, "Ci": ci, "H": h, "W": h},
                      "forward_ms": round(t_f, 4), "backward_ms": round(t_b, 4),
                      "forward_GBs": round(fwd_bytes / t_f / 1e6, 1), "backward_GBs": round(bwd_bytes / t_b / 1e6, 1),
                      "forward_frac_of_8TBs": round(fwd_bytes / t_f / 1e6 / 8000, 3),
                      "backward_frac_of_8TBs": round(bwd_bytes / t_b / 1e6 / 8000, 3)}))


if __name__ == "__main__":
    main()
