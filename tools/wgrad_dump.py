#!/usr/bin/env python3
"""Writes the weight gradients of a few layer shapes (3x3 and 2x2-transposed) to an .npz: run once with
UNET_LAB_WGRAD_DMA=0 (the register-staged kernel) and once without, then compare the two files bit for bit (tools/wgrad_dump.py a.npz b.npz)."""
import os
import sys
import time
import numpy as np

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("arrays:", len(a.files), "differing:", bad)
    sys.exit(1 if bad else 0)

import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from unet_amd import capi
dev = "cuda:0"
capi.unet_set_device(0)
rng = np.random.default_rng(3)
out = {}
shapes = [(16, 64, 64, 256, 256), (16, 128, 64, 256, 256), (16, 128, 128, 128, 128), (16, 256, 256, 64, 64), (16, 512, 512, 32, 32),
          (16, 1024, 1024, 16, 16), (2, 8, 64, 40, 24), (3, 96, 32, 17, 33)]
for (n, ci, co, h, w) in shapes:
    cip = max(ci, 8) if ci < 32 else (ci + 31) // 32 * 32
    cop = (co + 31) // 32 * 32
    x = torch.from_numpy(rng.standard_normal((n, h, w, cip)).astype(np.float32)).to(dev).bfloat16()
    dz = torch.from_numpy(rng.standard_normal((n, h, w, cop)).astype(np.float32)).to(dev).bfloat16()
    g = torch.zeros(ci * co * 9, device=dev)
    capi.unet_conv3x3_wgrad_bf16(dz.data_ptr(), cop, x.data_ptr(), cip, g.data_ptr(), n, h, w, cip, cop, ci, co, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        capi.unet_conv3x3_wgrad_bf16(dz.data_ptr(), cop, x.data_ptr(), cip, g.data_ptr(), n, h, w, cip, cop, ci, co, None)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"3x3 {ci:4d}->{co:4d} {h}x{w} n={n}: {us:7.1f} us  {2.0 * n * h * w * ci * co * 9 / us / 1e6:7.1f} TFLOP/s", flush=True)
    out[f"c3_{n}_{ci}_{co}_{h}_{w}"] = g.cpu().numpy()
for (n, ci, co, h, w) in [(16, 128, 64, 128, 128), (16, 1024, 512, 16, 16), (2, 64, 32, 9, 20)]:
    x = torch.from_numpy(rng.standard_normal((n, h, w, ci)).astype(np.float32)).to(dev).bfloat16()
    dy = torch.from_numpy(rng.standard_normal((n, 2 * h, 2 * w, co)).astype(np.float32)).to(dev).bfloat16()
    g = torch.zeros(ci * co * 4, device=dev)
    capi.unet_tconv2x2_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, g.data_ptr(), n, h, w, ci, co, ci, co, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        capi.unet_tconv2x2_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, g.data_ptr(), n, h, w, ci, co, ci, co, None)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"2x2T {ci:4d}->{co:4d} {h}x{w} n={n}: {us:7.1f} us", flush=True)
    out[f"t2_{n}_{ci}_{co}_{h}_{w}"] = g.cpu().numpy()
np.savez(sys.argv[1], **out)
