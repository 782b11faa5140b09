#!/usr/bin/env python3
"""The transposed-convolution forward entry point on the four decoder levels of the benchmark graph (batch 16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi
dev = torch.device("cuda", 0); capi.unet_set_device(0)
N = 16
for ci, co, h, out_ld_mul in [(1024, 512, 16, 2), (512, 256, 32, 2), (256, 128, 64, 2), (128, 64, 128, 2)]:
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    w = (torch.randn(4 * co * ci, device=dev) * 0.02).bfloat16()
    bias = torch.zeros(co, device=dev)
    out = torch.empty(N, 2 * h, 2 * h, co * out_ld_mul, device=dev, dtype=torch.bfloat16)
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, h)
    stats = torch.zeros(4 * N * tiles * co * 2, device=dev)
    def run():
        capi.unet_tconv2x2_bf16(x.data_ptr(), ci, w.data_ptr(), bias.data_ptr(), co, out.data_ptr() + 2 * co * (out_ld_mul - 1),
                                co * out_ld_mul, stats.data_ptr(), N, h, h, ci, co, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * N * h * h * 4 * ci * co
    print(f"{ci:5d}->{co:<4d} {h:3d}x{h:<3d} {us:7.1f} us  {fl / us * 1e-6:6.0f} TFLOP/s")
