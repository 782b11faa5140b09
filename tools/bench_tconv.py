#!/usr/bin/env python3
"""The transposed-convolution entry points (forward with statistics, data gradient with the fused reduction, weight
gradient) on the four decoder levels of the benchmark graph (batch 16), each alone on the device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi
if os.environ.get("UNET_LAB_LIB"):  # a laboratory build of the library (tools/bin/libunet_hip_*.so) instead of the one in the tree
    capi.LIB_PATH = os.path.join(ROOT, os.environ["UNET_LAB_LIB"])
dev = torch.device("cuda", 0); capi.unet_set_device(0)
N = 16


def timed(run):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


for ci, co, h in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    wf = (torch.randn(4 * co * ci, device=dev) * 0.02).bfloat16()
    wb = (torch.randn(4 * co * ci, device=dev) * 0.02).bfloat16()
    bias = torch.zeros(co, device=dev)
    out = torch.empty(N, 2 * h, 2 * h, 2 * co, device=dev, dtype=torch.bfloat16)   # into the second half of a concat buffer
    dy = torch.randn(N, 2 * h, 2 * h, co, device=dev).bfloat16()
    dx = torch.empty(N, h, h, ci, device=dev, dtype=torch.bfloat16)
    z = torch.randn(N, h, h, ci, device=dev).bfloat16()
    coef = torch.rand(ci * 4, device=dev) + 0.5
    tiles = capi.lib().unet_conv3x3_stat_tiles(h, h)
    stats = torch.zeros(4 * N * tiles * co * 2, device=dev)
    parts = torch.zeros(N * tiles * ci * 2, device=dev)
    gw = torch.zeros(ci * co * 4, device=dev)
    fwd = timed(lambda: capi.unet_tconv2x2_bf16(x.data_ptr(), ci, wf.data_ptr(), bias.data_ptr(), co, out.data_ptr() + 2 * co,
                                                2 * co, stats.data_ptr(), N, h, h, ci, co, None))
    dg = timed(lambda: capi.unet_tconv2x2_dgrad_bnred_bf16(dy.data_ptr(), co, wb.data_ptr(), dx.data_ptr(), ci, z.data_ptr(), ci,
                                                           coef.data_ptr(), parts.data_ptr(), N, h, h, ci, co, None))
    wg = timed(lambda: capi.unet_tconv2x2_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, gw.data_ptr(), N, h, h, ci, co, ci, co, None))
    fl = 2.0 * N * h * h * 4 * ci * co
    print(f"{ci:5d}->{co:<4d} {h:3d}x{h:<3d} fwd {fwd:6.1f} us {fl / fwd * 1e-6:5.0f} TF | dgrad+red {dg:6.1f} us {fl / dg * 1e-6:5.0f} TF"
          f" | wgrad {wg:6.1f} us {fl / wg * 1e-6:5.0f} TF", flush=True)
