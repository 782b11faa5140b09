#!/usr/bin/env python3
"""Times the bandwidth-bound bf16 kernels alone at the U-Net's tensor shapes (batch 16)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi

dev = torch.device("cuda", 0)
capi.unet_set_device(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3  # us


N = 16
for H, C in ((256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)):
    pixels = N * H * H
    z = (torch.randn(pixels, C, device=dev)).to(torch.bfloat16)
    dy1 = torch.randn(pixels, C, device=dev).to(torch.bfloat16)
    dy2 = torch.randn(pixels, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(z)
    dz = torch.empty_like(z)
    coef = torch.rand(4 * C, device=dev) + 0.5
    gb = torch.rand(2 * C, device=dev) + 0.5
    ggb = torch.zeros(2 * C, device=dev)
    gbias = torch.zeros(C, device=dev)
    p = lambda t: t.data_ptr()
    t_act = timeit(lambda: capi.unet_bn_act_bf16(p(z), C, p(y), C, p(coef), C, pixels, 0, None))
    capi.unet_prof_enable(1)
    for _ in range(10):
        capi.unet_bn_act_backward_bf16(p(dy1), C, None, 0, p(z), C, p(coef), p(gb), p(dz), C, p(ggb), p(gbias), C, C, pixels, 0, None)
    res = {}
    kinds = ctypes.c_int(); capi.unet_prof_kinds(ctypes.byref(kinds))
    for k in range(kinds.value):
        n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        capi.unet_prof_get(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            res[capi.lib().unet_prof_name(k).decode()] = (ms.value / n.value * 1e3, by.value / n.value)
    capi.unet_prof_enable(1)
    for _ in range(10):
        capi.unet_bn_act_backward_bf16(p(dy1), C, p(dy2), C, p(z), C, p(coef), p(gb), p(dz), C, p(ggb), p(gbias), C, C, pixels, 0, None)
    res2 = {}
    for k in range(kinds.value):
        n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        capi.unet_prof_get(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            res2[capi.lib().unet_prof_name(k).decode()] = (ms.value / n.value * 1e3, by.value / n.value)
    capi.unet_prof_enable(0)
    e = pixels * C
    line = f"{H}x{H}x{C}: bn_act {t_act:6.1f} us {4*e/t_act*1e-6:5.2f} TB/s"
    for name, (us, by) in res.items():
        line += f" | {name} {us:6.1f} us {by/us*1e-6:5.2f} TB/s"
    for name, (us, by) in res2.items():
        line += f" | +dy2 {name} {us:6.1f} us {by/us*1e-6:5.2f} TB/s"
    print(line, flush=True)
