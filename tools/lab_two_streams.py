#!/usr/bin/env python3
"""Data gradient and weight gradient of one 3x3 layer, one after the other on one stream against side by side on
two streams (both read the same output gradient; neither reads what the other writes).  Batch 16, the layer
shapes of the benchmark graph.  Prints us per pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi

dev = torch.device("cuda", 0)
capi.unet_set_device(0)
N = 16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for ci, co, h in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16)]:
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    dy = torch.randn(N, h, h, co, device=dev).bfloat16()
    wb = (torch.randn(9 * ci * co, device=dev) * 0.02).bfloat16()
    dx = torch.empty(N, h, h, ci, device=dev, dtype=torch.bfloat16)
    gw = torch.empty(9 * ci * co, device=dev)

    def dgrad(st):
        capi.unet_conv3x3_bf16(dy.data_ptr(), co, wb.data_ptr(), None, 0, dx.data_ptr(), ci, None, N, h, h, co, ci, 0, st.cuda_stream)

    def wgrad(st):
        capi.unet_conv3x3_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, gw.data_ptr(), N, h, h, ci, co, ci, co, st.cuda_stream)

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(reps):
            fn()
        e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def serial():
        dgrad(s1)
        wgrad(s1)

    def side_by_side():
        ev = torch.cuda.Event()
        ev.record(s1)
        s2.wait_event(ev)
        dgrad(s1)
        wgrad(s2)
        ev2 = torch.cuda.Event()
        ev2.record(s2)
        s1.wait_event(ev2)

    a, b = timed(serial), timed(side_by_side)
    print(f"{ci:5d}->{co:<5d} {h:3d}x{h:<3d}  one stream {a:7.1f} us   two streams {b:7.1f} us  ({100 * (b / a - 1):+.1f} %)")
