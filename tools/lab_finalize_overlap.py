#!/usr/bin/env python3
"""A layer's backward tail -- weight gradient, then the small finalize kernel of the parent's batch norm, then the pass
that needs its result -- on one stream against the finalize on a second stream beside the weight gradient (events:
the side stream starts behind the data gradient, the main stream waits for it before the dependent pass).  The
forward finalize and normalisation entry points stand in for the backward pair (same kernel sizes).  us per trip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi

dev = torch.device("cuda", 0)
capi.unet_set_device(0)
N = 16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for ci, co, h in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 1024, 16)]:
    x = torch.randn(N, h, h, ci, device=dev).bfloat16()
    dy = torch.randn(N, h, h, co, device=dev).bfloat16()
    wb = (torch.randn(9 * ci * co, device=dev) * 0.02).bfloat16()
    dx = torch.empty(N, h, h, ci, device=dev, dtype=torch.bfloat16)
    y = torch.empty(N, h, h, ci, device=dev, dtype=torch.bfloat16)
    gw = torch.empty(9 * ci * co, device=dev)
    rows = capi.lib().unet_conv3x3_stat_rows(N, h, h, co, ci)
    parts = torch.rand(rows * ci * 2, device=dev)
    gb = torch.rand(2 * ci, device=dev) + 0.5
    bstats, run = torch.zeros(2 * ci, device=dev), torch.zeros(2 * ci, device=dev)
    coef = torch.zeros(capi.lib().unet_bn_coef_floats(ci), device=dev)

    def dgrad(st):
        capi.unet_conv3x3_bf16(dy.data_ptr(), co, wb.data_ptr(), None, 0, dx.data_ptr(), ci, None, N, h, h, co, ci, 0, st.cuda_stream)

    def wgrad(st):
        capi.unet_conv3x3_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, gw.data_ptr(), N, h, h, ci, co, ci, co, st.cuda_stream)

    def finalize(st):
        capi.unet_bn_finalize_bf16(parts.data_ptr(), rows, ci, ci, N * h * h, gb.data_ptr(), bstats.data_ptr(), run.data_ptr(),
                                   coef.data_ptr(), st.cuda_stream)

    def apply(st):
        capi.unet_bn_act_bf16(dx.data_ptr(), ci, y.data_ptr(), ci, coef.data_ptr(), ci, N * h * h, 0, st.cuda_stream)

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
        dgrad(s1); wgrad(s1); finalize(s1); apply(s1)

    def no_finalize():
        dgrad(s1); wgrad(s1); apply(s1)

    def overlapped():
        dgrad(s1)
        ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
        finalize(s2)
        ev2 = torch.cuda.Event(); ev2.record(s2)
        wgrad(s1)
        s1.wait_event(ev2)
        apply(s1)

    a, b, c = timed(serial), timed(overlapped), timed(no_finalize)
    print(f"{ci:5d}->{co:<5d} {h:3d}x{h:<3d}  one stream {a:7.1f} us   finalize beside the weight gradient {b:7.1f} us  ({b - a:+.1f})   without it {c:7.1f} us ({c - a:+.1f})", flush=True)
