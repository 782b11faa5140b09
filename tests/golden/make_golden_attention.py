#!/usr/bin/env python3
"""Generates tests/golden/attention_small.npz with PyTorch-CPU (float64 math,
stored as float32): the attention gate of the reference
(src/cuda_library/attention/attention.cu:9-499) restated with torch operators --
a stride-2 1x1 convolution of x plus a 1x1 convolution of g, ReLU, a 1x1
convolution to one channel, sigmoid, the reference's 4x4-block blend back to the
resolution of x, and the product with x -- with autograd for the backward.
The reference never tests this layer (src/cuda_library/test/attention_test.h:12-22
is empty), so these vectors are what pins the oracle for it.  Run from the repo root:

    python tests/golden/make_golden_attention.py

Only torch and numpy are used; nothing from the reference is imported or read.
Parameter layout (layer.h:558-566): W[(Cx+Cg)][Ci] | psi[Ci] | b1[Ci] | bpsi.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(4321)
D = torch.float64
out = {}


def rnd(*shape):
    return torch.randn(*shape, dtype=torch.float32).to(D)


def blend_axis(n_out, n_cells):
    """Cell indices and the float32 fraction the reference uses along one axis (attention.cu:61-78, with
    the '+1' cell clamped as its lines 65-66 compute)."""
    i = torch.arange(n_out)
    c0 = (i // 2) // 2 * 2
    c1 = torch.clamp(c0 + 1, max=n_cells - 1)
    frac = ((i % 4).to(torch.float32) / torch.tensor(3.0, dtype=torch.float32)).to(D)
    return c0, c1, frac


def attention(x, g, params, ci):
    n, cx, hx, wx = x.shape
    cg = g.shape[1]
    hg, wg = hx // 2, wx // 2
    W = params[:(cx + cg) * ci].view(cx + cg, ci)
    psi = params[(cx + cg) * ci:(cx + cg + 1) * ci]
    b1 = params[(cx + cg + 1) * ci:(cx + cg + 2) * ci]
    bpsi = params[(cx + cg + 2) * ci]
    wx_ = W[:cx].t().reshape(ci, cx, 1, 1)
    wg_ = W[cx:].t().reshape(ci, cg, 1, 1)
    int1 = F.relu(F.conv2d(x, wx_, stride=2) + F.conv2d(g, wg_) + b1.view(1, -1, 1, 1))
    alpha = torch.sigmoid((int1 * psi.view(1, -1, 1, 1)).sum(1) + bpsi)
    h0, h1, fh = blend_axis(hx, hg)
    w0, w1, fw = blend_axis(wx, wg)
    A, B = alpha[:, h0][:, :, w0], alpha[:, h0][:, :, w1]
    C, Dd = alpha[:, h1][:, :, w0], alpha[:, h1][:, :, w1]
    r1 = B * fw + A * (1 - fw)
    r2 = Dd * fw + C * (1 - fw)
    interp = r2 * fh[:, None] + r1 * (1 - fh[:, None])
    return x * interp[:, None], int1, alpha


cases = {
    # name: (N, Cx, Cg, Ci, Hx, Wx)
    "attn_a": (2, 8, 6, 4, 8, 12),
    "attn_odd": (1, 4, 4, 2, 6, 10),     # H/2 and W/2 odd: the clamped last cells
    "attn_b": (2, 16, 32, 8, 16, 16),    # the layer's own ratio: Cg = 2*Cx, Ci = Cx/2
}
for name, (n, cx, cg, ci, hx, wx) in cases.items():
    x = rnd(n, cx, hx, wx).requires_grad_(True)
    g = rnd(n, cg, hx // 2, wx // 2).requires_grad_(True)
    params = (rnd((cx + cg + 2) * ci + 1) * 0.3).to(torch.float32).to(D).requires_grad_(True)
    y, int1, alpha = attention(x, g, params, ci)
    dy = rnd(*y.shape)
    y.backward(dy)
    for key, t in dict(cfg=np.array([n, cx, cg, ci, hx, wx], np.int32), x=x, g=g, params=params, y=y, int1=int1,
                       alpha=alpha, dy=dy, dx=x.grad, dg=g.grad, dparams=params.grad).items():
        if isinstance(t, torch.Tensor):
            t = t.detach().numpy().astype(np.float32)
        out[f"{name}_{key}"] = t

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention_small.npz")
np.savez_compressed(path, **out)
print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")
