#!/usr/bin/env python3
"""Generates tests/golden/ops_small.npz with PyTorch-CPU (float64 math, stored
as float32) -- the oracle named by the reference's test/torch_compare.py:109-146
(F.conv2d + autograd + optim.Adam).  Run from the repo root:

    python tests/golden/make_golden.py

Only torch and numpy are used; nothing from the reference is imported or read.
Layout conversions: the reference stores conv weights [C_in][C_out][k][k]
(torch: [C_out][C_in][k][k]); conv_transpose2d weights are [C_in][C_out][k][k]
in both; BN params are interleaved [gamma0, beta0, ...].
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(1234)
D = torch.float64
out = {}


def put(name, t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    t = np.asarray(t)
    if t.dtype == np.float64:
        t = t.astype(np.float32)
    out[name] = t


def rnd(*shape):
    # values rounded to float32 so inputs are exactly representable
    return torch.randn(*shape, dtype=torch.float32).to(D)


# ---------------------------------------------------------------- conv ----
conv_cases = {
    # name: (N, Ci, Co, H, W, k, pad, stride)
    "conv_a": (2, 3, 8, 16, 16, 3, 1, 1),
    "conv_b": (1, 4, 5, 7, 9, 3, 1, 1),      # odd, non-square
    "conv_c": (2, 2, 3, 12, 10, 3, 1, 2),    # stride 2
    "conv_d": (1, 3, 2, 9, 9, 5, 2, 1),      # k=5
    "conv_e": (2, 5, 4, 6, 6, 1, 0, 1),      # 1x1
    "conv_f": (1, 2, 2, 8, 8, 3, 0, 1),      # valid conv
    "conv_g": (1, 1, 1, 64, 64, 3, 1, 1),    # BASELINE config 1 shape
}
for name, (n, ci, co, h, w, k, p, s) in conv_cases.items():
    x = rnd(n, ci, h, w).requires_grad_(True)
    wt = (rnd(ci, co, k, k) * 0.3).requires_grad_(True)   # reference layout
    b = rnd(co).requires_grad_(True)
    if name == "conv_g":
        with torch.no_grad():
            wt.copy_(torch.tensor([[-1, 0, 1]] * 3, dtype=D).reshape(1, 1, 3, 3))
            b.zero_()
    y = F.conv2d(x, wt.permute(1, 0, 2, 3), b, stride=s, padding=p)
    dy = rnd(*y.shape)
    y.backward(dy)
    put(name + "_cfg", np.array([n, ci, co, h, w, k, p, s], np.int32))
    put(name + "_x", x)
    put(name + "_w", torch.cat([wt.reshape(-1), b]))       # weights then biases
    put(name + "_y", y)
    put(name + "_dy", dy)
    put(name + "_dx", x.grad)
    put(name + "_dw", torch.cat([wt.grad.reshape(-1), b.grad]))

# ----------------------------------------------------------- batch norm ----
for name, (n, c, h, w) in {"bn_a": (4, 6, 8, 8), "bn_b": (2, 3, 5, 7)}.items():
    x = (rnd(n, c, h, w) * 1.7 + 0.4).requires_grad_(True)
    gamma = (rnd(c) * 0.5 + 1.0).requires_grad_(True)
    beta = rnd(c).requires_grad_(True)
    run_mean = rnd(c) * 0.1
    run_var = rnd(c).abs() + 0.5
    M = n * h * w
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    y = F.batch_norm(x, None, None, gamma, beta, training=True, eps=1e-8)
    dy = rnd(*y.shape)
    y.backward(dy)
    gb = torch.stack([gamma, beta], dim=1).reshape(-1)
    put(name + "_x", x)
    put(name + "_gb", gb)
    put(name + "_y", y)
    put(name + "_stats", torch.stack([x.sum(dim=(0, 2, 3)), (x * x).sum(dim=(0, 2, 3))], dim=1).reshape(-1))
    put(name + "_run_in", torch.stack([run_mean, run_var], dim=1).reshape(-1))
    # reference EMA uses the BIASED variance (batch_norm.cu:136-139)
    put(name + "_run_out", torch.stack([0.9 * run_mean + 0.1 * mean, 0.9 * run_var + 0.1 * var], dim=1).reshape(-1))
    put(name + "_dy", dy)
    put(name + "_dx", x.grad)
    put(name + "_dgb", torch.stack([gamma.grad, beta.grad], dim=1).reshape(-1))
    yi = F.batch_norm(x.detach(), run_mean, run_var, gamma.detach(), beta.detach(), training=False, eps=1e-8)
    put(name + "_yinf", yi)

# ------------------------------------------------------------- max pool ----
for name, (n, c, h, w, ties) in {"pool_a": (2, 3, 8, 8, False), "pool_b": (2, 4, 6, 10, True)}.items():
    x = rnd(n, c, h, w)
    if ties:  # post-ReLU-like input with many exact ties (zeros) and repeated values
        x = torch.relu(torch.round(x * 2) / 2)
    x = x.requires_grad_(True)
    y, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    dy = rnd(*y.shape)
    y.backward(dy)
    put(name + "_x", x)
    put(name + "_y", y)
    put(name + "_idx", idx.to(torch.int32))
    put(name + "_dy", dy)
    put(name + "_dx_first", x.grad)              # torch: gradient to the first maximum
    yu = y.detach().repeat_interleave(2, 2).repeat_interleave(2, 3)
    dyu = dy.repeat_interleave(2, 2).repeat_interleave(2, 3)
    put(name + "_dx_ties", (x.detach() == yu).to(D) * dyu)   # reference: every tied maximum
    xw = x.detach().reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    bits = ((xw == y.detach().unsqueeze(-1)).to(torch.int32) * torch.tensor([1, 2, 4, 8], dtype=torch.int32)).sum(-1)
    put(name + "_tiemask", bits.to(torch.uint8))

# ------------------------------------------------------------- upsample ----
for name, (n, ci, co, h, w, s) in {"up_a": (2, 6, 4, 5, 5, 2), "up_b": (1, 3, 5, 4, 6, 2), "up_c": (1, 2, 2, 3, 3, 3)}.items():
    x = rnd(n, ci, h, w).requires_grad_(True)
    wt = (rnd(ci, co, s, s) * 0.4).requires_grad_(True)
    b = rnd(co).requires_grad_(True)
    y = F.conv_transpose2d(x, wt, b, stride=s)
    dy = rnd(*y.shape)
    y.backward(dy)
    put(name + "_cfg", np.array([n, ci, co, h, w, s], np.int32))
    put(name + "_x", x)
    put(name + "_w", torch.cat([wt.reshape(-1), b]))
    put(name + "_y", y)
    put(name + "_dy", dy)
    put(name + "_dx", x.grad)
    put(name + "_dw", torch.cat([wt.grad.reshape(-1), b.grad]))

# --------------------------------------------- concat / activations ----
a, b = rnd(3, 2, 4, 5), rnd(3, 5, 4, 5)
put("cat_a", a)
put("cat_b", b)
put("cat_y", torch.cat([a, b], dim=1))
x = rnd(2, 3, 5, 5).requires_grad_(True)
dy = rnd(2, 3, 5, 5)
y = torch.relu(x)
y.backward(dy)
put("act_x", x)
put("act_dy", dy)
put("relu_y", y)
put("relu_dx", x.grad)
x.grad = None
y = torch.sigmoid(x)
y.backward(dy)
put("sigm_y", y)
put("sigm_dx", x.grad)

# --------------------------------------------------------------- losses ----
n, h, w = 3, 8, 8
pred = torch.sigmoid(rnd(n, h, w)).requires_grad_(True)
mask = (torch.rand(n, h, w) < 0.3).to(D)
I = (mask * pred).sum(dim=(1, 2))
sm = mask.sum(dim=(1, 2))
sp = pred.sum(dim=(1, 2))
dice = (2 * I / (sm + sp + 1e-5)).mean()
# the reference's backward is d(dice_n)/d(pred) per sample (not divided by N, not negated)
(dice * n).backward()
put("dice_pred", pred)
put("dice_mask", mask)
put("dice_score", dice)
put("dice_lv", torch.stack([I, sm, sp], dim=1).reshape(-1))
put("dice_grad_full", pred.grad)                     # exact derivative: 2(mD - I)/D^2 with I, D depending on p
# reference kernel formula given loss_values (cuda_ops.cu:308-318)
Dn = (sm + sp + 1e-5).reshape(n, 1, 1)
put("dice_grad_ref", 2 * (mask * Dn - I.reshape(n, 1, 1)) / (Dn * Dn))
y = rnd(2, 1, 6, 6).requires_grad_(True)
m = rnd(2, 1, 6, 6)
loss = 0.5 * ((y - m) ** 2).sum() / y.numel()
loss.backward()
put("diff_y", y)
put("diff_m", m)
put("diff_grad", y.grad)
logits = rnd(2, 4, 5, 5).requires_grad_(True)
tgt = torch.randint(0, 4, (2, 5, 5))
ce = F.cross_entropy(logits, tgt)
ce.backward()
put("xent_logits", logits)
put("xent_target", tgt.to(torch.int32))
put("xent_loss", ce)
put("xent_grad", logits.grad)

# ----------------------------------------------------------------- adam ----
w0 = rnd(37)
p = w0.clone().requires_grad_(True)
opt = torch.optim.AdamW([p], lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
grads = [rnd(37) for _ in range(3)]
for g in grads:
    p.grad = g.clone()
    opt.step()
put("adam_w0", w0)
put("adam_grads", torch.stack(grads))
put("adam_w3", p)
st = opt.state[p]
put("adam_state3", torch.stack([st["exp_avg"], st["exp_avg_sq"]], dim=1).reshape(-1))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops_small.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "arrays")
