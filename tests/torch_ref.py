"""PyTorch-CPU (float64) restatement of a `neural_net` graph, used as the
network-level oracle (the reference names PyTorch-CPU as its oracle in
test/torch_compare.py:109-146).  Test infrastructure only.

A graph is a list of (kind, shape, parents, activation) tuples with the same
field orders as neural_net.LayerDesc; `params` is the flat fp32 parameter
vector in the module's layout: per CONV/UPSAMPLING layer
[weights (C_in,C_out,k,k) | biases (C_out) | gamma0,beta0,gamma1,beta1,...].
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-8


def main_py_graph(h=256, w=256, cin=3, f1=64):
    """The 16-layer graph of the reference's main.py:187-224 (3 down / 3 up,
    single conv per stage), scalable for tests."""
    f2, f3, f4 = 2 * f1, 4 * f1, 8 * f1
    h2, h3, h4, w2, w3, w4 = h // 2, h // 4, h // 8, w // 2, w // 4, w // 8
    return [
        ("conv", [h, w, cin, f1, 3, 1, 1], [], "relu"),      # 0
        ("pool", [f1, h, w, 2, 2], [0], None),               # 1
        ("conv", [h2, w2, f1, f2, 3, 1, 1], [1], "relu"),    # 2
        ("pool", [f2, h2, w2, 2, 2], [2], None),             # 3
        ("conv", [h3, w3, f2, f3, 3, 1, 1], [3], "relu"),    # 4
        ("pool", [f3, h3, w3, 2, 2], [4], None),             # 5
        ("conv", [h4, w4, f3, f4, 3, 1, 1], [5], "relu"),    # 6
        ("up", [h4, w4, f4, f3, 2], [6], None),              # 7
        ("concat", [], [4, 7], None),                        # 8
        ("conv", [h3, w3, f4, f2, 3, 1, 1], [8], "relu"),    # 9
        ("up", [h3, w3, f2, f2, 2], [9], None),              # 10
        ("concat", [], [2, 10], None),                       # 11
        ("conv", [h2, w2, f3, f1, 3, 1, 1], [11], "relu"),   # 12
        ("up", [h2, w2, f1, f1, 2], [12], None),             # 13
        ("concat", [], [0, 13], None),                       # 14
        ("conv", [h, w, f2, 1, 3, 1, 1], [14], "sigmoid"),   # 15
    ]


def unet_graph(h=256, w=256, cin=3, base=64, depth=4, double_conv=True):
    """Classic U-Net: `depth` down / `depth` up levels, (double) 3x3 conv + BN +
    ReLU per stage, 2x2 max pool, 2x2 transposed-conv upsampling, skip concat,
    final 3x3 conv to one sigmoid channel."""
    g = []

    def conv(hh, ww, ci, co, parent, act="relu"):
        g.append(("conv", [hh, ww, ci, co, 3, 1, 1], [] if parent is None else [parent], act))
        return len(g) - 1

    def stage(hh, ww, ci, co, parent):
        i = conv(hh, ww, ci, co, parent)
        if double_conv:
            i = conv(hh, ww, co, co, i)
        return i

    skips = []
    cur, c, hh, ww = None, cin, h, w
    for d in range(depth):
        co = base << d
        cur = stage(hh, ww, c, co, cur)
        skips.append((cur, co, hh, ww))
        g.append(("pool", [co, hh, ww, 2, 2], [cur], None))
        cur, c, hh, ww = len(g) - 1, co, hh // 2, ww // 2
    cur = stage(hh, ww, c, base << depth, cur)
    c = base << depth
    for d in reversed(range(depth)):
        skip, co, sh, sw = skips[d]
        g.append(("up", [hh, ww, c, co, 2], [cur], None))
        up = len(g) - 1
        g.append(("concat", [], [skip, up], None))
        cur = stage(sh, sw, 2 * co, co, len(g) - 1)
        c, hh, ww = co, sh, sw
    conv(hh, ww, c, 1, cur, "sigmoid")
    return g


def valid_unet_graph(h=60, w=60, cin=3, base=8, depth=2, acts=("relu", "relu")):
    """The original U-Net geometry (Ronneberger et al.): unpadded 3x3 convolutions, 2x2 max pool, 2x2
    transposed convolutions, the larger skip tensor centre-cropped at every concat; BASELINE config 4 at
    572 x 572 / base 64 / depth 4 gives the 388 x 388 output.  `acts` = activations of the two convolutions of
    a stage.  Returns (graph, output height, output width)."""
    g = []

    def conv(hh, ww, ci, co, parent, act):
        g.append(("conv", [hh, ww, ci, co, 3, 0, 1], [] if parent is None else [parent], act))
        return len(g) - 1, hh - 2, ww - 2

    def stage(hh, ww, ci, co, parent):
        i, hh, ww = conv(hh, ww, ci, co, parent, acts[0])
        return conv(hh, ww, co, co, i, acts[1])

    skips = []
    cur, c, hh, ww = None, cin, h, w
    for d in range(depth):
        co = base << d
        cur, hh, ww = stage(hh, ww, c, co, cur)
        skips.append((cur, co))
        assert hh % 2 == 0 and ww % 2 == 0, (hh, ww)
        g.append(("pool", [co, hh, ww, 2, 2], [cur], None))
        cur, c, hh, ww = len(g) - 1, co, hh // 2, ww // 2
    cur, hh, ww = stage(hh, ww, c, base << depth, cur)
    c = base << depth
    for d in reversed(range(depth)):
        skip, co = skips[d]
        g.append(("up", [hh, ww, c, co, 2], [cur], None))
        hh, ww = 2 * hh, 2 * ww
        g.append(("concat", [], [skip, len(g) - 1], None))
        cur, hh, ww = stage(hh, ww, 2 * co, co, len(g) - 1)
        c = co
    g.append(("conv", [hh, ww, c, 1, 3, 1, 1], [cur], "sigmoid"))
    return g, hh, ww


def attention_unet_graph(h=64, w=64, cin=3, f1=8):
    """The "Attention U-Net" graph the reference keeps commented out in main.py:144-184: the main.py
    graph with an attention gate on every skip connection (x = the encoder output, g = the decoder
    tensor one level below, before upsampling)."""
    f2, f3, f4 = 2 * f1, 4 * f1, 8 * f1
    h2, h3, h4, w2, w3, w4 = h // 2, h // 4, h // 8, w // 2, w // 4, w // 8
    return [
        ("conv", [h, w, cin, f1, 3, 1, 1], [], "relu"),      # 0
        ("pool", [f1, h, w, 2, 2], [0], None),               # 1
        ("conv", [h2, w2, f1, f2, 3, 1, 1], [1], "relu"),    # 2
        ("pool", [f2, h2, w2, 2, 2], [2], None),             # 3
        ("conv", [h3, w3, f2, f3, 3, 1, 1], [3], "relu"),    # 4
        ("pool", [f3, h3, w3, 2, 2], [4], None),             # 5
        ("conv", [h4, w4, f3, f4, 3, 1, 1], [5], "relu"),    # 6
        ("up", [h4, w4, f4, f3, 2], [6], None),              # 7
        ("attention", [], [4, 6], None),                     # 8
        ("concat", [], [8, 7], None),                        # 9
        ("conv", [h3, w3, f4, f2, 3, 1, 1], [9], "relu"),    # 10
        ("up", [h3, w3, f2, f2, 2], [10], None),             # 11
        ("attention", [], [2, 10], None),                    # 12
        ("concat", [], [12, 11], None),                      # 13
        ("conv", [h2, w2, f3, f1, 3, 1, 1], [13], "relu"),   # 14
        ("up", [h2, w2, f1, f1, 2], [14], None),             # 15
        ("attention", [], [0, 14], None),                    # 16
        ("concat", [], [16, 15], None),                      # 17
        ("conv", [h, w, f2, 1, 3, 1, 1], [17], "sigmoid"),   # 18
    ]


def out_channels(graph):
    """Output channel count of every layer."""
    c = []
    for kind, s, parents, _ in graph:
        if kind in ("conv", "up"):
            c.append(s[3])
        elif kind == "pool":
            c.append(s[0])
        elif kind == "concat":
            c.append(c[parents[0]] + c[parents[1]])
        elif kind == "attention":
            c.append(c[parents[0]])
        else:
            raise ValueError(kind)
    return c


def attention_dims(graph, i):
    """(Cx, Cg, Ci) of attention layer i: Ci = Cx / 2 (attention_layer_functions.h:18)."""
    c = out_channels(graph)
    cx, cg = c[graph[i][2][0]], c[graph[i][2][1]]
    return cx, cg, cx // 2


def attention_gate(x, g, flat, ci):
    """The reference's attention gate (src/cuda_library/attention/attention.cu:9-203) with torch operators;
    flat = W[(Cx+Cg)][Ci] | psi[Ci] | b1[Ci] | bpsi (layer.h:558-566).  The blend back to the resolution of
    x mixes, per 4x4 block of output pixels, the 2x2 coefficient cells at (2*(h//4), 2*(w//4)) with
    fractions (h%4)/3, (w%4)/3 (:61-78), the "+1" cells clamped to the last row / column (:65-66)."""
    n, cx, hx, wx = x.shape
    cg = g.shape[1]
    W = flat[:(cx + cg) * ci].view(cx + cg, ci)
    psi = flat[(cx + cg) * ci:(cx + cg + 1) * ci]
    b1 = flat[(cx + cg + 1) * ci:(cx + cg + 2) * ci]
    bpsi = flat[(cx + cg + 2) * ci]
    int1 = F.relu(F.conv2d(x, W[:cx].t().reshape(ci, cx, 1, 1), stride=2) +
                  F.conv2d(g, W[cx:].t().reshape(ci, cg, 1, 1)) + b1.view(1, -1, 1, 1))
    alpha = torch.sigmoid((int1 * psi.view(1, -1, 1, 1)).sum(1) + bpsi)

    def axis(n_out, n_cells):
        i = torch.arange(n_out)
        c0 = (i // 2) // 2 * 2
        c1 = torch.clamp(c0 + 1, max=n_cells - 1)
        frac = ((i % 4).to(torch.float32) / torch.tensor(3.0, dtype=torch.float32)).to(x.dtype)
        return c0, c1, frac

    h0, h1, fh = axis(hx, hx // 2)
    w0, w1, fw = axis(wx, wx // 2)
    A, B = alpha[:, h0][:, :, w0], alpha[:, h0][:, :, w1]
    C, D = alpha[:, h1][:, :, w0], alpha[:, h1][:, :, w1]
    interp = (D * fw + C * (1 - fw)) * fh[:, None] + (B * fw + A * (1 - fw)) * (1 - fh[:, None])
    return x * interp[:, None]


def to_layer_descs(nn, graph):
    kinds = {"conv": nn.LayerType.CONV_LAYER, "pool": nn.LayerType.MAX_POOL_LAYER,
             "up": nn.LayerType.UPSAMPLING_LAYER, "concat": nn.LayerType.CONCAT_LAYER,
             "attention": nn.LayerType.ATTENTION_LAYER}
    out = []
    for kind, shape, parents, act in graph:
        if act == "sigmoid":
            out.append(nn.LayerDesc(kinds[kind], shape, parents, nn.ActivationType.SIGMOID))
        elif act == "tanh":
            out.append(nn.LayerDesc(kinds[kind], shape, parents, nn.ActivationType.TANH))
        elif isinstance(act, str) and act.startswith("leaky:"):
            out.append(nn.LayerDesc(kinds[kind], shape, parents, nn.ActivationType.LEAKY_ReLU, float(act[6:])))
        else:
            out.append(nn.LayerDesc(kinds[kind], shape, parents))
    return out


def param_count(graph):
    n = 0
    for kind, s, _, _ in graph:
        if kind == "conv":
            n += s[2] * s[3] * s[4] * s[4] + s[3] + 2 * s[3]
        elif kind == "up":
            n += s[2] * s[3] * s[4] * s[4] + s[3] + 2 * s[3]
    for i, layer in enumerate(graph):
        if layer[0] == "attention":
            cx, cg, ci = attention_dims(graph, i)
            n += (cx + cg + 2) * ci + 1
    return n


def split_params(graph, flat):
    """-> list (per layer) of dicts of torch leaf tensors (float64, requires_grad) or None."""
    flat = torch.as_tensor(np.asarray(flat, dtype=np.float64))
    out, off = [], 0
    for i, (kind, s, _, _) in enumerate(graph):
        if kind == "attention":
            cx, cg, ci = attention_dims(graph, i)
            cnt = (cx + cg + 2) * ci + 1
            out.append({"att": flat[off:off + cnt].clone().requires_grad_(True)})
            off += cnt
        elif kind in ("conv", "up"):
            ci, co, k = s[2], s[3], s[4]
            nw = ci * co * k * k
            d = {
                "w": flat[off:off + nw].reshape(ci, co, k, k).clone().requires_grad_(True),
                "b": flat[off + nw:off + nw + co].clone().requires_grad_(True),
                "gamma": flat[off + nw + co:off + nw + 3 * co:2].clone().requires_grad_(True),
                "beta": flat[off + nw + co + 1:off + nw + 3 * co:2].clone().requires_grad_(True),
            }
            off += nw + 3 * co
            out.append(d)
        else:
            out.append(None)
    assert off == flat.numel(), (off, flat.numel())
    return out


def flat_grads(graph, params):
    parts = []
    for p in params:
        if p is None:
            continue
        if "att" in p:
            parts.append(p["att"].grad.reshape(-1))
            continue
        parts += [p["w"].grad.reshape(-1), p["b"].grad.reshape(-1),
                  torch.stack([p["gamma"].grad, p["beta"].grad], dim=1).reshape(-1)]
    return torch.cat(parts).numpy()


class _RoundBF16(torch.autograd.Function):
    """Rounds to bf16 in forward AND rounds the gradient to bf16 in backward:
    models a tensor (and its gradient) that is stored in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundBF16Value(torch.autograd.Function):
    """Rounds to bf16 in forward, passes the gradient through unchanged: models
    bf16 copies of fp32 master weights."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _MaxPoolTies(torch.autograd.Function):
    """k x k / stride k max pool whose backward gives the gradient to EVERY
    element equal to the window maximum (the reference's rule,
    src/cuda_library/max_pool/max_pool.cu:172), not just the first."""

    @staticmethod
    def forward(ctx, x, k):
        y = F.max_pool2d(x, k, k)
        ctx.save_for_backward(x, y)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        k = ctx.k
        up = lambda t: t.repeat_interleave(k, 2).repeat_interleave(k, 3)
        return (x == up(y)).to(g.dtype) * up(g), None


def _ulp_nudge(z, gen):
    """z moved by one fp32 unit in the last place, up or down at random per element (no gradient through
    the nudge): the smallest difference two correct fp32 accumulations of the same sum can show."""
    z32 = z.detach().to(torch.float32)
    up = torch.rand(z32.shape, generator=gen) < 0.5
    inf = torch.full_like(z32, float("inf"))
    moved = torch.where(up, torch.nextafter(z32, inf), torch.nextafter(z32, -inf))
    return z + (moved.to(z.dtype) - z.detach())


def forward(graph, params, x, training=True, running=None, emulate_bf16=False, nudge_seed=None, round_fp32=False,
            pre=None):
    """x float64 [N,C,H,W] -> list of per-layer activations.  `running`
    (optional list of [mean,var] interleaved arrays per layer) is used when
    training=False.

    emulate_bf16=True restates the module's bf16 path: the input, the weights
    (values only), every convolution output and every activation are rounded
    to bf16 where the module stores them in bf16, gradients are rounded where
    the module stores them in bf16, batch-norm statistics are taken from the
    rounded convolution output, attention gates compute in full precision
    on the rounded inputs and round their output, the final (narrow) layer's batch norm,
    activation and loss stay in full precision, and max pool routes gradients
    to every tied maximum.  Arithmetic stays float64.

    nudge_seed: every convolution output is moved by +-1 fp32 ulp (seeded, per element) before it is
    rounded -- a second, equally valid evaluation of the same network, used to measure how far two correct
    bf16 runs drift apart (tools/bf16_grad_sensitivity.py).

    pre: a list that receives the batch-norm output (the activation's argument) of every conv / up layer."""
    gen = None if nudge_seed is None else torch.Generator().manual_seed(int(nudge_seed))
    if emulate_bf16:
        rnd = _RoundBF16.apply
    elif round_fp32:   # every convolution output rounded to fp32 (what an fp32 engine stores), values only
        rnd = lambda t: t + (t.detach().to(torch.float32).to(t.dtype) - t.detach())
    else:
        rnd = lambda t: t
    rndw = _RoundBF16Value.apply if emulate_bf16 else (lambda t: t)
    if emulate_bf16:
        x = x.to(torch.bfloat16).to(x.dtype)
    acts = []
    last = len(graph) - 1
    for i, (kind, s, parents, act) in enumerate(graph):
        inp = x if not parents else acts[parents[0]]
        if kind in ("conv", "up"):
            p = params[i]
            if kind == "conv":
                z = F.conv2d(inp, rndw(p["w"]).permute(1, 0, 2, 3), p["b"], stride=s[6], padding=s[5])
            else:
                z = F.conv_transpose2d(inp, rndw(p["w"]), p["b"], stride=s[4])
            if gen is not None:
                z = _ulp_nudge(z, gen)
            z = rnd(z)
            if training:
                y = F.batch_norm(z, None, None, p["gamma"], p["beta"], training=True, eps=BN_EPS)
            else:
                r = torch.as_tensor(np.asarray(running[i], dtype=np.float64))
                y = F.batch_norm(z, r[0::2], r[1::2], p["gamma"], p["beta"], training=False, eps=BN_EPS)
            if pre is not None:
                pre.append(y.detach())
            if act == "sigmoid":
                y = torch.sigmoid(y)
            elif act == "logits":      # the final layer under the cross-entropy loss: no activation
                pass
            elif act == "tanh" and kind == "conv":
                y = torch.tanh(y)
            elif isinstance(act, str) and act.startswith("leaky:") and kind == "conv":
                y = F.leaky_relu(y, float(act[6:]))
            else:
                y = torch.relu(y)
            acts.append(y if i == last else rnd(y))
        elif kind == "pool":
            acts.append(_MaxPoolTies.apply(inp, s[3]) if emulate_bf16 else F.max_pool2d(inp, s[3], s[4]))
        elif kind == "concat":
            a, b = acts[parents[0]], acts[parents[1]]
            if a.shape[2:] != b.shape[2:]:   # the larger skip tensor is centre-cropped to the size of the one from below
                oy, ox = (a.shape[2] - b.shape[2]) // 2, (a.shape[3] - b.shape[3]) // 2
                a = a[:, :, oy:oy + b.shape[2], ox:ox + b.shape[3]]
            acts.append(torch.cat([a, b], dim=1))
        elif kind == "attention":
            acts.append(rnd(attention_gate(acts[parents[0]], acts[parents[1]], params[i]["att"],
                                           attention_dims(graph, i)[2])))
        else:
            raise ValueError(kind)
    return acts


def dice_loss(pred, mask):
    """L = 1 - mean_n 2*I/(sum m + sum p + 1e-5); pred [N,1,H,W], mask [N,H,W]."""
    p = pred[:, 0]
    inter = (p * mask).sum(dim=(1, 2))
    d = mask.sum(dim=(1, 2)) + p.sum(dim=(1, 2)) + 1e-5
    return 1.0 - (2.0 * inter / d).mean()


def difference_loss(pred, mask):
    return 0.5 * ((pred[:, 0] - mask) ** 2).mean()


def adam_step(flat_params, flat_grad, state, lr, t):
    """The module's optimiser: Adam beta (.9,.999), eps 1e-8, decoupled weight
    decay 1e-4 (src/cuda_library/cuda_ops.cu:405-441).  float64 numpy."""
    m, v = state
    m[:] = 0.9 * m + 0.1 * flat_grad
    v[:] = 0.999 * v + 0.001 * flat_grad * flat_grad
    mh, vh = m / (1 - 0.9 ** t), v / (1 - 0.999 ** t)
    return flat_params - lr * (1e-4 * flat_params + mh / (np.sqrt(vh) + 1e-8))
