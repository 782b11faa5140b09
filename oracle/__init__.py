"""numpy front-end of the CPU oracle (oracle/unet_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (unet_amd) never imports it.

Every function takes/returns C-contiguous float32 numpy arrays in the
reference's layouts (NCHW; weights [C_in][C_out][k][k] + C_out biases; BN
params/stats interleaved) and cites the reference loop it restates in
unet_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libunet_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i32 = ctypes.POINTER(ctypes.c_int32)
_u8 = ctypes.POINTER(ctypes.c_uint8)
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "unet_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libunet_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_dice_score.restype = _F
        _lib.oracle_softmax_xent.restype = _F
    return _lib


def _p(a, ty=_f):
    return None if a is None else a.ctypes.data_as(ty)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def conv_out(h, k, p, s):
    return (h + 2 * p - k) // s + 1


def convolve(x, w_and_b, c_out, k, padding, stride):
    """x [N,Ci,H,W]; w_and_b flat [Ci*Co*k*k + Co] -> y [N,Co,Ho,Wo]."""
    x, w_and_b = _c(x), _c(w_and_b)
    n, ci, h, w = x.shape
    ho, wo = conv_out(h, k, padding, stride), conv_out(w, k, padding, stride)
    y = np.empty((n, c_out, ho, wo), np.float32)
    lib().oracle_convolve(_p(x), _p(w_and_b), _p(y), _I(h), _I(w), _I(padding),
                          _I(stride), _I(ci), _I(c_out), _I(k), _I(n))
    return y


def convolve_backward(x, dy, w_and_b, k, padding, stride):
    """-> (dx [N,Ci,H,W], dw_and_db flat)."""
    x, dy, w_and_b = _c(x), _c(dy), _c(w_and_b)
    n, ci, h, w = x.shape
    co = dy.shape[1]
    dx = np.empty_like(x)
    dw = np.empty(ci * co * k * k + co, np.float32)
    lib().oracle_convolve_backward(_p(x), _p(dy), _p(w_and_b), _p(dx), _p(dw),
                                   _I(h), _I(w), _I(padding), _I(stride), _I(ci),
                                   _I(co), _I(k), _I(n))
    return dx, dw


def bn_train(x, gamma_beta, running=None):
    """-> (y, batch_stats[2C] raw sums, running_after[2C] or None)."""
    x, gamma_beta = _c(x), _c(gamma_beta)
    n, c, h, w = x.shape
    y = np.empty_like(x)
    stats = np.empty(2 * c, np.float32)
    run = None if running is None else _c(running).copy()
    lib().oracle_bn_train(_p(x), _p(gamma_beta), _p(y), _p(stats), _p(run),
                          _I(c), _I(h), _I(w), _I(n))
    return y, stats, run


def bn_inference(x, gamma_beta, running):
    x, gamma_beta, running = _c(x), _c(gamma_beta), _c(running)
    n, c, h, w = x.shape
    y = np.empty_like(x)
    lib().oracle_bn_inference(_p(x), _p(y), _p(gamma_beta), _p(running), _I(c),
                              _I(h), _I(w), _I(n))
    return y


def bn_backward(x, dy, gamma_beta, batch_stats):
    """-> (dx, grad_gamma_beta[2C] interleaved)."""
    x, dy, gamma_beta, batch_stats = _c(x), _c(dy), _c(gamma_beta), _c(batch_stats)
    n, c, h, w = x.shape
    dx = np.empty_like(x)
    g = np.empty(2 * c, np.float32)
    lib().oracle_bn_backward(_p(x), _p(dy), _p(gamma_beta), _p(batch_stats),
                             _p(dx), _p(g), _I(c), _I(h), _I(w), _I(n))
    return dx, g


def upsample(x, w_and_b, c_out, scale):
    x, w_and_b = _c(x), _c(w_and_b)
    n, ci, h, w = x.shape
    y = np.empty((n, c_out, h * scale, w * scale), np.float32)
    lib().oracle_upsample(_p(x), _p(y), _p(w_and_b), _I(n), _I(ci), _I(c_out),
                          _I(scale), _I(h), _I(w))
    return y


def upsample_backward(x, w_and_b, dy, scale):
    x, w_and_b, dy = _c(x), _c(w_and_b), _c(dy)
    n, ci, h, w = x.shape
    co = dy.shape[1]
    dx = np.empty_like(x)
    dw = np.empty(ci * co * scale * scale + co, np.float32)
    lib().oracle_upsample_backward(_p(x), _p(w_and_b), _p(dy), _p(dx), _p(dw),
                                   _I(n), _I(co), _I(ci), _I(h), _I(w), _I(scale))
    return dx, dw


def maxpool(x, k, stride, want_indices=False, want_tiemask=False):
    x = _c(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, h // stride, w // stride), np.float32)
    idx = np.empty(y.shape, np.int32) if want_indices else None
    tm = np.empty(y.shape, np.uint8) if want_tiemask else None
    lib().oracle_maxpool(_p(x), _p(y), _p(idx, _i32), _p(tm, _u8), _I(n), _I(c),
                         _I(h), _I(w), _I(k), _I(stride))
    out = [y]
    if want_indices:
        out.append(idx)
    if want_tiemask:
        out.append(tm)
    return out[0] if len(out) == 1 else tuple(out)


def maxpool_backward(dy, x, y, stride):
    dy, x, y = _c(dy), _c(x), _c(y)
    n, c, h, w = x.shape
    dx = np.empty_like(x)
    lib().oracle_maxpool_backward(_p(dy), _p(x), _p(y), _p(dx), _I(n), _I(stride),
                                  _I(c), _I(h), _I(w))
    return dx


def maxpool_backward_indices(dy, indices, h, w):
    dy, indices = _c(dy), _c(indices, np.int32)
    n, c, ho, wo = dy.shape
    dx = np.empty((n, c, h, w), np.float32)
    lib().oracle_maxpool_backward_indices(_p(dy), _p(indices, _i32), _p(dx), _I(n),
                                          _I(c), _I(h), _I(w), _I(ho), _I(wo))
    return dx


def concat(a, b):
    a, b = _c(a), _c(b)
    n, c1, h, w = a.shape
    c2 = b.shape[1]
    out = np.empty((n, c1 + c2, h, w), np.float32)
    lib().oracle_concat(_p(a), _p(b), _p(out), _I(n), _L(h * w), _I(c1), _I(c2))
    return out


def concat_backward(dy, c1):
    dy = _c(dy)
    n, ct, h, w = dy.shape
    c2 = ct - c1
    da = np.empty((n, c1, h, w), np.float32)
    db = np.empty((n, c2, h, w), np.float32)
    lib().oracle_concat_backward(_p(dy), _p(da), _p(db), _I(n), _L(h * w), _I(c1), _I(c2))
    return da, db


def _ew(fn, *arrs):
    arrs = [_c(a) for a in arrs]
    out = np.empty_like(arrs[0])
    fn(*[_p(a) for a in arrs], _p(out), _L(out.size))
    return out


def relu(x):
    return _ew(lib().oracle_relu, x)


def relu_backward(dy, x):
    return _ew(lib().oracle_relu_backward, dy, x)


def sigmoid(x):
    return _ew(lib().oracle_sigmoid, x)


def sigmoid_backward(dy, y):
    return _ew(lib().oracle_sigmoid_backward, dy, y)


def leaky_relu(x, slope):
    x = _c(x)
    out = np.empty_like(x)
    lib().oracle_leaky_relu(_p(x), _p(out), _L(out.size), ctypes.c_float(slope))
    return out


def leaky_relu_backward(dy, x, slope):
    dy, x = _c(dy), _c(x)
    out = np.empty_like(x)
    lib().oracle_leaky_relu_backward(_p(dy), _p(x), _p(out), _L(out.size), ctypes.c_float(slope))
    return out


def tanh(x):
    return _ew(lib().oracle_tanh, x)


def tanh_backward(dy, y):
    return _ew(lib().oracle_tanh_backward, dy, y)


def concat_crop(a, b):
    """a [N,C1,Ha,Wa] centre-cropped to b's [Hb,Wb], then concatenated with b along the channels."""
    a, b = _c(a), _c(b)
    n, c1, ha, wa = a.shape
    _, c2, hb, wb = b.shape
    out = np.empty((n, c1 + c2, hb, wb), np.float32)
    lib().oracle_concat_crop(_p(a), _p(b), _p(out), _I(n), _I(c1), _I(c2), _I(ha), _I(wa), _I(hb), _I(wb))
    return out


def concat_crop_backward(dy, c1, ha, wa):
    """-> (dA [N,C1,Ha,Wa] zero outside the cropped window, dB [N,C2,Hb,Wb])."""
    dy = _c(dy)
    n, c, hb, wb = dy.shape
    c2 = c - c1
    da = np.empty((n, c1, ha, wa), np.float32)
    db = np.empty((n, c2, hb, wb), np.float32)
    lib().oracle_concat_crop_backward(_p(dy), _p(da), _p(db), _I(n), _I(c1), _I(c2), _I(ha), _I(wa), _I(hb), _I(wb))
    return da, db


def dice_score(mask, pred):
    """mask, pred [N,H,W] -> (mean dice, loss_values[3N])."""
    mask, pred = _c(mask), _c(pred)
    n, h, w = mask.shape[0], mask.shape[-2], mask.shape[-1]
    lv = np.empty(3 * n, np.float32)
    d = lib().oracle_dice_score(_p(mask), _p(pred), _I(h), _I(w), _I(n), _p(lv))
    return float(d), lv


def dice_backward(mask, loss_values):
    mask, loss_values = _c(mask), _c(loss_values)
    n, h, w = mask.shape[0], mask.shape[-2], mask.shape[-1]
    g = np.empty_like(mask)
    lib().oracle_dice_backward(_p(mask), _p(loss_values), _p(g), _I(h), _I(w), _I(n))
    return g


def difference_backward(mask, y):
    mask, y = _c(mask), _c(y)
    g = np.empty_like(y)
    lib().oracle_difference_backward(_p(mask), _p(y), _p(g), _L(y.size))
    return g


def softmax_xent(logits, target, want_grad=True):
    logits, target = _c(logits), _c(target, np.int32)
    n, c, h, w = logits.shape
    g = np.empty_like(logits) if want_grad else None
    loss = lib().oracle_softmax_xent(_p(logits), _p(target, _i32), _p(g), _I(n),
                                     _I(c), _I(h), _I(w))
    return float(loss), g


def adam(w, g, state, lr, t):
    """Returns updated copies (w, state)."""
    w, g, state = _c(w).copy(), _c(g), _c(state).copy()
    lib().oracle_adam(_p(w), _p(g), _p(state), _F(lr), _L(w.size), _I(t))
    return w, state


def sgd(w, g, lr):
    w, g = _c(w).copy(), _c(g)
    lib().oracle_sgd(_p(w), _p(g), _F(lr), _L(w.size))
    return w


def add_b_to_a(a, b):
    a, b = _c(a).copy(), _c(b)
    lib().oracle_add_b_to_a(_p(a), _p(b), _L(a.size))
    return a


def threshold(x, thr):
    x = _c(x).copy()
    lib().oracle_threshold(_p(x), _F(thr), _L(x.size))
    return x


def attention_params(cx, cg, ci):
    """Number of parameters of an attention gate (layer.h:558-566)."""
    return (cx + cg + 2) * ci + 1


def attention(x, g, params, ci):
    """Attention gate forward -> (out [N,Cx,H,W], int1 [N,Ci,H/2,W/2], alpha [N,H/2,W/2])."""
    x, g, params = _c(x), _c(g), _c(params)
    n, cx, h, w = x.shape
    cg = g.shape[1]
    assert g.shape == (n, cg, h // 2, w // 2) and params.size == attention_params(cx, cg, ci)
    int1 = np.empty((n, ci, h // 2, w // 2), np.float32)
    alpha = np.empty((n, h // 2, w // 2), np.float32)
    out = np.empty_like(x)
    lib().oracle_attention(_p(x), _p(g), _p(params), _p(int1), _p(alpha), _p(out), _I(n), _I(h), _I(w),
                           _I(cx), _I(cg), _I(ci))
    return out, int1, alpha


def attention_backward(grad_out, int1, alpha, x, g, params):
    """-> (grad_x, grad_g, grad_params)."""
    grad_out, int1, alpha, x, g, params = _c(grad_out), _c(int1), _c(alpha), _c(x), _c(g), _c(params)
    n, cx, h, w = x.shape
    cg, ci = g.shape[1], int1.shape[1]
    gx, gg, gp = np.empty_like(x), np.empty_like(g), np.empty_like(params)
    lib().oracle_attention_backward(_p(grad_out), _p(int1), _p(alpha), _p(x), _p(g), _p(params), _p(gx), _p(gg),
                                    _p(gp), _I(n), _I(h), _I(w), _I(cx), _I(cg), _I(ci))
    return gx, gg, gp
