"""Helpers for the GPU parity tests: torch is used only to hold device memory;
every operator call goes through the C ABI (unet_amd.capi)."""
import numpy as np
import torch

from unet_amd import capi

DEV = "cuda:0"


def dev(a, dtype=None):
    a = np.ascontiguousarray(a)
    t = torch.from_numpy(a).to(DEV)
    if dtype is not None:
        t = t.to(dtype)
    return t


def empty(shape, dtype=torch.float32):
    # poison so that unwritten elements are noticed
    t = torch.empty(shape, dtype=dtype, device=DEV)
    if dtype.is_floating_point:
        t.fill_(float("nan"))
    else:
        t.fill_(-7 if dtype != torch.uint8 else 255)
    return t


def host(t):
    torch.cuda.synchronize()
    if t.dtype == torch.bfloat16:
        t = t.float()
    return t.cpu().numpy()


def p(t):
    return None if t is None else t.data_ptr()


def conv_out(h, k, pad, s):
    return (h + 2 * pad - k) // s + 1


def convolve(x, w, co, k, pad, s):
    n, ci, h, wd = x.shape
    ho, wo = conv_out(h, k, pad, s), conv_out(wd, k, pad, s)
    y = empty((n, co, ho, wo))
    capi.unet_convolve_f32(p(w), p(x), p(y), y.numel(), n, k, ci, co, ho, wo, h, wd, pad, s, None)
    return y


def convolve_backward(dy, x, w, k, pad, s, want_dx=True):
    n, ci, h, wd = x.shape
    co, ho, wo = dy.shape[1:]
    dx = empty(x.shape) if want_dx else None
    dw = empty((ci * co * k * k + co,))
    capi.unet_convolve_backward_f32(p(dy), p(x), p(w), p(dx), p(dw), n, k, ci, co, ho, wo, h, wd, pad, s, None)
    return dx, dw
