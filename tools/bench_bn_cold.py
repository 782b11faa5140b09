#!/usr/bin/env python3
"""bn_act / bn backward at the largest shape with rotating buffers (working set > 1 GB, so neither the
L2 nor the 256 MB Infinity Cache serves re-reads): the in-situ HBM rate of these kernels."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unet_amd import capi
dev = torch.device("cuda", 0); capi.unet_set_device(0)
N, H, C, SETS = 16, 256, 64, 6
pixels = N * H * H
mk = lambda: torch.randn(pixels, C, device=dev).to(torch.bfloat16)
zs, dys, ys = [mk() for _ in range(SETS)], [mk() for _ in range(SETS)], [torch.empty(pixels, C, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
coef = torch.rand(4 * C, device=dev) + 0.5; gb = torch.rand(2 * C, device=dev) + 0.5
ggb = torch.zeros(2 * C, device=dev); gbias = torch.zeros(C, device=dev)
p = lambda t: t.data_ptr()
def run(fn, reps=24):
    for i in range(SETS): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): fn(r % SETS)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
e = pixels * C
t = run(lambda i: capi.unet_bn_act_bf16(p(zs[i]), C, p(ys[i]), C, p(coef), C, pixels, 0, None))
print(f"bn_act           {t:6.1f} us  {4*e/t*1e-6:5.2f} TB/s")
t = run(lambda i: capi.unet_bn_act_backward_bf16(p(dys[i]), C, None, 0, p(zs[i]), C, p(coef), p(gb), p(ys[i]), C, p(ggb), p(gbias), C, C, pixels, 0, None))
print(f"bn bwd (reduce+finalize+apply+colsum) {t:6.1f} us  {10*e/t*1e-6:5.2f} TB/s of 10 B/elt")
