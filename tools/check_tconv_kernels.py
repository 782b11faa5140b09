#!/usr/bin/env python3
"""Random shapes through the transposed-convolution entry points twice -- the kernels the library selects, and
(second process, UNET_LAB_NO_TCONV_GEMM=1 UNET_LAB_TCONV_WGRAD_PHASE_WGS=1) the general ones -- and compares
forward output, its batch-norm statistics, data gradient (plain, accumulated, with the fused reduction) and weight
gradient.  The two families add in different orders: bf16 outputs may differ by one rounding, sums by 1e-4."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def shapes(seed=0, count=24):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        ci = int(rng.choice([64, 128, 192, 256, 320, 512]))
        co = int(rng.choice([32, 64, 96, 128, 256]))
        out.append((int(rng.integers(1, 4)), ci, co, int(rng.integers(3, 41)), int(rng.integers(3, 41))))
    return out


def run(path):
    import torch
    from unet_amd import capi
    dev = torch.device("cuda", 0)
    capi.unet_set_device(0)
    res = {}
    for k, (n, ci, co, h, w) in enumerate(shapes()):
        g = torch.Generator(device="cpu").manual_seed(k)
        x = torch.randn(n, h, w, ci, generator=g).to(dev).bfloat16()
        dy = torch.randn(n, 2 * h, 2 * w, co, generator=g).to(dev).bfloat16()
        wref = (torch.randn(ci * co * 4, generator=g) * (1.0 / np.sqrt(ci))).to(dev)
        wf = torch.zeros(4 * co * ci, dtype=torch.bfloat16, device=dev)
        wb = torch.zeros(4 * co * ci, dtype=torch.bfloat16, device=dev)
        capi.unet_pack_weights_bf16(wref.data_ptr(), wf.data_ptr(), wb.data_ptr(), ci, co, ci, co, 2, None)
        bias = torch.randn(co, generator=g).to(dev)
        tiles = capi.lib().unet_conv3x3_stat_tiles(h, w)
        out = torch.zeros(n, 2 * h, 2 * w, co, dtype=torch.bfloat16, device=dev)
        stats = torch.zeros(4, n, tiles, co, 2, device=dev)
        capi.unet_tconv2x2_bf16(x.data_ptr(), ci, wf.data_ptr(), bias.data_ptr(), co, out.data_ptr(), co, stats.data_ptr(), n, h, w,
                                ci, co, None)
        dx = torch.zeros(n, h, w, ci, dtype=torch.bfloat16, device=dev)
        capi.unet_tconv2x2_dgrad_bf16(dy.data_ptr(), co, wb.data_ptr(), dx.data_ptr(), ci, n, h, w, ci, co, 0, None)
        acc = torch.randn(n, h, w, ci, generator=g).to(dev).bfloat16()
        capi.unet_tconv2x2_dgrad_bf16(dy.data_ptr(), co, wb.data_ptr(), acc.data_ptr(), ci, n, h, w, ci, co, 1, None)
        z = torch.randn(n, h, w, ci, generator=g).to(dev).bfloat16()
        coef = (torch.rand(ci * 4, generator=g) + 0.5).to(dev)
        parts = torch.zeros(n, tiles, ci, 2, device=dev)
        dx2 = torch.zeros_like(dx)
        capi.unet_tconv2x2_dgrad_bnred_bf16(dy.data_ptr(), co, wb.data_ptr(), dx2.data_ptr(), ci, z.data_ptr(), ci, coef.data_ptr(),
                                            parts.data_ptr(), n, h, w, ci, co, None)
        gw = torch.zeros(ci * co * 4, device=dev)
        capi.unet_tconv2x2_wgrad_bf16(dy.data_ptr(), co, x.data_ptr(), ci, gw.data_ptr(), n, h, w, ci, co, ci, co, None)
        torch.cuda.synchronize()
        for name, t in [("out", out), ("stats", stats.sum(dim=(1, 2))), ("dx", dx), ("acc", acc), ("dx2", dx2),
                        ("parts", parts.sum(dim=(0, 1))), ("gw", gw)]:
            res[f"{k}_{name}"] = t.float().cpu().numpy()
    np.savez(path, **res)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    a, b = "/tmp/tconv_new.npz", "/tmp/tconv_old.npz"
    env = {k: v for k, v in os.environ.items() if k not in ("UNET_LAB_NO_TCONV_GEMM", "UNET_LAB_TCONV_WGRAD_PHASE_WGS")}
    subprocess.check_call([sys.executable, __file__, a], env=env)
    subprocess.check_call([sys.executable, __file__, b], env=dict(env, UNET_LAB_NO_TCONV_GEMM="1", UNET_LAB_TCONV_WGRAD_PHASE_WGS="1"))
    A, B = np.load(a), np.load(b)
    bad = 0
    for k, shp in enumerate(shapes()):
        line = []
        for name in ("out", "stats", "dx", "acc", "dx2", "parts", "gw"):
            u, v = A[f"{k}_{name}"], B[f"{k}_{name}"]
            scale = np.abs(v).max() + 1e-30
            err = np.abs(u - v).max() / scale
            tol = 1.0 / 128 if name in ("out", "dx", "acc", "dx2") else 2e-4   # one bf16 rounding / reordered fp32 sums
            ok = np.isfinite(u).all() and err <= tol
            bad += not ok
            line.append(f"{name} {err:.1e}{'' if ok else ' !!'}")
        print(shp, " ".join(line), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
