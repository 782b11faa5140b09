#!/usr/bin/env python3
"""How far do two CORRECT bf16 evaluations of the benchmark network drift apart?

The float64 oracle that rounds to bf16 wherever the engine stores bf16 (tests/torch_ref.py,
emulate_bf16=True) is run twice on one image of the 4-down/4-up base-64 graph: once as it is, once with
every convolution output moved by +-1 fp32 ulp before the bf16 rounding (the difference two valid fp32
accumulation orders can show).  Everything else is identical and exact.  The per-layer cosine between the two
gradients is the noise floor any bf16 implementation has against any other on this network: an engine
whose gradient agrees with the oracle about as well as the oracle agrees with its nudged self has no
systematic error.  Also prints both runs against the plain float64 gradient.

    python tools/bf16_grad_sensitivity.py [--size 256] [--out profiles/r2_bf16_grad_sensitivity.json]
CPU only; a 256x256 image takes a few minutes on 8 cores."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref  # noqa: E402


def he_params(graph, seed):
    """Flat parameter vector as the engine initialises it: He(fan-in) weights, zero biases, gamma 1, beta 0."""
    rng = np.random.default_rng(seed)
    parts = []
    for kind, s, _, _ in graph:
        if kind in ("conv", "up"):
            ci, co, k = s[2], s[3], s[4]
            fan = ci * k * k if kind == "conv" else ci
            parts += [rng.standard_normal(ci * co * k * k) * np.sqrt(2.0 / fan), np.zeros(co),
                      np.tile([1.0, 0.0], co)]
    return np.concatenate(parts)


def run(graph, flat, x, m, emulate, nudge):
    params = torch_ref.split_params(graph, flat)
    acts = torch_ref.forward(graph, params, x, training=True, emulate_bf16=emulate, nudge_seed=nudge)
    loss = torch_ref.dice_loss(acts[-1], m)
    loss.backward()
    return loss.item(), torch_ref.flat_grads(graph, params)


def per_layer(graph, a, b):
    cos = lambda u, v: float(np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v) + 1e-300))
    out, off = [], 0
    for i, (kind, s, _, _) in enumerate(graph):
        if kind in ("conv", "up"):
            nw = s[2] * s[3] * s[4] * s[4]
            out.append({"layer": i, "kind": kind, "shape": s, "cos": round(cos(a[off:off + nw], b[off:off + nw]), 5),
                        "norm_ratio": round(float(np.linalg.norm(a[off:off + nw]) / np.linalg.norm(b[off:off + nw])), 4)})
            off += nw + 3 * s[3]
    return out, cos(a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    graph = torch_ref.unet_graph(args.size, args.size, 3, args.base, 4, True)
    flat = he_params(graph, 11)
    rng = np.random.default_rng(23)
    x = torch.from_numpy(rng.standard_normal((1, 3, args.size, args.size)))
    m = torch.from_numpy((rng.random((1, args.size, args.size)) < 0.03).astype(np.float64))
    res = {"size": args.size, "base": args.base}
    t0 = time.time()
    l_a, g_a = run(graph, flat, x, m, True, None)
    print(f"emulating oracle: loss {l_a:.6f} ({time.time() - t0:.0f} s)", flush=True)
    l_b, g_b = run(graph, flat, x, m, True, 1)
    print(f"nudged oracle:    loss {l_b:.6f} ({time.time() - t0:.0f} s)", flush=True)
    l_c, g_c = run(graph, flat, x, m, True, 2)
    print(f"nudged oracle 2:  loss {l_c:.6f} ({time.time() - t0:.0f} s)", flush=True)
    l_f, g_f = run(graph, flat, x, m, False, None)
    print(f"float64:          loss {l_f:.6f} ({time.time() - t0:.0f} s)", flush=True)
    res["loss"] = {"emulating": l_a, "nudged_1": l_b, "nudged_2": l_c, "float64": l_f}
    for name, (u, v) in {"emulating_vs_nudged_1": (g_a, g_b), "nudged_1_vs_nudged_2": (g_b, g_c),
                         "emulating_vs_float64": (g_a, g_f), "nudged_1_vs_float64": (g_b, g_f)}.items():
        layers, whole = per_layer(graph, u, v)
        res[name] = {"whole_gradient_cos": round(whole, 5), "layers": layers}
        print(name, "whole-gradient cosine", round(whole, 4))
        print("  per layer:", " ".join(f"{l['cos']:.3f}" for l in layers))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
