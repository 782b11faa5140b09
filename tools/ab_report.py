#!/usr/bin/env python3
"""Prints step time and the kernel-family table of the A/B runs under gpurun_out/ab/ (tools/ab_run.sh)."""
import glob, json, sys
keys = sys.argv[1:] or ["conv3x3_fwd_bf16", "conv3x3_dgrad_bf16", "conv3x3_wgrad_bf16", "bn_act_bf16", "bn_bwd_apply_bf16",
                        "tconv2x2_bf16", "tconv2x2_wgrad_bf16", "wgrad_reduce"]
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1]); k = d["kernel_families"]
    print(f.split("/")[-1], "step", d["ms_per_step"], "ms;", " ".join(f"{n.replace('_bf16','')} {k[n]['ms_per_step']}" for n in keys if n in k))
