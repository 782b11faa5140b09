#!/bin/bash
# Same-box A/B of the streaming policy of the batch-norm passes (cache policy x trips per workgroup).
# usage (on a GPU box): tools/ab_bn_stream.sh > gpurun_out/ab_bn_stream.txt
set -e
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 20 --no-cpu-baseline | python -c '
import json,sys
d=json.loads(sys.stdin.readline()); f=d["kernel_families"]
print(d["ms_per_step"], "ms/step;", {k: f[k]["ms_per_step"] for k in ("bn_act_bf16","bn_bwd_apply_bf16","bn_bwd_reduce_bf16","conv3x3_fwd_bf16","conv3x3_dgrad_bf16","conv3x3_wgrad_bf16")})'
}
for rep in 1 2; do
  run UNET_LAB_BN_NT=0 UNET_LAB_BN_TRIPS=2
  run UNET_LAB_BN_NT=1 UNET_LAB_BN_TRIPS=2
  run UNET_LAB_BN_NT=1 UNET_LAB_BN_TRIPS=1
  run UNET_LAB_BN_NT=1 UNET_LAB_BN_TRIPS=4
  run UNET_LAB_BN_TRIPS=2
done
