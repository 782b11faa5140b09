#!/bin/bash
# Same-box A/B of a laboratory environment switch: alternates `VAR=1 python bench.py` and plain `python bench.py`,
# N rounds, and prints step time and the 3x3 convolution families of each run.
#   tools/ab_env.sh UNET_LAB_NO_XCD_ORDER [rounds]
set -e
cd "$(dirname "$0")/.."
var=$1; n=${2:-2}
mkdir -p gpurun_out/ab_env
for i in $(seq 1 "$n"); do
  env $var=1 timeout -k 10 100 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/ab_env/with_$i.json
  timeout -k 10 100 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/ab_env/without_$i.json
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/ab_env/*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1]); k = d["kernel_families"]
    print(f.split("/")[-1], "step", d["ms_per_step"], "ms; 3x3 fwd", k["conv3x3_fwd_bf16"]["ms_per_step"], "dgrad",
          k["conv3x3_dgrad_bf16"]["ms_per_step"], "wgrad", k["conv3x3_wgrad_bf16"]["ms_per_step"])
PY
