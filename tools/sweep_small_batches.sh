#!/bin/bash
# Step time by batch size, training and inference, for the benchmark graph and main.py's graph
# (main.py:229 trains with batches of 4).  Prints one line per run; run it on the GPU box:
#   gpurun -- 'bash tools/sweep_small_batches.sh > gpurun_out/small_batches.txt'
set -e
cd "$(dirname "$0")/.."
for mode in train infer; do
  for g in unet4_base64 mainpy; do
    for b in 1 2 4 8 16; do
      flag=""; [ $mode = infer ] && flag="--infer"
      timeout -k 10 200 python bench.py --graph $g --batch $b $flag --no-cpu-baseline --no-prof --steps 50 --warmup 5 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', '$g', 'batch', $b, 'ms/step', d['ms_per_step'], 'images/s', d['value'])"
    done
  done
done
