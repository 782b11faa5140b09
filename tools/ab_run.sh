#!/bin/bash
# Alternates the old (ab_old/) and the working-tree bench, N rounds, one JSON file per run under gpurun_out/ab/.
set -e
cd "$(dirname "$0")/.."
n=${1:-2}
mkdir -p gpurun_out/ab
for i in $(seq 1 "$n"); do
  (cd ab_old && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > ../gpurun_out/ab/old_$i.json)
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/ab/new_$i.json
done
