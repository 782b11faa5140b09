#!/bin/bash
# Builds (anywhere) or runs (on a GPU box) the ablation variants of the pipelined 3x3 convolution laboratory:
#   tools/run_pipe_ablations.sh build      -> tools/bin/conv_lab2_abl{0..4}
#   tools/run_pipe_ablations.sh            -> runs them, one table per variant
# 0 = complete kernel, 1 = no output stores, 2 = no MFMAs, 3 = no halo loads, 4 = no operand loads after the first stage.
# Ablations change what the chip clocks at, so they bound a component's share from above, no more.
set -e
cd "$(dirname "$0")"
if [ "$1" = build ]; then
  mkdir -p bin
  for v in 0 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -DPIPE_LAB_ABLATE=$v conv_lab2.hip \
      -o bin/conv_lab2_abl$v -L../unet_amd/lib -lunet_hip -Wl,-rpath,'$ORIGIN/../../unet_amd/lib' 2>&1 | grep -v "recognized feature" || true
  done
  exit 0
fi
for v in 0 1 2 3 4; do
  echo "== PIPE_LAB_ABLATE=$v"
  timeout -k 10 120 ./bin/conv_lab2_abl$v
done
