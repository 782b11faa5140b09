#!/bin/sh
# Builds the kernel laboratory (tools/conv_lab.hip) against the in-tree libunet_hip.so.
# Run from anywhere; the binary lands in tools/bin/ (not tracked) and is started on a GPU box with
#   gpurun -- 'timeout -k 10 120 ./tools/bin/conv_lab'
set -e
cd "$(dirname "$0")"
mkdir -p bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form conv_lab.hip -o bin/conv_lab \
  -L../unet_amd/lib -lunet_hip -Wl,-rpath,'$ORIGIN/../../unet_amd/lib'
