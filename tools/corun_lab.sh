#!/bin/bash
# Laboratory copies of the library whose 3x3 weight-gradient kernel (wgrad_bf16.hip) differs in one ingredient,
# used as the co-runner of tools/pk_hazard_lab2 (UNET_LAB_LIB=tools/bin/<lib> ./tools/bin/pk_hazard_lab2 200 8 -40):
#   libunet_wg_agpr.so     MFMA accumulators in the accumulation registers (no -amdgpu-mfma-vgpr-form)
#   libunet_wg_nomfma.so   the MFMAs replaced by one VALU multiply-add each
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin /tmp/corun_lab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
OBJS=""
for f in unet_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b != wgrad_bf16 ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/corun_lab/$b.o &
    OBJS="$OBJS /tmp/corun_lab/$b.o"
  fi
done
/opt/rocm/bin/hipcc $FLAGS -c unet_amd/csrc/wgrad_bf16.hip -o /tmp/corun_lab/wg_agpr.o &
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form -DWGRAD_LAB_NO_MFMA -c unet_amd/csrc/wgrad_bf16.hip -o /tmp/corun_lab/wg_nomfma.o &
wait
for v in agpr nomfma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libunet_wg_$v.so $OBJS /tmp/corun_lab/wg_$v.o
done
ls -la tools/bin/libunet_wg_*.so
