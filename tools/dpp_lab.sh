#!/bin/bash
# Builds laboratory copies of the library that differ only in what dpp_operand() (bf16_util.h) puts between
# the producer of a value and the DPP row rotation that reads it:
#   tools/bin/libunet_dpp_plain.so   nothing (the compiler is free to pair the producers into packed fp32 ops)
#   tools/bin/libunet_dpp_empty.so   an empty asm statement tied to the register (no wait states)
#   tools/bin/libunet_dpp_nop0.so    s_nop 0
# Run on the GPU with:  UNET_LAB_LIB=tools/bin/<lib> REPS=40 python tools/diag_tconv_bred.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin /tmp/dpp_lab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
OBJS=""
for f in unet_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b != conv_bf16 ]; then
    extra=""; [ $b = wgrad_bf16 ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o /tmp/dpp_lab/$b.o &
    OBJS="$OBJS /tmp/dpp_lab/$b.o"
  fi
done
/opt/rocm/bin/hipcc $FLAGS -DUNET_DPP_OPERAND_PLAIN -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_plain.o &
/opt/rocm/bin/hipcc $FLAGS '-DUNET_DPP_OPERAND_ASM=""' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_empty.o &
/opt/rocm/bin/hipcc $FLAGS '-DUNET_DPP_OPERAND_ASM="s_nop 0"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_nop0.o &
wait
for v in plain empty nop0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libunet_dpp_$v.so $OBJS /tmp/dpp_lab/conv_$v.o
done
ls -la tools/bin/libunet_dpp_*.so
