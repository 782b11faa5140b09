#!/bin/bash
# Builds laboratory copies of the library that differ only in the epilogue of the transposed-convolution data
# gradient with the fused batch-norm reduction (conv_bf16.hip), to study a run-to-run difference of its sums:
#   libunet_dpp_plain.so       keep_scalar() (bf16_util.h) does nothing: the products are compiled to packed fp32
#   libunet_dpp_nop0/nop4.so   keep_scalar() holds s_nop 0 / s_nop 4 instead of an empty asm statement
#   libunet_dpp_nodpp.so       plain + every reduction stage through ds_bpermute_b32 (no DPP instruction)
#   libunet_dpp_fence_*.so     plain + one instruction pinned in front of the packed block
#                              (s_nop 7, s_waitcnt vmcnt(0), s_waitcnt lgkmcnt(0), s_barrier)
# all under tools/bin/.  Run on the GPU with
#   UNET_LAB_LIB=tools/bin/<lib> REPS=60 python tools/diag_tconv_bred.py [n cp cn h w]
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
/opt/rocm/bin/hipcc $FLAGS '-DUNET_DPP_OPERAND_ASM="s_nop 4"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_nop4.o &
/opt/rocm/bin/hipcc $FLAGS '-DUNET_DPP_OPERAND_ASM="s_nop 0"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_nop0.o &
P="-DUNET_DPP_OPERAND_PLAIN"
/opt/rocm/bin/hipcc $FLAGS $P '-DUNET_LAB_BRED_FENCE="s_nop 7"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_fence_nop.o &
/opt/rocm/bin/hipcc $FLAGS $P '-DUNET_LAB_BRED_FENCE="s_waitcnt vmcnt(0)"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_fence_vm.o &
/opt/rocm/bin/hipcc $FLAGS $P '-DUNET_LAB_BRED_FENCE="s_waitcnt lgkmcnt(0)"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_fence_lgkm.o &
/opt/rocm/bin/hipcc $FLAGS $P '-DUNET_LAB_BRED_FENCE="s_barrier"' -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_fence_bar.o &
/opt/rocm/bin/hipcc $FLAGS $P -DUNET_LAB_NO_DPP -c unet_amd/csrc/conv_bf16.hip -o /tmp/dpp_lab/conv_nodpp.o &
wait
for v in plain nop4 nop0 nodpp fence_nop fence_vm fence_lgkm fence_bar; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libunet_dpp_$v.so $OBJS /tmp/dpp_lab/conv_$v.o
done
ls -la tools/bin/libunet_dpp_*.so
