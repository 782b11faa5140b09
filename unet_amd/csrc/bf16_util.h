// Types and small device helpers shared by the bf16 MFMA kernels (gfx950).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return (uint32_t) __builtin_bit_cast(unsigned short, h);
}

// Two transposing 64-bit LDS reads (ds_read_b64_tr_b16) -> one 8-element K-contiguous MFMA fragment.
__device__ __forceinline__ bf16x8 tr_read_pair(const unsigned char *p0, const unsigned char *p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(__attribute__((address_space(3))) void *)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(__attribute__((address_space(3))) void *)(p1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// Sum over the lanes of a wavefront that share (lane % CH), CH a power of two in [4, 32]; every lane gets
// the total.  The stages inside a 16-lane row are DPP row rotations (VALU, no LDS traffic); the two stages
// across rows are __shfl_xor (ds_bpermute_b32).  The gfx950 lane-swap instructions (v_permlane16/32_swap)
// make those VALU-only as well; without a wait on their operand the sums varied from run to run (see
// dpp_operand below), with it they are exact but no faster on the whole step, so they are not used.
// The operand passes through a wait (s_nop 4, tied to the register): whatever VALU instruction produced it then
// retires at least five wait states before the DPP instruction reads it across lanes.  Two is the documented
// distance from a VALU write to a DPP read; the margin is wider because without the wait one build of the
// transposed-convolution data-gradient kernel -- a packed fp32 multiply scheduled right before the rotation --
// returned sums that differed between runs on identical inputs.
#ifndef UNET_DPP_OPERAND_ASM
#define UNET_DPP_OPERAND_ASM "s_nop 4"  // tools/dpp_lab.sh builds the library with other bodies ("" = no wait)
#endif
__device__ __forceinline__ float dpp_operand(float v) {
#ifndef UNET_DPP_OPERAND_PLAIN
  asm volatile(UNET_DPP_OPERAND_ASM : "+v"(v));
#endif
  return v;
}
__device__ __forceinline__ float lane_rot_sum8(float v) {  // + lane (i + 8) % 16 of the same 16-lane row
  v = dpp_operand(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_rot_sum4(float v) {  // + lane (i + 4) % 16 of the same 16-lane row
  v = dpp_operand(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));
}
template <int CH>
__device__ __forceinline__ float lane_class_sum(float v) {
  static_assert(CH == 4 || CH == 8 || CH == 16 || CH == 32, "lane_class_sum: CH must be 4, 8, 16 or 32");
  if (CH <= 4) {
    v = lane_rot_sum8(v);  // {i, i+8}
    v = lane_rot_sum4(v);  // {i, i+4, i+8, i+12}: the row's lanes of this class
  } else if (CH <= 8) {
    v = lane_rot_sum8(v);
  }
  if (CH <= 16) v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
