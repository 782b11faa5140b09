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
// across rows are __shfl_xor (ds_bpermute_b32).
//
// keep_scalar(): an empty asm statement tied to the value's register.  Its only effect is on code shape: the
// arithmetic that produced the value is emitted as scalar fp32 instructions instead of being paired with a
// neighbour's into packed ones (v_pk_fma_f32 / v_pk_mul_f32).  It is here because of a measured failure of the
// packed form on this hardware: packed fp32 arithmetic of one wave returned wrong results while waves of an
// MFMA kernel of this library ran on the same SIMD.
//   - First seen in the epilogue of the transposed-convolution data gradient with the fused batch-norm
//     reduction (conv_mfma_kernel<1, 64, true, 64>, three workgroups per CU): the products
//     (sum g*z - mean * sum g) * invstd, compiled to
//         v_pk_fma_f32 ... op_sel:[0,1,1] op_sel_hi:[1,0,0]   (low result from the operands' high halves)
//         v_mov_b32    vN, w
//         v_pk_mul_f32 v[M:M+1], v[N:N+1], v[M:M+1]            (upper halves unused)
//     differed between runs on identical inputs -- only sum g*xhat, only the channels whose product went
//     through the op_sel form, only in workgroups that retire while others (in their MFMA loops) run on the CU;
//     batches that fit one round of resident workgroups never showed it.
//   - Then in the pooled batch-norm backward reduction (fused_bf16.hip) running beside the 3x3 weight-gradient
//     kernel on a second stream: wrong sums of g*xhat for half of the lanes, every input intact.
//   - Reproduced outside the library: the 34 instructions of the first case replayed verbatim in a kernel of
//     their own give 0 wrong products of 2.5e9 alone, beside loops of MFMAs (either accumulator form), of
//     transposing or plain LDS reads, and beside the weight-gradient kernel with its MFMAs removed -- and about
//     4500 wrong ones (even channels only: the op_sel form) beside the real weight-gradient kernel
//     (tools/pk_hazard_lab2.hip, tools/corun_lab.sh).
// What it is not: a DPP hazard (the nearest producer sits 19 instructions before the first row rotation, and a
// build without any DPP instruction fails alike), a matter of wait states (s_nop 0, s_nop 4 and the empty asm all
// cure it; s_nop 7, s_waitcnt vmcnt(0) / lgkmcnt(0) or s_barrier pinned in front of the packed block do not), or
// clobbered inputs.  The mechanism inside the SIMD is not isolated further; the packed form is avoided where a
// result feeds a reduction (here, conv3x3_pipe.h, the pooled reduction), and two tests keep watch:
// test_fused_bn_reduction_is_bit_identical_between_runs and tests/test_concurrency_gpu.py.
#ifndef UNET_DPP_OPERAND_ASM
#define UNET_DPP_OPERAND_ASM ""  // tools/dpp_lab.sh builds the library with other bodies
#endif
__device__ __forceinline__ float keep_scalar(float v) {
#ifndef UNET_DPP_OPERAND_PLAIN
  asm volatile(UNET_DPP_OPERAND_ASM : "+v"(v));
#endif
  return v;
}
__device__ __forceinline__ float lane_rot_sum8(float v) {  // + lane (i + 8) % 16 of the same 16-lane row
  v = keep_scalar(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_rot_sum4(float v) {  // + lane (i + 4) % 16 of the same 16-lane row
  v = keep_scalar(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));
}
template <int CH>
__device__ __forceinline__ float lane_class_sum(float v) {
  static_assert(CH == 4 || CH == 8 || CH == 16 || CH == 32, "lane_class_sum: CH must be 4, 8, 16 or 32");
#ifdef UNET_LAB_NO_DPP  // tools/dpp_lab.sh: every stage through ds_bpermute_b32
  if (CH <= 4) v += __shfl_xor(v, 4, 64);
  if (CH <= 8) v += __shfl_xor(v, 8, 64);
  if (CH <= 16) v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
#endif
  if (CH <= 4) {
    v = lane_rot_sum8(v);  // {i, i+8}
    v = lane_rot_sum4(v);  // {i, i+4, i+8, i+12}: the row's lanes of this class
  } else if (CH <= 8) {
    v = lane_rot_sum8(v);
  }
  if (CH <= 16) v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
