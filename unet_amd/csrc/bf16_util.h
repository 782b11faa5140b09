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
