// Laboratory for one question: on gfx950, does a packed-fp32 instruction that reads a 64-bit register pair see
// the new value of the pair's low register when an ordinary VALU instruction wrote it in the instruction right
// before?  The sequence is the one hipcc emitted in the epilogue of the transposed-convolution data-gradient
// kernel (fused batch-norm reduction):
//     v_pk_fma_f32 v[2:3], ...            t = sum - mean * sumsq      (packed, upper half unused)
//     v_mov_b32    v4, w                  the multiplier, into the low half of the pair v[4:5]
//     v_pk_mul_f32 v[2:3], v[4:5], v[2:3] lo = w * t
// Every lane repeats it ITERS times on its own operands and counts results that differ from the scalar
// v_fma_f32 / v_mul_f32 evaluation of the same numbers.  Variant 1 puts s_nop 0 between v_mov and v_pk_mul.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_lab.hip -o tools/bin/pk_hazard_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void lab_kernel(unsigned long long *bad, unsigned *first_bad, int iters) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long n_bad = 0;
  unsigned state = tid * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    const float z = 0.5f + (float)(state >> 8) * (1.0f / 16777216.0f);
    state = state * 1664525u + 1013904223u;
    const float q = 1.0f + (float)(state >> 8) * (1.0f / 16777216.0f);
    state = state * 1664525u + 1013904223u;
    const float s = 3.0f + (float)(state >> 8) * (1.0f / 16777216.0f);
    state = state * 1664525u + 1013904223u;
    const float w = 0.25f + (float)(state >> 8) * (1.0f / 16777216.0f);
    const float stale = 1000.0f + (float)it;   // what v4 holds before the v_mov: makes a stale read visible
    f32x2 zz = {z, 7.0f}, qq = {11.0f, q}, ss = {13.0f, s};
    float out;
    if (VARIANT == 0)
      asm volatile(
          "v_mov_b32 v4, %[stale]\n\t"
          "v_mov_b32 v5, %[stale]\n\t"
          "s_nop 7\n\t"
          "v_pk_fma_f32 v[2:3], %[zz], %[qq], %[ss] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
          "v_mov_b32 v4, %[w]\n\t"
          "v_pk_mul_f32 v[2:3], v[4:5], v[2:3]\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %[out], v2\n\t"
          : [out] "=v"(out)
          : [stale] "v"(stale), [zz] "v"(zz), [qq] "v"(qq), [ss] "v"(ss), [w] "v"(w)
          : "v2", "v3", "v4", "v5");
    else
      asm volatile(
          "v_mov_b32 v4, %[stale]\n\t"
          "v_mov_b32 v5, %[stale]\n\t"
          "s_nop 7\n\t"
          "v_pk_fma_f32 v[2:3], %[zz], %[qq], %[ss] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
          "v_mov_b32 v4, %[w]\n\t"
          "s_nop 0\n\t"
          "v_pk_mul_f32 v[2:3], v[4:5], v[2:3]\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %[out], v2\n\t"
          : [out] "=v"(out)
          : [stale] "v"(stale), [zz] "v"(zz), [qq] "v"(qq), [ss] "v"(ss), [w] "v"(w)
          : "v2", "v3", "v4", "v5");
    const float want = w * __builtin_fmaf(-z, q, s);
    if (out != want) {
      if (n_bad == 0) first_bad[tid] = (unsigned)it + 1u;
      ++n_bad;
    }
  }
  if (n_bad) atomicAdd(bad, n_bad);
}

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = 256 * 3 * 4, threads = 256;
  unsigned long long *bad;
  unsigned *first_bad;
  CHECK(hipMalloc(&bad, sizeof(*bad)));
  CHECK(hipMalloc(&first_bad, sizeof(unsigned) * blocks * threads));
  for (int variant = 0; variant < 2; ++variant) {
    CHECK(hipMemset(bad, 0, sizeof(*bad)));
    CHECK(hipMemset(first_bad, 0, sizeof(unsigned) * blocks * threads));
    if (variant == 0)
      hipLaunchKernelGGL(lab_kernel<0>, dim3(blocks), dim3(threads), 0, 0, bad, first_bad, iters);
    else
      hipLaunchKernelGGL(lab_kernel<1>, dim3(blocks), dim3(threads), 0, 0, bad, first_bad, iters);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("variant %d (%s): %llu wrong results of %llu\n", variant,
           variant == 0 ? "v_mov then v_pk_mul back to back" : "s_nop 0 between them", h,
           (unsigned long long)blocks * threads * iters);
  }
  return 0;
}
