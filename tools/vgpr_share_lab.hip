// Laboratory: do two kernels that each hold ~248 vector registers per lane keep their register contents when
// their waves share a SIMD (2 x 248 <= 512)?  Kernel "hold" fills NREG registers with a pattern, idles for a
// while (re-touching them so that they stay allocated in registers), and counts registers that no longer hold
// the pattern.  Kernel "churn" on a second stream rewrites as many registers of its own in a loop.
//   hipcc --offload-arch=gfx950 -O2 tools/vgpr_share_lab.hip -o tools/bin/vgpr_share_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int NREG = 232;

__global__ __launch_bounds__(256, 1) void hold_kernel(unsigned long long *bad, int rounds) {
  unsigned r[NREG];
  const unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    r[i] = seed + 0x9E3779B9u * (unsigned)i;
    asm volatile("" : "+v"(r[i]));
  }
  for (int it = 0; it < rounds; ++it) {
    __builtin_amdgcn_s_sleep(8);
#pragma unroll
    for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));
  }
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    unsigned s2 = seed;
    asm volatile("" : "+v"(s2));  // the expected value is recomputed here, not kept in a register of its own
    n += (r[i] != s2 + 0x9E3779B9u * (unsigned)i);
  }
  if (n) atomicAdd(bad, (unsigned long long)n);
}

__global__ __launch_bounds__(256, 1) void churn_kernel(unsigned *out, int rounds) {
  unsigned r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = threadIdx.x + i;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      r[i] = r[i] * 1664525u + 1013904223u;
      asm volatile("" : "+v"(r[i]));
    }
  }
  unsigned t = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) t ^= r[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

#define CHECK(x)                                              \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long *bad, h = 0;
  unsigned *out;
  CHECK(hipMalloc(&bad, sizeof(*bad)));
  CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * 1024));
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (int with_churn = 0; with_churn < 2; ++with_churn) {
    CHECK(hipMemset(bad, 0, sizeof(*bad)));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(hold_kernel, dim3(256), dim3(256), 0, s0, bad, rounds);
    if (with_churn) hipLaunchKernelGGL(churn_kernel, dim3(1024), dim3(256), 0, s1, out, rounds / 8);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("%s: %llu registers lost their value (of %llu)\n", with_churn ? "beside the churn kernel" : "alone", h,
           (unsigned long long)256 * 256 * NREG);
  }
  return 0;
}
