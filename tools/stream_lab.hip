// Streaming-pass laboratory: what rate a read + write pass over a 134 MB bf16 tensor (16 x 256 x 256 pixels
// x 64 channels, the largest activation of the benchmark graph) reaches on this part, by access shape and cache
// policy.  The batch-norm passes of the library (bn_act, bn_bwd_apply) are such passes; this program tells
// how far below the copy rate they sit and which knob moves them.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/bin/stream_lab
//   gpurun -- 'timeout -k 10 120 ./tools/bin/stream_lab'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ u32x4 ld(const u32x4 *p) {
  if (NT & 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
  if (NT & 2)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
__device__ __forceinline__ u32x4 touch(u32x4 v) {  // a little arithmetic so that the pass is not a pure copy
  v.x ^= 0x00010001u;
  return v;
}

// grid-stride, U independent 16-byte loads in flight per lane (the shape of bn_act_kernel)
template <int NT, int U>
__global__ __launch_bounds__(256) void copy_stride(const u32x4 *__restrict__ a, u32x4 *__restrict__ o, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(o + i + u * stride, touch(v[u]));
  }
  for (; i < n; i += stride) st<NT>(o + i, touch(ld<NT>(a + i)));
}

// every workgroup owns one contiguous range; U loads in flight per lane, consecutive 4 KB rows
template <int NT, int U>
__global__ __launch_bounds__(256) void copy_block(const u32x4 *__restrict__ a, u32x4 *__restrict__ o, long n) {
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long b0 = (long)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  long i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(o + i + u * 256, touch(v[u]));
  }
  for (; i < b1; i += 256) st<NT>(o + i, touch(ld<NT>(a + i)));
}

// two inputs, one output (the shape of bn_bwd_apply: dy and z in, dz out), block-contiguous or strided
template <int NT, int U, bool BLOCK>
__global__ __launch_bounds__(256) void axpy2(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
                                              u32x4 *__restrict__ o, long n) {
  long i, end, step;
  if (BLOCK) {
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long b0 = (long)blockIdx.x * per;
    end = b0 + per < n ? b0 + per : n;
    i = b0 + threadIdx.x;
    step = 256;
  } else {
    step = (long)gridDim.x * 256;
    i = (long)blockIdx.x * 256 + threadIdx.x;
    end = n;
  }
  for (; i + (U - 1) * step < end; i += U * step) {
    u32x4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = ld<NT>(a + i + u * step);
      w[u] = ld<NT>(b + i + u * step);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(o + i + u * step, touch(v[u] ^ w[u]));
  }
  for (; i < end; i += step) st<NT>(o + i, touch(ld<NT>(a + i) ^ ld<NT>(b + i)));
}

int main(int argc, char **argv) {
  const long bytes = 16L * 256 * 256 * 64 * 2, n = bytes / 16;
  const int SETS = 6, REPS = 24;
  std::vector<u32x4 *> A(SETS), B(SETS), O(SETS);
  for (int s = 0; s < SETS; ++s) {
    CK(hipMalloc(&A[s], bytes));
    CK(hipMalloc(&B[s], bytes));
    CK(hipMalloc(&O[s], bytes));
    CK(hipMemset(A[s], 1, bytes));
    CK(hipMemset(B[s], 2, bytes));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](const char *name, int streams, auto launch) {
    for (int s = 0; s < SETS; ++s) launch(s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < REPS; ++r) launch(r % SETS);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / REPS;
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, us, streams * (double)bytes / us * 1e-6);
  };
  time("hipMemcpyAsync D2D", 2, [&](int s) { CK(hipMemcpyAsync(O[s], A[s], bytes, hipMemcpyDeviceToDevice, 0)); });
  const int grids[] = {1024, 2048, 4096, 8192, 32768};
  char name[128];
  for (int g : grids) {
#define RUN1(K, NT, U)                                                                     \
  snprintf(name, sizeof name, #K " nt=%d U=%d grid=%d", NT, U, g);                         \
  time(name, 2, [&](int s) { K<NT, U><<<g, 256>>>(A[s], O[s], n); });
    RUN1(copy_stride, 0, 4)
    RUN1(copy_stride, 3, 4)
    RUN1(copy_stride, 1, 4)
    RUN1(copy_stride, 2, 4)
    RUN1(copy_stride, 0, 8)
    RUN1(copy_stride, 3, 8)
    RUN1(copy_block, 0, 4)
    RUN1(copy_block, 3, 4)
    RUN1(copy_block, 0, 8)
    RUN1(copy_block, 3, 8)
#define RUN2(NT, U, BL)                                                                    \
  snprintf(name, sizeof name, "axpy2 nt=%d U=%d %s grid=%d", NT, U, BL ? "block" : "stride", g); \
  time(name, 3, [&](int s) { axpy2<NT, U, BL><<<g, 256>>>(A[s], B[s], O[s], n); });
    RUN2(0, 2, false)
    RUN2(3, 2, false)
    RUN2(0, 4, false)
    RUN2(3, 4, false)
    RUN2(0, 2, true)
    RUN2(3, 2, true)
    RUN2(0, 4, true)
    RUN2(3, 4, true)
  }
  return 0;
}
