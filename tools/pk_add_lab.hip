// Laboratory: plain packed fp32 adds / multiply-adds (v_pk_add_f32, v_pk_fma_f32 without op_sel -- what a
// vectorised reduction kernel such as a collective library's is made of) beside the library's 3x3
// weight-gradient kernel on a second stream.  Each lane checks every packed result against the scalar one.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_add_lab.hip -o tools/bin/pk_add_lab -ldl
//   ./tools/bin/pk_add_lab <iters> <wgrad launches on the second stream, 0 = none>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void pk_kernel(unsigned long long *bad, const float *src, int iters) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long n_bad = 0;
  unsigned state = tid * 2654435761u + 777u;
  f32x2 acc_p = {0.f, 0.f};
  float acc_s0 = 0.f, acc_s1 = 0.f;
  for (int it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    const float a0 = src[state & 0xffff], a1 = src[(state >> 16) & 0xffff];
    state = state * 1664525u + 1013904223u;
    const float b0 = src[state & 0xffff], b1 = src[(state >> 16) & 0xffff];
    f32x2 a = {a0, a1}, b = {b0, b1}, sum, fma;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(sum) : "v"(a), "v"(b));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(fma) : "v"(a), "v"(b), "v"(acc_p));
    float s0, s1;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(a0), "v"(b0));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a1), "v"(b1));
    const float f0 = __builtin_fmaf(a0, b0, acc_s0), f1 = __builtin_fmaf(a1, b1, acc_s1);
    n_bad += (sum.x != s0) + (sum.y != s1) + (fma.x != f0) + (fma.y != f1);
    acc_p = fma * 0.5f;
    acc_s0 = f0 * 0.5f;
    acc_s1 = f1 * 0.5f;
    acc_p.x = acc_s0;  // resynchronise, so that one wrong result is counted once
    acc_p.y = acc_s1;
  }
  if (n_bad) atomicAdd(bad, n_bad);
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
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int corun = argc > 2 ? atoi(argv[2]) : 0;
  unsigned long long *bad, h = 0;
  float *src;
  CHECK(hipMalloc(&bad, sizeof(*bad)));
  CHECK(hipMalloc(&src, sizeof(float) * 65536));
  float *hs = (float *)malloc(sizeof(float) * 65536);
  for (int i = 0; i < 65536; ++i) hs[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  CHECK(hipMemcpy(src, hs, sizeof(float) * 65536, hipMemcpyHostToDevice));
  hipStream_t s_main, s_side;
  CHECK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
  typedef int (*wgrad_fn)(const void *, int, const void *, int, float *, int, int, int, int, int, int, int, void *);
  wgrad_fn wgrad = nullptr;
  void *wg_dz = nullptr, *wg_x = nullptr;
  float *wg_g = nullptr;
  if (corun) {
    void *lib = dlopen("unet_amd/lib/libunet_hip.so", RTLD_NOW);
    if (!lib) {
      fprintf(stderr, "dlopen: %s\n", dlerror());
      return 1;
    }
    wgrad = (wgrad_fn)dlsym(lib, "unet_conv3x3_wgrad_bf16");
    CHECK(hipMalloc(&wg_dz, 2 * 16 * 16 * 16 * 1024));
    CHECK(hipMalloc(&wg_x, 2 * 16 * 16 * 16 * 512));
    CHECK(hipMalloc(&wg_g, sizeof(float) * 512 * 1024 * 9));
    CHECK(hipMemset(wg_dz, 0x3c, 2 * 16 * 16 * 16 * 1024));
    CHECK(hipMemset(wg_x, 0x3c, 2 * 16 * 16 * 16 * 512));
  }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(bad, 0, sizeof(*bad)));
    CHECK(hipDeviceSynchronize());
    for (int i = 0; i < corun; ++i)
      if (wgrad(wg_dz, 1024, wg_x, 512, wg_g, 16, 16, 16, 512, 1024, 512, 1024, s_side)) return 1;
    hipLaunchKernelGGL(pk_kernel, dim3(256 * 8), dim3(256), 0, s_main, bad, src, iters);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("%s: %llu wrong results of %llu\n", corun ? "beside the weight-gradient kernel" : "alone", h,
           (unsigned long long)256 * 8 * 256 * iters * 4);
  }
  return 0;
}
