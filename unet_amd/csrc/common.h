// Shared helpers for the HIP sources of libunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/unet_hip.h"

typedef int64_t i64;

// Records `msg` as the calling thread's last error and returns `code`.
int unet_set_error(int code, const char *fmt, ...);

#define UNET_HIP_TRY(expr)                                                          \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      return unet_set_error(UNET_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                            hipGetErrorString(e_), __FILE__, __LINE__);             \
  } while (0)

#define UNET_REQUIRE(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) return unet_set_error(UNET_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// After a kernel launch: report launch-configuration errors.
#define UNET_LAUNCH_CHECK(name)                                                     \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess)                                                           \
      return unet_set_error(UNET_ERR_HIP, "launch of %s failed: %s", name,          \
                            hipGetErrorString(e_));                                 \
  } while (0)

// Grow-only scratch used by reductions that need partial results: one block per (device, stream),
// so launchers working on different streams never share it.  The returned memory is valid until
// the next unet_workspace() call for the same device and stream that asks for more bytes.
int unet_workspace(void **ptr, size_t bytes, hipStream_t stream);
void unet_workspace_release_all();

// Batch-norm statistics shared across data-parallel ranks (comm.hip): whether a communicator was selected with
// unet_bn_set_sync, and the in-place cross-rank sum of `count` doubles on stream `s` (*world = number of ranks).
bool unet_bn_sync_on();
int unet_bn_sync_sum_f64(double *data, i64 count, int *world, hipStream_t s);

static inline hipStream_t as_stream(unet_stream_t s) { return (hipStream_t)s; }

// ---- per-kernel timing (unet_prof_* in the C ABI) ---------------------------
// When enabled, a ProfScope around a launch records two HIP events on the
// launch stream; unet_prof_get() later sums the elapsed times per kind.
enum UnetProfKind {
  PROF_CONV_FWD_F32 = 0,
  PROF_CONV_DGRAD_F32,
  PROF_CONV_WGRAD_F32,
  PROF_TCONV_F32,
  PROF_BN_F32,
  PROF_CONV_FWD_BF16,      // conv_mfma_kernel<9,*> as forward
  PROF_CONV_DGRAD_BF16,    // conv_mfma_kernel<9,*> as data gradient
  PROF_CONV_WGRAD_BF16,    // wgrad_mfma_kernel<9>
  PROF_WGRAD_REDUCE,       // wgrad_reduce(_tile)_kernel
  PROF_TCONV_BF16,         // conv_mfma_kernel<1,*> (transposed conv forward and data gradient)
  PROF_TCONV_WGRAD_BF16,   // wgrad_mfma_kernel<1>
  PROF_BN_ACT_BF16,        // bn_act_kernel
  PROF_BN_BWD_REDUCE_BF16, // bn_bwd_reduce_kernel
  PROF_BN_BWD_APPLY_BF16,  // bn_bwd_apply_kernel
  PROF_POOL_BF16,
  PROF_CONV_HEAD_BF16,     // head_fwd / head_dgrad / head_wgrad kernels (one-channel 3x3 convolution)
  PROF_KINDS
};
bool unet_prof_on();
void unet_prof_mark(int kind, double flops, double bytes, hipStream_t s, bool begin);
struct ProfScope {
  int kind;
  hipStream_t s;
  bool on;
  ProfScope(int k, double flops, double bytes, hipStream_t st) : kind(k), s(st), on(unet_prof_on()) {
    if (on) unet_prof_mark(kind, flops, bytes, s, true);
  }
  ~ProfScope() {
    if (on) unet_prof_mark(kind, 0, 0, s, false);
  }
};

static inline unsigned cdiv_u(i64 a, i64 b) { return (unsigned)((a + b - 1) / b); }

// Grid for grid-stride elementwise kernels: enough blocks to fill 256 CUs
// several times over, capped so very large tensors loop instead.
static inline unsigned ew_grid(i64 work_items, int block = 256) {
  i64 g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (unsigned)g;
}

#ifdef __HIPCC__
// 64-lane wavefront sum via xor shuffles; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of NV values per thread for blocks of up to 1024 threads.
// `red` must hold NV * (blockDim.x/64) floats.  Result valid in thread 0
// (all lanes of wave 0 hold it).
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (nw == 1) return;
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wid * NV + i] = v[i];
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[w * NV + i];
      v[i] = s;
    }
  }
}
#endif
