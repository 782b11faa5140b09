// fp32 NCHW operator set: one entry point per host function of the reference's
// operator API (src/cuda_library/cuda_lib.h).  Deterministic (no float
// atomics), 64-bit indexing, no host synchronisation unless documented.
//
// Part 1 of 2: elementwise ops, batch norm, pooling, concat, losses,
// optimiser.  Convolution and transposed convolution live in conv_f32.hip.
#include <math.h>

#include <mutex>

#include "common.h"

#define BN_EPS 1e-8f
#define BN_MOMENTUM 0.9f

// ---------------------------------------------------------------------------
// Elementwise kernels.  Vector path: 16 bytes per lane when every pointer is
// 16-byte aligned; the scalar tail (and the unaligned case) goes element-wise.
// ---------------------------------------------------------------------------
namespace {

struct ReluF { __device__ float operator()(float x) const { return x > 0.f ? x : 0.f; } };
struct SigmF { __device__ float operator()(float x) const { return 1.0f / (1.0f + expf(-x)); } };
struct ReluBwdF { __device__ float operator()(float dy, float x) const { return x > 0.f ? dy : 0.f; } };
struct SigmBwdF { __device__ float operator()(float dy, float y) const { return dy * y * (1.0f - y); } };
struct LeakyF {
  float slope;
  __device__ float operator()(float x) const { return x > 0.f ? x : slope * x; }
};
struct LeakyBwdF {
  float slope;
  __device__ float operator()(float dy, float x) const { return x > 0.f ? dy : slope * dy; }
};
struct TanhF { __device__ float operator()(float x) const { return tanhf(x); } };
struct TanhBwdF { __device__ float operator()(float dy, float y) const { return dy * (1.0f - y * y); } };
struct AddF { __device__ float operator()(float a, float b) const { return a + b; } };
struct ThreshF {
  float t;
  __device__ float operator()(float x) const { return x >= t ? 1.0f : 0.0f; }
};
struct ScaleF {
  float a;
  __device__ float operator()(float x) const { return x * a; }
};
struct SgdF {
  float lr;
  __device__ float operator()(float w, float g) const { return w - g * lr; }
};
struct DiffLossF {
  float inv_n;
  // dLdY = (y - m)/n ; called as f(mask, y)
  __device__ float operator()(float m, float y) const { return (y - m) * inv_n; }
};

template <class F>
__global__ __launch_bounds__(256) void ew1_kernel(const float *a, float *o,
                                                  i64 n, int vec, F f) {
  const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 nth = (i64)gridDim.x * blockDim.x;
  i64 done = 0;
  if (vec) {
    const i64 n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    float4 *o4 = reinterpret_cast<float4 *>(o);
    for (i64 i = tid; i < n4; i += nth) {
      float4 v = a4[i];
      v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
      o4[i] = v;
    }
    done = n4 << 2;
  }
  for (i64 i = done + tid; i < n; i += nth) o[i] = f(a[i]);
}

template <class F>
__global__ __launch_bounds__(256) void ew2_kernel(const float *a, const float *b, float *o, i64 n,
                                                  int vec, F f) {
  const i64 tid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 nth = (i64)gridDim.x * blockDim.x;
  i64 done = 0;
  if (vec) {
    const i64 n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o4 = reinterpret_cast<float4 *>(o);
    for (i64 i = tid; i < n4; i += nth) {
      float4 u = a4[i], v = b4[i], r;
      r.x = f(u.x, v.x); r.y = f(u.y, v.y); r.z = f(u.z, v.z); r.w = f(u.w, v.w);
      o4[i] = r;
    }
    done = n4 << 2;
  }
  for (i64 i = done + tid; i < n; i += nth) o[i] = f(a[i], b[i]);
}

inline int aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

template <class F>
int launch_ew1(const float *a, float *o, i64 n, F f, unet_stream_t s, const char *name) {
  UNET_REQUIRE(n >= 0, "%s: negative element count", name);
  if (n == 0) return UNET_OK;
  UNET_REQUIRE(a && o, "%s: NULL pointer", name);
  const int vec = aligned16(a) && aligned16(o);
  hipLaunchKernelGGL(ew1_kernel<F>, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, as_stream(s), a, o, n, vec, f);
  UNET_LAUNCH_CHECK(name);
  return UNET_OK;
}
template <class F>
int launch_ew2(const float *a, const float *b, float *o, i64 n, F f, unet_stream_t s, const char *name) {
  UNET_REQUIRE(n >= 0, "%s: negative element count", name);
  if (n == 0) return UNET_OK;
  UNET_REQUIRE(a && b && o, "%s: NULL pointer", name);
  const int vec = aligned16(a) && aligned16(b) && aligned16(o);
  hipLaunchKernelGGL(ew2_kernel<F>, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, as_stream(s), a, b, o, n, vec, f);
  UNET_LAUNCH_CHECK(name);
  return UNET_OK;
}

// ---- Adam -------------------------------------------------------------------
// One parameter's update (cuda_ops.cu:405-441: Adam with the weight decay added to the step).  The fused
// multiply-adds are written out so that both kernels below round alike: left to the compiler's contraction
// choices they differed in the last bit for about one parameter in ten thousand.
__device__ __forceinline__ void adam_update(float &w, float g, float &m, float &v, float lr, float c1, float c2) {
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, wd = 1e-4f;
  m = fmaf(b1, m, (1.0f - b1) * g);
  v = fmaf(b2, v, (1.0f - b2) * (g * g));
  const float mh = m / c1, vh = v / c2;
  w = fmaf(-lr, fmaf(wd, w, mh / (sqrtf(vh) + eps)), w);
}
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ w, const float *__restrict__ g,
                                                   float2 *__restrict__ st, i64 n, float lr, float c1,
                                                   float c2) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    float2 mv = st[i];
    float wi = w[i];
    adam_update(wi, g[i], mv.x, mv.y, lr, c1, c2);
    st[i] = mv;
    w[i] = wi;
  }
}

// Four parameters per lane (16-byte accesses), a contiguous piece of the vectors per workgroup.  The moments are
// touched once per step and are twice the size of the parameters: they stream past the caches (non-temporal),
// as does the gradient; the parameters themselves are written normally (the weight repack reads them next).
typedef float adam_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void adam_vec4_kernel(adam_f32x4 *__restrict__ w, const adam_f32x4 *__restrict__ g,
                                                        adam_f32x4 *__restrict__ st, i64 n4, int tail, float lr,
                                                        float c1, float c2) {
  constexpr int U = 2;
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) {  // the up to three values behind the last 16-byte piece
    float *ws = reinterpret_cast<float *>(w + n4) + threadIdx.x;
    float2 *ss = reinterpret_cast<float2 *>(st + 2 * n4) + threadIdx.x;
    float2 mv = *ss;
    float wi = *ws;
    adam_update(wi, reinterpret_cast<const float *>(g + n4)[threadIdx.x], mv.x, mv.y, lr, c1, c2);
    *ss = mv;
    *ws = wi;
  }
  const i64 base = (i64)blockIdx.x * (256 * U);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const i64 i = base + u * 256 + threadIdx.x;
    if (i >= n4) break;
    const adam_f32x4 gi = __builtin_nontemporal_load(g + i);
    adam_f32x4 s0 = __builtin_nontemporal_load(st + 2 * i), s1 = __builtin_nontemporal_load(st + 2 * i + 1);
    adam_f32x4 wi = w[i];
    float m[4] = {s0.x, s0.z, s1.x, s1.z}, v[4] = {s0.y, s0.w, s1.y, s1.w};
    float wv[4] = {wi.x, wi.y, wi.z, wi.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_update(wv[j], gi[j], m[j], v[j], lr, c1, c2);
    wi = adam_f32x4{wv[0], wv[1], wv[2], wv[3]};
    s0 = adam_f32x4{m[0], v[0], m[1], v[1]};
    s1 = adam_f32x4{m[2], v[2], m[3], v[3]};
    __builtin_nontemporal_store(s0, st + 2 * i);
    __builtin_nontemporal_store(s1, st + 2 * i + 1);
    w[i] = wi;
  }
}

// ---- init ---------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void init_he_kernel(float *w, i64 n, float scale, uint64_t seed) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix64(seed * 0xD6E8FEB86659FD93ull + (uint64_t)i);
    // two uniforms in (0,1] and [0,1) from the 64 random bits -> Box-Muller
    const float u1 = ((float)((r >> 40) + 1)) * (1.0f / 16777216.0f);
    const float u2 = ((float)((r >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);
    w[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2) * scale;
  }
}
__global__ __launch_bounds__(256) void init_bn_kernel(float *w, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
    w[i] = (i & 1) ? 0.0f : 1.0f;
}
__global__ __launch_bounds__(256) void has_nan_kernel(const float *x, i64 n, int *flag) {
  int bad = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
    bad |= (x[i] != x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ __launch_bounds__(256) void f32_to_i32_kernel(const float *__restrict__ in, int32_t *__restrict__ out, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
  {
    const float v = in[i];  // NaN, infinities and anything beyond the int32 range: -1 (no class)
    out[i] = (v >= -2147483648.f && v < 2147483648.f) ? (int32_t)lrintf(v) : -1;
  }
}
// out[n][i] = index of the largest of x[n][0..C)[i] (first on ties), as a float
__global__ __launch_bounds__(256) void argmax_channels_kernel(const float *__restrict__ x, float *__restrict__ out, int N,
                                                              int C, i64 HW) {
  const i64 total = (i64)N * HW;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const i64 n = i / HW, p = i - n * HW;
    const float *xp = x + n * C * HW + p;
    float best = xp[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = xp[(i64)c * HW];
      if (v > best) {
        best = v;
        arg = c;
      }
    }
    out[i] = (float)arg;
  }
}
// *flag <- value if *flag was 0 and some x[i] is NaN or +-Inf
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float *__restrict__ x, i64 n, int *flag, int value) {
  int bad = 0;
  const i64 n4 = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (i64)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    // x - x is 0 for finite x and NaN otherwise
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad |= (t != 0.0f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    bad |= (v - v != 0.0f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicCAS(flag, 0, value);
}

// ---- batch norm ---------------------------------------------------------------
// Channel c's elements are the N planes x[(n*C+c)*HW .. +HW).  A block reduces
// slice `s` of the flattened (n, i) range; partial[(c*S+s)*2 + {0,1}].
// RELU (MODE 1): dy is the gradient behind a ReLU whose output is ymask: dy counts where ymask > 0.
template <int MODE, bool RELU = false>  // 0: (sum x, sum x^2)   1: (sum dy*xhat, sum dy)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x,
                                                         const float *__restrict__ dy,
                                                         const float *__restrict__ stats,
                                                         float *__restrict__ partial, int C, i64 HW,
                                                         i64 M, int S, float Mstat,
                                                         const float *__restrict__ ymask = nullptr) {
  __shared__ float red[2 * 4];
  const int c = blockIdx.y, s = blockIdx.x;
  const i64 chunk = (M + S - 1) / S;
  const i64 j0 = (i64)s * chunk;
  const i64 j1 = (j0 + chunk < M) ? j0 + chunk : M;
  float mean = 0.f, inv = 0.f;
  if (MODE == 1) {
    mean = stats[2 * c] / Mstat;  // Mstat: the element count the statistics were taken over (all ranks when synchronised)
    float var = stats[2 * c + 1] / Mstat - mean * mean;
    var = var < 0.f ? 0.f : var;
    inv = 1.0f / sqrtf(var + BN_EPS);
  }
  float acc[2] = {0.f, 0.f};
  // (n, i) of element j advance with j: one 64-bit division per thread, not per element
  i64 j = j0 + threadIdx.x;
  i64 n = j / HW, i = j - n * HW;
  // eight elements per trip: their loads are in flight together, the additions stay in element order
  constexpr int U = 8;
  for (; j + (U - 1) * (i64)blockDim.x < j1; j += U * (i64)blockDim.x) {
    float v[U], g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      while (i >= HW) {
        i -= HW;
        ++n;
      }
      const i64 idx = (n * C + c) * HW + i;
      v[u] = x[idx];
      g[u] = MODE == 1 ? dy[idx] : 0.f;
      if (MODE == 1 && RELU) g[u] = ymask[idx] > 0.f ? g[u] : 0.f;
      i += blockDim.x;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) {
        acc[0] += v[u];
        acc[1] += v[u] * v[u];
      } else {
        acc[0] += g[u] * ((v[u] - mean) * inv);
        acc[1] += g[u];
      }
    }
  }
  for (; j < j1; j += blockDim.x, i += blockDim.x) {
    while (i >= HW) {
      i -= HW;
      ++n;
    }
    const i64 idx = (n * C + c) * HW + i;
    const float v = x[idx];
    if (MODE == 0) {
      acc[0] += v;
      acc[1] += v * v;
    } else {
      float g = dy[idx];
      if (RELU) g = ymask[idx] > 0.f ? g : 0.f;
      acc[0] += g * ((v - mean) * inv);
      acc[1] += g;
    }
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    partial[((i64)c * S + s) * 2] = acc[0];
    partial[((i64)c * S + s) * 2 + 1] = acc[1];
  }
}

// One wavefront per channel: lanes sum strided slices of the S partials in double, a fixed-order
// butterfly combines them; lane 0 writes the pair and (training forward only) updates the running
// statistics.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float *__restrict__ partial, float *__restrict__ out2,
                                                         float *__restrict__ running, int C, int S, float M) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double a = 0, b = 0;
  for (int s = lane; s < S; s += 64) {
    a += partial[((i64)c * S + s) * 2];
    b += partial[((i64)c * S + s) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (lane != 0) return;
  out2[2 * c] = (float)a;
  out2[2 * c + 1] = (float)b;
  if (running) {
    const float mean = (float)(a / M);
    float var = (float)(b / M - (a / M) * (a / M));
    var = var < 0.f ? 0.f : var;
    running[2 * c] = running[2 * c] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * mean;
    running[2 * c + 1] = running[2 * c + 1] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * var;
  }
}

// Statistics synchronised across data-parallel ranks: the per-channel sums leave this rank as doubles
// (sums[2c], sums[2c+1]), are added across ranks, and bn_publish_kernel then does what the tail of
// bn_finalize_kernel does, with the global element count M.
__global__ __launch_bounds__(64) void bn_finalize_f64_kernel(const float *__restrict__ partial, double *__restrict__ sums,
                                                             int C, int S) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double a = 0, b = 0;
  for (int s = lane; s < S; s += 64) {
    a += partial[((i64)c * S + s) * 2];
    b += partial[((i64)c * S + s) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (lane != 0) return;
  sums[2 * c] = a;
  sums[2 * c + 1] = b;
}
__global__ __launch_bounds__(64) void bn_publish_kernel(const double *__restrict__ sums, float *__restrict__ out2,
                                                        float *__restrict__ running, int C, float M) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const double a = sums[2 * c], b = sums[2 * c + 1];
  out2[2 * c] = (float)a;
  out2[2 * c + 1] = (float)b;
  if (running) {
    const float mean = (float)(a / M);
    float var = (float)(b / M - (a / M) * (a / M));
    var = var < 0.f ? 0.f : var;
    running[2 * c] = running[2 * c] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * mean;
    running[2 * c + 1] = running[2 * c + 1] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * var;
  }
}

// MODE 0: train normalise (eps outside sqrt, stats = raw sums)
// MODE 1: inference (eps inside sqrt, stats = [mean,var])
// MODE 2: backward dx (stats = raw sums, gw = [dgamma,dbeta])
// MODE 3: MODE 1 followed by ReLU (the same two roundings as the two passes, one pass over the tensor)
// RELU: MODE 0 followed by ReLU likewise; MODE 2 with dy masked by the ReLU's output ymask (ReLU backward folded in)
template <int MODE, bool RELU = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ dy,
                                                       float *__restrict__ out,
                                                       const float *__restrict__ gb,
                                                       const float *__restrict__ stats,
                                                       const float *__restrict__ gw, int C, i64 HW,
                                                       i64 planes, float M,
                                                       const float *__restrict__ ymask = nullptr) {
  // grid.y strides over planes (n*C + c); grid.x over the plane
  for (i64 p = blockIdx.y; p < planes; p += gridDim.y) {
    const int c = (int)(p % C);
    float a, b;  // out = a*x + b  (MODE 2: see below)
    float mean, inv, k1 = 0.f, k2 = 0.f;
    if (MODE == 1 || MODE == 3) {
      mean = stats[2 * c];
      inv = 1.0f / sqrtf(stats[2 * c + 1] + BN_EPS);
    } else {
      mean = stats[2 * c] / M;
      float var = stats[2 * c + 1] / M - mean * mean;
      var = var < 0.f ? 0.f : var;
      inv = (MODE == 0) ? 1.0f / (sqrtf(var) + BN_EPS) : 1.0f / sqrtf(var + BN_EPS);
    }
    const float gamma = gb[2 * c];
    if (MODE == 2) {
      k1 = gw[2 * c + 1] / M;  // dbeta/M
      k2 = gw[2 * c] / M;      // dgamma/M
      a = gamma * inv;
      b = 0.f;
    } else {
      a = gamma * inv;
      b = gb[2 * c + 1] - a * mean;
    }
    const float *xp = x + p * HW;
    float *op = out + p * HW;
    const float *gp = (MODE == 2) ? dy + p * HW : nullptr;
    const float *mp = (MODE == 2 && RELU) ? ymask + p * HW : nullptr;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (i64)gridDim.x * blockDim.x) {
      if (MODE == 2) {
        const float xh = (xp[i] - mean) * inv;
        float g = gp[i];
        if (RELU) g = mp[i] > 0.f ? g : 0.f;
        op[i] = a * (g - k1 - xh * k2);
      } else {
        const float y = MODE == 0 ? gamma * ((xp[i] - mean) * inv) + gb[2 * c + 1] : a * xp[i] + b;
        op[i] = (MODE == 3 || (MODE == 0 && RELU)) ? (y > 0.f ? y : 0.f) : y;
      }
    }
  }
}

// Slices of the (n, i) range per channel: about 16 K elements each, more when few channels
// would otherwise leave most of the chip idle (slices stay above 1 K elements).
inline int bn_splits(i64 M, int C) {
  i64 s = (M + 16383) / 16384;
  const i64 want = (1024 + C - 1) / C, cap = (M + 1023) / 1024;
  if (s < want) s = want < cap ? want : cap;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

// ---- raw image input ---------------------------------------------------------------
// uint8 [N][H][W][C] (the layout image files decode to) -> fp32 [N][C][H][W], out = in*scale[c] + shift[c]:
// the /255 and per-channel z-score of main.py:89-92 done on the device, so a batch crosses PCIe as bytes.
__global__ __launch_bounds__(256) void normalize_u8_nhwc_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift, int C, i64 HW) {
  const i64 n = blockIdx.y;
  const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const uint8_t *src = in + (n * HW + p) * C;
  for (int c = 0; c < C; ++c) out[(n * C + c) * HW + p] = fmaf((float)src[c], scale[c], shift[c]);
}
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, i64 n,
                                                        float scale) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = (float)in[i] * scale;
}

// ---- max pool -------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x,
                                                          float *__restrict__ y,
                                                          int32_t *__restrict__ indices,
                                                          uint8_t *__restrict__ tiemask, i64 total,
                                                          int H, int W, int Ho, int Wo, int k, int s) {
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (i64)gridDim.x * blockDim.x) {
    const int wo = (int)(o % Wo);
    const i64 t = o / Wo;
    const int ho = (int)(t % Ho);
    const i64 plane = t / Ho;
    const float *xp = x + plane * (i64)H * W;
    float m = -INFINITY;
    int arg = -1;
    for (int i = 0; i < k; ++i) {
      const int hi = ho * s + i;
      if (hi >= H) break;
      for (int j = 0; j < k; ++j) {
        const int wi = wo * s + j;
        if (wi >= W) break;
        const float v = xp[(i64)hi * W + wi];
        if (arg < 0 || v > m) {
          m = v;
          arg = hi * W + wi;
        }
      }
    }
    y[o] = m;
    if (indices) indices[o] = arg;
    if (tiemask) {
      unsigned bits = 0;
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
          const int hi = ho * s + i, wi = wo * s + j, bit = i * k + j;
          if (hi < H && wi < W && bit < 8 && xp[(i64)hi * W + wi] == m) bits |= 1u << bit;
        }
      tiemask[o] = (uint8_t)bits;
    }
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_ties_kernel(const float *__restrict__ dy,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ y,
                                                               float *__restrict__ dx, i64 total, int H,
                                                               int W, int Ho, int Wo, int s) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const i64 t = i / W;
    const int h = (int)(t % H);
    const i64 plane = t / H;
    const int ho = h / s, wo = w / s;
    float r = 0.f;
    if (ho < Ho && wo < Wo) {
      const i64 o = (plane * Ho + ho) * Wo + wo;
      r = (x[i] == y[o]) ? dy[o] : 0.f;
    }
    dx[i] = r;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const float *__restrict__ dy,
                                                              const int32_t *__restrict__ indices,
                                                              float *__restrict__ dx, i64 total, int H,
                                                              int W, int Ho, int Wo) {
  // dx was zeroed beforehand; windows do not overlap, so each target element
  // has a single writer.
  for (i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (i64)gridDim.x * blockDim.x) {
    const i64 plane = o / ((i64)Ho * Wo);
    const int idx = indices[o];
    if (idx >= 0 && idx < H * W) dx[plane * (i64)H * W + idx] = dy[o];
  }
}

// ---- concat ----------------------------------------------------------------------
// dst[n][coff + c][i] <- src[n][c][i]   (dst has Cd channels, src has Cs)
__global__ __launch_bounds__(256) void chan_copy_kernel(const float *__restrict__ src,
                                                        float *__restrict__ dst, i64 HW, int Cs,
                                                        int Cd, int coff, int batch, int to_dst) {
  const i64 per = (i64)Cs * HW;  // contiguous run per sample
  const i64 total = per * batch;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const i64 n = i / per, r = i - n * per;
    const i64 j = n * (i64)Cd * HW + (i64)coff * HW + r;
    if (to_dst)
      dst[j] = src[i];
    else
      dst[i] = src[j];
  }
}

// Element i of the cropped window [N][Cs][Hb][Wb] <-> big[n][c][y + oy][x + ox] (big: [N][Cs][Ha][Wa]) and
// cat[n][c][y][x] (cat: [N][Cd][Hb][Wb], channels 0..Cs-1).  BWD = false: cat <- big; true: big <- cat.
template <bool BWD>
__global__ __launch_bounds__(256) void crop_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, i64 total,
                                                        int Cs, int Cd, int Ha, int Wa, int Hb, int Wb, int oy, int ox) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wb);
    i64 r = i / Wb;
    const int y = (int)(r % Hb);
    r /= Hb;
    const int c = (int)(r % Cs);
    const i64 n = r / Cs;
    const i64 ib = ((n * Cs + c) * Ha + y + oy) * Wa + x + ox, ic = ((n * Cd + c) * Hb + y) * Wb + x;
    if (BWD)
      dst[ib] = src[ic];
    else
      dst[ic] = src[ib];
  }
}

// ---- dice ------------------------------------------------------------------------
// Sample n, slice blockIdx.y of its pixels -> out[(n*S + slice)*3 + {0,1,2}] (S = gridDim.y; with
// S == 1 `out` is the loss_values array itself).  SQ: slot 0 holds sum (p - m)^2 instead of sum m p (the
// difference loss).
template <bool SQ>
__global__ __launch_bounds__(1024) void dice_sums_kernel(const float *__restrict__ mask,
                                                         const float *__restrict__ pred,
                                                         float *__restrict__ out, i64 HW) {
  __shared__ float red[3 * 16];
  const int n = blockIdx.x, S = gridDim.y;
  const i64 chunk = (HW + S - 1) / S, i0 = (i64)blockIdx.y * chunk, i1 = (i0 + chunk < HW) ? i0 + chunk : HW;
  float acc[3] = {0.f, 0.f, 0.f};
  for (i64 i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float m = mask[n * HW + i], p = pred[n * HW + i];
    acc[0] += SQ ? (p - m) * (p - m) : m * p;
    acc[1] += m;
    acc[2] += p;
  }
  block_sum<3>(acc, red);
  if (threadIdx.x == 0) {
    float *o = out + ((i64)n * S + blockIdx.y) * 3;
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
  }
}
// lv[3n + k] = sum over slices, in slice order
__global__ void dice_sums_finish_kernel(const float *__restrict__ part, float *__restrict__ lv, int N, int S) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * N) return;
  const int n = t / 3, k = t - 3 * n;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += part[((i64)n * S + s) * 3 + k];
  lv[t] = a;
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const float *__restrict__ mask,
                                                       const float *__restrict__ lv,
                                                       float *__restrict__ g, i64 HW, float scale) {
  const int n = blockIdx.y;
  const float I = lv[3 * n];
  const float D = lv[3 * n + 1] + lv[3 * n + 2] + 1e-5f;
  const float inv = scale * 2.0f / (D * D);
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (i64)gridDim.x * blockDim.x)
    g[n * HW + i] = (mask[n * HW + i] * D - I) * inv;
}

// ---- softmax cross entropy ---------------------------------------------------------
// A label outside [0, C) marks its pixel as ignored: no loss, zero gradient, and the mean runs over the other
// pixels only (what torch's ignore_index does for one value, here for every value that is not a class).
__global__ __launch_bounds__(256) void xent_count_kernel(const int32_t *__restrict__ target, int *__restrict__ counts,
                                                         int C, i64 P) {
  __shared__ int red[4];
  int n = 0;
  for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x)
    n += (unsigned)target[p] < (unsigned)C;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void xent_kernel(const float *__restrict__ logits,
                                                   const int32_t *__restrict__ target,
                                                   float *__restrict__ dlogits,
                                                   float *__restrict__ partial, const int *__restrict__ counts,
                                                   int C, i64 HW, i64 P) {
  __shared__ float red[4];
  __shared__ float inv_s;
  if (threadIdx.x == 0) {  // every block adds up the per-block counts of valid labels (at most 1024, in order)
    i64 valid = 0;
    for (unsigned b = 0; b < gridDim.x; ++b) valid += counts[b];
    inv_s = valid > 0 ? 1.0f / (float)valid : 0.f;
  }
  __syncthreads();
  const float inv_p = inv_s;
  float acc[1] = {0.f};
  for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x) {
    const i64 n = p / HW, i = p - n * HW;
    const float *lp = logits + n * C * HW + i;
    const int t = target[p];
    const bool valid = (unsigned)t < (unsigned)C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(i64)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lp[(i64)c * HW] - mx);
    if (valid) acc[0] += logf(se) + mx - lp[(i64)t * HW];
    if (dlogits) {
      float *gp = dlogits + n * C * HW + i;
      const float r = valid ? inv_p / se : 0.f;
      for (int c = 0; c < C; ++c)
        gp[(i64)c * HW] = valid ? expf(lp[(i64)c * HW] - mx) * r - (c == t ? inv_p : 0.0f) : 0.f;
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
__global__ void xent_finalize_kernel(const float *partial, const int *counts, int nblocks, float *loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0;
    i64 valid = 0;
    for (int i = 0; i < nblocks; ++i) {
      s += partial[i];
      valid += counts[i];
    }
    *loss = valid > 0 ? (float)(s / (double)valid) : 0.f;
  }
}

}  // namespace

// ===========================================================================
extern "C" {

int unet_relu_f32(const float *inputs, float *activations, int64_t n, unet_stream_t s) {
  return launch_ew1(inputs, activations, n, ReluF(), s, "unet_relu_f32");
}
int unet_sigmoid_f32(const float *inputs, float *activations, int64_t n, unet_stream_t s) {
  return launch_ew1(inputs, activations, n, SigmF(), s, "unet_sigmoid_f32");
}
int unet_relu_backward_f32(const float *grad_activations, const float *inputs, float *grad_inputs,
                           int64_t n, unet_stream_t s) {
  return launch_ew2(grad_activations, inputs, grad_inputs, n, ReluBwdF(), s, "unet_relu_backward_f32");
}
int unet_sigmoid_backward_f32(const float *grad_activations, const float *activations,
                              float *grad_inputs, int64_t n, unet_stream_t s) {
  return launch_ew2(grad_activations, activations, grad_inputs, n, SigmBwdF(), s,
                    "unet_sigmoid_backward_f32");
}
int unet_leaky_relu_f32(const float *inputs, float *activations, int64_t n, float slope, unet_stream_t s) {
  return launch_ew1(inputs, activations, n, LeakyF{slope}, s, "unet_leaky_relu_f32");
}
int unet_leaky_relu_backward_f32(const float *grad_activations, const float *inputs, float *grad_inputs, int64_t n,
                                 float slope, unet_stream_t s) {
  return launch_ew2(grad_activations, inputs, grad_inputs, n, LeakyBwdF{slope}, s, "unet_leaky_relu_backward_f32");
}
int unet_tanh_f32(const float *inputs, float *activations, int64_t n, unet_stream_t s) {
  return launch_ew1(inputs, activations, n, TanhF(), s, "unet_tanh_f32");
}
int unet_tanh_backward_f32(const float *grad_activations, const float *activations, float *grad_inputs, int64_t n,
                           unet_stream_t s) {
  return launch_ew2(grad_activations, activations, grad_inputs, n, TanhBwdF(), s, "unet_tanh_backward_f32");
}
int unet_f32_to_i32(const float *in, int32_t *out, int64_t n, unet_stream_t s) {
  UNET_REQUIRE(in && out && n > 0, "unet_f32_to_i32: bad argument");
  hipLaunchKernelGGL(f32_to_i32_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(s), in, out, (i64)n);
  UNET_LAUNCH_CHECK("f32_to_i32_kernel");
  return UNET_OK;
}
int unet_argmax_channels_f32(const float *x, float *out, int N, int C, int64_t HW, unet_stream_t s) {
  UNET_REQUIRE(x && out && N > 0 && C > 0 && HW > 0, "unet_argmax_channels_f32: bad argument");
  hipLaunchKernelGGL(argmax_channels_kernel, dim3(ew_grid((i64)N * HW)), dim3(256), 0, as_stream(s), x, out, N, C, (i64)HW);
  UNET_LAUNCH_CHECK("argmax_channels_kernel");
  return UNET_OK;
}
int unet_add_b_to_a_f32(float *a, const float *b, int64_t n, unet_stream_t s) {
  return launch_ew2(a, b, a, n, AddF(), s, "unet_add_b_to_a_f32");
}
int unet_normalize_u8_nhwc_to_nchw_f32(const uint8_t *in, float *out, const float *scale, const float *shift, int N,
                                       int C, int H, int W, unet_stream_t s) {
  UNET_REQUIRE(in && out && scale && shift, "unet_normalize_u8_nhwc_to_nchw_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0, "unet_normalize_u8_nhwc_to_nchw_f32: bad sizes");
  const i64 HW = (i64)H * W;
  hipLaunchKernelGGL(normalize_u8_nhwc_kernel, dim3(cdiv_u(HW, 256), (unsigned)N), dim3(256), 0, as_stream(s), in, out,
                     scale, shift, C, HW);
  UNET_LAUNCH_CHECK("normalize_u8_nhwc_kernel");
  return UNET_OK;
}
int unet_u8_to_f32(const uint8_t *in, float *out, int64_t n, float scale, unet_stream_t s) {
  UNET_REQUIRE(in && out && n > 0, "unet_u8_to_f32: bad arguments");
  const i64 blocks = (n + 255) / 256;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(s), in,
                     out, (i64)n, scale);
  UNET_LAUNCH_CHECK("u8_to_f32_kernel");
  return UNET_OK;
}
int unet_threshold_f32(float *x, float threshold, int64_t n, unet_stream_t s) {
  return launch_ew1(x, x, n, ThreshF{threshold}, s, "unet_threshold_f32");
}
int unet_scale_f32(float *x, float alpha, int64_t n, unet_stream_t s) {
  return launch_ew1(x, x, n, ScaleF{alpha}, s, "unet_scale_f32");
}
int unet_update_weights_sgd_f32(float *weights, const float *dLdW, float lr, int64_t n, unet_stream_t s) {
  return launch_ew2(weights, dLdW, weights, n, SgdF{lr}, s, "unet_update_weights_sgd_f32");
}
int unet_difference_loss_backward_f32(const float *mask, const float *y, float *dLdY, int64_t n,
                                      unet_stream_t s) {
  UNET_REQUIRE(n > 0, "unet_difference_loss_backward_f32: total_size must be positive");
  return launch_ew2(mask, y, dLdY, n, DiffLossF{1.0f / (float)n}, s, "unet_difference_loss_backward_f32");
}

int unet_update_weights_adam_f32(float *weights, const float *dLdW, float *adam_parameters, float lr,
                                 int64_t n, int t, unet_stream_t s) {
  UNET_REQUIRE(n >= 0 && t >= 1, "unet_update_weights_adam_f32: need n >= 0 and step t >= 1");
  if (n == 0) return UNET_OK;
  UNET_REQUIRE(weights && dLdW && adam_parameters, "unet_update_weights_adam_f32: NULL pointer");
  const float c1 = (float)(1.0 - pow(0.9, (double)t));
  const float c2 = (float)(1.0 - pow(0.999, (double)t));
  // 16-byte pieces when the vectors are aligned for it (the up to three values behind the last piece go to
  // three lanes of the first workgroup), else one value per lane; both kernels do the same arithmetic per parameter
  const bool aligned = ((((uintptr_t)weights) | ((uintptr_t)dLdW)) & 15) == 0 && (((uintptr_t)adam_parameters) & 31) == 0;
  const i64 n4 = aligned ? (i64)n / 4 : 0;
  if (n4 > 0) {
    hipLaunchKernelGGL(adam_vec4_kernel, dim3((unsigned)((n4 + 511) / 512)), dim3(256), 0, as_stream(s),
                       reinterpret_cast<adam_f32x4 *>(weights), reinterpret_cast<const adam_f32x4 *>(dLdW),
                       reinterpret_cast<adam_f32x4 *>(adam_parameters), n4, (int)(n - 4 * n4), lr, c1, c2);
    UNET_LAUNCH_CHECK("adam_vec4_kernel");
  } else {
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(s), weights, dLdW,
                       reinterpret_cast<float2 *>(adam_parameters), (i64)n, lr, c1, c2);
    UNET_LAUNCH_CHECK("adam_kernel");
  }
  return UNET_OK;
}

int unet_init_he_f32(float *weights, int64_t n, int64_t fan, uint64_t seed, unet_stream_t s) {
  UNET_REQUIRE(fan > 0, "Can't pass N=0 when initializing with HE or XAVIER");
  if (n <= 0) return UNET_OK;
  UNET_REQUIRE(weights, "unet_init_he_f32: NULL pointer");
  hipLaunchKernelGGL(init_he_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(s), weights, (i64)n,
                     sqrtf(2.0f / (float)fan), seed);
  UNET_LAUNCH_CHECK("init_he_kernel");
  return UNET_OK;
}
int unet_init_bn_f32(float *weights, int64_t n, unet_stream_t s) {
  if (n <= 0) return UNET_OK;
  UNET_REQUIRE(weights, "unet_init_bn_f32: NULL pointer");
  hipLaunchKernelGGL(init_bn_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(s), weights, (i64)n);
  UNET_LAUNCH_CHECK("init_bn_kernel");
  return UNET_OK;
}
int unet_has_nans_f32(const float *ptr, int64_t n, int *has_nan_host, unet_stream_t s) {
  UNET_REQUIRE(has_nan_host, "unet_has_nans_f32: NULL result pointer");
  *has_nan_host = 0;
  if (!ptr || n <= 0) return UNET_OK;
  void *ws;
  int rc = unet_workspace(&ws, 256, as_stream(s));
  if (rc) return rc;
  UNET_HIP_TRY(hipMemsetAsync(ws, 0, 4, as_stream(s)));
  hipLaunchKernelGGL(has_nan_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(s), ptr, (i64)n, (int *)ws);
  UNET_LAUNCH_CHECK("has_nan_kernel");
  UNET_HIP_TRY(hipMemcpyAsync(has_nan_host, ws, 4, hipMemcpyDeviceToHost, as_stream(s)));
  UNET_HIP_TRY(hipStreamSynchronize(as_stream(s)));
  return UNET_OK;
}
int unet_flag_nonfinite_f32(const float *ptr, int64_t n, int *flag_device, int value, unet_stream_t s) {
  UNET_REQUIRE(flag_device && value != 0, "unet_flag_nonfinite_f32: need a flag pointer and a non-zero value");
  if (!ptr || n <= 0) return UNET_OK;
  UNET_REQUIRE(((uintptr_t)ptr & 15) == 0, "unet_flag_nonfinite_f32: the array must be 16-byte aligned");
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, as_stream(s), ptr, (i64)n,
                     flag_device, value);
  UNET_LAUNCH_CHECK("nonfinite_flag_kernel");
  return UNET_OK;
}

// ---- batch norm -----------------------------------------------------------------
static int bn_check(const char *name, int C, int H, int W, int N) {
  UNET_REQUIRE(C > 0 && H > 0 && W > 0 && N > 0, "%s: need positive C,H,W,batch (got %d,%d,%d,%d)", name, C, H,
               W, N);
  return UNET_OK;
}
static dim3 bn_apply_grid(i64 HW, i64 planes) {
  // up to 16 elements per thread: a workgroup first fetches its plane's coefficients (a chain of dependent loads),
  // and with one element per thread that prologue was most of its life
  static const int lab_ept = getenv("UNET_LAB_BN_APPLY_EPT") ? atoi(getenv("UNET_LAB_BN_APPLY_EPT")) : 16;
  unsigned gx = (unsigned)((HW + 256 * (i64)lab_ept - 1) / (256 * (i64)lab_ept));
  if (gx > 64) gx = 64;
  unsigned gy = planes > 16384 ? 16384u : (unsigned)planes;
  return dim3(gx, gy);
}

// The launches of the training forward; `partials` holds 2*C*S floats, every one of which the first kernel
// overwrites.  `sync_sums` (2*C doubles) is needed when the statistics are synchronised across ranks
// (unet_bn_set_sync): the channel sums are then added over all ranks and M counts all ranks' elements.
static int bn_train_launch(const float *inputs, float *activations, const float *weights, float *BN_batch_stats,
                           float *BN_running_stats, int C, i64 HW, int N, int S, float *partials, double *sync_sums,
                           unet_stream_t s, bool relu = false) {
  const i64 M = HW * N;
  float Mg = (float)M;
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(S, C), dim3(256), 0, as_stream(s), inputs,
                     (const float *)nullptr, (const float *)nullptr, partials, C, HW, M, S, Mg);
  UNET_LAUNCH_CHECK("bn_partial_kernel<0>");
  if (sync_sums) {
    hipLaunchKernelGGL(bn_finalize_f64_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(s), (const float *)partials,
                       sync_sums, C, S);
    UNET_LAUNCH_CHECK("bn_finalize_f64_kernel");
    int world = 1;
    int rc = unet_bn_sync_sum_f64(sync_sums, 2 * (i64)C, &world, as_stream(s));
    if (rc) return rc;
    Mg = (float)(M * world);
    hipLaunchKernelGGL(bn_publish_kernel, dim3(cdiv_u(C, 64)), dim3(64), 0, as_stream(s), (const double *)sync_sums,
                       BN_batch_stats, BN_running_stats, C, Mg);
    UNET_LAUNCH_CHECK("bn_publish_kernel");
  } else {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(s), (const float *)partials,
                       BN_batch_stats, BN_running_stats, C, S, Mg);
    UNET_LAUNCH_CHECK("bn_finalize_kernel");
  }
  if (relu)
    hipLaunchKernelGGL((bn_apply_kernel<0, true>), bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                       (const float *)nullptr, activations, weights, (const float *)BN_batch_stats,
                       (const float *)nullptr, C, HW, (i64)N * C, Mg, (const float *)nullptr);
  else
    hipLaunchKernelGGL((bn_apply_kernel<0, false>), bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                       (const float *)nullptr, activations, weights, (const float *)BN_batch_stats,
                       (const float *)nullptr, C, HW, (i64)N * C, Mg, (const float *)nullptr);
  UNET_LAUNCH_CHECK("bn_apply_kernel<0>");
  return UNET_OK;
}

static int bn_train_impl(const char *who, const float *inputs, float *activations, const float *weights,
                         float *BN_batch_stats, float *BN_running_stats, int C, int H, int W, int N, bool relu,
                         unet_stream_t s) {
  int rc = bn_check(who, C, H, W, N);
  if (rc) return rc;
  UNET_REQUIRE(inputs && activations && weights && BN_batch_stats, "%s: NULL pointer", who);
  const i64 HW = (i64)H * W, M = HW * N;
  const int S = bn_splits(M, C);
  // workspace: cross-rank sums [2C] double (synchronised statistics only) | partials [C][S][2]
  const size_t n_sync = unet_bn_sync_on() ? 4 * (size_t)C : 0;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (n_sync + 2 * (size_t)C * S), as_stream(s));
  if (rc) return rc;
  return bn_train_launch(inputs, activations, weights, BN_batch_stats, BN_running_stats, C, HW, N, S,
                         (float *)ws + n_sync, n_sync ? (double *)ws : nullptr, s, relu);
}

int unet_batch_norm_train_f32(const float *inputs, float *activations, const float *weights,
                              float *BN_batch_stats, float *BN_running_stats, int C, int H, int W,
                              int N, unet_stream_t s) {
  return bn_train_impl("unet_batch_norm_train_f32", inputs, activations, weights, BN_batch_stats, BN_running_stats, C,
                       H, W, N, false, s);
}

int unet_batch_norm_train_relu_f32(const float *inputs, float *activations, const float *weights,
                                   float *BN_batch_stats, float *BN_running_stats, int C, int H, int W,
                                   int N, unet_stream_t s) {
  return bn_train_impl("unet_batch_norm_train_relu_f32", inputs, activations, weights, BN_batch_stats,
                       BN_running_stats, C, H, W, N, true, s);
}

// ---- batch norm training forward for callers that must not allocate --------------------------------
// The per-slice partial sums live in a fixed array in the library's device image instead of the grow-only
// workspace.  Every GPU the library runs on holds its own copy of the array at its own address, looked up
// once per device; host code only ever uses that looked-up address.
#define UNET_BN_SCRATCH_FLOATS (256 * 1024) /* 1 MiB per GPU */
static __device__ __align__(256) float g_bn_scratch[UNET_BN_SCRATCH_FLOATS];
namespace {
constexpr int kMaxDevices = 64;
float *g_bn_scratch_ptr[kMaxDevices] = {};
std::mutex g_bn_scratch_mu;
}  // namespace

static int bn_scratch_of_current_device(float **out) {
  int dev = -1;
  UNET_HIP_TRY(hipGetDevice(&dev));
  UNET_REQUIRE(dev >= 0 && dev < kMaxDevices, "unet_batch_norm_train_noalloc_f32: device %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_bn_scratch_mu);
  if (!g_bn_scratch_ptr[dev]) {
    float *ptr = nullptr;
    UNET_HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(g_bn_scratch)));
    UNET_REQUIRE(ptr, "unet_batch_norm_train_noalloc_f32: the scratch array has no address on device %d", dev);
    g_bn_scratch_ptr[dev] = ptr;
  }
  *out = g_bn_scratch_ptr[dev];
  return UNET_OK;
}

int unet_batch_norm_train_noalloc_max_partials(void) { return UNET_BN_SCRATCH_FLOATS / 2; }

int unet_batch_norm_train_noalloc_f32(const float *inputs, float *activations, const float *weights,
                                      float *BN_batch_stats, float *BN_running_stats, int C, int H, int W,
                                      int N, unet_stream_t s) {
  int rc = bn_check("unet_batch_norm_train_noalloc_f32", C, H, W, N);
  if (rc) return rc;
  UNET_REQUIRE(inputs && activations && weights && BN_batch_stats,
               "unet_batch_norm_train_noalloc_f32: NULL pointer");
  const i64 HW = (i64)H * W, M = HW * N;
  const int S = bn_splits(M, C);
  UNET_REQUIRE((i64)2 * C * S <= UNET_BN_SCRATCH_FLOATS,
               "unet_batch_norm_train_noalloc_f32: %d channels x %d slices need %lld scratch floats, the fixed "
               "scratch holds %d; use unet_batch_norm_train_f32",
               C, S, (long long)2 * C * S, UNET_BN_SCRATCH_FLOATS);
  UNET_REQUIRE(!unet_bn_sync_on(), "unet_batch_norm_train_noalloc_f32: statistics synchronised across ranks "
                                   "(unet_bn_set_sync) are not supported here; use unet_batch_norm_train_f32");
  float *scratch = nullptr;
  rc = bn_scratch_of_current_device(&scratch);
  if (rc) return rc;
  return bn_train_launch(inputs, activations, weights, BN_batch_stats, BN_running_stats, C, HW, N, S, scratch, nullptr,
                         s);
}

int unet_batch_norm_inference_f32(const float *inputs, float *activations, const float *weights,
                                  const float *BN_stats, int H, int W, int C, int N, unet_stream_t s) {
  int rc = bn_check("unet_batch_norm_inference_f32", C, H, W, N);
  if (rc) return rc;
  UNET_REQUIRE(inputs && activations && weights && BN_stats, "unet_batch_norm_inference_f32: NULL pointer");
  const i64 HW = (i64)H * W;
  hipLaunchKernelGGL(bn_apply_kernel<1>, bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                     (const float *)nullptr, activations, weights, BN_stats, (const float *)nullptr, C, HW,
                     (i64)N * C, 1.0f);
  UNET_LAUNCH_CHECK("bn_apply_kernel<1>");
  return UNET_OK;
}

int unet_batch_norm_inference_relu_f32(const float *inputs, float *activations, const float *weights,
                                       const float *BN_stats, int H, int W, int C, int N, unet_stream_t s) {
  int rc = bn_check("unet_batch_norm_inference_relu_f32", C, H, W, N);
  if (rc) return rc;
  UNET_REQUIRE(inputs && activations && weights && BN_stats, "unet_batch_norm_inference_relu_f32: NULL pointer");
  const i64 HW = (i64)H * W;
  hipLaunchKernelGGL(bn_apply_kernel<3>, bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                     (const float *)nullptr, activations, weights, BN_stats, (const float *)nullptr, C, HW,
                     (i64)N * C, 1.0f);
  UNET_LAUNCH_CHECK("bn_apply_kernel<3>");
  return UNET_OK;
}

// relu_out != NULL: grad_activations is the gradient behind a ReLU with output relu_out (ReLU backward folded in)
static int bn_backward_impl(const char *who, const float *grad_activations, const float *relu_out, const float *inputs,
                            const float *BN_batch_stats, const float *weights, float *grad_inputs,
                            float *grad_weights, int C, int H, int W, int N, unet_stream_t s) {
  int rc = bn_check(who, C, H, W, N);
  if (rc) return rc;
  UNET_REQUIRE(grad_activations && inputs && BN_batch_stats && weights && grad_inputs && grad_weights,
               "%s: NULL pointer", who);
  const i64 HW = (i64)H * W, M = HW * N;
  const int S = bn_splits(M, C);
  const bool sync = unet_bn_sync_on();
  const size_t n_sync = sync ? 4 * (size_t)C : 0;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (n_sync + 2 * (size_t)C * S), as_stream(s));
  if (rc) return rc;
  float *partials = (float *)ws + n_sync;
  int world = 1;
  if (sync) {  // element count of the statistics: all ranks (the same count on every rank)
    rc = unet_bn_sync_sum_f64(nullptr, 0, &world, as_stream(s));
    if (rc) return rc;
  }
  const float Mg = (float)(M * world);
  if (relu_out)
    hipLaunchKernelGGL((bn_partial_kernel<1, true>), dim3(S, C), dim3(256), 0, as_stream(s), inputs, grad_activations,
                       BN_batch_stats, partials, C, HW, M, S, Mg, relu_out);
  else
    hipLaunchKernelGGL((bn_partial_kernel<1, false>), dim3(S, C), dim3(256), 0, as_stream(s), inputs, grad_activations,
                       BN_batch_stats, partials, C, HW, M, S, Mg, (const float *)nullptr);
  UNET_LAUNCH_CHECK("bn_partial_kernel<1>");
  if (sync) {
    double *sums = (double *)ws;
    hipLaunchKernelGGL(bn_finalize_f64_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(s), (const float *)partials,
                       sums, C, S);
    UNET_LAUNCH_CHECK("bn_finalize_f64_kernel");
    rc = unet_bn_sync_sum_f64(sums, 2 * (i64)C, &world, as_stream(s));
    if (rc) return rc;
    hipLaunchKernelGGL(bn_publish_kernel, dim3(cdiv_u(C, 64)), dim3(64), 0, as_stream(s), (const double *)sums,
                       grad_weights, (float *)nullptr, C, Mg);
    UNET_LAUNCH_CHECK("bn_publish_kernel");
  } else {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(s), (const float *)partials,
                       grad_weights, (float *)nullptr, C, S, Mg);
    UNET_LAUNCH_CHECK("bn_finalize_kernel");
  }
  if (relu_out)
    hipLaunchKernelGGL((bn_apply_kernel<2, true>), bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                       grad_activations, grad_inputs, weights, BN_batch_stats, (const float *)grad_weights, C, HW,
                       (i64)N * C, Mg, relu_out);
  else
    hipLaunchKernelGGL((bn_apply_kernel<2, false>), bn_apply_grid(HW, (i64)N * C), dim3(256), 0, as_stream(s), inputs,
                       grad_activations, grad_inputs, weights, BN_batch_stats, (const float *)grad_weights, C, HW,
                       (i64)N * C, Mg, (const float *)nullptr);
  UNET_LAUNCH_CHECK("bn_apply_kernel<2>");
  // the gradient all-reduce takes the mean over ranks: hand it this rank's share of the global sums
  if (sync && world > 1) return unet_scale_f32(grad_weights, 1.0f / (float)world, 2 * (i64)C, s);
  return UNET_OK;
}

int unet_batch_norm_backward_f32(const float *grad_activations, const float *inputs,
                                 const float *BN_batch_stats, const float *weights, float *grad_inputs,
                                 float *grad_weights, int C, int H, int W, int N, unet_stream_t s) {
  return bn_backward_impl("unet_batch_norm_backward_f32", grad_activations, nullptr, inputs, BN_batch_stats, weights,
                          grad_inputs, grad_weights, C, H, W, N, s);
}

int unet_batch_norm_relu_backward_f32(const float *grad_activations, const float *activations, const float *inputs,
                                      const float *BN_batch_stats, const float *weights, float *grad_inputs,
                                      float *grad_weights, int C, int H, int W, int N, unet_stream_t s) {
  UNET_REQUIRE(activations, "unet_batch_norm_relu_backward_f32: NULL pointer");
  return bn_backward_impl("unet_batch_norm_relu_backward_f32", grad_activations, activations, inputs, BN_batch_stats,
                          weights, grad_inputs, grad_weights, C, H, W, N, s);
}

// ---- max pool -------------------------------------------------------------------
int unet_max_pool_indices_f32(const float *inputs, float *activations, int32_t *indices,
                              uint8_t *tiemask, int N, int stride, int k, int C, int H, int W,
                              unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && stride > 0 && k > 0,
               "max pool: need positive batch, channels, H, W, stride, kernel");
  UNET_REQUIRE(inputs && activations, "max pool: NULL pointer");
  const int Ho = H / stride, Wo = W / stride;
  UNET_REQUIRE(Ho > 0 && Wo > 0, "max pool: output would be empty (H=%d W=%d stride=%d)", H, W, stride);
  const i64 total = (i64)N * C * Ho * Wo;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), inputs, activations,
                     indices, tiemask, total, H, W, Ho, Wo, k, stride);
  UNET_LAUNCH_CHECK("maxpool_fwd_kernel");
  return UNET_OK;
}
int unet_max_pool_f32(const float *inputs, float *activations, int N, int stride, int k, int C, int H,
                      int W, unet_stream_t s) {
  return unet_max_pool_indices_f32(inputs, activations, nullptr, nullptr, N, stride, k, C, H, W, s);
}
int unet_max_pool_backward_f32(const float *grad_activations, const float *inputs,
                               const float *activations, float *grad_inputs, int N, int stride, int C,
                               int H, int W, unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && stride > 0, "max pool backward: need positive sizes");
  UNET_REQUIRE(grad_activations && inputs && activations && grad_inputs, "max pool backward: NULL pointer");
  const i64 total = (i64)N * C * H * W;
  hipLaunchKernelGGL(maxpool_bwd_ties_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s),
                     grad_activations, inputs, activations, grad_inputs, total, H, W, H / stride, W / stride,
                     stride);
  UNET_LAUNCH_CHECK("maxpool_bwd_ties_kernel");
  return UNET_OK;
}
int unet_max_pool_backward_indices_f32(const float *grad_activations, const int32_t *indices,
                                       float *grad_inputs, int N, int C, int H, int W, int Ho, int Wo,
                                       unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "max pool backward: need positive sizes");
  UNET_REQUIRE(grad_activations && indices && grad_inputs, "max pool backward: NULL pointer");
  UNET_HIP_TRY(hipMemsetAsync(grad_inputs, 0, sizeof(float) * (size_t)N * C * H * W, as_stream(s)));
  const i64 total = (i64)N * C * Ho * Wo;
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s),
                     grad_activations, indices, grad_inputs, total, H, W, Ho, Wo);
  UNET_LAUNCH_CHECK("maxpool_bwd_idx_kernel");
  return UNET_OK;
}

// ---- concat ----------------------------------------------------------------------
int unet_concat_f32(const float *a1, const float *a2, float *out, int N, int64_t HxW, int C1, int C2,
                    unet_stream_t s) {
  UNET_REQUIRE(N > 0 && HxW > 0 && C1 >= 0 && C2 >= 0 && C1 + C2 > 0, "concat: bad sizes");
  UNET_REQUIRE(out && (a1 || C1 == 0) && (a2 || C2 == 0), "concat: NULL pointer");
  if (C1) {
    hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C1 * HxW)), dim3(256), 0, as_stream(s), a1, out,
                       (i64)HxW, C1, C1 + C2, 0, N, 1);
    UNET_LAUNCH_CHECK("chan_copy_kernel");
  }
  if (C2) {
    hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C2 * HxW)), dim3(256), 0, as_stream(s), a2, out,
                       (i64)HxW, C2, C1 + C2, C1, N, 1);
    UNET_LAUNCH_CHECK("chan_copy_kernel");
  }
  return UNET_OK;
}
int unet_concat_backward_f32(const float *dLdY, float *dLdX, int H, int W, int C1, int C2, int N,
                             unet_stream_t s) {
  UNET_REQUIRE(N > 0 && H > 0 && W > 0 && C1 >= 0 && C2 >= 0 && C1 + C2 > 0, "concat backward: bad sizes");
  UNET_REQUIRE(dLdY && dLdX, "concat backward: NULL pointer");
  const i64 HW = (i64)H * W;
  if (C1) {
    hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C1 * HW)), dim3(256), 0, as_stream(s), dLdY,
                       dLdX, HW, C1, C1 + C2, 0, N, 0);
    UNET_LAUNCH_CHECK("chan_copy_kernel");
  }
  if (C2) {
    hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C2 * HW)), dim3(256), 0, as_stream(s), dLdY,
                       dLdX + (i64)N * C1 * HW, HW, C2, C1 + C2, C1, N, 0);
    UNET_LAUNCH_CHECK("chan_copy_kernel");
  }
  return UNET_OK;
}

// Concat with the first tensor centre-cropped to the second one's size (see the header).
int unet_concat_crop_f32(const float *a1, const float *a2, float *out, int N, int C1, int C2, int Ha, int Wa, int Hb,
                         int Wb, unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C1 > 0 && C2 > 0 && Hb > 0 && Wb > 0 && Ha >= Hb && Wa >= Wb, "concat crop: bad sizes");
  UNET_REQUIRE(a1 && a2 && out, "concat crop: NULL pointer");
  const i64 total = (i64)N * C1 * Hb * Wb;
  hipLaunchKernelGGL(crop_copy_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), a1, out, total, C1, C1 + C2,
                     Ha, Wa, Hb, Wb, (Ha - Hb) / 2, (Wa - Wb) / 2);
  UNET_LAUNCH_CHECK("crop_copy_kernel");
  hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C2 * Hb * Wb)), dim3(256), 0, as_stream(s), a2, out,
                     (i64)Hb * Wb, C2, C1 + C2, C1, N, 1);
  UNET_LAUNCH_CHECK("chan_copy_kernel");
  return UNET_OK;
}
int unet_concat_crop_backward_f32(const float *dLdY, float *dA, float *dB, int N, int C1, int C2, int Ha, int Wa, int Hb,
                                  int Wb, unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C1 > 0 && C2 > 0 && Hb > 0 && Wb > 0 && Ha >= Hb && Wa >= Wb, "concat crop backward: bad sizes");
  UNET_REQUIRE(dLdY && dA && dB, "concat crop backward: NULL pointer");
  UNET_HIP_TRY(hipMemsetAsync(dA, 0, sizeof(float) * (size_t)N * C1 * Ha * Wa, as_stream(s)));
  const i64 total = (i64)N * C1 * Hb * Wb;
  hipLaunchKernelGGL(crop_copy_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), dLdY, dA, total, C1, C1 + C2,
                     Ha, Wa, Hb, Wb, (Ha - Hb) / 2, (Wa - Wb) / 2);
  UNET_LAUNCH_CHECK("crop_copy_kernel");
  hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid((i64)N * C2 * Hb * Wb)), dim3(256), 0, as_stream(s), dLdY, dB,
                     (i64)Hb * Wb, C2, C1 + C2, C1, N, 0);
  UNET_LAUNCH_CHECK("chan_copy_kernel");
  return UNET_OK;
}

// ---- losses ------------------------------------------------------------------------
static int loss_sums_launch(bool sq, const float *mask, const float *prediction, int H, int W, int N,
                            float *loss_values, unet_stream_t s) {
  auto kernel = sq ? dice_sums_kernel<true> : dice_sums_kernel<false>;
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "dice: need positive sizes");
  UNET_REQUIRE(mask && prediction && loss_values, "dice: NULL pointer");
  const i64 HW = (i64)H * W;
  int S = (int)((256 + N - 1) / N);                       // about one workgroup per CU in total
  const i64 cap = (HW + 4095) / 4096;                     // at least 4 K pixels per slice
  if (S > cap) S = (int)cap;
  if (S <= 1) {
    hipLaunchKernelGGL(kernel, dim3(N, 1), dim3(1024), 0, as_stream(s), mask, prediction, loss_values, HW);
    UNET_LAUNCH_CHECK("dice_sums_kernel");
    return UNET_OK;
  }
  void *ws;
  int rc = unet_workspace(&ws, sizeof(float) * 3 * (size_t)N * S, as_stream(s));
  if (rc) return rc;
  hipLaunchKernelGGL(kernel, dim3(N, S), dim3(1024), 0, as_stream(s), mask, prediction, (float *)ws, HW);
  UNET_LAUNCH_CHECK("dice_sums_kernel");
  hipLaunchKernelGGL(dice_sums_finish_kernel, dim3(cdiv_u(3 * N, 64)), dim3(64), 0, as_stream(s), (const float *)ws,
                     loss_values, N, S);
  UNET_LAUNCH_CHECK("dice_sums_finish_kernel");
  return UNET_OK;
}
int unet_dice_sums_f32(const float *mask, const float *prediction, int H, int W, int N, float *loss_values,
                       unet_stream_t s) {
  return loss_sums_launch(false, mask, prediction, H, W, N, loss_values, s);
}
int unet_difference_loss_sums_f32(const float *mask, const float *prediction, int H, int W, int N, float *loss_values,
                                  unet_stream_t s) {
  return loss_sums_launch(true, mask, prediction, H, W, N, loss_values, s);
}
int unet_dice_score_f32(const float *mask, const float *prediction, int H, int W, int N,
                        float *loss_values, float *dice_host, unet_stream_t s) {
  UNET_REQUIRE(dice_host, "dice: NULL result pointer");
  int rc = unet_dice_sums_f32(mask, prediction, H, W, N, loss_values, s);
  if (rc) return rc;
  float *h = new float[3 * (size_t)N];
  hipError_t e = hipMemcpyAsync(h, loss_values, sizeof(float) * 3 * (size_t)N, hipMemcpyDeviceToHost, as_stream(s));
  if (e == hipSuccess) e = hipStreamSynchronize(as_stream(s));
  if (e != hipSuccess) {
    delete[] h;
    return unet_set_error(UNET_ERR_HIP, "dice: copy to host failed: %s", hipGetErrorString(e));
  }
  float dice = 0.f;
  for (int i = 0; i < N; ++i) dice += 2.0f * h[3 * i] / (h[3 * i + 1] + h[3 * i + 2] + 1e-5f);
  delete[] h;
  *dice_host = dice / (float)N;
  return UNET_OK;
}
int unet_dice_loss_backward_f32(const float *mask, const float *prediction, const float *loss_values,
                                float *dLdY, int H, int W, int N, float scale, unet_stream_t s) {
  (void)prediction;
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "dice backward: need positive sizes");
  UNET_REQUIRE(mask && loss_values && dLdY, "dice backward: NULL pointer");
  const i64 HW = (i64)H * W;
  unsigned gx = (unsigned)((HW + 255) / 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(gx, N), dim3(256), 0, as_stream(s), mask, loss_values, dLdY, HW,
                     scale);
  UNET_LAUNCH_CHECK("dice_bwd_kernel");
  return UNET_OK;
}
int unet_softmax_xent_f32(const float *logits, const int32_t *target, float *dlogits, float *loss_out,
                          int N, int C, int H, int W, unet_stream_t s) {
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "cross entropy: need positive sizes");
  UNET_REQUIRE(logits && target && loss_out, "cross entropy: NULL pointer");
  const i64 HW = (i64)H * W, P = HW * N;
  unsigned nb = ew_grid(P);
  if (nb > 1024) nb = 1024;
  void *ws;
  int rc = unet_workspace(&ws, (sizeof(float) + sizeof(int)) * nb, as_stream(s));
  if (rc) return rc;
  float *partial = (float *)ws;
  int *counts = (int *)(partial + nb);
  hipLaunchKernelGGL(xent_count_kernel, dim3(nb), dim3(256), 0, as_stream(s), target, counts, C, P);
  UNET_LAUNCH_CHECK("xent_count_kernel");
  hipLaunchKernelGGL(xent_kernel, dim3(nb), dim3(256), 0, as_stream(s), logits, target, dlogits, partial,
                     (const int *)counts, C, HW, P);
  UNET_LAUNCH_CHECK("xent_kernel");
  hipLaunchKernelGGL(xent_finalize_kernel, dim3(1), dim3(64), 0, as_stream(s), (const float *)partial, (const int *)counts,
                     (int)nb, loss_out);
  UNET_LAUNCH_CHECK("xent_finalize_kernel");
  return UNET_OK;
}

}  // extern "C"
