// Bandwidth-bound kernels of the bf16 / NHWC path: batch-norm finalise,
// fused BN + ReLU|Sigmoid apply, fused activation-backward + BN-backward (two
// passes: reduce, apply), 2x2 max pool with a saved tie mask, and its backward.
//
// Every kernel moves 16 bytes (8 bf16 channels of one pixel) per lane; channel
// reductions keep 8 channels' partial sums in registers per lane, combine the
// lanes that share channels through LDS, write one partial per block, and a
// tiny finalise kernel adds the partials in a fixed order (no atomics).
#include "common.h"

typedef uint16_t bf16_t;
#define BN_EPS 1e-8f
#define BN_MOMENTUM 0.9f

namespace {

__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return (uint32_t) __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void unpack8(const uint4 &u, float (&v)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = bf2f(w[j] & 0xffffu);
    v[2 * j + 1] = bf2f(w[j] >> 16);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = f2bf(v[2 * j]) | (f2bf(v[2 * j + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16-byte loads and stores of the streaming passes.  NT: non-temporal.  Measured on 134 MB bf16 tensors, read +
// write, a pass alone: 5.0 -> 6.1 TB/s with non-temporal loads and stores and contiguous 8-32 KiB pieces per
// workgroup.  Inside the training step the passes take their INPUTS non-temporal (read once, then dead until the
// backward pass) and store their OUTPUTS normally: the convolution that reads an output right after the pass
// finds part of it in the caches and gains more (3x3 forward -3 %, weight gradient -1 %) than the pass loses
// (bn_act +5 %) -- step 6.75 -> 6.69 ms, same-box A/B of the four combinations.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16(const bf16_t *p) {
  const u32x4_t *q = reinterpret_cast<const u32x4_t *>(p);
  const u32x4_t v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void st16(bf16_t *p, const uint4 &u) {
  const u32x4_t v = {u.x, u.y, u.z, u.w};
  u32x4_t *q = reinterpret_cast<u32x4_t *>(p);
  if (NT)
    __builtin_nontemporal_store(v, q);
  else
    *q = v;
}

// Per-channel record produced by the forward finalise and consumed by the
// apply / backward kernels.
struct ChanFwd {
  float a, b;        // y = act(a*z + b): a = gamma/(sqrt(var)+eps) [train] or gamma/sqrt(var+eps) [inference]
  float mean, invb;  // batch mean and 1/sqrt(var+eps) (backward's x-hat)
};
struct ChanBwd {
  float c0, k1, k2, pad;  // dz = c0*(g - k1 - xhat*k2): c0 = gamma*invb, k1 = dbeta/M, k2 = dgamma/M
};

// Sums column pair (2c, 2c+1) of partials [P][CoP][2] with one workgroup of 256 threads per
// channel: threads stride over the P partials with eight independent loads in flight, accumulate
// in double, combine with xor shuffles and a fixed-order sum over the four wavefronts
// (reproducible).  Every thread of the workgroup returns the totals.
template <typename T>  // float: per-tile / per-block partials; double: sums already combined across ranks
__device__ __forceinline__ void block_sum_partials(const T *__restrict__ partials, int P, int CoP, int c,
                                                    double &s, double &q) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a = 0, b = 0;
  int p = tid;
  for (; p + 7 * 256 < P; p += 8 * 256) {
    T v[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u][0] = partials[((i64)(p + u * 256) * CoP + c) * 2];
      v[u][1] = partials[((i64)(p + u * 256) * CoP + c) * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += v[u][0];
      b += v[u][1];
    }
  }
  for (; p < P; p += 256) {
    a += partials[((i64)p * CoP + c) * 2];
    b += partials[((i64)p * CoP + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = a;
    red[1][wave] = b;
  }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

// partials [P][CoP][2] (sum, sum of squares) -> batch_stats, running stats, ChanFwd
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const T *__restrict__ partials, int P, int C,
                                                              int CoP, float M, const float *__restrict__ gb,
                                                              float *__restrict__ batch_stats,
                                                              float *__restrict__ running,
                                                              ChanFwd *__restrict__ out) {
  const int c = blockIdx.x;  // one workgroup per channel
  ChanFwd r{0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    double s, q;
    block_sum_partials(partials, P, CoP, c, s, q);
    const float mean = (float)(s / M);
    float var = (float)(q / M - (s / M) * (s / M));
    var = var < 0.f ? 0.f : var;
    if (threadIdx.x == 0) {
      batch_stats[2 * c] = (float)s;
      batch_stats[2 * c + 1] = (float)q;
      if (running) {
        running[2 * c] = running[2 * c] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * mean;
        running[2 * c + 1] = running[2 * c + 1] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * var;
      }
    }
    r.a = gb[2 * c] / (sqrtf(var) + BN_EPS);
    r.b = gb[2 * c + 1] - r.a * mean;
    r.mean = mean;
    r.invb = 1.0f / sqrtf(var + BN_EPS);
  }
  if (threadIdx.x == 0) out[c] = r;
}

// inference: a, b from the running statistics
__global__ void bn_inference_coef_kernel(const float *__restrict__ gb, const float *__restrict__ running, int C,
                                         int CoP, ChanFwd *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CoP) return;
  ChanFwd r{0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    r.invb = 1.0f / sqrtf(running[2 * c + 1] + BN_EPS);
    r.a = gb[2 * c] * r.invb;
    r.mean = running[2 * c];
    r.b = gb[2 * c + 1] - r.a * r.mean;
  }
  out[c] = r;
}

// y = act(a*z + b), 8 channels per lane.  ACT: 0 ReLU, 1 sigmoid, 2 identity, 3 leaky ReLU (slope ap), 4 tanh.
template <int ACT>
__device__ __forceinline__ float act_value(float t, float ap) {
  if (ACT == 0) return t > 0.f ? t : 0.f;
  if (ACT == 1) return 1.0f / (1.0f + __expf(-t));
  if (ACT == 3) return t > 0.f ? t : ap * t;
  if (ACT == 4) return 1.0f - 2.0f / (1.0f + __expf(2.0f * t));
  return t;
}
// A lane keeps its channel group for all of its pixels, so the 16 coefficients
// live in registers.  A workgroup owns `ppb` consecutive pixels (a contiguous piece of the tensor when the
// pixel stride equals the channel count).
template <int ACT, int NT>  // NT bit 0: non-temporal loads, bit 1: non-temporal stores
__global__ __launch_bounds__(256) void bn_act_kernel(const bf16_t *__restrict__ z, int z_ld, bf16_t *__restrict__ y,
                                                     int y_ld, const ChanFwd *__restrict__ cf, int CG, int PL,
                                                     i64 pixels, i64 ppb, float ap) {
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= PL) return;
  float ca[8], cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const ChanFwd c = cf[cg * 8 + j];
    ca[j] = c.a;
    cb[j] = c.b;
  }
  auto apply = [&](const uint4 &in) {
    float v[8];
    unpack8(in, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = act_value<ACT>(fmaf(ca[j], v[j], cb[j]), ap);
    }
    return pack8(v);
  };
  const i64 p0 = (i64)blockIdx.x * ppb, p1 = min(pixels, p0 + ppb);
  i64 p = p0 + pl;
  // four pixel rows per trip: four independent 16-byte loads in flight per lane
  for (; p + 3 * PL < p1; p += 4 * PL) {
    uint4 in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) in[u] = ld16<(NT & 1) != 0>(z + (p + u * PL) * z_ld + cg * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) st16<(NT & 2) != 0>(y + (p + u * PL) * y_ld + cg * 8, apply(in[u]));
  }
  for (; p < p1; p += PL) st16<(NT & 2) != 0>(y + p * y_ld + cg * 8, apply(ld16<(NT & 1) != 0>(z + p * z_ld + cg * 8)));
}

// ---- backward -------------------------------------------------------------------
// g = (dy1 + dy2) * act'(.)   with act' from the recomputed BN output a*z + b
template <int ACT>
__device__ __forceinline__ float act_grad(float dy, float t, float ap = 0.f) {
  if (ACT == 0) return t > 0.f ? dy : 0.f;
  if (ACT == 1) {
    const float s = 1.0f / (1.0f + __expf(-t));
    return dy * s * (1.0f - s);
  }
  if (ACT == 3) return t > 0.f ? dy : ap * dy;
  if (ACT == 4) {
    const float y = 1.0f - 2.0f / (1.0f + __expf(2.0f * t));
    return dy * (1.0f - y * y);
  }
  return dy;
}

// pass 1: partial[b][c][2] = (sum g*xhat, sum g) over block b's pixels
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t *__restrict__ dy1, int ld1,
                                                            const bf16_t *__restrict__ dy2, int ld2,
                                                            const bf16_t *__restrict__ z, int z_ld,
                                                            const ChanFwd *__restrict__ cf, int CG, int PL,
                                                            i64 pixels, i64 ppb, float *__restrict__ partial,
                                                            int CoP, float ap) {
  extern __shared__ float red[];  // [PL][CG*16]
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  const bool active = pl < PL;
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  if (active) {
    ChanFwd c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = cf[cg * 8 + j];
    const i64 p0 = (i64)blockIdx.x * ppb, p1 = min(pixels, p0 + ppb);
    auto accumulate = [&](const uint4 &r1, const uint4 &r2, const uint4 &rz) {
      float g[8], zv[8];
      unpack8(r1, g);
      if (dy2) {
        float g2[8];
        unpack8(r2, g2);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += g2[j];
      }
      unpack8(rz, zv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gj = act_grad<ACT>(g[j], fmaf(c[j].a, zv[j], c[j].b), ap);
        s[2 * j] += gj * ((zv[j] - c[j].mean) * c[j].invb);
        s[2 * j + 1] += gj;
      }
    };
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    i64 p = p0 + pl;
    // two pixels per trip keep four to six independent 16-byte loads in flight per lane
    for (; p + PL < p1; p += 2 * PL) {
      const uint4 a1 = *reinterpret_cast<const uint4 *>(dy1 + p * ld1 + cg * 8);
      const uint4 b1 = *reinterpret_cast<const uint4 *>(dy1 + (p + PL) * ld1 + cg * 8);
      const uint4 az = *reinterpret_cast<const uint4 *>(z + p * z_ld + cg * 8);
      const uint4 bz = *reinterpret_cast<const uint4 *>(z + (p + PL) * z_ld + cg * 8);
      uint4 a2 = zero4, b2 = zero4;
      if (dy2) {
        a2 = *reinterpret_cast<const uint4 *>(dy2 + p * ld2 + cg * 8);
        b2 = *reinterpret_cast<const uint4 *>(dy2 + (p + PL) * ld2 + cg * 8);
      }
      accumulate(a1, a2, az);
      accumulate(b1, b2, bz);
    }
    if (p < p1)
      accumulate(*reinterpret_cast<const uint4 *>(dy1 + p * ld1 + cg * 8),
                 dy2 ? *reinterpret_cast<const uint4 *>(dy2 + p * ld2 + cg * 8) : zero4,
                 *reinterpret_cast<const uint4 *>(z + p * z_ld + cg * 8));
#pragma unroll
    for (int j = 0; j < 16; ++j) red[(pl * CG + cg) * 16 + j] = s[j];
  }
  __syncthreads();
  // thread t < CG*16 sums column t over the PL pixel lanes
  for (int t = tid; t < CG * 16; t += 256) {
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[l * CG * 16 + t];
    // t = cg*16 + 2*j + which  ->  channel cg*8 + j
    const int ch = (t >> 4) * 8 + ((t & 15) >> 1);
    partial[((i64)blockIdx.x * CoP + ch) * 2 + (t & 1)] = acc;
  }
}

// sums partials; writes grad_gb[2c] = dgamma, [2c+1] = dbeta (times grad_scale) and the ChanBwd record
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const T *__restrict__ partials, int P, int C,
                                                              int CoP, float M, const float *__restrict__ gb,
                                                              const ChanFwd *__restrict__ cf,
                                                              float *__restrict__ grad_gb, ChanBwd *__restrict__ out,
                                                              float grad_scale) {
  const int c = blockIdx.x;  // one workgroup per channel
  ChanBwd r{0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    double dg, db;
    block_sum_partials(partials, P, CoP, c, dg, db);
    if (threadIdx.x == 0) {
      grad_gb[2 * c] = (float)dg * grad_scale;
      grad_gb[2 * c + 1] = (float)db * grad_scale;
    }
    r.c0 = gb[2 * c] * cf[c].invb;
    r.k1 = (float)(db / M);
    r.k2 = (float)(dg / M);
  }
  if (threadIdx.x == 0) out[c] = r;
}

// pass 2: dz = c0*(g - k1 - xhat*k2) (bf16) and, with BIAS, per-block sums of dz (bias gradient)
template <int ACT, bool BIAS, int NT>  // NT bit 0: non-temporal loads, bit 1: non-temporal stores
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t *__restrict__ dy1, int ld1,
                                                           const bf16_t *__restrict__ dy2, int ld2,
                                                           const bf16_t *__restrict__ z, int z_ld,
                                                           const ChanFwd *__restrict__ cf,
                                                           const ChanBwd *__restrict__ cb, bf16_t *__restrict__ dz,
                                                           int dz_ld, int CG, int PL, i64 pixels, i64 ppb,
                                                           float *__restrict__ partial, int CoP, float ap) {
  extern __shared__ float red[];  // [PL][CG*8]
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  const bool active = pl < PL;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (active) {
    ChanFwd c[8];
    ChanBwd b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c[j] = cf[cg * 8 + j];
      b[j] = cb[cg * 8 + j];
    }
    const i64 p0 = (i64)blockIdx.x * ppb, p1 = min(pixels, p0 + ppb);
    auto apply = [&](const uint4 &r1, const uint4 &r2, const uint4 &rz) {
      float g[8], zv[8], o[8];
      unpack8(r1, g);
      if (dy2) {
        float g2[8];
        unpack8(r2, g2);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += g2[j];
      }
      unpack8(rz, zv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gj = act_grad<ACT>(g[j], fmaf(c[j].a, zv[j], c[j].b), ap);
        const float xh = (zv[j] - c[j].mean) * c[j].invb;
        o[j] = b[j].c0 * (gj - b[j].k1 - xh * b[j].k2);
      }
      const uint4 packed = pack8(o);
      if (BIAS) {
        float r[8];
        unpack8(packed, r);  // the bias gradient sums what the weight-gradient kernels will read
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += r[j];
      }
      return packed;
    };
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    i64 p = p0 + pl;
    // U pixel rows per trip: 2 U (3 U with a second gradient) independent 16-byte loads in flight per lane
    constexpr int U = BIAS ? 2 : 4;
    for (; p + (U - 1) * PL < p1; p += U * PL) {
      uint4 r1[U], r2[U], rz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        r1[u] = ld16<(NT & 1) != 0>(dy1 + (p + u * PL) * ld1 + cg * 8);
        rz[u] = ld16<(NT & 1) != 0>(z + (p + u * PL) * z_ld + cg * 8);
        r2[u] = dy2 ? ld16<(NT & 1) != 0>(dy2 + (p + u * PL) * ld2 + cg * 8) : zero4;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) st16<(NT & 2) != 0>(dz + (p + u * PL) * dz_ld + cg * 8, apply(r1[u], r2[u], rz[u]));
    }
    for (; p < p1; p += PL)
      st16<(NT & 2) != 0>(dz + p * dz_ld + cg * 8, apply(ld16<(NT & 1) != 0>(dy1 + p * ld1 + cg * 8),
                                              dy2 ? ld16<(NT & 1) != 0>(dy2 + p * ld2 + cg * 8) : zero4,
                                              ld16<(NT & 1) != 0>(z + p * z_ld + cg * 8)));
    if (BIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(pl * CG + cg) * 8 + j] = s[j];
    }
  }
  if (!BIAS) return;
  __syncthreads();
  for (int t = tid; t < CG * 8; t += 256) {
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[l * CG * 8 + t];
    partial[(i64)blockIdx.x * CoP + t] = acc;
  }
}

// ---- BN backward of a layer that also feeds a 2x2 max pool ---------------------------
// The gradient through the pool is read where it is needed instead of being scattered into a full-resolution
// tensor first: a lane owns one pooling window of 8 channels, loads the pooled gradient and the tie mask once
// and adds the pooled gradient to the window positions whose mask bit is set.  dy1 (the other consumer's
// gradient, full resolution) may be NULL.
struct PoolSrc {
  const bf16_t *dy;     // [N][H/2][W/2] pixels, ld channels apart
  const uint8_t *mask;  // [N][H/2][W/2][CG*8]
  int ld, W;            // W: width of the full-resolution image (even)
};

template <int ACT, bool NT>
__device__ __forceinline__ void pool_window_grads(const PoolSrc &ps, const bf16_t *__restrict__ dy1, int ld1,
                                                  const bf16_t *__restrict__ z, int z_ld, i64 o, int cg, int CG,
                                                  const ChanFwd (&c)[8], float (&g)[4][8], float (&zv)[4][8],
                                                  i64 &pix) {
  const int Wo = ps.W >> 1;
  const i64 t = o / Wo;  // = n*H/2 + ho: pooled row, counted over the whole batch
  const int wo = (int)(o - t * Wo);
  pix = 2 * t * ps.W + 2 * wo;
  const i64 off[4] = {0, 1, (i64)ps.W, (i64)ps.W + 1};
  uint4 rz[4], r1[4];
  const uint4 rp = *reinterpret_cast<const uint4 *>(ps.dy + o * ps.ld + cg * 8);
  const uint2 mk = *reinterpret_cast<const uint2 *>(ps.mask + o * (CG * 8) + cg * 8);
#pragma unroll
  for (int k = 0; k < 4; ++k) rz[k] = ld16<NT>(z + (pix + off[k]) * z_ld + cg * 8);
  if (dy1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r1[k] = ld16<NT>(dy1 + (pix + off[k]) * ld1 + cg * 8);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) r1[k] = make_uint4(0, 0, 0, 0);
  }
  float pg[8];
  unpack8(rp, pg);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unpack8(r1[k], g[k]);
    unpack8(rz[k], zv[k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t bit = ((j < 4 ? mk.x : mk.y) >> (8 * (j & 3) + k)) & 1u;
      const float gin = g[k][j] + (bit ? pg[j] : 0.f);
      g[k][j] = act_grad<ACT>(gin, fmaf(c[j].a, zv[k][j], c[j].b));
    }
  }
}

// pass 1 over windows: partial[b][c][2] = (sum g*xhat, sum g) over block b's windows
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const bf16_t *__restrict__ dy1, int ld1, PoolSrc ps,
                                                                 const bf16_t *__restrict__ z, int z_ld,
                                                                 const ChanFwd *__restrict__ cf, int CG, int PL,
                                                                 i64 windows, i64 wpb, float *__restrict__ partial,
                                                                 int CoP) {
  extern __shared__ float red[];  // [PL][CG*16]
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  if (pl < PL) {
    ChanFwd c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = cf[cg * 8 + j];
    const i64 o0 = (i64)blockIdx.x * wpb, o1 = min(windows, o0 + wpb);
    for (i64 o = o0 + pl; o < o1; o += PL) {
      float g[4][8], zv[4][8];
      i64 pix;
      pool_window_grads<ACT, false>(ps, dy1, ld1, z, z_ld, o, cg, CG, c, g, zv, pix);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // xhat stays a scalar product (empty asm tied to its register): compiled to packed fp32 instructions
          // this kernel needed 245 registers instead of 162 and, worse, returned wrong sums of g*xhat whenever
          // its waves shared a SIMD with waves of one of the MFMA kernels running on another stream (the
          // sum of g, computed beside it, stayed right); see keep_scalar in bf16_util.h
          float xh = (zv[k][j] - c[j].mean) * c[j].invb;
          asm volatile("" : "+v"(xh));
          s[2 * j] += g[k][j] * xh;
          s[2 * j + 1] += g[k][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) red[(pl * CG + cg) * 16 + j] = s[j];
  }
  __syncthreads();
  for (int t = tid; t < CG * 16; t += 256) {
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[l * CG * 16 + t];
    const int ch = (t >> 4) * 8 + ((t & 15) >> 1);
    partial[((i64)blockIdx.x * CoP + ch) * 2 + (t & 1)] = acc;
  }
}

// pass 2 over windows: dz = c0*(g - k1 - xhat*k2) and, with BIAS, per-block sums of dz
template <int ACT, bool BIAS, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const bf16_t *__restrict__ dy1, int ld1, PoolSrc ps,
                                                                const bf16_t *__restrict__ z, int z_ld,
                                                                const ChanFwd *__restrict__ cf,
                                                                const ChanBwd *__restrict__ cb,
                                                                bf16_t *__restrict__ dz, int dz_ld, int CG, int PL,
                                                                i64 windows, i64 wpb, float *__restrict__ partial,
                                                                int CoP) {
  extern __shared__ float red[];  // [PL][CG*8]
  const int tid = threadIdx.x;
  const int cg = tid % CG, pl = tid / CG;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (pl < PL) {
    ChanFwd c[8];
    ChanBwd b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c[j] = cf[cg * 8 + j];
      b[j] = cb[cg * 8 + j];
    }
    const i64 o0 = (i64)blockIdx.x * wpb, o1 = min(windows, o0 + wpb);
    for (i64 o = o0 + pl; o < o1; o += PL) {
      float g[4][8], zv[4][8];
      i64 pix;
      pool_window_grads<ACT, NT>(ps, dy1, ld1, z, z_ld, o, cg, CG, c, g, zv, pix);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (zv[k][j] - c[j].mean) * c[j].invb;
          out[j] = b[j].c0 * (g[k][j] - b[j].k1 - xh * b[j].k2);
        }
        const uint4 packed = pack8(out);
        st16<false>(dz + (pix + (k >> 1) * (i64)ps.W + (k & 1)) * dz_ld + cg * 8, packed);
        if (BIAS) {
          float r[8];
          unpack8(packed, r);
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += r[j];
        }
      }
    }
    if (BIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(pl * CG + cg) * 8 + j] = s[j];
    }
  }
  if (!BIAS) return;
  __syncthreads();
  for (int t = tid; t < CG * 8; t += 256) {
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[l * CG * 8 + t];
    partial[(i64)blockIdx.x * CoP + t] = acc;
  }
}

// out[c] = sum_p partial[p][c], c < C: one workgroup per channel
__global__ __launch_bounds__(256) void column_sum_kernel(const float *__restrict__ partial, int P, int C, int stride,
                                                         float *__restrict__ out) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0;
  int p = tid;
  for (; p + 3 * 256 < P; p += 4 * 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = partial[(i64)(p + u * 256) * stride + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; p < P; p += 256) s += partial[(i64)p * stride + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) out[c] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// ---- 2x2 / stride 2 max pool -------------------------------------------------------
// mask[n,ho,wo,c] bit (2*i+j) set when x[n,2ho+i,2wo+j,c] equals the maximum
// (every tied maximum, the reference's backward rule; the lowest set bit is the
// first maximum in row-major order).
__global__ __launch_bounds__(256) void maxpool2_kernel(const bf16_t *__restrict__ x, int x_ld,
                                                       bf16_t *__restrict__ y, int y_ld,
                                                       uint8_t *__restrict__ mask, int CG, int H, int W, i64 total) {
  const int Ho = H / 2, Wo = W / 2;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    i64 t = i / CG;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const i64 n = t / Ho;
    const bf16_t *xp = x + ((n * H + 2 * ho) * W + 2 * wo) * (i64)x_ld + cg * 8;
    float v[4][8];
    unpack8(*reinterpret_cast<const uint4 *>(xp), v[0]);
    unpack8(*reinterpret_cast<const uint4 *>(xp + x_ld), v[1]);
    unpack8(*reinterpret_cast<const uint4 *>(xp + (i64)W * x_ld), v[2]);
    unpack8(*reinterpret_cast<const uint4 *>(xp + (i64)W * x_ld + x_ld), v[3]);
    float m[8];
    uint32_t mk[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      uint32_t bits = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) bits |= (v[k][j] == m[j]) ? (1u << k) : 0u;
      mk[j >> 2] |= bits << (8 * (j & 3));
    }
    const i64 o = ((n * Ho + ho) * Wo + wo);
    *reinterpret_cast<uint4 *>(y + o * y_ld + cg * 8) = pack8(m);
    *reinterpret_cast<uint2 *>(mask + o * (CG * 8) + cg * 8) = make_uint2(mk[0], mk[1]);
  }
}

// bn_act and the 2x2 max pool that follows it in one pass: a lane owns one pooling window of 8 channels, writes the
// four activations, their maximum and the tie mask (same values as bn_act_kernel followed by maxpool2_kernel:
// the maximum is taken over the rounded bf16 activations).
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const bf16_t *__restrict__ z, int z_ld,
                                                          bf16_t *__restrict__ y, int y_ld,
                                                          const ChanFwd *__restrict__ cf,
                                                          bf16_t *__restrict__ yp, int yp_ld,
                                                          uint8_t *__restrict__ mask, int CG, int H, int W,
                                                          i64 total) {
  const int Ho = H / 2, Wo = W / 2;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    i64 t = i / CG;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const i64 n = t / Ho;
    const i64 pix = (n * H + 2 * ho) * W + 2 * wo;
    const bf16_t *zp = z + pix * z_ld + cg * 8;
    uint4 in[4];
    in[0] = ld16<NT>(zp);
    in[1] = ld16<NT>(zp + z_ld);
    in[2] = ld16<NT>(zp + (i64)W * z_ld);
    in[3] = ld16<NT>(zp + (i64)W * z_ld + z_ld);
    float ca[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const ChanFwd c = cf[cg * 8 + j];
      ca[j] = c.a;
      cb[j] = c.b;
    }
    float v[4][8];
    bf16_t *yq = y + pix * y_ld + cg * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float r[8];
      unpack8(in[k], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = fmaf(ca[j], r[j], cb[j]);
        if (ACT == 0) a = a > 0.f ? a : 0.f;
        if (ACT == 1) a = 1.0f / (1.0f + __expf(-a));
        r[j] = a;
      }
      const uint4 packed = pack8(r);
      st16<false>(yq + (i64)(k >> 1) * W * y_ld + (k & 1) * y_ld, packed);
      unpack8(packed, v[k]);
    }
    float m[8];
    uint32_t mk[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      uint32_t bits = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) bits |= (v[k][j] == m[j]) ? (1u << k) : 0u;
      mk[j >> 2] |= bits << (8 * (j & 3));
    }
    const i64 o = ((n * Ho + ho) * Wo + wo);
    *reinterpret_cast<uint4 *>(yp + o * yp_ld + cg * 8) = pack8(m);
    *reinterpret_cast<uint2 *>(mask + o * (CG * 8) + cg * 8) = make_uint2(mk[0], mk[1]);
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16_t *__restrict__ dy, int dy_ld,
                                                           const uint8_t *__restrict__ mask,
                                                           bf16_t *__restrict__ dx, int dx_ld, int CG, int H, int W,
                                                           i64 total, int accumulate) {
  const int Ho = H / 2, Wo = W / 2;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    i64 t = i / CG;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const i64 n = t / Ho;
    const i64 o = ((n * Ho + ho) * Wo + wo);
    float g[8];
    unpack8(*reinterpret_cast<const uint4 *>(dy + o * dy_ld + cg * 8), g);
    const uint2 mk = *reinterpret_cast<const uint2 *>(mask + o * (CG * 8) + cg * 8);
    bf16_t *xp = dx + ((n * H + 2 * ho) * W + 2 * wo) * (i64)dx_ld + cg * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16_t *dst = xp + (i64)(k >> 1) * W * dx_ld + (k & 1) * dx_ld;
      float r[8];
      if (accumulate)
        unpack8(*reinterpret_cast<const uint4 *>(dst), r);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t bits = ((j < 4 ? mk.x : mk.y) >> (8 * (j & 3))) & 0xffu;
        if (bits & (1u << k)) r[j] += g[j];
      }
      *reinterpret_cast<uint4 *>(dst) = pack8(r);
    }
  }
}

// sums[c] = sum_p partial[p][c] in double, c < ncols (the statistics a rank contributes to the
// cross-rank sum when batch-norm statistics are synchronised)
__global__ __launch_bounds__(256) void column_sum_f64_kernel(const float *__restrict__ partial, int P, int ncols,
                                                             double *__restrict__ out) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0;
  for (int p = tid; p < P; p += 256) s += partial[(i64)p * ncols + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The two finalize steps.  With statistics synchronised across data-parallel ranks (unet_bn_set_sync) each
// rank first adds up its own partials, the [CoP][2] sums are added across ranks in double, and the finalize
// kernel then sees one row of global sums and the global element count; gamma/beta gradients are divided by
// the number of ranks so that the mean over ranks taken by the gradient all-reduce is the global gradient.
// `sync_ws` holds 2*CoP doubles.  Synchronised statistics require the same element count on every rank.
int launch_finalize_fwd(const float *partials, int P, int C, int CoP, i64 count, const float *gb, float *batch_stats,
                        float *running, ChanFwd *out, double *sync_ws, hipStream_t s) {
  if (!unet_bn_sync_on()) {
    hipLaunchKernelGGL(bn_finalize_fwd_kernel<float>, dim3(CoP), dim3(256), 0, s, partials, P, C, CoP, (float)count, gb,
                       batch_stats, running, out);
    UNET_LAUNCH_CHECK("bn_finalize_fwd_kernel");
    return UNET_OK;
  }
  hipLaunchKernelGGL(column_sum_f64_kernel, dim3(2 * CoP), dim3(256), 0, s, partials, P, 2 * CoP, sync_ws);
  UNET_LAUNCH_CHECK("column_sum_f64_kernel");
  int world = 1;
  int rc = unet_bn_sync_sum_f64(sync_ws, 2 * (i64)CoP, &world, s);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_fwd_kernel<double>, dim3(CoP), dim3(256), 0, s, (const double *)sync_ws, 1, C, CoP,
                     (float)(count * world), gb, batch_stats, running, out);
  UNET_LAUNCH_CHECK("bn_finalize_fwd_kernel");
  return UNET_OK;
}

int launch_finalize_bwd(const float *partials, int P, int C, int CoP, i64 count, const float *gb, const ChanFwd *cf,
                        float *grad_gb, ChanBwd *out, double *sync_ws, hipStream_t s) {
  if (!unet_bn_sync_on()) {
    hipLaunchKernelGGL(bn_finalize_bwd_kernel<float>, dim3(CoP), dim3(256), 0, s, partials, P, C, CoP, (float)count, gb,
                       cf, grad_gb, out, 1.0f);
    UNET_LAUNCH_CHECK("bn_finalize_bwd_kernel");
    return UNET_OK;
  }
  hipLaunchKernelGGL(column_sum_f64_kernel, dim3(2 * CoP), dim3(256), 0, s, partials, P, 2 * CoP, sync_ws);
  UNET_LAUNCH_CHECK("column_sum_f64_kernel");
  int world = 1;
  int rc = unet_bn_sync_sum_f64(sync_ws, 2 * (i64)CoP, &world, s);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_bwd_kernel<double>, dim3(CoP), dim3(256), 0, s, (const double *)sync_ws, 1, C, CoP,
                     (float)(count * world), gb, cf, grad_gb, out, 1.0f / (float)world);
  UNET_LAUNCH_CHECK("bn_finalize_bwd_kernel");
  return UNET_OK;
}

struct RedPlan {
  int CG, PL, blocks;
  i64 ppb;
};
RedPlan plan_reduce(int CoP, i64 pixels) {
  RedPlan r;
  r.CG = CoP / 8;
  r.PL = 256 / r.CG;
  i64 b = (pixels + (i64)r.PL * 8 - 1) / ((i64)r.PL * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  r.ppb = (pixels + b - 1) / b;
  r.blocks = (int)((pixels + r.ppb - 1) / r.ppb);
  return r;
}

// The streaming passes without per-block partial sums (bn_act, bn_bwd_apply without a bias gradient): every
// workgroup takes `trips` main-loop trips over consecutive pixels, i.e. a contiguous 8-32 KiB piece per tensor,
// and there are as many workgroups as that takes.
RedPlan plan_stream(int CoP, i64 pixels, int rows_per_trip, int trips) {
  RedPlan r;
  r.CG = CoP / 8;
  r.PL = 256 / r.CG;
  r.ppb = (i64)r.PL * rows_per_trip * trips;
  r.blocks = (int)((pixels + r.ppb - 1) / r.ppb);
  return r;
}

// Non-temporal loads and stores for passes over tensors that no cache holds anyway.  UNET_LAB_BN_NT=0 / 1 forces the
// choice (laboratory switch); by default a pass streams when it moves at least 32 MB.
bool stream_nontemporal(double bytes) {
  static const int forced = [] {
    const char *e = getenv("UNET_LAB_BN_NT");
    return e && *e ? atoi(e) : -1;
  }();
  if (forced >= 0) return forced != 0;
  static const double min_bytes = getenv("UNET_LAB_BN_NT_BYTES") ? atof(getenv("UNET_LAB_BN_NT_BYTES")) : 32e6;
  return bytes >= min_bytes;
}
int apply_nt_mask() {  // bn_bwd_apply: 3 = loads and stores non-temporal, 1 = loads only (laboratory switch)
  static const int m = getenv("UNET_LAB_BN_APPLY_NT_MASK") ? atoi(getenv("UNET_LAB_BN_APPLY_NT_MASK")) : 1;
  return m == 1 ? 1 : (m == 0 ? 0 : 3);
}
int stream_trips() {  // main-loop trips per workgroup of the streaming passes (UNET_LAB_BN_TRIPS: laboratory switch)
  static const int t = [] {
    const char *e = getenv("UNET_LAB_BN_TRIPS");
    const int v = e && *e ? atoi(e) : 2;
    return v < 1 ? 1 : v;
  }();
  return t;
}

int check_view(const char *name, const void *p, int ld, int CoP) {
  UNET_REQUIRE(p, "%s: NULL pointer", name);
  UNET_REQUIRE(ld >= CoP && ld % 8 == 0 && (((uintptr_t)p) & 15) == 0,
               "%s: pixel stride must be a multiple of 8 and >= the channel count, pointer 16-byte aligned", name);
  return UNET_OK;
}

}  // namespace

extern "C" {

int unet_bn_coef_floats(int CoP) { return 4 * CoP; }

int unet_bn_finalize_bf16(const float *stat_partials, int num_partials, int C, int CoP, int64_t count,
                          const float *gamma_beta, float *batch_stats, float *running_stats, float *coef,
                          unet_stream_t s) {
  UNET_REQUIRE(stat_partials && gamma_beta && batch_stats && coef, "unet_bn_finalize_bf16: NULL pointer");
  UNET_REQUIRE(num_partials > 0 && C > 0 && CoP >= C && count > 0, "unet_bn_finalize_bf16: bad sizes");
  double *sync_ws = nullptr;
  if (unet_bn_sync_on()) {
    void *ws;
    int rc = unet_workspace(&ws, sizeof(double) * 2 * (size_t)CoP, as_stream(s));
    if (rc) return rc;
    sync_ws = (double *)ws;
  }
  return launch_finalize_fwd(stat_partials, num_partials, C, CoP, count, gamma_beta, batch_stats, running_stats,
                             reinterpret_cast<ChanFwd *>(coef), sync_ws, as_stream(s));
}

int unet_bn_inference_coef_bf16(const float *gamma_beta, const float *running_stats, int C, int CoP, float *coef,
                                unet_stream_t s) {
  UNET_REQUIRE(gamma_beta && running_stats && coef && C > 0 && CoP >= C, "unet_bn_inference_coef_bf16: bad argument");
  hipLaunchKernelGGL(bn_inference_coef_kernel, dim3(cdiv_u(CoP, 64)), dim3(64), 0, as_stream(s), gamma_beta,
                     running_stats, C, CoP, reinterpret_cast<ChanFwd *>(coef));
  UNET_LAUNCH_CHECK("bn_inference_coef_kernel");
  return UNET_OK;
}

int unet_bn_act_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, int CoP,
                     int64_t pixels, int act, unet_stream_t s) {
  UNET_REQUIRE(act >= 0 && act <= 2, "unet_bn_act_bf16: act must be 0 (ReLU), 1 (sigmoid) or 2 (identity)");
  return unet_bn_act_ex_bf16(z, z_ld, y, y_ld, coef, CoP, pixels, act, 0.f, s);
}

int unet_bn_act_ex_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, int CoP,
                        int64_t pixels, int act, float act_param, unet_stream_t s) {
  int rc = check_view("unet_bn_act_bf16 (z)", z, z_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_bf16 (y)", y, y_ld, CoP);
  if (rc) return rc;
  UNET_REQUIRE(coef && CoP > 0 && CoP % 8 == 0 && CoP <= 2048 && pixels > 0 && act >= 0 && act <= 4,
               "unet_bn_act_bf16: bad argument (CoP must be a multiple of 8, at most 2048)");
  const RedPlan pl = plan_stream(CoP, pixels, 4, stream_trips());
  const dim3 grid(pl.blocks);
  const ChanFwd *cf = reinterpret_cast<const ChanFwd *>(coef);
  const double bytes = 4.0 * pixels * CoP;
  ProfScope prof(PROF_BN_ACT_BF16, 0, bytes, as_stream(s));
#define ACT_LAUNCH(ACT_, NT_)                                                                                   \
  hipLaunchKernelGGL((bn_act_kernel<ACT_, NT_>), grid, dim3(256), 0, as_stream(s), z, z_ld, y, y_ld, cf, pl.CG, \
                     pl.PL, (i64)pixels, pl.ppb, act_param)
  static const int lab_mask = getenv("UNET_LAB_BN_ACT_NT_MASK") ? atoi(getenv("UNET_LAB_BN_ACT_NT_MASK")) : 1;  // laboratory switch
  const int mask = stream_nontemporal(bytes) ? lab_mask : 0;
#define ACT_DISPATCH(NT_)                \
  do {                                   \
    if (act == 0) ACT_LAUNCH(0, NT_);    \
    else if (act == 1) ACT_LAUNCH(1, NT_); \
    else if (act == 3) ACT_LAUNCH(3, NT_); \
    else if (act == 4) ACT_LAUNCH(4, NT_); \
    else ACT_LAUNCH(2, NT_);             \
  } while (0)
  if (mask == 3) ACT_DISPATCH(3);
  else if (mask == 2) ACT_DISPATCH(2);
  else if (mask == 1) ACT_DISPATCH(1);
  else ACT_DISPATCH(0);
#undef ACT_DISPATCH
#undef ACT_LAUNCH
  UNET_LAUNCH_CHECK("bn_act_kernel");
  return UNET_OK;
}

int unet_bn_act_backward_bf16(const uint16_t *dy1, int ld1, const uint16_t *dy2, int ld2, const uint16_t *z, int z_ld,
                              const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld, float *grad_gb,
                              float *grad_bias, int C, int CoP, int64_t pixels, int act, unet_stream_t s) {
  UNET_REQUIRE(act >= 0 && act <= 2, "unet_bn_act_backward_bf16: act must be 0 (ReLU), 1 (sigmoid) or 2 (identity)");
  return unet_bn_act_backward_ex_bf16(dy1, ld1, dy2, ld2, z, z_ld, coef, gamma_beta, dz, dz_ld, grad_gb, grad_bias, C, CoP,
                                      pixels, act, 0.f, s);
}

int unet_bn_act_backward_ex_bf16(const uint16_t *dy1, int ld1, const uint16_t *dy2, int ld2, const uint16_t *z, int z_ld,
                                 const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld, float *grad_gb,
                                 float *grad_bias, int C, int CoP, int64_t pixels, int act, float act_param,
                                 unet_stream_t s) {
  int rc = check_view("unet_bn_act_backward_bf16 (dy1)", dy1, ld1, CoP);
  if (!rc && dy2) rc = check_view("unet_bn_act_backward_bf16 (dy2)", dy2, ld2, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_bf16 (z)", z, z_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_bf16 (dz)", dz, dz_ld, CoP);
  if (rc) return rc;
  UNET_REQUIRE(coef && gamma_beta && grad_gb && C > 0 && CoP >= C && CoP % 8 == 0 && CoP <= 2048 && pixels > 0 &&
                   act >= 0 && act <= 4,
               "unet_bn_act_backward_bf16: bad argument (CoP must be a multiple of 8, at most 2048)");
  const RedPlan pl = plan_reduce(CoP, pixels);
  // workspace: cross-rank sums [CoP][2] double | partials [blocks][CoP][2] | ChanBwd [CoP] | bias partials [blocks][CoP]
  const size_t n_sync = (size_t)CoP * 4, n_part = (size_t)pl.blocks * CoP * 2, n_cb = (size_t)CoP * 4,
               n_bias = (size_t)pl.blocks * CoP;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (n_sync + n_part + n_cb + n_bias), as_stream(s));
  if (rc) return rc;
  double *sync_ws = (double *)ws;
  float *partial = (float *)ws + n_sync, *cbf = partial + n_part, *bias_part = cbf + n_cb;
  const ChanFwd *cf = reinterpret_cast<const ChanFwd *>(coef);
  ChanBwd *cb = reinterpret_cast<ChanBwd *>(cbf);
  const size_t lds1 = sizeof(float) * (size_t)pl.PL * pl.CG * 16, lds2 = sizeof(float) * (size_t)pl.PL * pl.CG * 8;
  const double elems = (double)pixels * CoP;
  const RedPlan ps = plan_stream(CoP, pixels, 4, stream_trips());  // the apply pass when it writes no partials
  const int nt = stream_nontemporal(2.0 * elems * (dy2 ? 4.0 : 3.0)) ? apply_nt_mask() : 0;
#define BWD_LAUNCH(ACT_)                                                                                            \
  do {                                                                                                              \
    {                                                                                                               \
      ProfScope prof(PROF_BN_BWD_REDUCE_BF16, 0, 2.0 * elems * (dy2 ? 3.0 : 2.0), as_stream(s));                    \
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<ACT_>, dim3(pl.blocks), dim3(256), lds1, as_stream(s), dy1, ld1,     \
                         dy2, ld2, z, z_ld, cf, pl.CG, pl.PL, (i64)pixels, pl.ppb, partial, CoP, act_param);        \
    }                                                                                                               \
    UNET_LAUNCH_CHECK("bn_bwd_reduce_kernel");                                                                      \
    rc = launch_finalize_bwd(partial, pl.blocks, C, CoP, (i64)pixels, gamma_beta, cf, grad_gb, cb, sync_ws,        \
                             as_stream(s));                                                                         \
    if (rc) return rc;                                                                                              \
    {                                                                                                               \
      ProfScope prof(PROF_BN_BWD_APPLY_BF16, 0, 2.0 * elems * (dy2 ? 4.0 : 3.0), as_stream(s));                     \
      if (grad_bias)                                                                                               \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT_, true, 0>), dim3(pl.blocks), dim3(256), lds2, as_stream(s), \
                           dy1, ld1, dy2, ld2, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, pl.CG, pl.PL,          \
                           (i64)pixels, pl.ppb, bias_part, CoP, act_param);                                         \
      else if (nt == 3)                                                                                             \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT_, false, 3>), dim3(ps.blocks), dim3(256), 0, as_stream(s),      \
                           dy1, ld1, dy2, ld2, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,          \
                           (i64)pixels, ps.ppb, (float *)nullptr, CoP, act_param);                                  \
      else if (nt == 1)                                                                                             \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT_, false, 1>), dim3(ps.blocks), dim3(256), 0, as_stream(s),      \
                           dy1, ld1, dy2, ld2, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,          \
                           (i64)pixels, ps.ppb, (float *)nullptr, CoP, act_param);                                  \
      else                                                                                                          \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT_, false, 0>), dim3(ps.blocks), dim3(256), 0, as_stream(s),      \
                           dy1, ld1, dy2, ld2, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,          \
                           (i64)pixels, ps.ppb, (float *)nullptr, CoP, act_param);                                  \
    }                                                                                                               \
    UNET_LAUNCH_CHECK("bn_bwd_apply_kernel");                                                                       \
  } while (0)
  if (act == 0) BWD_LAUNCH(0);
  else if (act == 1) BWD_LAUNCH(1);
  else if (act == 3) BWD_LAUNCH(3);
  else if (act == 4) BWD_LAUNCH(4);
  else BWD_LAUNCH(2);
#undef BWD_LAUNCH
  if (grad_bias) {
    hipLaunchKernelGGL(column_sum_kernel, dim3(C), dim3(256), 0, as_stream(s), (const float *)bias_part,
                       pl.blocks, C, CoP, grad_bias);
    UNET_LAUNCH_CHECK("column_sum_kernel");
  }
  return UNET_OK;
}

int unet_bn_act_backward_pooled_bf16(const uint16_t *dy, int dy_ld, const uint16_t *pooled_dy, int pooled_ld,
                                     const uint8_t *mask, int N, int H, int W, const uint16_t *z, int z_ld,
                                     const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld,
                                     float *grad_gb, float *grad_bias, int C, int CoP, int act, unet_stream_t s) {
  int rc = UNET_OK;
  if (dy) rc = check_view("unet_bn_act_backward_pooled_bf16 (dy)", dy, dy_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_pooled_bf16 (pooled_dy)", pooled_dy, pooled_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_pooled_bf16 (z)", z, z_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_pooled_bf16 (dz)", dz, dz_ld, CoP);
  if (rc) return rc;
  UNET_REQUIRE(mask && (((uintptr_t)mask) & 7) == 0 && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && coef &&
                   gamma_beta && grad_gb && C > 0 && CoP >= C && CoP % 8 == 0 && CoP <= 2048 && act >= 0 && act <= 2,
               "unet_bn_act_backward_pooled_bf16: need an 8-byte aligned mask, even H and W, CoP a multiple of 8 "
               "(at most 2048)");
  const i64 windows = (i64)N * (H / 2) * (W / 2), pixels = 4 * windows;
  const RedPlan pl = plan_reduce(CoP, windows);
  // workspace: cross-rank sums [CoP][2] double | partials [blocks][CoP][2] | ChanBwd [CoP] | bias partials [blocks][CoP]
  const size_t n_sync = (size_t)CoP * 4, n_part = (size_t)pl.blocks * CoP * 2, n_cb = (size_t)CoP * 4,
               n_bias = (size_t)pl.blocks * CoP;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (n_sync + n_part + n_cb + n_bias), as_stream(s));
  if (rc) return rc;
  double *sync_ws = (double *)ws;
  float *partial = (float *)ws + n_sync, *cbf = partial + n_part, *bias_part = cbf + n_cb;
  const ChanFwd *cf = reinterpret_cast<const ChanFwd *>(coef);
  ChanBwd *cb = reinterpret_cast<ChanBwd *>(cbf);
  const bf16_t *d1 = reinterpret_cast<const bf16_t *>(dy);
  const bf16_t *zz = reinterpret_cast<const bf16_t *>(z);
  const PoolSrc ps{reinterpret_cast<const bf16_t *>(pooled_dy), mask, pooled_ld, W};
  const size_t lds1 = sizeof(float) * (size_t)pl.PL * pl.CG * 16, lds2 = sizeof(float) * (size_t)pl.PL * pl.CG * 8;
  // bytes per element: z (+ dz), the full-resolution gradient, a quarter of a pooled gradient and of a mask byte
  const double elems = (double)pixels * CoP, grads = (dy ? 2.0 : 0.0) + 0.75;
  const RedPlan pa = plan_stream(CoP, windows, 1, 2 * stream_trips());  // the apply pass when it writes no partials
  const bool nt = stream_nontemporal(elems * (4.0 + grads));
#define BWD_LAUNCH(ACT_)                                                                                            \
  do {                                                                                                              \
    {                                                                                                               \
      ProfScope prof(PROF_BN_BWD_REDUCE_BF16, 0, elems * (2.0 + grads), as_stream(s));                              \
      hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<ACT_>, dim3(pl.blocks), dim3(256), lds1, as_stream(s), d1,       \
                         dy_ld, ps, zz, z_ld, cf, pl.CG, pl.PL, windows, pl.ppb, partial, CoP);                     \
    }                                                                                                               \
    UNET_LAUNCH_CHECK("bn_bwd_reduce_pool_kernel");                                                                 \
    rc = launch_finalize_bwd(partial, pl.blocks, C, CoP, pixels, gamma_beta, cf, grad_gb, cb, sync_ws,              \
                             as_stream(s));                                                                         \
    if (rc) return rc;                                                                                              \
    {                                                                                                               \
      ProfScope prof(PROF_BN_BWD_APPLY_BF16, 0, elems * (4.0 + grads), as_stream(s));                               \
      if (grad_bias)                                                                                                \
        hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<ACT_, true, false>), dim3(pl.blocks), dim3(256), lds2,          \
                           as_stream(s), d1, dy_ld, ps, zz, z_ld, cf, (const ChanBwd *)cb,                          \
                           reinterpret_cast<bf16_t *>(dz), dz_ld, pl.CG, pl.PL, windows, pl.ppb, bias_part, CoP);   \
      else if (nt)                                                                                                  \
        hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<ACT_, false, true>), dim3(pa.blocks), dim3(256), 0,            \
                           as_stream(s), d1, dy_ld, ps, zz, z_ld, cf, (const ChanBwd *)cb,                          \
                           reinterpret_cast<bf16_t *>(dz), dz_ld, pa.CG, pa.PL, windows, pa.ppb, (float *)nullptr,  \
                           CoP);                                                                                    \
      else                                                                                                          \
        hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<ACT_, false, false>), dim3(pa.blocks), dim3(256), 0,           \
                           as_stream(s), d1, dy_ld, ps, zz, z_ld, cf, (const ChanBwd *)cb,                          \
                           reinterpret_cast<bf16_t *>(dz), dz_ld, pa.CG, pa.PL, windows, pa.ppb, (float *)nullptr,  \
                           CoP);                                                                                    \
    }                                                                                                               \
    UNET_LAUNCH_CHECK("bn_bwd_apply_pool_kernel");                                                                  \
  } while (0)
  if (act == 0) BWD_LAUNCH(0);
  else if (act == 1) BWD_LAUNCH(1);
  else BWD_LAUNCH(2);
#undef BWD_LAUNCH
  if (grad_bias) {
    hipLaunchKernelGGL(column_sum_kernel, dim3(C), dim3(256), 0, as_stream(s), (const float *)bias_part,
                       pl.blocks, C, CoP, grad_bias);
    UNET_LAUNCH_CHECK("column_sum_kernel");
  }
  return UNET_OK;
}

int unet_bn_act_backward_from_partials_bf16(const uint16_t *dy, int dy_ld, const uint16_t *z, int z_ld,
                                            const float *coef, const float *gamma_beta,
                                            const float *red_partials, int num_partials, uint16_t *dz,
                                            int dz_ld, float *grad_gb, float *grad_bias, int C, int CoP,
                                            int64_t pixels, unet_stream_t s) {
  int rc = check_view("unet_bn_act_backward_from_partials_bf16 (dy)", dy, dy_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_from_partials_bf16 (z)", z, z_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_backward_from_partials_bf16 (dz)", dz, dz_ld, CoP);
  if (rc) return rc;
  UNET_REQUIRE(coef && gamma_beta && grad_gb && red_partials && num_partials > 0 && C > 0 && CoP >= C &&
                   CoP % 8 == 0 && CoP <= 2048 && pixels > 0,
               "unet_bn_act_backward_from_partials_bf16: bad argument (CoP must be a multiple of 8, at most 2048)");
  const RedPlan pl = plan_reduce(CoP, pixels);
  // workspace: cross-rank sums [CoP][2] double | ChanBwd [CoP] | bias partials [blocks][CoP]
  const size_t n_sync = (size_t)CoP * 4, n_cb = (size_t)CoP * 4, n_bias = (size_t)pl.blocks * CoP;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (n_sync + n_cb + n_bias), as_stream(s));
  if (rc) return rc;
  float *cbf = (float *)ws + n_sync, *bias_part = cbf + n_cb;
  const ChanFwd *cf = reinterpret_cast<const ChanFwd *>(coef);
  ChanBwd *cb = reinterpret_cast<ChanBwd *>(cbf);
  rc = launch_finalize_bwd(red_partials, num_partials, C, CoP, (i64)pixels, gamma_beta, cf, grad_gb, cb, (double *)ws,
                           as_stream(s));
  if (rc) return rc;
  {
    ProfScope prof(PROF_BN_BWD_APPLY_BF16, 0, 2.0 * (double)pixels * CoP * 3.0, as_stream(s));
    const RedPlan ps = plan_stream(CoP, pixels, 4, stream_trips());
    const int nt = stream_nontemporal(2.0 * (double)pixels * CoP * 3.0) ? apply_nt_mask() : 0;
    if (grad_bias)
      hipLaunchKernelGGL((bn_bwd_apply_kernel<0, true, 0>), dim3(pl.blocks), dim3(256),
                         sizeof(float) * (size_t)pl.PL * pl.CG * 8, as_stream(s), dy, dy_ld, (const bf16_t *)nullptr, 0, z,
                         z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, pl.CG, pl.PL, (i64)pixels, pl.ppb, bias_part, CoP, 0.f);
    else if (nt == 3)
      hipLaunchKernelGGL((bn_bwd_apply_kernel<0, false, 3>), dim3(ps.blocks), dim3(256), 0, as_stream(s), dy, dy_ld,
                         (const bf16_t *)nullptr, 0, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,
                         (i64)pixels, ps.ppb, (float *)nullptr, CoP, 0.f);
    else if (nt == 1)
      hipLaunchKernelGGL((bn_bwd_apply_kernel<0, false, 1>), dim3(ps.blocks), dim3(256), 0, as_stream(s), dy, dy_ld,
                         (const bf16_t *)nullptr, 0, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,
                         (i64)pixels, ps.ppb, (float *)nullptr, CoP, 0.f);
    else
      hipLaunchKernelGGL((bn_bwd_apply_kernel<0, false, 0>), dim3(ps.blocks), dim3(256), 0, as_stream(s), dy, dy_ld,
                         (const bf16_t *)nullptr, 0, z, z_ld, cf, (const ChanBwd *)cb, dz, dz_ld, ps.CG, ps.PL,
                         (i64)pixels, ps.ppb, (float *)nullptr, CoP, 0.f);
  }
  UNET_LAUNCH_CHECK("bn_bwd_apply_kernel");
  if (grad_bias) {
    hipLaunchKernelGGL(column_sum_kernel, dim3(C), dim3(256), 0, as_stream(s), (const float *)bias_part,
                       pl.blocks, C, CoP, grad_bias);
    UNET_LAUNCH_CHECK("column_sum_kernel");
  }
  return UNET_OK;
}

int unet_bn_act_pool_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, uint16_t *pooled,
                          int pooled_ld, uint8_t *mask, int N, int H, int W, int CoP, int act, unet_stream_t s) {
  int rc = check_view("unet_bn_act_pool_bf16 (z)", z, z_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_pool_bf16 (y)", y, y_ld, CoP);
  if (!rc) rc = check_view("unet_bn_act_pool_bf16 (pooled)", pooled, pooled_ld, CoP);
  if (rc) return rc;
  UNET_REQUIRE(coef && mask && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && CoP > 0 && CoP % 8 == 0 &&
                   CoP <= 2048 && act >= 0 && act <= 2 && (((uintptr_t)mask) & 7) == 0,
               "unet_bn_act_pool_bf16: need even H and W, channels a multiple of 8 (at most 2048), an 8-byte "
               "aligned mask");
  const i64 total = (i64)N * (H / 2) * (W / 2) * (CoP / 8);
  const ChanFwd *cf = reinterpret_cast<const ChanFwd *>(coef);
  // read z, write y, write a quarter of y again (pooled) and a quarter of a byte of mask per element
  ProfScope prof(PROF_BN_ACT_BF16, 0, (double)N * H * W * CoP * 4.75, as_stream(s));
  const dim3 grid(ew_grid(total));
#define POOL_LAUNCH(ACT_, NT_)                                                                                  \
  hipLaunchKernelGGL((bn_act_pool_kernel<ACT_, NT_>), grid, dim3(256), 0, as_stream(s), z, z_ld, y, y_ld, cf,   \
                     pooled, pooled_ld, mask, CoP / 8, H, W, total)
  if (stream_nontemporal((double)N * H * W * CoP * 4.75)) {
    if (act == 0) POOL_LAUNCH(0, true);
    else if (act == 1) POOL_LAUNCH(1, true);
    else POOL_LAUNCH(2, true);
  } else {
    if (act == 0) POOL_LAUNCH(0, false);
    else if (act == 1) POOL_LAUNCH(1, false);
    else POOL_LAUNCH(2, false);
  }
#undef POOL_LAUNCH
  UNET_LAUNCH_CHECK("bn_act_pool_kernel");
  return UNET_OK;
}

int unet_max_pool2x2_bf16(const uint16_t *x, int x_ld, uint16_t *y, int y_ld, uint8_t *mask, int N, int H, int W,
                          int C, unet_stream_t s) {
  int rc = check_view("unet_max_pool2x2_bf16 (x)", x, x_ld, C);
  if (!rc) rc = check_view("unet_max_pool2x2_bf16 (y)", y, y_ld, C);
  if (rc) return rc;
  UNET_REQUIRE(mask && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0 &&
                   (((uintptr_t)mask) & 7) == 0,
               "unet_max_pool2x2_bf16: need even H and W, channels a multiple of 8, an 8-byte aligned mask");
  const i64 total = (i64)N * (H / 2) * (W / 2) * (C / 8);
  ProfScope prof(PROF_POOL_BF16, 0, (double)N * H * W * C * 2.0 * 1.25 + (double)total * 8, as_stream(s));
  hipLaunchKernelGGL(maxpool2_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), x, x_ld, y, y_ld, mask, C / 8,
                     H, W, total);
  UNET_LAUNCH_CHECK("maxpool2_kernel");
  return UNET_OK;
}

int unet_max_pool2x2_backward_bf16(const uint16_t *dy, int dy_ld, const uint8_t *mask, uint16_t *dx, int dx_ld, int N,
                                   int H, int W, int C, int accumulate, unet_stream_t s) {
  int rc = check_view("unet_max_pool2x2_backward_bf16 (dy)", dy, dy_ld, C);
  if (!rc) rc = check_view("unet_max_pool2x2_backward_bf16 (dx)", dx, dx_ld, C);
  if (rc) return rc;
  UNET_REQUIRE(mask && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
               "unet_max_pool2x2_backward_bf16: need even H and W and channels a multiple of 8");
  const i64 total = (i64)N * (H / 2) * (W / 2) * (C / 8);
  ProfScope prof(PROF_POOL_BF16, 0, (double)N * H * W * C * 2.0 * 1.25 + (double)total * 8, as_stream(s));
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), dy, dy_ld, mask, dx, dx_ld,
                     C / 8, H, W, total, accumulate);
  UNET_LAUNCH_CHECK("maxpool2_bwd_kernel");
  return UNET_OK;
}

}  // extern "C"
