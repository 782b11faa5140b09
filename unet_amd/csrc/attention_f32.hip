// Additive attention gate, fp32 NCHW (the reference's cudaAttention / cudaAttentionBackward,
// src/cuda_library/attention/attention.cu:9-499).
//
//   int1[n,o,h,w] = relu(b1[o] + sum_i x[n,i,2h,2w] W[i][o] + sum_j g[n,j,h,w] W[Cx+j][o])
//   alpha[n,h,w]  = sigmoid(bpsi + sum_o int1[n,o,h,w] psi[o])
//   out[n,c,h,w]  = x[n,c,h,w] * blend(alpha)(h,w)
//
// x [N,Cx,Hx,Wx], g [N,Cg,Hx/2,Wx/2]; "cells" below are the N*Hx/2*Wx/2 pixels of g.
// Parameters (layer.h:558-566): W[(Cx+Cg)][Ci] | psi[Ci] | b1[Ci] | bpsi.
// blend: the 4x4 block of output pixels that contains (h,w) mixes the 2x2 cells whose top-left is
// (2*(h/4), 2*(w/4)) with the fractions (h%4)/3 and (w%4)/3 (attention.cu:61-78); the "+1" cells are
// clamped to the last row / column, which the reference computes (:65-66) but does not use (it reads
// out of bounds when Hx/2 or Wx/2 is odd).
//
// Unlike the reference's backward (fp32 atomicAdd everywhere) every sum here has a fixed order:
// gathers instead of scatters, one workgroup per output tile for the weight gradient, split
// partials added in index order.
#include "bf16_util.h"
#include "common.h"

namespace {

constexpr int CELLS = 64;  // cells per workgroup in the gate kernels (one wavefront wide)

struct Blend {
  int c0, c1;
  float f;
};
__device__ __forceinline__ Blend blend_axis(int i, int cells) {
  Blend b;
  b.c0 = (i >> 2) << 1;
  b.c1 = min(b.c0 + 1, cells - 1);
  b.f = (float)(i & 3) / 3.0f;
  return b;
}
__device__ __forceinline__ float blend_value(const float *__restrict__ alpha_n, int h, int w, int Hg, int Wg) {
  const Blend bh = blend_axis(h, Hg), bw = blend_axis(w, Wg);
  const float A = alpha_n[(i64)bh.c0 * Wg + bw.c0], B = alpha_n[(i64)bh.c0 * Wg + bw.c1];
  const float C = alpha_n[(i64)bh.c1 * Wg + bw.c0], D = alpha_n[(i64)bh.c1 * Wg + bw.c1];
  const float r1 = B * bw.f + A * (1.0f - bw.f), r2 = D * bw.f + C * (1.0f - bw.f);
  return r2 * bh.f + r1 * (1.0f - bh.f);
}

// Input channel k of cell (n, h, w): x at the even pixels for k < Cx, g otherwise.  x_dense: x was
// gathered beforehand and is [N,Cx,Hg,Wg] like g.
struct GateIn {
  const float *x, *g;
  int Cx, Cg, Hg, Wg;
  int x_dense;
};
// Two per-cell base pointers: channel k is one plane stride further.
struct GateCell {
  const float *xp, *gp;
  i64 xplane, gplane;
  int Cx;
};
__device__ __forceinline__ GateCell gate_cell(const GateIn &in, i64 n, int h, int w) {
  GateCell c;
  c.gplane = (i64)in.Hg * in.Wg;
  c.xplane = in.x_dense ? c.gplane : 4 * c.gplane;
  c.xp = in.x + n * in.Cx * c.xplane + (in.x_dense ? (i64)h * in.Wg + w : (i64)(2 * h) * (2 * in.Wg) + 2 * w);
  c.gp = in.g + n * in.Cg * c.gplane + (i64)h * in.Wg + w;
  c.Cx = in.Cx;
  return c;
}
__device__ __forceinline__ float gate_input(const GateCell &c, int k) {
  return k < c.Cx ? c.xp[k * c.xplane] : c.gp[(k - c.Cx) * c.gplane];
}

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---- forward ---------------------------------------------------------------------------------
// The projections on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands and accumulator), for Ci a
// multiple of 32 and K a multiple of 16.  One wavefront owns 32 consecutive cells; per 32 intermediate
// channels it accumulates D[o][cell] += W[k][o] * in[k][cell] two input channels at a time: lane l supplies
// W[k + l/32][o0 + l%32] (a 128-byte row of the weights) and in[k + l/32][cell l%32] (128 bytes of a plane),
// eight such steps with their sixteen loads in flight per trip.  Accumulator register r of lane l is
// (o = o0 + 8*(r/4) + 4*(l/32) + r%4, cell l%32), so the int_1 stores are coalesced over the cells.
__global__ __launch_bounds__(256) void attn_gate_fwd_mfma_kernel(GateIn in, const float *__restrict__ params, int Ci,
                                                                 float *__restrict__ int1, float *__restrict__ alpha,
                                                                 i64 cells) {
  const int K = in.Cx + in.Cg;
  const float *psi = params + (i64)K * Ci, *b1 = psi + Ci;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
  const i64 cell = ((i64)blockIdx.x * 4 + wave) * 32 + j;
  const bool live = cell < cells;
  const i64 hw = (i64)in.Hg * in.Wg;
  const i64 n = live ? cell / hw : 0;
  const int rem = live ? (int)(cell - n * hw) : 0;
  const int h = rem / in.Wg, w = rem - h * in.Wg;
  const GateCell gc = gate_cell(in, n, h, w);
  float z = 0.f;
  for (int o0 = 0; o0 < Ci; o0 += 32) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb < K; kb += 16) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + 2 * u + half;
        a[u] = params[(i64)k * Ci + o0 + j];
        b[u] = live ? gate_input(gc, k) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + 8 * (r >> 2) + 4 * half + (r & 3);
      const float t = acc[r] + b1[o];
      const float v = t > 0.f ? t : 0.f;
      if (live) int1[(n * Ci + o) * hw + rem] = v;
      z = fmaf(v, psi[o], z);
    }
  }
  z += __shfl_xor(z, 32, 64);  // the two halves hold the other sixteen channels of the same cell
  if (half == 0 && live) alpha[cell] = 1.0f / (1.0f + expf(-(b1[Ci] + z)));
}

// 256 threads = 64 consecutive cells x 4 channel groups; a thread walks its group's intermediate
// channels eight at a time (the weights of a group are uniform over a wavefront).  The psi dot
// product is summed over the four groups through LDS in group order.
// FULL: every group has a multiple of eight channels and K is a multiple of eight, so the loops carry no
// bounds tests (with them the compiler emits a branch per channel and trip).
template <bool FULL>
__global__ __launch_bounds__(256) void attn_gate_fwd_kernel(GateIn in, const float *__restrict__ params, int Ci,
                                                            float *__restrict__ int1, float *__restrict__ alpha,
                                                            i64 cells) {
  __shared__ float zpart[4][CELLS];
  const int K = in.Cx + in.Cg;
  const float *psi = params + (i64)K * Ci, *b1 = psi + Ci;
  // the channel group is the wavefront index: read it as a scalar so that the weight addresses are uniform
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i64 cell = (i64)blockIdx.x * CELLS + lane;
  const bool live = cell < cells;
  const i64 hw = (i64)in.Hg * in.Wg;
  const i64 n = live ? cell / hw : 0;
  const int rem = live ? (int)(cell - n * hw) : 0;
  const int h = rem / in.Wg, w = rem - h * in.Wg;
  const int per = (Ci + 3) / 4, o_begin = grp * per, o_end = min(Ci, o_begin + per);
  const GateCell gc = gate_cell(in, n, h, w);
  float z = 0.f;
  for (int o0 = o_begin; o0 < o_end; o0 += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (FULL || o0 + j < o_end) ? b1[o0 + j] : 0.f;
    // eight input channels per trip: their loads are in flight together (x and g planes are hw apart)
    for (int kb = 0; kb < K; kb += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (live && (FULL || kb + u < K)) ? gate_input(gc, kb + u) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (!FULL && kb + u >= K) break;
        const float *wr = params + (i64)(kb + u) * Ci + o0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (FULL || o0 + j < o_end) acc[j] = fmaf(v[u], wr[j], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (FULL || o0 + j < o_end) {
        const float r = acc[j] > 0.f ? acc[j] : 0.f;
        if (live) int1[(n * Ci + o0 + j) * hw + rem] = r;
        z = fmaf(r, psi[o0 + j], z);
      }
  }
  zpart[grp][lane] = z;
  __syncthreads();
  if (grp == 0 && live) {
    const float t = b1[Ci] + ((zpart[0][lane] + zpart[1][lane]) + (zpart[2][lane] + zpart[3][lane]));
    alpha[cell] = 1.0f / (1.0f + expf(-t));
  }
}

// out = x * blend(alpha); one thread per (n, h, w) pixel, looping over the channels.
__global__ __launch_bounds__(256) void attn_resample_mul_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ alpha,
                                                                float *__restrict__ out, int Cx, int Hx, int Wx,
                                                                i64 pixels) {
  const int Hg = Hx / 2, Wg = Wx / 2;
  const i64 hw = (i64)Hx * Wx;
  for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (i64)gridDim.x * blockDim.x) {
    const i64 n = p / hw;
    const int rem = (int)(p - n * hw);
    const int h = rem / Wx, w = rem - h * Wx;
    const float a = blend_value(alpha + n * (i64)Hg * Wg, h, w, Hg, Wg);
    const i64 base = n * Cx * hw + rem;
    int c = 0;
    for (; c + 3 < Cx; c += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = x[base + (c + u) * hw];
#pragma unroll
      for (int u = 0; u < 4; ++u) out[base + (c + u) * hw] = v[u] * a;
    }
    for (; c < Cx; ++c) out[base + c * hw] = x[base + c * hw] * a;
  }
}

// ---- backward --------------------------------------------------------------------------------
// grad_x = grad_out * blend(alpha) and s[n,h,w] = sum_c grad_out * x (what the blend weights of
// pixel (h,w) receive); one thread per pixel, channels in order.
__global__ __launch_bounds__(256) void attn_resample_bwd_kernel(const float *__restrict__ grad_out,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ alpha,
                                                                float *__restrict__ grad_x, float *__restrict__ s,
                                                                int Cx, int Hx, int Wx, i64 pixels) {
  const int Hg = Hx / 2, Wg = Wx / 2;
  const i64 hw = (i64)Hx * Wx;
  for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (i64)gridDim.x * blockDim.x) {
    const i64 n = p / hw;
    const int rem = (int)(p - n * hw);
    const int h = rem / Wx, w = rem - h * Wx;
    const float a = blend_value(alpha + n * (i64)Hg * Wg, h, w, Hg, Wg);
    const i64 base = n * Cx * hw + rem;
    float acc = 0.f;
    int c = 0;
    for (; c + 3 < Cx; c += 4) {
      float go[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        go[u] = grad_out[base + (c + u) * hw];
        xv[u] = x[base + (c + u) * hw];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        grad_x[base + (c + u) * hw] = go[u] * a;
        acc = fmaf(go[u], xv[u], acc);
      }
    }
    for (; c < Cx; ++c) {
      const float go = grad_out[base + c * hw];
      grad_x[base + c * hw] = go * a;
      acc = fmaf(go, x[base + c * hw], acc);
    }
    s[p] = acc;
  }
}

// One thread per cell: gathers grad_alpha from the 4x4 pixel block that blends this cell (each cell
// belongs to exactly one block; with the clamp it can hold two corner roles of it), then sigmoid
// backward, and the gradient at the ReLU input for every intermediate channel:
//   grad_alpha -> galpha,  gpre[n,o,cell] = int1 > 0 ? grad_alpha * alpha * (1 - alpha) * psi[o] : 0
__global__ __launch_bounds__(256) void attn_cell_bwd_kernel(const float *__restrict__ s,
                                                            const float *__restrict__ alpha,
                                                            const float *__restrict__ int1,
                                                            const float *__restrict__ psi, float *__restrict__ galpha,
                                                            float *__restrict__ gpre, int Ci, int Hx, int Wx,
                                                            i64 cells) {
  const int Hg = Hx / 2, Wg = Wx / 2;
  const i64 hw = (i64)Hg * Wg;
  for (i64 cell = (i64)blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += (i64)gridDim.x * blockDim.x) {
    const i64 n = cell / hw;
    const int rem = (int)(cell - n * hw);
    const int hs = rem / Wg, ws = rem - hs * Wg;
    const int bh = hs >> 1, bw = ws >> 1;
    const float *sn = s + n * (i64)Hx * Wx;
    float ga = 0.f;
    for (int rh = 0; rh < 2; ++rh) {
      if ((rh == 0 ? 2 * bh : min(2 * bh + 1, Hg - 1)) != hs) continue;
      for (int rw = 0; rw < 2; ++rw) {
        if ((rw == 0 ? 2 * bw : min(2 * bw + 1, Wg - 1)) != ws) continue;
        for (int dh = 0; dh < 4; ++dh) {
          const int h = 4 * bh + dh;
          if (h >= Hx) break;
          const float fh = (float)dh / 3.0f, wh = rh ? fh : 1.0f - fh;
          for (int dw = 0; dw < 4; ++dw) {
            const int w = 4 * bw + dw;
            if (w >= Wx) break;
            const float fw = (float)dw / 3.0f, ww = rw ? fw : 1.0f - fw;
            ga = fmaf(sn[(i64)h * Wx + w], ww * wh, ga);
          }
        }
      }
    }
    galpha[cell] = ga;
    const float al = alpha[cell];
    const float gz = ga * al * (1.0f - al);
    for (int o = 0; o < Ci; ++o) {
      const i64 idx = (n * Ci + o) * hw + rem;
      gpre[idx] = int1[idx] > 0.f ? gz * psi[o] : 0.f;
    }
  }
}

// Input gradients of the two 1x1 projections: dst[k] (+)= sum_o gpre[o] * W[k][o] per cell.
// Same thread layout as the forward: 64 cells x 4 groups of input channels, eight at a time.
// The x part lands on the even pixels of grad_x, which already holds the resample term (or, x_dense, is written
// alone at the resolution of g).
// FULL: every group has a multiple of eight input channels and Ci is a multiple of eight.
template <bool FULL>
__global__ __launch_bounds__(256) void attn_gate_dgrad_kernel(const float *__restrict__ gpre,
                                                              const float *__restrict__ params, int Ci, int Cx,
                                                              int Cg, int Hg, int Wg, float *__restrict__ grad_x,
                                                              float *__restrict__ grad_g, i64 cells, int x_dense) {
  const int K = Cx + Cg;
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i64 cell = (i64)blockIdx.x * CELLS + lane;
  if (cell >= cells) return;
  const i64 hw = (i64)Hg * Wg;
  const i64 n = cell / hw;
  const int rem = (int)(cell - n * hw);
  const int h = rem / Wg, w = rem - h * Wg;
  const int per = (K + 3) / 4, k_begin = grp * per, k_end = min(K, k_begin + per);
  const float *gp = gpre + n * Ci * hw + rem;
  for (int k0 = k_begin; k0 < k_end; k0 += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ob = 0; ob < Ci; ob += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (FULL || ob + u < Ci) ? gp[(ob + u) * hw] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (!FULL && ob + u >= Ci) break;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (FULL || k0 + j < k_end) acc[j] = fmaf(v[u], params[(i64)(k0 + j) * Ci + ob + u], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      if (!FULL && k >= k_end) break;
      if (k < Cx) {
        if (x_dense) {  // the projection's term alone, [N,Cx,Hg,Wg]
          grad_x[((n * Cx + k) * Hg + h) * (i64)Wg + w] = acc[j];
        } else {
          const i64 idx = ((n * Cx + k) * (2 * Hg) + 2 * h) * (i64)(2 * Wg) + 2 * w;
          grad_x[idx] += acc[j];
        }
      } else {
        grad_g[((n * Cg + (k - Cx)) * Hg + h) * (i64)Wg + w] = acc[j];
      }
    }
  }
}

// wt[o][k] = W[k][o]: the input-gradient kernel below reads the weights with k across the lanes
__global__ __launch_bounds__(256) void attn_transpose_w_kernel(const float *__restrict__ params, int K, int Ci,
                                                               float *__restrict__ wt) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K * Ci; e += gridDim.x * blockDim.x) {
    const int o = e / K, k = e - o * K;
    wt[e] = params[(i64)k * Ci + o];
  }
}

// The input gradients on the matrix cores, for K a multiple of 32 and Ci a multiple of 16: one wavefront
// owns 32 cells; per 32 input channels D[k][cell] += W[k][o] * gpre[o][cell], two intermediate channels at a
// time (lane l supplies wt[o + l/32][k0 + l%32] and gpre[o + l/32][cell l%32]).  Outputs as in the kernel above.
__global__ __launch_bounds__(256) void attn_gate_dgrad_mfma_kernel(const float *__restrict__ gpre,
                                                                   const float *__restrict__ wt, int Ci, int Cx,
                                                                   int Cg, int Hg, int Wg, float *__restrict__ grad_x,
                                                                   float *__restrict__ grad_g, i64 cells,
                                                                   int x_dense) {
  const int K = Cx + Cg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
  const i64 cell = ((i64)blockIdx.x * 4 + wave) * 32 + j;
  const bool live = cell < cells;
  const i64 hw = (i64)Hg * Wg;
  const i64 n = live ? cell / hw : 0;
  const int rem = live ? (int)(cell - n * hw) : 0;
  const int h = rem / Wg, w = rem - h * Wg;
  const float *gp = gpre + n * Ci * hw + rem;
  for (int k0 = 0; k0 < K; k0 += 32) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ob = 0; ob < Ci; ob += 16) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = ob + 2 * u + half;
        a[u] = wt[(i64)o * K + k0 + j];
        b[u] = live ? gp[o * hw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    if (!live) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
      if (k < Cx) {
        if (x_dense)
          grad_x[((n * Cx + k) * Hg + h) * (i64)Wg + w] = acc[r];
        else
          grad_x[((n * Cx + k) * (2 * Hg) + 2 * h) * (i64)(2 * Wg) + 2 * w] += acc[r];
      } else {
        grad_g[((n * Cg + (k - Cx)) * Hg + h) * (i64)Wg + w] = acc[r];
      }
    }
  }
}

// Weight gradient of the projections: dW[k][o] = sum_cells in[k] * gpre[o], a GEMM whose reduction runs
// over the cells.  A workgroup owns a 32(k) x 32(o) tile over a slice of the cells, staged 64 cells at a
// time through LDS; each of its four wavefronts multiplies a quarter of the staged cells with
// v_mfma_f32_32x32x2_f32 (fp32 operands and accumulator) and the four accumulators are added in
// wavefront order at the end.  part[split][K][Ci].
__global__ __launch_bounds__(256) void attn_wgrad_kernel(GateIn in, const float *__restrict__ gpre, int Ci,
                                                         float *__restrict__ part, i64 cells, i64 cells_per_split) {
  constexpr int PITCH = CELLS + 1;
  __shared__ float sm[2 * 32 * PITCH];  // A[32][PITCH] (inputs), B[32][PITCH] (gpre); reused for the final sum
  float *A = sm, *B = sm + 32 * PITCH;
  const int K = in.Cx + in.Cg;
  const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32, split = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 hw = (i64)in.Hg * in.Wg;
  const i64 c_begin = split * cells_per_split, c_end = min(cells, c_begin + cells_per_split);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (i64 cb = c_begin; cb < c_end; cb += CELLS) {
    {
      // thread t stages cell t % 64 of the chunk for the eight channels t / 64 + 4 i: one index
      // computation per chunk, sixteen loads in flight
      const int cl = lane;
      const i64 cell = cb + cl;
      const bool in_range = cell < c_end;
      const unsigned ucell = in_range ? (unsigned)cell : 0u;  // cells < 2^31 (checked by the caller)
      const unsigned n = ucell / (unsigned)hw, rem = ucell - n * (unsigned)hw;
      const int h = (int)(rem / (unsigned)in.Wg), w = (int)(rem - (unsigned)h * (unsigned)in.Wg);
      const GateCell gc = gate_cell(in, n, h, w);
      float a[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ch = wave + 4 * i;
        a[i] = (in_range && k0 + ch < K) ? gate_input(gc, k0 + ch) : 0.f;
        b[i] = (in_range && o0 + ch < Ci) ? gpre[((i64)n * Ci + o0 + ch) * hw + rem] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        A[(wave + 4 * i) * PITCH + cl] = a[i];
        B[(wave + 4 * i) * PITCH + cl] = b[i];
      }
    }
    __syncthreads();
    // lane l supplies A[row l%32][cell] and B[row l%32][cell] of cell 2q + l/32 of this wavefront's sixteen
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cl = wave * 16 + 2 * q + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(lane & 31) * PITCH + cl], B[(lane & 31) * PITCH + cl], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // accumulator register r of lane l is element (k = 8*(r/4) + 4*(l/32) + r%4, o = l%32) of the tile
  float *red = sm;  // [wave][32][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kk = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
    red[(wave * 32 + kk) * 32 + (lane & 31)] = acc[r];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 256) {
    const int k = k0 + (e >> 5), o = o0 + (e & 31);
    if (k < K && o < Ci)
      part[((i64)split * K + k) * Ci + o] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
  }
}

// grad[e] = sum_split part[split][e], splits in order
__global__ __launch_bounds__(256) void attn_wgrad_reduce_kernel(const float *__restrict__ part, int splits, i64 count,
                                                                float *__restrict__ grad) {
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (i64)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int sidx = 0; sidx < splits; ++sidx) t += part[(i64)sidx * count + e];
    grad[e] = t;
  }
}

// The vector-shaped gradients in two stages (fixed order in both):
//   o <  Ci : dpsi[o] = sum_cells gz * int1[o]   and   db1[o] = sum_cells gpre[o]
//   o == Ci : dbpsi   = sum_cells gz,            gz = galpha * alpha * (1 - alpha)
// Stage 1: workgroup (slice, o) sums its slice of the cells into part[o][slice][2].
__global__ __launch_bounds__(256) void attn_vec_grad_kernel(const float *__restrict__ galpha,
                                                            const float *__restrict__ alpha,
                                                            const float *__restrict__ int1,
                                                            const float *__restrict__ gpre, int Ci, i64 hw,
                                                            i64 cells, i64 cells_per_slice, float *__restrict__ part) {
  __shared__ float red[2 * 4];
  const int slice = blockIdx.x, o = blockIdx.y;
  const i64 c_begin = slice * cells_per_slice, c_end = min(cells, c_begin + cells_per_slice);
  float v[2] = {0.f, 0.f};
  for (i64 cell = c_begin + threadIdx.x; cell < c_end; cell += 256) {
    const float al = alpha[cell];
    const float gz = galpha[cell] * al * (1.0f - al);
    if (o < Ci) {
      const i64 n = cell / hw;
      const i64 idx = (n * Ci + o) * hw + (cell - n * hw);
      v[0] = fmaf(gz, int1[idx], v[0]);
      v[1] += gpre[idx];
    } else {
      v[0] += gz;
    }
  }
  block_sum<2>(v, red);
  if (threadIdx.x == 0) {
    part[((i64)o * gridDim.x + slice) * 2] = v[0];
    part[((i64)o * gridDim.x + slice) * 2 + 1] = v[1];
  }
}
// Stage 2: one wavefront per o adds the slices in order.
__global__ __launch_bounds__(64) void attn_vec_grad_finish_kernel(const float *__restrict__ part, int slices, int Ci,
                                                                  float *__restrict__ grad_psi,
                                                                  float *__restrict__ grad_b1,
                                                                  float *__restrict__ grad_bpsi) {
  const int o = blockIdx.x;
  if (threadIdx.x != 0) return;
  float a = 0.f, b = 0.f;
  for (int sl = 0; sl < slices; ++sl) {
    a += part[((i64)o * slices + sl) * 2];
    b += part[((i64)o * slices + sl) * 2 + 1];
  }
  if (o < Ci) {
    grad_psi[o] = a;
    grad_b1[o] = b;
  } else {
    grad_bpsi[0] = a;
  }
}

// ---- the gate between NHWC bf16 tensors ---------------------------------------------------------
// The projections, psi and their gradients stay on the fp32 kernels above (on x gathered at the even pixels);
// the passes over the full-resolution tensors read and write NHWC bf16 directly.

// xs[n,c,h,w] (fp32, resolution of g) = x[n,2h,2w,c] (NHWC bf16, pixel stride ld): a 32-cell x 32-channel
// tile transposed through LDS.
__global__ __launch_bounds__(256) void attn_gather_x_kernel(const bf16_t *__restrict__ x, int ld,
                                                            float *__restrict__ xs, int Cx, int Hg, int Wg) {
  __shared__ float t[32][33];
  const i64 n = blockIdx.z, HWg = (i64)Hg * Wg;
  const i64 p0 = (i64)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const i64 p = p0 + r;
    const int c = c0 + tx;
    float v = 0.f;
    if (p < HWg && c < Cx) {
      const int h = (int)(p / Wg), w = (int)(p - (i64)h * Wg);
      v = bf16_to_f32(x[((n * 2 * Hg + 2 * h) * (i64)(2 * Wg) + 2 * w) * ld + c]);
    }
    t[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const i64 p = p0 + tx;
    if (c < Cx && p < HWg) xs[(n * Cx + c) * HWg + p] = t[tx][r];
  }
}

__device__ __forceinline__ void unpack8f(const uint4 &r, float (&v)[8]) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = bf16_to_f32(w[j] & 0xffffu);
    v[2 * j + 1] = bf16_to_f32(w[j] >> 16);
  }
}
__device__ __forceinline__ uint4 pack8f(const float (&v)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = f32_to_bf16(v[2 * j]) | (f32_to_bf16(v[2 * j + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// out = bf16(x * blend(alpha)), one lane per (pixel, 8 channels)
__global__ __launch_bounds__(256) void attn_apply_bf16_kernel(const bf16_t *__restrict__ x, int x_ld,
                                                              const float *__restrict__ alpha,
                                                              bf16_t *__restrict__ out, int out_ld, int CG, int Hx,
                                                              int Wx, i64 total) {
  const int Hg = Hx / 2, Wg = Wx / 2;
  const i64 hw = (i64)Hx * Wx;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    const i64 p = i / CG, n = p / hw;
    const int rem = (int)(p - n * hw);
    const int h = rem / Wx, w = rem - h * Wx;
    const float a = blend_value(alpha + n * (i64)Hg * Wg, h, w, Hg, Wg);
    float v[8];
    unpack8f(*reinterpret_cast<const uint4 *>(x + p * x_ld + cg * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= a;
    *reinterpret_cast<uint4 *>(out + p * out_ld + cg * 8) = pack8f(v);
  }
}

// s[pixel] = sum_c (dy1 + dy2) * x: eight lanes per pixel, each walking every eighth 8-channel group; the
// eight partial sums are added by a fixed xor butterfly.
__global__ __launch_bounds__(256) void attn_sums_bf16_kernel(const bf16_t *__restrict__ dy1, int ld1,
                                                             const bf16_t *__restrict__ dy2, int ld2,
                                                             const bf16_t *__restrict__ x, int x_ld,
                                                             float *__restrict__ s, int CG, i64 pixels) {
  const int sub = threadIdx.x & 7;
  // the eight lanes of a group share p, so they leave the loop together
  for (i64 p = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 3; p < pixels; p += ((i64)gridDim.x * blockDim.x) >> 3) {
    float acc = 0.f;
    for (int cg = sub; cg < CG; cg += 8) {
      float g[8], xv[8];
      unpack8f(*reinterpret_cast<const uint4 *>(dy1 + p * ld1 + cg * 8), g);
      if (dy2) {
        float g2[8];
        unpack8f(*reinterpret_cast<const uint4 *>(dy2 + p * ld2 + cg * 8), g2);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += g2[j];
      }
      unpack8f(*reinterpret_cast<const uint4 *>(x + p * x_ld + cg * 8), xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(g[j], xv[j], acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) s[p] = acc;
  }
}

// dx (+)= bf16((dy1 + dy2) * blend(alpha) + [even pixel] grad_xs[n,c,h/2,w/2]), one lane per (pixel, 8 channels)
__global__ __launch_bounds__(256) void attn_apply_bwd_bf16_kernel(const bf16_t *__restrict__ dy1, int ld1,
                                                                  const bf16_t *__restrict__ dy2, int ld2,
                                                                  const float *__restrict__ alpha,
                                                                  const float *__restrict__ grad_xs,
                                                                  bf16_t *__restrict__ dx, int dx_ld, int accumulate,
                                                                  int CG, int Hx, int Wx, i64 total) {
  const int Hg = Hx / 2, Wg = Wx / 2, Cx = CG * 8;
  const i64 hw = (i64)Hx * Wx, hwg = (i64)Hg * Wg;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    const i64 p = i / CG, n = p / hw;
    const int rem = (int)(p - n * hw);
    const int h = rem / Wx, w = rem - h * Wx;
    const float a = blend_value(alpha + n * hwg, h, w, Hg, Wg);
    float g[8];
    unpack8f(*reinterpret_cast<const uint4 *>(dy1 + p * ld1 + cg * 8), g);
    if (dy2) {
      float g2[8];
      unpack8f(*reinterpret_cast<const uint4 *>(dy2 + p * ld2 + cg * 8), g2);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += g2[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= a;
    if (((h | w) & 1) == 0) {
      const float *gx = grad_xs + (n * Cx + cg * 8) * hwg + (i64)(h >> 1) * Wg + (w >> 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += gx[j * hwg];
    }
    bf16_t *dst = dx + p * dx_ld + cg * 8;
    if (accumulate) {
      float o[8];
      unpack8f(*reinterpret_cast<const uint4 *>(dst), o);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += o[j];
    }
    *reinterpret_cast<uint4 *>(dst) = pack8f(g);
  }
}

int launch_gate_fwd(const GateIn &in, const float *weights, int Ci, float *int1, float *alpha, i64 cells,
                    hipStream_t st) {
  const int K = in.Cx + in.Cg;
  if (Ci % 32 == 0 && K % 16 == 0)
    hipLaunchKernelGGL(attn_gate_fwd_mfma_kernel, dim3(cdiv_u(cells, 128)), dim3(256), 0, st, in, weights, Ci, int1,
                       alpha, cells);
  else if (Ci % 32 == 0 && K % 8 == 0)
    hipLaunchKernelGGL(attn_gate_fwd_kernel<true>, dim3(cdiv_u(cells, CELLS)), dim3(256), 0, st, in, weights, Ci, int1,
                       alpha, cells);
  else
    hipLaunchKernelGGL(attn_gate_fwd_kernel<false>, dim3(cdiv_u(cells, CELLS)), dim3(256), 0, st, in, weights, Ci,
                       int1, alpha, cells);
  UNET_LAUNCH_CHECK("attn_gate_fwd_kernel");
  return UNET_OK;
}

int check_attention_shape(const char *who, int batch, int H_x, int W_x, int Cx, int Cg, int Ci) {
  UNET_REQUIRE(batch > 0 && H_x >= 2 && W_x >= 2 && H_x % 2 == 0 && W_x % 2 == 0 && Cx > 0 && Cg > 0 && Ci > 0,
               "%s: need batch, channel counts > 0 and even H_x, W_x >= 2 (g is half the size of x)", who);
  UNET_REQUIRE((i64)batch * H_x * W_x < ((i64)1 << 31), "%s: batch * H_x * W_x must stay below 2^31", who);
  return UNET_OK;
}

}  // namespace

extern "C" {

int unet_attention_param_count(int channels_in_x, int channels_in_g, int int_channels) {
  return (channels_in_x + channels_in_g + 2) * int_channels + 1;
}

int unet_attention_f32(float *activations_int_1, float *activations_int_2, float *activations, const float *weights,
                       const float *input_x, const float *input_g, int batch_size, int H_x, int W_x,
                       int channels_in_x, int channels_in_g, int int_channels, unet_stream_t s) {
  UNET_REQUIRE(activations_int_1 && activations_int_2 && activations && weights && input_x && input_g,
               "unet_attention_f32: NULL pointer");
  int rc = check_attention_shape("unet_attention_f32", batch_size, H_x, W_x, channels_in_x, channels_in_g, int_channels);
  if (rc) return rc;
  const int Hg = H_x / 2, Wg = W_x / 2;
  const i64 cells = (i64)batch_size * Hg * Wg, pixels = 4 * cells;
  const GateIn in{input_x, input_g, channels_in_x, channels_in_g, Hg, Wg, 0};
  rc = launch_gate_fwd(in, weights, int_channels, activations_int_1, activations_int_2, cells, as_stream(s));
  if (rc) return rc;
  hipLaunchKernelGGL(attn_resample_mul_kernel, dim3(ew_grid(pixels)), dim3(256), 0, as_stream(s), input_x,
                     (const float *)activations_int_2, activations, channels_in_x, H_x, W_x, pixels);
  UNET_LAUNCH_CHECK("attn_resample_mul_kernel");
  return UNET_OK;
}

// Everything of the backward after s = sum_c grad_out * x is known: cell gathers, sigmoid / ReLU backward,
// input and weight gradients of the projections.  grad_x is the full-resolution tensor (the projection's
// term is added at its even pixels) or, in.x_dense, the [N,Cx,Hg,Wg] term alone.
struct AttnPlan {
  i64 cells, splits, cells_per_split, cells_per_slice;
  int slices;
  size_t part_floats, vec_floats;
};
static AttnPlan attention_plan(int batch, int Hg, int Wg, int K, int Ci) {
  AttnPlan p;
  p.cells = (i64)batch * Hg * Wg;
  // weight-gradient split over the cells: enough workgroups to fill the device when the tile count is small
  const int tiles = (int)cdiv_u(K, 32) * (int)cdiv_u(Ci, 32);
  i64 splits = (512 + tiles - 1) / tiles;
  const i64 max_splits = (p.cells + 4 * CELLS - 1) / (4 * CELLS);
  if (splits > max_splits) splits = max_splits;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  p.cells_per_split = (p.cells + splits - 1) / splits;
  p.cells_per_split = (p.cells_per_split + CELLS - 1) / CELLS * CELLS;
  p.splits = (p.cells + p.cells_per_split - 1) / p.cells_per_split;
  // slices of the cells for the vector-shaped gradients
  p.slices = (int)((p.cells + 4095) / 4096);
  if (p.slices > 64) p.slices = 64;
  p.cells_per_slice = (p.cells + p.slices - 1) / p.slices;
  p.part_floats = (size_t)p.splits * K * Ci;
  p.vec_floats = (size_t)(Ci + 1) * p.slices * 2;
  return p;
}
static int attention_backward_core(const AttnPlan &pl, const float *sbuf, const float *int1, const float *alpha,
                                   const GateIn &in, const float *weights, float *gpre, float *galpha, float *grad_x,
                                   float *grad_g, float *grad_weights, float *part, float *vec_part, int Ci, int H_x,
                                   int W_x, hipStream_t st) {
  const int Cx = in.Cx, Cg = in.Cg, K = Cx + Cg, Hg = in.Hg, Wg = in.Wg;
  const i64 cells = pl.cells;
  const float *psi = weights + (i64)K * Ci;
  float *grad_psi = grad_weights + (i64)K * Ci, *grad_b1 = grad_psi + Ci, *grad_bpsi = grad_b1 + Ci;
  hipLaunchKernelGGL(attn_cell_bwd_kernel, dim3(ew_grid(cells)), dim3(256), 0, st, sbuf, alpha, int1, psi, galpha,
                     gpre, Ci, H_x, W_x, cells);
  UNET_LAUNCH_CHECK("attn_cell_bwd_kernel");
  if (K % 32 == 0 && Ci % 16 == 0) {
    // part (the weight-gradient partials, at least K*Ci floats) is free until the weight-gradient kernel runs
    hipLaunchKernelGGL(attn_transpose_w_kernel, dim3(ew_grid((i64)K * Ci)), dim3(256), 0, st, weights, K, Ci, part);
    UNET_LAUNCH_CHECK("attn_transpose_w_kernel");
    hipLaunchKernelGGL(attn_gate_dgrad_mfma_kernel, dim3(cdiv_u(cells, 128)), dim3(256), 0, st, (const float *)gpre,
                       (const float *)part, Ci, Cx, Cg, Hg, Wg, grad_x, grad_g, cells, in.x_dense);
  } else if (K % 32 == 0 && Ci % 8 == 0)
    hipLaunchKernelGGL(attn_gate_dgrad_kernel<true>, dim3(cdiv_u(cells, CELLS)), dim3(256), 0, st, (const float *)gpre,
                       weights, Ci, Cx, Cg, Hg, Wg, grad_x, grad_g, cells, in.x_dense);
  else
    hipLaunchKernelGGL(attn_gate_dgrad_kernel<false>, dim3(cdiv_u(cells, CELLS)), dim3(256), 0, st,
                       (const float *)gpre, weights, Ci, Cx, Cg, Hg, Wg, grad_x, grad_g, cells, in.x_dense);
  UNET_LAUNCH_CHECK("attn_gate_dgrad_kernel");
  hipLaunchKernelGGL(attn_wgrad_kernel, dim3(cdiv_u(K, 32), cdiv_u(Ci, 32), (unsigned)pl.splits), dim3(256), 0, st, in,
                     (const float *)gpre, Ci, part, cells, pl.cells_per_split);
  UNET_LAUNCH_CHECK("attn_wgrad_kernel");
  hipLaunchKernelGGL(attn_wgrad_reduce_kernel, dim3(ew_grid((i64)K * Ci)), dim3(256), 0, st, (const float *)part,
                     (int)pl.splits, (i64)K * Ci, grad_weights);
  UNET_LAUNCH_CHECK("attn_wgrad_reduce_kernel");
  hipLaunchKernelGGL(attn_vec_grad_kernel, dim3(pl.slices, Ci + 1), dim3(256), 0, st, (const float *)galpha, alpha,
                     int1, (const float *)gpre, Ci, (i64)Hg * Wg, cells, pl.cells_per_slice, vec_part);
  UNET_LAUNCH_CHECK("attn_vec_grad_kernel");
  hipLaunchKernelGGL(attn_vec_grad_finish_kernel, dim3(Ci + 1), dim3(64), 0, st, (const float *)vec_part, pl.slices,
                     Ci, grad_psi, grad_b1, grad_bpsi);
  UNET_LAUNCH_CHECK("attn_vec_grad_finish_kernel");
  return UNET_OK;
}

int unet_attention_backward_f32(const float *grad_activations, const float *activations_int_1,
                                const float *activations_int_2, const float *input_x, const float *input_g,
                                const float *weights, float *grad_activations_int_1, float *grad_activations_int_2,
                                float *grad_input_x, float *grad_input_g, float *grad_weights, int batch_size, int H_x,
                                int W_x, int channels_in_x, int channels_in_g, int int_channels, unet_stream_t s) {
  UNET_REQUIRE(grad_activations && activations_int_1 && activations_int_2 && input_x && input_g && weights &&
                   grad_activations_int_1 && grad_activations_int_2 && grad_input_x && grad_input_g && grad_weights,
               "unet_attention_backward_f32: NULL pointer");
  int rc = check_attention_shape("unet_attention_backward_f32", batch_size, H_x, W_x, channels_in_x, channels_in_g,
                                 int_channels);
  if (rc) return rc;
  const int Cx = channels_in_x, Cg = channels_in_g, Ci = int_channels;
  const int Hg = H_x / 2, Wg = W_x / 2;
  const AttnPlan pl = attention_plan(batch_size, Hg, Wg, Cx + Cg, Ci);
  const i64 pixels = 4 * pl.cells;
  // workspace: s [pixels] | weight-gradient partials [splits][K][Ci] | vector partials [Ci+1][slices][2]
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * ((size_t)pixels + pl.part_floats + pl.vec_floats), as_stream(s));
  if (rc) return rc;
  float *sbuf = (float *)ws, *part = sbuf + pixels, *vec_part = part + pl.part_floats;
  const GateIn in{input_x, input_g, Cx, Cg, Hg, Wg, 0};
  hipLaunchKernelGGL(attn_resample_bwd_kernel, dim3(ew_grid(pixels)), dim3(256), 0, as_stream(s), grad_activations,
                     input_x, activations_int_2, grad_input_x, sbuf, Cx, H_x, W_x, pixels);
  UNET_LAUNCH_CHECK("attn_resample_bwd_kernel");
  return attention_backward_core(pl, sbuf, activations_int_1, activations_int_2, in, weights, grad_activations_int_1,
                                 grad_activations_int_2, grad_input_x, grad_input_g, grad_weights, part, vec_part, Ci,
                                 H_x, W_x, as_stream(s));
}

static int check_attention_views(const char *who, const void *x, int x_ld, const void *g, int g_ld, int Cx, int Cg) {
  UNET_REQUIRE(x && g && Cx % 8 == 0 && Cg % 8 == 0 && x_ld >= Cx && g_ld >= Cg && x_ld % 8 == 0 && g_ld % 8 == 0 &&
                   (((uintptr_t)x | (uintptr_t)g) & 15) == 0,
               "%s: x and g must be 16-byte aligned NHWC bf16 views with channel counts and pixel strides in "
               "multiples of 8", who);
  return UNET_OK;
}

int unet_attention_bf16(const uint16_t *x, int x_ld, const uint16_t *g, int g_ld, const float *weights,
                        float *activations_int_1, float *activations_int_2, uint16_t *out, int out_ld, int batch_size,
                        int H_x, int W_x, int channels_in_x, int channels_in_g, int int_channels, unet_stream_t s) {
  UNET_REQUIRE(weights && activations_int_1 && activations_int_2 && out, "unet_attention_bf16: NULL pointer");
  int rc = check_attention_shape("unet_attention_bf16", batch_size, H_x, W_x, channels_in_x, channels_in_g, int_channels);
  if (!rc) rc = check_attention_views("unet_attention_bf16", x, x_ld, g, g_ld, channels_in_x, channels_in_g);
  if (rc) return rc;
  UNET_REQUIRE(out_ld >= channels_in_x && out_ld % 8 == 0 && ((uintptr_t)out & 15) == 0 && batch_size <= 65535,
               "unet_attention_bf16: out must be a 16-byte aligned view with a pixel stride multiple of 8");
  const int Cx = channels_in_x, Cg = channels_in_g, Ci = int_channels, Hg = H_x / 2, Wg = W_x / 2;
  const i64 cells = (i64)batch_size * Hg * Wg, pixels = 4 * cells;
  // workspace: xs [cells][Cx] | g in fp32 [cells][Cg]
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (size_t)cells * (Cx + Cg), as_stream(s));
  if (rc) return rc;
  float *xs = (float *)ws, *gf = xs + (size_t)cells * Cx;
  hipLaunchKernelGGL(attn_gather_x_kernel, dim3(cdiv_u((i64)Hg * Wg, 32), cdiv_u(Cx, 32), (unsigned)batch_size), dim3(256), 0,
                     as_stream(s), x, x_ld, xs, Cx, Hg, Wg);
  UNET_LAUNCH_CHECK("attn_gather_x_kernel");
  rc = unet_nhwc_bf16_to_nchw_f32(g, gf, batch_size, Cg, Hg, Wg, g_ld, s);
  if (rc) return rc;
  const GateIn in{xs, gf, Cx, Cg, Hg, Wg, 1};
  rc = launch_gate_fwd(in, weights, Ci, activations_int_1, activations_int_2, cells, as_stream(s));
  if (rc) return rc;
  const i64 total = pixels * (Cx / 8);
  hipLaunchKernelGGL(attn_apply_bf16_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), x, x_ld,
                     (const float *)activations_int_2, out, out_ld, Cx / 8, H_x, W_x, total);
  UNET_LAUNCH_CHECK("attn_apply_bf16_kernel");
  return UNET_OK;
}

int unet_attention_backward_bf16(const uint16_t *dy, int dy_ld, const uint16_t *dy2, int dy2_ld, const uint16_t *x,
                                 int x_ld, const uint16_t *g, int g_ld, const float *weights,
                                 const float *activations_int_1, const float *activations_int_2, uint16_t *dx,
                                 int dx_ld, int dx_accumulate, uint16_t *dg, int dg_ld, int dg_accumulate,
                                 float *grad_weights, int batch_size, int H_x, int W_x, int channels_in_x,
                                 int channels_in_g, int int_channels, unet_stream_t s) {
  UNET_REQUIRE(dy && weights && activations_int_1 && activations_int_2 && dx && dg && grad_weights,
               "unet_attention_backward_bf16: NULL pointer");
  int rc = check_attention_shape("unet_attention_backward_bf16", batch_size, H_x, W_x, channels_in_x, channels_in_g,
                                 int_channels);
  if (!rc) rc = check_attention_views("unet_attention_backward_bf16", x, x_ld, g, g_ld, channels_in_x, channels_in_g);
  if (!rc) rc = check_attention_views("unet_attention_backward_bf16 (gradients)", dx, dx_ld, dg, dg_ld, channels_in_x, channels_in_g);
  if (rc) return rc;
  UNET_REQUIRE(dy_ld >= channels_in_x && dy_ld % 8 == 0 && ((uintptr_t)dy & 15) == 0 &&
                   (!dy2 || (dy2_ld >= channels_in_x && dy2_ld % 8 == 0 && ((uintptr_t)dy2 & 15) == 0)) &&
                   batch_size <= 65535,
               "unet_attention_backward_bf16: dy (and dy2) must be 16-byte aligned views with pixel strides in multiples of 8");
  const int Cx = channels_in_x, Cg = channels_in_g, Ci = int_channels, Hg = H_x / 2, Wg = W_x / 2;
  const AttnPlan pl = attention_plan(batch_size, Hg, Wg, Cx + Cg, Ci);
  const i64 cells = pl.cells, pixels = 4 * cells;
  // workspace: xs | g fp32 | s | galpha | gpre | grad_xs | grad_g fp32 | weight-gradient partials | vector partials
  const size_t n_xs = (size_t)cells * Cx, n_g = (size_t)cells * Cg, n_pre = (size_t)cells * Ci;
  void *ws;
  rc = unet_workspace(&ws, sizeof(float) * (2 * n_xs + 2 * n_g + (size_t)pixels + (size_t)cells + n_pre +
                                            pl.part_floats + pl.vec_floats), as_stream(s));
  if (rc) return rc;
  float *xs = (float *)ws, *gf = xs + n_xs, *sbuf = gf + n_g, *galpha = sbuf + pixels, *gpre = galpha + cells;
  float *grad_xs = gpre + n_pre, *grad_gf = grad_xs + n_xs, *part = grad_gf + n_g, *vec_part = part + pl.part_floats;
  hipLaunchKernelGGL(attn_gather_x_kernel, dim3(cdiv_u((i64)Hg * Wg, 32), cdiv_u(Cx, 32), (unsigned)batch_size), dim3(256), 0,
                     as_stream(s), x, x_ld, xs, Cx, Hg, Wg);
  UNET_LAUNCH_CHECK("attn_gather_x_kernel");
  rc = unet_nhwc_bf16_to_nchw_f32(g, gf, batch_size, Cg, Hg, Wg, g_ld, s);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_sums_bf16_kernel, dim3(ew_grid(pixels * 8)), dim3(256), 0, as_stream(s), dy, dy_ld, dy2,
                     dy2_ld, x, x_ld, sbuf, Cx / 8, pixels);
  UNET_LAUNCH_CHECK("attn_sums_bf16_kernel");
  const GateIn in{xs, gf, Cx, Cg, Hg, Wg, 1};
  rc = attention_backward_core(pl, sbuf, activations_int_1, activations_int_2, in, weights, gpre, galpha, grad_xs,
                               grad_gf, grad_weights, part, vec_part, Ci, H_x, W_x, as_stream(s));
  if (rc) return rc;
  const i64 total = pixels * (Cx / 8);
  hipLaunchKernelGGL(attn_apply_bwd_bf16_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), dy, dy_ld, dy2,
                     dy2_ld, activations_int_2, (const float *)grad_xs, dx, dx_ld, dx_accumulate, Cx / 8, H_x, W_x,
                     total);
  UNET_LAUNCH_CHECK("attn_apply_bwd_bf16_kernel");
  if (dg_accumulate) return unet_nchw_f32_add_to_nhwc_bf16(grad_gf, dg, batch_size, Cg, Hg, Wg, dg_ld, s);
  return unet_nchw_f32_to_nhwc_bf16(grad_gf, dg, batch_size, Cg, Hg, Wg, dg_ld, Cg, s);
}

}  // extern "C"
