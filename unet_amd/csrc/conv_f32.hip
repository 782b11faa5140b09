// fp32 NCHW convolution and transposed convolution (part 2 of the fp32
// operator set).  LDS-tiled direct kernels on the fp32 vector pipe: every
// product is an fmaf in fp32, reductions over pixels are tree sums with a
// fixed order (deterministic, no atomics).
//
// Weight layout everywhere: [C_in][C_out][k][k] followed by C_out biases
// (src/cuda_library/2Dconvolution/convolution.cu:44-46,77-78).
#include "common.h"

namespace {

constexpr int TILE = 16;  // a block covers a 16x16 pixel tile (256 threads)

__device__ __forceinline__ void fma4x4(float (&acc)[16], float v, const float *w) {
  const float4 *w4 = reinterpret_cast<const float4 *>(w);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = w4[q];
    acc[4 * q + 0] = fmaf(v, t.x, acc[4 * q + 0]);
    acc[4 * q + 1] = fmaf(v, t.y, acc[4 * q + 1]);
    acc[4 * q + 2] = fmaf(v, t.z, acc[4 * q + 2]);
    acc[4 * q + 3] = fmaf(v, t.w, acc[4 * q + 3]);
  }
}

// ---------------------------------------------------------------- forward --
// Block: one image, a 16x16 output tile, COB=16 output channels.  Loops over
// input channels in chunks of CIB=8, staging the input halo patch
// ((15*stride + k)^2 per channel) and the weight slice [ci][tap][co] in LDS.
constexpr int F_COB = 16, F_CIB = 8;

// COB = output channels per workgroup: F_COB, or 1 for single-channel layers (the segmentation head), which
// would otherwise spend fifteen of sixteen multiplies on zero padding.  Per channel the sums are the same.
template <int K, int COB = F_COB>  // K: compile-time kernel size, 0 = runtime
__global__ __launch_bounds__(256) void conv_fwd_f32_kernel(
    const float *__restrict__ Wt, const float *__restrict__ X, float *__restrict__ Y, int Ci, int Co,
    int Ho, int Wo, int Hi, int Wi, int pad, int stride, int k_rt, int tilesW) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = K ? K : k_rt;
  const int kk = k * k;
  const int PH = (TILE - 1) * stride + k;
  const int PWp = PH + 1;
  float *patch = smem;                                                 // [CIB][PH][PWp]
  float *wts = smem + ((F_CIB * PH * PWp + 3) & ~3);                   // [CIB][kk][COB]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int th0 = (blockIdx.x / tilesW) * TILE, tw0 = (blockIdx.x % tilesW) * TILE;
  const int co0 = blockIdx.y * COB;
  const i64 n = blockIdx.z;
  const int ih0 = th0 * stride - pad, iw0 = tw0 * stride - pad;
  float acc[COB];
#pragma unroll
  for (int i = 0; i < COB; ++i) acc[i] = 0.f;

  for (int ci0 = 0; ci0 < Ci; ci0 += F_CIB) {
    __syncthreads();
    for (int e = tid; e < F_CIB * PH * PH; e += 256) {
      const int ci = e / (PH * PH), r = e - ci * PH * PH, ph = r / PH, pw = r - ph * PH;
      const int ih = ih0 + ph, iw = iw0 + pw;
      float v = 0.f;
      if (ci0 + ci < Ci && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi)
        v = X[((n * Ci + ci0 + ci) * Hi + ih) * Wi + iw];
      patch[(ci * PH + ph) * PWp + pw] = v;
    }
    for (int e = tid; e < F_CIB * kk * COB; e += 256) {
      const int co = e % COB, r = e / COB, t = r % kk, ci = r / kk;
      float v = 0.f;
      if (ci0 + ci < Ci && co0 + co < Co) v = Wt[((i64)(ci0 + ci) * Co + co0 + co) * kk + t];
      wts[e] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int ci = 0; ci < F_CIB; ++ci) {
      const float *prow = patch + (ci * PH + ty * stride) * PWp + tx * stride;
      const float *wrow = wts + ci * kk * COB;
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
          if constexpr (COB == 16) {
            fma4x4(acc, prow[kh * PWp + kw], wrow + (kh * k + kw) * COB);
          } else {
#pragma unroll
            for (int c = 0; c < COB; ++c) acc[c] = fmaf(prow[kh * PWp + kw], wrow[(kh * k + kw) * COB + c], acc[c]);
          }
        }
    }
  }
  const int ho = th0 + ty, wo = tw0 + tx;
  if (ho < Ho && wo < Wo) {
    const float *bias = Wt + (i64)Ci * Co * kk;
#pragma unroll
    for (int co = 0; co < COB; ++co)
      if (co0 + co < Co) Y[((n * Co + co0 + co) * Ho + ho) * Wo + wo] = acc[co] + bias[co0 + co];
  }
}

// ------------------------------------------------- forward on the matrix cores --
// 3x3 / pad 1 / stride 1 as an implicit GEMM on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: every
// product and sum is an exact fp32 fma, like the vector kernel above, at the matrix pipe's rate):
//   D[co][pixel] += W[co][(ci, tap)] * X[(ci, tap)][pixel],  K = (input channel, filter tap), two K per MFMA.
// A workgroup (4 waves) owns 256 output pixels of one image x (32 MB) output channels; a wave owns two
// 32-pixel blocks (TW = 32: two tile rows of 32 pixels; TW = 16: four tile rows of 16) and MB 32-channel
// blocks.  Input channels go through LDS eight at a time: the halo patch [ci][rows + 2][pitch] and the weight
// slice [ci][tap][co].  Operand fragments are single floats per lane: lane l holds K index l / 32 and
// row / column l % 32, so a fragment read is 32 consecutive floats per K index (the patch pitch puts the
// second pixel row of a TW = 16 block on the other half of the banks).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int M_CIB = 8;

template <int TW, int MB>
__global__ __launch_bounds__(256) void conv3x3_fwd_mfma_f32_kernel(const float *__restrict__ Wt,
                                                                   const float *__restrict__ X,
                                                                   float *__restrict__ Y, int Ci, int Co, int H,
                                                                   int W, int tilesW, int splits, int ci_per_split,
                                                                   int Hi, int Wi, int pad) {
  constexpr int TH = 256 / TW;                  // tile rows
  constexpr int PR = TH + 2, PC = TW + 2;       // patch rows / columns
  constexpr int PITCH = TW == 32 ? 36 : 48;     // floats; 48 = 16 mod 32
  constexpr int COB = 32 * MB;
  constexpr int RPB = 32 / TW;                  // tile rows per 32-pixel block
  __shared__ float patch[M_CIB][PR][PITCH];
  __shared__ float wts[M_CIB][9][COB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int th0 = (blockIdx.x / tilesW) * TH, tw0 = (blockIdx.x % tilesW) * TW;
  const int co0 = blockIdx.y * COB;
  // splits > 1: blockIdx.z = image * splits + split; a split covers ci_per_split input channels and writes its
  // partial sums (no bias) to Y[split][image][...], which conv_split_reduce_f32_kernel adds up in a fixed order
  const int split = blockIdx.z % splits;
  const i64 n = blockIdx.z / splits, n_out = (i64)split * (gridDim.z / splits) + n;
  const int ci_begin = split * ci_per_split, ci_end = min(Ci, ci_begin + ci_per_split);
  const int kl = lane >> 5, l32 = lane & 31;
  const int brow = l32 / TW, bcol = l32 % TW;   // pixel of this lane inside a 32-pixel block
  f32x16_t acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][b][r] = 0.f;

  // Staging plan of this thread, the same for every stage: which patch and weight elements it fetches (offsets
  // relative to the stage's first input channel; -1 = outside the image / past C_out: zero) and where they go in
  // LDS.  A stage's loads are all issued together, and one stage ahead: they are in flight while the matrix cores
  // work on the stage before (a load -> store loop per element cost a memory round trip per element).
  constexpr int NPE = M_CIB * PR * PC, NWE = M_CIB * COB * 9;
  constexpr int NP = (NPE + 255) / 256, NW = (NWE + 255) / 256;
  int poff[NP], woff[NW];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = tid + 256 * i, ci = e / (PR * PC), r = e - ci * PR * PC, ph = r / PC, pw = r - ph * PC;
    const int ih = th0 - pad + ph, iw = tw0 - pad + pw;  // pad 1: same size; pad 0: the unpadded convolution
    poff[i] = (e < NPE && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? (ci * Hi + ih) * Wi + iw : -1;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {  // global order (ci, co, tap): contiguous runs of COB * 9 floats
    const int e = tid + 256 * i, ci = e / (COB * 9), r = e - ci * COB * 9, co = r / 9;
    woff[i] = (e < NWE && co0 + co < Co) ? ci * Co * 9 + r : -1;
  }
  float pv[NP], wv[NW];
  auto load_stage = [&](int ci0) {
    const float *xs = X + (n * Ci + ci0) * (i64)Hi * Wi;
    const float *ws = Wt + ((i64)ci0 * Co + co0) * 9;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int ci = (tid + 256 * i) / (PR * PC);
      pv[i] = (poff[i] >= 0 && ci0 + ci < ci_end) ? xs[poff[i]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int ci = (tid + 256 * i) / (COB * 9);
      wv[i] = (woff[i] >= 0 && ci0 + ci < ci_end) ? ws[woff[i]] : 0.f;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 256 * i, ci = e / (PR * PC), r = e - ci * PR * PC, ph = r / PC, pw = r - ph * PC;
      if (NPE % 256 == 0 || e < NPE) patch[ci][ph][pw] = pv[i];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i, ci = e / (COB * 9), r = e - ci * COB * 9, co = r / 9, t = r - co * 9;
      if (NWE % 256 == 0 || e < NWE) wts[ci][t][co] = wv[i];
    }
  };
  load_stage(ci_begin);
  for (int ci0 = ci_begin; ci0 < ci_end; ci0 += M_CIB) {
    __syncthreads();  // the previous stage's fragment reads are done
    store_stage();
    __syncthreads();
    if (ci0 + M_CIB < ci_end) load_stage(ci0 + M_CIB);
#pragma unroll
    for (int pair = 0; pair < M_CIB / 2; ++pair) {
      const int ci = 2 * pair + kl;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap % 3;
        float af[MB], bf[2];
#pragma unroll
        for (int m = 0; m < MB; ++m) af[m] = wts[ci][tap][m * 32 + l32];
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = patch[ci][(2 * wave + b) * RPB + brow + kh][bcol + kw];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[b], acc[m][b], 0, 0, 0);
      }
    }
  }
  const float *bias = Wt + (i64)Ci * Co * 9;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int y = th0 + (2 * wave + b) * RPB + brow, x = tw0 + bcol;
    if (y >= H || x >= W) continue;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // register r of a 32x32 block holds row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (co < Co) Y[((n_out * Co + co) * H + y) * W + x] = acc[m][b][r] + (splits > 1 ? 0.f : bias[co]);
      }
  }
}

// Y[i] = bias[channel of i] + sum over splits (ascending) of partial[split][i]
__global__ __launch_bounds__(256) void conv_split_reduce_f32_kernel(const float *__restrict__ partial,
                                                                    const float *__restrict__ bias,
                                                                    float *__restrict__ Y, i64 total, i64 HW, int Co,
                                                                    int splits) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    float s = bias[(i / HW) % Co];
    for (int k = 0; k < splits; ++k) s += partial[k * total + i];
    Y[i] = s;
  }
}

// ---------------------------------------------------------- backward data --
// dX[n,ci,hi,wi] = sum_co sum_kh,kw W[ci,co,kh,kw] * dY[n,co,(hi+p-kh)/s,(wi+p-kw)/s]
// over taps where the division is exact and the result is in range
// (convolution.cu:198-200).  Block: one image, 16x16 input tile, CIB=16 input
// channels; loops over output channels in chunks of COC=8 staging the dY patch.
constexpr int D_CIB = 16, D_COC = 8;

__global__ __launch_bounds__(256) void conv_bwd_data_f32_kernel(
    const float *__restrict__ Wt, const float *__restrict__ dY, float *__restrict__ dX, int Ci, int Co,
    int Ho, int Wo, int Hi, int Wi, int pad, int stride, int k, int tilesW, int PH) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int kk = k * k;
  const int PWp = PH + 1;
  float *patch = smem;                                     // [COC][PH][PWp]
  float *wts = smem + ((D_COC * PH * PWp + 3) & ~3);       // [COC][kk][CIB]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int h0 = (blockIdx.x / tilesW) * TILE, w0 = (blockIdx.x % tilesW) * TILE;
  const int ci0 = blockIdx.y * D_CIB;
  const i64 n = blockIdx.z;
  // lowest output row/col any tap of this tile can touch (floor division)
  const int a = h0 + pad - (k - 1), b = w0 + pad - (k - 1);
  const int ho_lo = (a >= 0) ? a / stride : -((-a + stride - 1) / stride);
  const int wo_lo = (b >= 0) ? b / stride : -((-b + stride - 1) / stride);
  const int hi = h0 + ty, wi = w0 + tx;
  float acc[D_CIB];
#pragma unroll
  for (int i = 0; i < D_CIB; ++i) acc[i] = 0.f;

  for (int co0 = 0; co0 < Co; co0 += D_COC) {
    __syncthreads();
    for (int e = tid; e < D_COC * PH * PH; e += 256) {
      const int co = e / (PH * PH), r = e - co * PH * PH, ph = r / PH, pw = r - ph * PH;
      const int ho = ho_lo + ph, wo = wo_lo + pw;
      float v = 0.f;
      if (co0 + co < Co && ho >= 0 && ho < Ho && wo >= 0 && wo < Wo)
        v = dY[((n * Co + co0 + co) * Ho + ho) * Wo + wo];
      patch[(co * PH + ph) * PWp + pw] = v;
    }
    for (int e = tid; e < D_COC * kk * D_CIB; e += 256) {
      const int ci = e % D_CIB, r = e / D_CIB, t = r % kk, co = r / kk;
      float v = 0.f;
      if (ci0 + ci < Ci && co0 + co < Co) v = Wt[((i64)(ci0 + ci) * Co + co0 + co) * kk + t];
      wts[e] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int co = 0; co < D_COC; ++co) {
      for (int kh = 0; kh < k; ++kh) {
        const int th = hi + pad - kh;
        if (th % stride) continue;
        const int ph = th / stride - ho_lo;
        for (int kw = 0; kw < k; ++kw) {
          const int tw = wi + pad - kw;
          if (tw % stride) continue;
          const int pw = tw / stride - wo_lo;
          fma4x4(acc, patch[(co * PH + ph) * PWp + pw], wts + (co * kk + kh * k + kw) * D_CIB);
        }
      }
    }
  }
  if (hi < Hi && wi < Wi) {
#pragma unroll
    for (int ci = 0; ci < D_CIB; ++ci)
      if (ci0 + ci < Ci) dX[((n * Ci + ci0 + ci) * Hi + hi) * Wi + wi] = acc[ci];
  }
}

// -------------------------------------------------------- backward weights --
// dW[ci,co,kh,kw] = sum_{n,ho,wo} dY[n,co,ho,wo] * X[n,ci,ho*s-p+kh,wo*s-p+kw];
// db[co] = sum dY.  Block (split, co-group, ci): each thread walks pixels of
// the split's rows with COB*K*K private accumulators, then the block reduces
// them (wave shuffles + LDS) into partial[split][...]; reduce_partials_kernel
// sums the splits in a fixed order.
template <int K, int COB>
__global__ __launch_bounds__(256) void conv_bwd_weight_f32_kernel(
    const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ partial, int N, int Ci,
    int Co, int Ho, int Wo, int Hi, int Wi, int pad, int stride, int rows_per_split, i64 nW) {
  constexpr int KK = K * K, NACC = COB * KK + COB;
  __shared__ float red[4 * NACC];
  const int split = blockIdx.x, co0 = blockIdx.y * COB, ci = blockIdx.z;
  const int rows = N * Ho;
  const int r0 = split * rows_per_split;
  const int r1 = min(rows, r0 + rows_per_split);
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const i64 npix = (i64)max(0, r1 - r0) * Wo;
  for (i64 p = threadIdx.x; p < npix; p += 256) {
    const int r = r0 + (int)(p / Wo), wo = (int)(p % Wo);
    const i64 n = r / Ho;
    const int ho = r % Ho;
    float g[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) {
      g[c] = (co0 + c < Co) ? dY[((n * Co + co0 + c) * Ho + ho) * Wo + wo] : 0.f;
      acc[COB * KK + c] += g[c];
    }
    const float *xp = X + (n * Ci + ci) * (i64)Hi * Wi;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int ih = ho * stride - pad + kh;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int iw = wo * stride - pad + kw;
        const float xv = (ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? xp[(i64)ih * Wi + iw] : 0.f;
#pragma unroll
        for (int c = 0; c < COB; ++c) acc[c * KK + kh * K + kw] = fmaf(g[c], xv, acc[c * KK + kh * K + kw]);
      }
    }
  }
  // wave totals -> LDS, then thread i adds the four wave totals of value i
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = wave_sum(acc[i]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[(threadIdx.x >> 6) * NACC + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < NACC) {
    const int i = threadIdx.x;
    const float v = (red[i] + red[NACC + i]) + (red[2 * NACC + i] + red[3 * NACC + i]);
    if (i < COB * KK) {
      const int c = i / KK, t = i % KK;
      if (co0 + c < Co) partial[(i64)split * (nW + Co) + ((i64)ci * Co + co0 + c) * KK + t] = v;
    } else if (ci == 0) {
      const int c = i - COB * KK;
      if (co0 + c < Co) partial[(i64)split * (nW + Co) + nW + co0 + c] = v;
    }
  }
}

// ------------------------------------------- backward weights on the matrix cores --
// 3x3 / stride 1 / pad 1:  dW[ci][co][tap] = sum over pixels of dY[co][pixel] * X[ci][pixel + off(tap)] as nine
// GEMMs D[co][ci] += A[co][pixel] * B[pixel][ci] on v_mfma_f32_32x32x2_f32 (fp32 products and sums, like the
// vector kernel).  A workgroup (4 waves) owns 64 output channels x 64 input channels -- one 32 x 32 block per
// wave, nine taps = 144 accumulator registers -- and walks its share of 64-pixel regions (TW = 32: 2 rows x 32
// columns; TW = 16: 4 rows x 16): the region's dY [64][64] and the halo patch of X [64][rows + 2][TW + 2] go
// through LDS, the next region's loads in flight while the matrix cores work on this one.  The two K indices of
// an MFMA are pixel j of the region's first and second half (lane half = region half), so a lane's operands of
// four consecutive steps are consecutive floats: one 16-byte LDS read of dY per four steps, and per filter row two
// 16-byte reads of X whose eight floats hold the sliding windows of all three filter columns.
// Regions are split over gridDim.x workgroups; partial[split] is added up by reduce_partials_kernel in a fixed
// order.  db[co] = sum dY comes from the staged dY tile (workgroups of the first input-channel block).
template <int TW>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_mfma_f32_kernel(const float *__restrict__ dY,
                                                                        const float *__restrict__ X,
                                                                        float *__restrict__ partial, int N, int Ci,
                                                                        int Co, int H, int W, int regions_per_split,
                                                                        i64 nW) {
  constexpr int TR = 64 / TW, PR = TR + 2, PCU = TW + 2;  // region rows, patch rows, patch columns in use
  constexpr int PCP = TW + 8;                             // patch row pitch: window reads reach column TW + 3
  constexpr int CIP = PR * PCP + 4;                       // floats per input channel (164 / 148: = 4, 20 mod 32)
  constexpr int DYP = 68;                                 // floats per output channel of the dY tile
  constexpr int NXE = 64 * PR * PCU, NX = (NXE + 255) / 256;
  __shared__ __attribute__((aligned(16))) float dys[64 * DYP];
  __shared__ __attribute__((aligned(16))) float xs[64 * CIP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 5, l32 = lane & 31;
  const int mo = wave >> 1, no = wave & 1;
  const int split = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int tilesX = (W + TW - 1) / TW, tilesY = (H + TR - 1) / TR;
  const int regions = N * tilesY * tilesX;
  const int r_begin = split * regions_per_split, r_end = min(regions, r_begin + regions_per_split);

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  float dv[16], xv[NX];
  auto load_region = [&](int r) {
    const int tx = r % tilesX, q = r / tilesX, ty = q % tilesY;
    const i64 n = q / tilesY;
    const int y0 = ty * TR, x0 = tx * TW;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, co = e >> 6, px = e & 63, y = y0 + px / TW, x = x0 + px % TW;
      dv[i] = (co0 + co < Co && y < H && x < W) ? dY[((n * Co + co0 + co) * H + y) * W + x] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, ci = e / (PR * PCU), rr = e - ci * (PR * PCU), prow = rr / PCU, pcol = rr - prow * PCU;
      const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
      xv[i] = (e < NXE && ci0 + ci < Ci && iy >= 0 && iy < H && ix >= 0 && ix < W)
                  ? X[((n * Ci + ci0 + ci) * H + iy) * W + ix]
                  : 0.f;
    }
  };
  auto store_region = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i;
      dys[(e >> 6) * DYP + (e & 63)] = dv[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, ci = e / (PR * PCU), rr = e - ci * (PR * PCU), prow = rr / PCU, pcol = rr - prow * PCU;
      if (NXE % 256 == 0 || e < NXE) xs[ci * CIP + prow * PCP + pcol] = xv[i];
    }
  };

  const float *a_lane = dys + (32 * mo + l32) * DYP + kl * 32;
  const float *b_lane = xs + (32 * no + l32) * CIP;
  if (r_begin < r_end) load_region(r_begin);
  for (int r = r_begin; r < r_end; ++r) {
    __syncthreads();  // the previous region's fragment reads are done
    store_region();
    __syncthreads();
    if (r + 1 < r_end) load_region(r + 1);
    if (blockIdx.z == 0 && tid < 64) {  // bias gradient: this channel's 64 staged values, in order
      const float4 *row = reinterpret_cast<const float4 *>(dys + tid * DYP);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = row[i];
        bsum += v.x; bsum += v.y; bsum += v.z; bsum += v.w;
      }
    }
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {  // four MFMA steps per trip: region pixels 4 jq .. 4 jq + 3 of this lane's half
      const float4 a4 = *reinterpret_cast<const float4 *>(a_lane + jq * 4);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      const int prow0 = TW == 32 ? kl : 2 * kl + (jq >> 2), col0 = TW == 32 ? jq * 4 : (jq & 3) * 4;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float4 *wp = reinterpret_cast<const float4 *>(b_lane + (prow0 + kh) * PCP + col0);
        const float4 w0 = wp[0], w1 = wp[1];
        const float win[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};  // patch columns col0 .. col0 + 7
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], win[st + kw], acc[kh * 3 + kw], 0, 0, 0);
      }
    }
  }
  float *out = partial + (i64)split * (nW + Co);
  const int ci = ci0 + 32 * no + l32;
  if (ci < Ci) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // register r holds row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of column lane % 32
        const int co = co0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (co < Co) out[((i64)ci * Co + co) * 9 + t] = acc[t][r];
      }
  }
  if (blockIdx.z == 0 && tid < 64 && co0 + tid < Co) out[nW + co0 + tid] = bsum;
}

// out[i] = sum_s partial[s*stride_s + i], i < n  (double accumulation)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial,
                                                              float *__restrict__ out, i64 n, int splits) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    double s = 0;
    for (int k = 0; k < splits; ++k) s += partial[(i64)k * n + i];
    out[i] = (float)s;
  }
}

// ------------------------------------------------------------- upsample ----
// Transposed convolution with kernel = stride = s (upsample.cu:9-59):
// Y[n,co,h,w] = b[co] + sum_ci W[ci,co,h%s,w%s] * X[n,ci,h/s,w/s].
// A chunk of U_CIC input channels per trip: its weights go to LDS, and a thread requests all of the chunk's inputs
// before the first multiply (one memory round trip per chunk instead of one per channel).
// (64 channels for the usual scales, 16 where 64 channels' weights would not fit the LDS budget.)
constexpr int U_COB = 8;

template <int U_CIC>
__global__ __launch_bounds__(256) void upsample_fwd_f32_kernel(
    const float *__restrict__ X, const float *__restrict__ Wt, float *__restrict__ Y, int Ci, int Co,
    int Hi, int Wi, int s) {
  extern __shared__ __attribute__((aligned(16))) float wts[];  // [CIC][ss][COB]
  const int ss = s * s, Ho = Hi * s, Wo = Wi * s;
  const i64 pix = (i64)blockIdx.x * 256 + threadIdx.x;  // output pixel within the image
  const int co0 = blockIdx.y * U_COB;
  const i64 n = blockIdx.z;
  const bool live = pix < (i64)Ho * Wo;
  const int ho = live ? (int)(pix / Wo) : 0, wo = live ? (int)(pix % Wo) : 0;
  const int pp = (ho % s) * s + (wo % s);
  const float *xp = X + n * Ci * (i64)Hi * Wi + (i64)(ho / s) * Wi + (wo / s);
  float acc[U_COB];
#pragma unroll
  for (int i = 0; i < U_COB; ++i) acc[i] = 0.f;
  for (int ci0 = 0; ci0 < Ci; ci0 += U_CIC) {
    __syncthreads();
#pragma unroll 8
    for (int e = threadIdx.x; e < U_CIC * ss * U_COB; e += 256) {
      const int co = e % U_COB, r = e / U_COB, t = r % ss, ci = r / ss;
      float v = 0.f;
      if (ci0 + ci < Ci && co0 + co < Co) v = Wt[((i64)(ci0 + ci) * Co + co0 + co) * ss + t];
      wts[e] = v;
    }
    __syncthreads();
    // channels past Ci have zero weights in LDS; their inputs are taken as zero without touching memory
    float xv[U_CIC];
#pragma unroll
    for (int ci = 0; ci < U_CIC; ++ci) xv[ci] = (live && ci0 + ci < Ci) ? xp[(i64)(ci0 + ci) * Hi * Wi] : 0.f;
#pragma unroll
    for (int ci = 0; ci < U_CIC; ++ci) {
      const float4 *w4 = reinterpret_cast<const float4 *>(wts + (ci * ss + pp) * U_COB);
      const float4 a = w4[0], b = w4[1];
      acc[0] = fmaf(xv[ci], a.x, acc[0]); acc[1] = fmaf(xv[ci], a.y, acc[1]);
      acc[2] = fmaf(xv[ci], a.z, acc[2]); acc[3] = fmaf(xv[ci], a.w, acc[3]);
      acc[4] = fmaf(xv[ci], b.x, acc[4]); acc[5] = fmaf(xv[ci], b.y, acc[5]);
      acc[6] = fmaf(xv[ci], b.z, acc[6]); acc[7] = fmaf(xv[ci], b.w, acc[7]);
    }
  }
  if (live) {
    const float *bias = Wt + (i64)Ci * Co * ss;
#pragma unroll
    for (int co = 0; co < U_COB; ++co)
      if (co0 + co < Co) Y[((n * Co + co0 + co) * Ho + ho) * Wo + wo] = acc[co] + bias[co0 + co];
  }
}

// Forward on the matrix cores for scale 2: per output phase t = 2i + j a GEMM
//   Y[co][2y + i][2x + j] = b[co] + sum_ci W[ci][co][t] * X[ci][y][x]   on v_mfma_f32_32x32x2_f32
// (A = weights, rows = output channels; B = input pixels; two input channels per MFMA).  A workgroup owns 64 output
// channels x 64 input pixels (TW = 32: 2 rows x 32 columns; TW = 16: 4 rows x 16) and all four phases -- a wave one
// 32 x 32 block, 64 accumulator registers -- with 32 input channels per stage through LDS, the next stage's loads
// in flight.  The two column phases of an output row leave as one 8-byte store per lane.  splits > 1: the input
// channels are divided over blockIdx.z = image * splits + split, partial sums go to Y[split][...] without the bias
// and conv_split_reduce_f32_kernel adds them in a fixed order (launches of few workgroups: single images).
constexpr int UM_CIB = 32;
template <int TW>
__global__ __launch_bounds__(256) void upsample2_fwd_mfma_f32_kernel(const float *__restrict__ X,
                                                                     const float *__restrict__ Wt,
                                                                     float *__restrict__ Y, int Ci, int Co, int Hi,
                                                                     int Wi, int tilesX, int splits, int ci_per_split) {
  constexpr int TR = 64 / TW;
  constexpr int WP = 4 * 64 + 8;  // floats per input channel of the weight stage [ci][t][64 co] (+8: = 8 mod 32)
  constexpr int XP = 64 + 8;      // floats per input channel of the pixel stage
  __shared__ float ws[UM_CIB * WP];
  __shared__ float xs[UM_CIB * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 5, l32 = lane & 31;
  const int mo = wave >> 1, no = wave & 1;
  const int y0 = (blockIdx.x / tilesX) * TR, x0 = (blockIdx.x % tilesX) * TW;
  const int co0 = blockIdx.y * 64;
  const int split = blockIdx.z % splits;
  const i64 n = blockIdx.z / splits, n_out = (i64)split * (gridDim.z / splits) + n;
  const int ci_begin = split * ci_per_split, ci_end = min(Ci, ci_begin + ci_per_split);
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  f32x16_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging plan: weights as float4 (the four phases of one (ci, co) are contiguous), pixels as floats
  constexpr int NWV = UM_CIB * 64 / 256, NXV = UM_CIB * 64 / 256;  // 8 + 8 per thread
  float4 wv[NWV];
  float xv[NXV];
  auto load_stage = [&](int ci0) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int e = tid + 256 * i, ci = e >> 6, co = e & 63;
      wv[i] = (ci0 + ci < ci_end && co0 + co < Co)
                  ? *reinterpret_cast<const float4 *>(Wt + ((i64)(ci0 + ci) * Co + co0 + co) * 4)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int e = tid + 256 * i, ci = e >> 6, px = e & 63, y = y0 + px / TW, x = x0 + px % TW;
      xv[i] = (ci0 + ci < ci_end && y < Hi && x < Wi) ? X[((n * Ci + ci0 + ci) * Hi + y) * Wi + x] : 0.f;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int e = tid + 256 * i, ci = e >> 6, co = e & 63;
      float *w = ws + ci * WP + co;
      w[0] = wv[i].x; w[64] = wv[i].y; w[128] = wv[i].z; w[192] = wv[i].w;
    }
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int e = tid + 256 * i;
      xs[(e >> 6) * XP + (e & 63)] = xv[i];
    }
  };
  const float *a_lane = ws + kl * WP + 32 * mo + l32;
  const float *b_lane = xs + kl * XP + 32 * no + l32;
  load_stage(ci_begin);
  for (int ci0 = ci_begin; ci0 < ci_end; ci0 += UM_CIB) {
    __syncthreads();  // the previous stage's fragment reads are done
    store_stage();
    __syncthreads();
    if (ci0 + UM_CIB < ci_end) load_stage(ci0 + UM_CIB);
#pragma unroll
    for (int kp = 0; kp < UM_CIB / 2; ++kp) {  // input channels 2 kp (lanes 0-31) and 2 kp + 1 (lanes 32-63)
      const float b = b_lane[2 * kp * XP];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lane[2 * kp * WP + t * 64], b, acc[t], 0, 0, 0);
    }
  }
  const int px = 32 * no + l32, y = y0 + px / TW, x = x0 + px % TW;
  if (y < Hi && x < Wi) {
    const float *bias = Wt + (i64)Ci * Co * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * kl;
      if (co >= Co) continue;
      const float bv = splits > 1 ? 0.f : bias[co];
      float *o = Y + ((n_out * Co + co) * Ho + 2 * y) * Wo + 2 * x;
      *reinterpret_cast<float2 *>(o) = make_float2(acc[0][r] + bv, acc[1][r] + bv);
      *reinterpret_cast<float2 *>(o + Wo) = make_float2(acc[2][r] + bv, acc[3][r] + bv);
    }
  }
}

// dX[n,ci,hi,wi] = sum_co sum_{i,j<s} W[ci,co,i,j] * dY[n,co,hi*s+i,wi*s+j]
constexpr int UD_CIB = 8, UD_COC = 8;

__global__ __launch_bounds__(256) void upsample_bwd_data_f32_kernel(
    const float *__restrict__ dY, const float *__restrict__ Wt, float *__restrict__ dX, int Ci, int Co,
    int Hi, int Wi, int s) {
  extern __shared__ __attribute__((aligned(16))) float wts[];  // [COC][ss][CIB]
  const int ss = s * s, Ho = Hi * s, Wo = Wi * s;
  const i64 pix = (i64)blockIdx.x * 256 + threadIdx.x;  // input pixel within the image
  const int ci0 = blockIdx.y * UD_CIB;
  const i64 n = blockIdx.z;
  const bool live = pix < (i64)Hi * Wi;
  const int hi = live ? (int)(pix / Wi) : 0, wi = live ? (int)(pix % Wi) : 0;
  const float *gp = dY + n * Co * (i64)Ho * Wo + (i64)(hi * s) * Wo + wi * s;
  float acc[UD_CIB];
#pragma unroll
  for (int i = 0; i < UD_CIB; ++i) acc[i] = 0.f;
  for (int co0 = 0; co0 < Co; co0 += UD_COC) {
    __syncthreads();
    for (int e = threadIdx.x; e < UD_COC * ss * UD_CIB; e += 256) {
      const int ci = e % UD_CIB, r = e / UD_CIB, t = r % ss, co = r / ss;
      float v = 0.f;
      if (ci0 + ci < Ci && co0 + co < Co) v = Wt[((i64)(ci0 + ci) * Co + co0 + co) * ss + t];
      wts[e] = v;
    }
    __syncthreads();
    const int cmax = min(UD_COC, Co - co0);
    if (live)
      for (int co = 0; co < cmax; ++co)
        for (int i = 0; i < s; ++i)
          for (int j = 0; j < s; ++j) {
            const float g = gp[(i64)(co0 + co) * Ho * Wo + (i64)i * Wo + j];
            const float4 *w4 = reinterpret_cast<const float4 *>(wts + (co * ss + i * s + j) * UD_CIB);
            const float4 a = w4[0], b = w4[1];
            acc[0] = fmaf(g, a.x, acc[0]); acc[1] = fmaf(g, a.y, acc[1]);
            acc[2] = fmaf(g, a.z, acc[2]); acc[3] = fmaf(g, a.w, acc[3]);
            acc[4] = fmaf(g, b.x, acc[4]); acc[5] = fmaf(g, b.y, acc[5]);
            acc[6] = fmaf(g, b.z, acc[6]); acc[7] = fmaf(g, b.w, acc[7]);
          }
  }
  if (live)
#pragma unroll
    for (int ci = 0; ci < UD_CIB; ++ci)
      if (ci0 + ci < Ci) dX[((n * Ci + ci0 + ci) * Hi + hi) * Wi + wi] = acc[ci];
}

// The data gradient on the matrix cores for scale 2:
//   dX[ci][y][x] = sum_co sum_{i,j} W[ci][co][2i + j] * dY[co][2y + i][2x + j]
// as one GEMM over K = (co, i, j); the two K indices of an MFMA are the column phases j = 0, 1 of one (co, i), so
// the B operands of a wave are 64 consecutive floats of a dY row.  A workgroup owns 64 input channels x 64 input
// pixels (a wave one 32 x 32 block), 16 output channels per stage through LDS, the next stage's loads in flight.
constexpr int UDM_COB = 16;
template <int TW>
__global__ __launch_bounds__(256) void upsample2_dgrad_mfma_f32_kernel(const float *__restrict__ dY,
                                                                       const float *__restrict__ Wt,
                                                                       float *__restrict__ dX, int Ci, int Co, int Hi,
                                                                       int Wi, int tilesX) {
  constexpr int TR = 64 / TW, OR = 2 * TR, OC = 2 * TW;
  constexpr int WP = 4 * 64 + 8, DP = OR * OC + 8;  // floats per output channel of a stage (= 8 mod 32)
  __shared__ float wl[UDM_COB * WP];   // [co][t][64 ci]
  __shared__ float dl[UDM_COB * DP];   // [co][2 TR rows][2 TW columns] of dY
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 5, l32 = lane & 31;
  const int mo = wave >> 1, no = wave & 1;
  const int y0 = (blockIdx.x / tilesX) * TR, x0 = (blockIdx.x % tilesX) * TW;
  const int ci0 = blockIdx.y * 64;
  const i64 n = blockIdx.z;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 wv[4];
  float dv[16];
  auto load_stage = [&](int cob) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, ci = e >> 4, co = e & 15;
      wv[i] = (ci0 + ci < Ci && cob + co < Co) ? *reinterpret_cast<const float4 *>(Wt + ((i64)(ci0 + ci) * Co + cob + co) * 4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, co = e / (OR * OC), rr = e - co * (OR * OC), oy = 2 * y0 + rr / OC, ox = 2 * x0 + rr % OC;
      dv[i] = (cob + co < Co && oy < Ho && ox < Wo) ? dY[((n * Co + cob + co) * Ho + oy) * Wo + ox] : 0.f;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, ci = e >> 4, co = e & 15;
      float *w = wl + co * WP + ci;
      w[0] = wv[i].x; w[64] = wv[i].y; w[128] = wv[i].z; w[192] = wv[i].w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, co = e / (OR * OC);
      dl[co * DP + (e - co * (OR * OC))] = dv[i];
    }
  };
  const int px = 32 * no + l32, row = px / TW, col = px % TW;
  const float *a_lane = wl + kl * 64 + 32 * mo + l32;          // + co * WP + i * 128
  const float *b_lane = dl + 2 * row * OC + 2 * col + kl;      // + co * DP + i * OC
  load_stage(0);
  for (int cob = 0; cob < Co; cob += UDM_COB) {
    __syncthreads();  // the previous stage's fragment reads are done
    store_stage();
    __syncthreads();
    if (cob + UDM_COB < Co) load_stage(cob + UDM_COB);
#pragma unroll
    for (int co = 0; co < UDM_COB; ++co)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lane[co * WP + i * 128], b_lane[co * DP + i * OC], acc, 0, 0, 0);
  }
  const int y = y0 + row, x = x0 + col;
  if (y < Hi && x < Wi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * kl;
      if (ci < Ci) dX[((n * Ci + ci) * Hi + y) * Wi + x] = acc[r];
    }
  }
}

// dW[ci,co,i,j] = sum_{n,hi,wi} dY[n,co,hi*s+i,wi*s+j] * X[n,ci,hi,wi]; db[co] = sum dY.
template <int S, int COB>
__global__ __launch_bounds__(256) void upsample_bwd_weight_f32_kernel(
    const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ partial, int N, int Ci,
    int Co, int Hi, int Wi, int rows_per_split, i64 nW) {
  constexpr int SS = S * S, NACC = COB * SS + COB;
  __shared__ float red[4 * NACC];
  const int split = blockIdx.x, co0 = blockIdx.y * COB, ci = blockIdx.z;
  const int Ho = Hi * S, Wo = Wi * S;
  const int rows = N * Hi;
  const int r0 = split * rows_per_split;
  const int r1 = min(rows, r0 + rows_per_split);
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const i64 npix = (i64)max(0, r1 - r0) * Wi;
  for (i64 p = threadIdx.x; p < npix; p += 256) {
    const int r = r0 + (int)(p / Wi), wi = (int)(p % Wi);
    const i64 n = r / Hi;
    const int hi = r % Hi;
    const float xv = X[((n * Ci + ci) * Hi + hi) * Wi + wi];
#pragma unroll
    for (int c = 0; c < COB; ++c) {
      if (co0 + c < Co) {
        const float *gp = dY + ((n * Co + co0 + c) * Ho + (i64)hi * S) * Wo + (i64)wi * S;
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const float g = gp[(i64)i * Wo + j];
            acc[c * SS + i * S + j] = fmaf(g, xv, acc[c * SS + i * S + j]);
            acc[COB * SS + c] += g;
          }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = wave_sum(acc[i]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[(threadIdx.x >> 6) * NACC + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < NACC) {
    const int i = threadIdx.x;
    const float v = (red[i] + red[NACC + i]) + (red[2 * NACC + i] + red[3 * NACC + i]);
    if (i < COB * SS) {
      const int c = i / SS, t = i % SS;
      if (co0 + c < Co) partial[(i64)split * (nW + Co) + ((i64)ci * Co + co0 + c) * SS + t] = v;
    } else if (ci == 0) {
      const int c = i - COB * SS;
      if (co0 + c < Co) partial[(i64)split * (nW + Co) + nW + co0 + c] = v;
    }
  }
}

// The same on the matrix cores for scale 2: four GEMMs D_t[co][ci] += dY[co][2y + i][2x + j] * X[ci][y][x], t = 2i + j,
// on v_mfma_f32_32x32x2_f32.  A workgroup owns 64 x 64 channels (a 32 x 32 block and four phases = 64 accumulator
// registers per wave) and walks 32-pixel regions of the input grid (TW = 32: one row x 32 columns; TW = 16: two
// rows x 16): X [64][32] and the matching raw piece of dY [64][2 TR][2 TW] go through LDS one region ahead.  Lane
// half = region half, so four steps of a lane are four consecutive input pixels: one 16-byte read of X, and per
// output row two 16-byte reads of dY whose eight floats interleave the two column phases.
template <int TW>
__global__ __launch_bounds__(256) void upsample2_wgrad_mfma_f32_kernel(const float *__restrict__ dY,
                                                                       const float *__restrict__ X,
                                                                       float *__restrict__ partial, int N, int Ci,
                                                                       int Co, int Hi, int Wi, int regions_per_split,
                                                                       i64 nW) {
  constexpr int TR = 32 / TW, OR = 2 * TR, OC = 2 * TW;  // region rows; rows and columns of its piece of dY
  constexpr int XP = 36, DP = OR * OC + 4;               // floats per channel in LDS (= 4 mod 32)
  __shared__ __attribute__((aligned(16))) float xs[64 * XP];
  __shared__ __attribute__((aligned(16))) float dyr[64 * DP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 5, l32 = lane & 31;
  const int mo = wave >> 1, no = wave & 1;
  const int split = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int tilesX = (Wi + TW - 1) / TW, tilesY = (Hi + TR - 1) / TR;
  const int regions = N * tilesY * tilesX;
  const int r_begin = split * regions_per_split, r_end = min(regions, r_begin + regions_per_split);
  f32x16_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  float xv[8], dv[32];
  auto load_region = [&](int r) {
    const int tx = r % tilesX, q = r / tilesX, ty = q % tilesY;
    const i64 n = q / tilesY;
    const int y0 = ty * TR, x0 = tx * TW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i, ci = e >> 5, px = e & 31, y = y0 + px / TW, x = x0 + px % TW;
      xv[i] = (ci0 + ci < Ci && y < Hi && x < Wi) ? X[((n * Ci + ci0 + ci) * Hi + y) * Wi + x] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int e = tid + 256 * i, co = e / (OR * OC), rr = e - co * (OR * OC), oy = 2 * y0 + rr / OC, ox = 2 * x0 + rr % OC;
      dv[i] = (co0 + co < Co && oy < Ho && ox < Wo) ? dY[((n * Co + co0 + co) * Ho + oy) * Wo + ox] : 0.f;
    }
  };
  auto store_region = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      xs[(e >> 5) * XP + (e & 31)] = xv[i];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int e = tid + 256 * i, co = e / (OR * OC);
      dyr[co * DP + (e - co * (OR * OC))] = dv[i];
    }
  };
  const float *a_lane = dyr + (32 * mo + l32) * DP;
  const float *b_lane = xs + (32 * no + l32) * XP + kl * 16;
  if (r_begin < r_end) load_region(r_begin);
  for (int r = r_begin; r < r_end; ++r) {
    __syncthreads();  // the previous region's fragment reads are done
    store_region();
    __syncthreads();
    if (r + 1 < r_end) load_region(r + 1);
    if (blockIdx.z == 0 && tid < 64) {  // bias gradient: this channel's staged values, in order
      const float4 *row = reinterpret_cast<const float4 *>(dyr + tid * DP);
#pragma unroll
      for (int i = 0; i < OR * OC / 4; ++i) {
        const float4 v = row[i];
        bsum += v.x; bsum += v.y; bsum += v.z; bsum += v.w;
      }
    }
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {  // input pixels 4 jq .. 4 jq + 3 of this lane's half of the region
      const float4 b4 = *reinterpret_cast<const float4 *>(b_lane + jq * 4);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w};
      const int row = TW == 32 ? 0 : kl, col = TW == 32 ? kl * 16 + jq * 4 : jq * 4;  // of the first of the four pixels
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 *gp = reinterpret_cast<const float4 *>(a_lane + (2 * row + i) * OC + 2 * col);
        const float4 g0 = gp[0], g1 = gp[1];
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};  // output columns 2 col .. 2 col + 7
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[2 * st + j], b[st], acc[i * 2 + j], 0, 0, 0);
      }
    }
  }
  float *out = partial + (i64)split * (nW + Co);
  const int ci = ci0 + 32 * no + l32;
  if (ci < Ci) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (co < Co) out[((i64)ci * Co + co) * 4 + t] = acc[t][r];
      }
  }
  if (blockIdx.z == 0 && tid < 64 && co0 + tid < Co) out[nW + co0 + tid] = bsum;
}

int pick_splits(i64 base_blocks, i64 rows, i64 pixels_per_row) {
  i64 s = (2048 + base_blocks - 1) / base_blocks;
  const i64 by_work = (rows * pixels_per_row + 4095) / 4096;
  if (s > by_work) s = by_work;
  if (s > rows) s = rows;
  if (s < 1) s = 1;
  return (int)s;
}

constexpr size_t kMaxDynLds = 64 * 1024;

}  // namespace

// ---- launch plan of the matrix-core 3x3 kernel (forward, and the data gradient as a forward pass over
// flipped, transposed weights) ----
struct MfmaConvPlan {
  bool wide, two;
  int tW, tH, cob, splits, ci_per_split;
  i64 total;  // output elements
};
inline bool mfma_conv_serves(int k, int pad, int stride, int Co, int Ho, int Wo) {
  return k == 3 && pad <= 1 && stride == 1 && Co >= 16 && Wo >= 8 && Ho >= 4;
}
inline MfmaConvPlan plan_mfma_conv(int N, int Ci, int Co, int Ho, int Wo) {
  MfmaConvPlan p;
  // 8 x 32 pixel tiles for wide images, 16 x 16 otherwise; 64 output channels per workgroup, 32 when that is
  // needed to give every CU a workgroup
  p.wide = Wo >= 32;
  p.tW = p.wide ? (Wo + 31) / 32 : (Wo + 15) / 16;
  p.tH = p.wide ? (Ho + 7) / 8 : (Ho + 15) / 16;
  p.two = Co > 32 && (i64)p.tW * p.tH * N * ((Co + 63) / 64) >= 256;
  p.cob = p.two ? (Co + 63) / 64 : (Co + 31) / 32;
  // few workgroups (single images, deep levels): the input channels are split over more workgroups, each
  // writing partial sums that a second kernel adds in a fixed order
  int splits = 1;
  const i64 wgs = (i64)p.tW * p.tH * N * p.cob;
  if (wgs < 192) {
    splits = (int)((384 + wgs - 1) / wgs);
    const int max_splits = Ci / 32 > 0 ? Ci / 32 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 16) splits = 16;
  }
  p.ci_per_split = ((Ci + splits - 1) / splits + M_CIB - 1) / M_CIB * M_CIB;
  p.splits = (Ci + p.ci_per_split - 1) / p.ci_per_split;
  p.total = (i64)N * Co * Ho * Wo;
  return p;
}
// `partial`: room for splits * total floats when the plan splits the input channels (else unused)
int launch_mfma_conv(const MfmaConvPlan &p, const float *weights, const float *inputs, float *activations,
                     float *partial, int N, int Ci, int Co, int Ho, int Wo, int Hi, int Wi, int pad, hipStream_t s) {
  float *dst = p.splits > 1 ? partial : activations;
  dim3 g((unsigned)(p.tW * p.tH), (unsigned)p.cob, (unsigned)(N * p.splits));
  if (p.wide && p.two)
    hipLaunchKernelGGL((conv3x3_fwd_mfma_f32_kernel<32, 2>), g, dim3(256), 0, s, weights, inputs, dst, Ci, Co, Ho, Wo, p.tW, p.splits, p.ci_per_split, Hi, Wi, pad);
  else if (p.wide)
    hipLaunchKernelGGL((conv3x3_fwd_mfma_f32_kernel<32, 1>), g, dim3(256), 0, s, weights, inputs, dst, Ci, Co, Ho, Wo, p.tW, p.splits, p.ci_per_split, Hi, Wi, pad);
  else if (p.two)
    hipLaunchKernelGGL((conv3x3_fwd_mfma_f32_kernel<16, 2>), g, dim3(256), 0, s, weights, inputs, dst, Ci, Co, Ho, Wo, p.tW, p.splits, p.ci_per_split, Hi, Wi, pad);
  else
    hipLaunchKernelGGL((conv3x3_fwd_mfma_f32_kernel<16, 1>), g, dim3(256), 0, s, weights, inputs, dst, Ci, Co, Ho, Wo, p.tW, p.splits, p.ci_per_split, Hi, Wi, pad);
  UNET_LAUNCH_CHECK("conv3x3_fwd_mfma_f32_kernel");
  if (p.splits > 1) {
    hipLaunchKernelGGL(conv_split_reduce_f32_kernel, dim3(ew_grid(p.total)), dim3(256), 0, s, (const float *)dst,
                       weights + (i64)Ci * Co * 9, activations, p.total, (i64)Ho * Wo, Co, p.splits);
    UNET_LAUNCH_CHECK("conv_split_reduce_f32_kernel");
  }
  return UNET_OK;
}

// Weights of the data gradient as a forward pass: out[co][ci][tap] = w[ci][co][8 - tap] (channels transposed,
// filter rotated by 180 degrees), followed by Ci zeros where the forward kernel looks for its bias.
__global__ __launch_bounds__(256) void flip_weights_f32_kernel(const float *__restrict__ w, float *__restrict__ out,
                                                               int Ci, int Co) {
  const i64 nW = (i64)Ci * Co * 9;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nW + Ci; i += (i64)gridDim.x * 256) {
    float v = 0.f;
    if (i < nW) {
      const int t = (int)(i % 9);
      const i64 r = i / 9;
      const int ci = (int)(r % Ci), co = (int)(r / Ci);
      v = w[((i64)ci * Co + co) * 9 + (8 - t)];
    }
    out[i] = v;
  }
}

extern "C" {

int unet_convolve_f32(const float *weights, const float *inputs, float *activations, int64_t act_len,
                      int N, int k, int Ci, int Co, int Ho, int Wo, int Hi, int Wi, int pad, int stride,
                      unet_stream_t s) {
  UNET_REQUIRE(weights && inputs && activations, "unet_convolve_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && Ci > 0 && Co > 0 && Hi > 0 && Wi > 0 && stride > 0 && pad >= 0,
               "unet_convolve_f32: need positive sizes");
  UNET_REQUIRE(k > 0 && (k & 1), "kernel must be square and odd.");
  UNET_REQUIRE(Ho == (Hi + 2 * pad - k) / stride + 1 && Wo == (Wi + 2 * pad - k) / stride + 1 && Ho > 0 && Wo > 0,
               "unet_convolve_f32: H_out/W_out (%d,%d) do not match (H_in + 2*padding - k)/stride + 1", Ho, Wo);
  UNET_REQUIRE(act_len <= 0 || act_len >= (int64_t)N * Co * Ho * Wo,
               "unet_convolve_f32: activation array too short");
  UNET_REQUIRE(N <= 65535, "unet_convolve_f32: batch above 65535");
  const int PH = (TILE - 1) * stride + k;
  const size_t lds = sizeof(float) * ((((size_t)F_CIB * PH * (PH + 1) + 3) & ~(size_t)3) + (size_t)F_CIB * k * k * F_COB);
  if (lds > kMaxDynLds)
    return unet_set_error(UNET_ERR_UNSUPPORTED, "unet_convolve_f32: kernel %d / stride %d needs %zu B of LDS", k,
                          stride, lds);
  const double flops = 2.0 * N * Co * Ho * Wo * (double)Ci * k * k;
  ProfScope prof(PROF_CONV_FWD_F32, flops, 4.0 * ((double)N * Ci * Hi * Wi + (double)N * Co * Ho * Wo), as_stream(s));
  if (mfma_conv_serves(k, pad, stride, Co, Ho, Wo)) {
    const MfmaConvPlan pl = plan_mfma_conv(N, Ci, Co, Ho, Wo);
    float *partial = nullptr;
    if (pl.splits > 1) {
      void *ws;
      int rc = unet_workspace(&ws, sizeof(float) * (size_t)pl.splits * (size_t)pl.total, as_stream(s));
      if (rc) return rc;
      partial = (float *)ws;
    }
    return launch_mfma_conv(pl, weights, inputs, activations, partial, N, Ci, Co, Ho, Wo, Hi, Wi, pad, as_stream(s));
  }
  const int tilesH = (Ho + TILE - 1) / TILE, tilesW = (Wo + TILE - 1) / TILE;
  dim3 grid((unsigned)(tilesH * tilesW), (unsigned)((Co + F_COB - 1) / F_COB), (unsigned)N);
  if (k == 3 && Co == 1)  // grid.y = 1 either way; the LDS request covers the smaller weight slice
    hipLaunchKernelGGL((conv_fwd_f32_kernel<3, 1>), grid, dim3(256), lds, as_stream(s), weights, inputs, activations,
                       Ci, Co, Ho, Wo, Hi, Wi, pad, stride, k, tilesW);
  else if (k == 3)
    hipLaunchKernelGGL(conv_fwd_f32_kernel<3>, grid, dim3(256), lds, as_stream(s), weights, inputs, activations,
                       Ci, Co, Ho, Wo, Hi, Wi, pad, stride, k, tilesW);
  else if (k == 1)
    hipLaunchKernelGGL(conv_fwd_f32_kernel<1>, grid, dim3(256), lds, as_stream(s), weights, inputs, activations,
                       Ci, Co, Ho, Wo, Hi, Wi, pad, stride, k, tilesW);
  else
    hipLaunchKernelGGL(conv_fwd_f32_kernel<0>, grid, dim3(256), lds, as_stream(s), weights, inputs, activations,
                       Ci, Co, Ho, Wo, Hi, Wi, pad, stride, k, tilesW);
  UNET_LAUNCH_CHECK("conv_fwd_f32_kernel");
  return UNET_OK;
}

int unet_convolve_backward_f32(const float *grad_activations, const float *inputs, const float *weights,
                               float *grad_inputs, float *grad_weights, int N, int k, int Ci, int Co,
                               int Ho, int Wo, int Hi, int Wi, int pad, int stride, unet_stream_t s) {
  UNET_REQUIRE(grad_activations && inputs && weights && grad_weights, "unet_convolve_backward_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && Ci > 0 && Co > 0 && Hi > 0 && Wi > 0 && stride > 0 && pad >= 0,
               "unet_convolve_backward_f32: need positive sizes");
  UNET_REQUIRE(k > 0 && (k & 1), "kernel must be square and odd.");
  UNET_REQUIRE(Ho == (Hi + 2 * pad - k) / stride + 1 && Wo == (Wi + 2 * pad - k) / stride + 1 && Ho > 0 && Wo > 0,
               "unet_convolve_backward_f32: H_out/W_out (%d,%d) do not match the input geometry", Ho, Wo);
  UNET_REQUIRE(N <= 65535 && Ci <= 65535, "unet_convolve_backward_f32: batch or C_in above 65535");
  if (k != 1 && k != 3 && k != 5 && k != 7)
    return unet_set_error(UNET_ERR_UNSUPPORTED, "unet_convolve_backward_f32: kernel size %d (supported: 1,3,5,7)", k);

  // ---- data gradient
  // 3x3 / stride 1 / pad 1: the forward matrix-core kernel on the gradient, with flipped and transposed weights
  // (built per call: 36 bytes per weight against 36 bytes per weight and OUTPUT PIXEL of the convolution itself)
  if (grad_inputs && pad == 1 && mfma_conv_serves(k, pad, stride, Ci, Hi, Wi)) {
    const MfmaConvPlan pl = plan_mfma_conv(N, Co, Ci, Hi, Wi);
    const size_t head = (((size_t)Ci * Co * 9 + Ci) + 63) & ~(size_t)63;
    void *ws;
    int rc = unet_workspace(&ws, sizeof(float) * (head + (pl.splits > 1 ? (size_t)pl.splits * (size_t)pl.total : 0)),
                            as_stream(s));
    if (rc) return rc;
    float *flipped = (float *)ws;
    ProfScope prof(PROF_CONV_DGRAD_F32, 2.0 * N * Co * Ho * Wo * (double)Ci * k * k,
                   4.0 * ((double)N * Ci * Hi * Wi + (double)N * Co * Ho * Wo), as_stream(s));
    hipLaunchKernelGGL(flip_weights_f32_kernel, dim3(ew_grid((i64)Ci * Co * 9 + Ci)), dim3(256), 0, as_stream(s), weights,
                       flipped, Ci, Co);
    UNET_LAUNCH_CHECK("flip_weights_f32_kernel");
    rc = launch_mfma_conv(pl, flipped, grad_activations, grad_inputs, flipped + head, N, Co, Ci, Hi, Wi, Ho, Wo, 1,
                          as_stream(s));
    if (rc) return rc;
  } else if (grad_inputs) {
    const int PH = (TILE - 1 + k - 1) / stride + 2;
    const size_t lds = sizeof(float) * ((((size_t)D_COC * PH * (PH + 1) + 3) & ~(size_t)3) + (size_t)D_COC * k * k * D_CIB);
    if (lds > kMaxDynLds)
      return unet_set_error(UNET_ERR_UNSUPPORTED, "unet_convolve_backward_f32: needs %zu B of LDS", lds);
    const int tilesH = (Hi + TILE - 1) / TILE, tilesW = (Wi + TILE - 1) / TILE;
    dim3 grid((unsigned)(tilesH * tilesW), (unsigned)((Ci + D_CIB - 1) / D_CIB), (unsigned)N);
    ProfScope prof(PROF_CONV_DGRAD_F32, 2.0 * N * Co * Ho * Wo * (double)Ci * k * k,
                   4.0 * ((double)N * Ci * Hi * Wi + (double)N * Co * Ho * Wo), as_stream(s));
    hipLaunchKernelGGL(conv_bwd_data_f32_kernel, grid, dim3(256), lds, as_stream(s), weights, grad_activations,
                       grad_inputs, Ci, Co, Ho, Wo, Hi, Wi, pad, stride, k, tilesW, PH);
    UNET_LAUNCH_CHECK("conv_bwd_data_f32_kernel");
  }

  // ---- weight and bias gradient
  const i64 nW = (i64)Ci * Co * k * k;
  if (k == 3 && stride == 1 && pad == 1 && Ci >= 16 && Co >= 16 && Wo >= 8 && Ho >= 4) {
    // matrix-core kernel: 64 x 64 channel blocks, the 64-pixel regions split over enough workgroups to fill the device
    const bool wide = Wo >= 32;
    const int tw = wide ? 32 : 16, tr = 64 / tw;
    const i64 regions = (i64)N * ((Ho + tr - 1) / tr) * ((Wo + tw - 1) / tw);
    UNET_REQUIRE(regions < (1ll << 30), "unet_convolve_backward_f32: too many pixels for one call");
    const int blocks = ((Co + 63) / 64) * ((Ci + 63) / 64);
    i64 splits = (256 + blocks - 1) / blocks;
    if (splits > regions) splits = regions;
    if (splits > 512) splits = 512;
    const int rps = (int)((regions + splits - 1) / splits);
    splits = (regions + rps - 1) / rps;
    void *ws;
    int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * (size_t)(nW + Co), as_stream(s));
    if (rc) return rc;
    dim3 grid((unsigned)splits, (unsigned)((Co + 63) / 64), (unsigned)((Ci + 63) / 64));
    ProfScope prof(PROF_CONV_WGRAD_F32, 2.0 * N * Co * Ho * Wo * (double)Ci * k * k,
                   4.0 * ((double)N * Ci * Hi * Wi + (double)N * Co * Ho * Wo), as_stream(s));
    if (wide)
      hipLaunchKernelGGL(conv3x3_wgrad_mfma_f32_kernel<32>, grid, dim3(256), 0, as_stream(s), grad_activations, inputs,
                         (float *)ws, N, Ci, Co, Ho, Wo, rps, nW);
    else
      hipLaunchKernelGGL(conv3x3_wgrad_mfma_f32_kernel<16>, grid, dim3(256), 0, as_stream(s), grad_activations, inputs,
                         (float *)ws, N, Ci, Co, Ho, Wo, rps, nW);
    UNET_LAUNCH_CHECK("conv3x3_wgrad_mfma_f32_kernel");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ew_grid(nW + Co)), dim3(256), 0, as_stream(s), (const float *)ws,
                       grad_weights, nW + Co, (int)splits);
    UNET_LAUNCH_CHECK("reduce_partials_kernel");
    return UNET_OK;
  }
  const int COB = k == 1 ? 16 : k == 3 ? 8 : k == 5 ? 2 : 1;
  const int cogroups = (Co + COB - 1) / COB;
  const int rows = N * Ho;
  const int splits = pick_splits((i64)Ci * cogroups, rows, Wo);
  const int rps = (rows + splits - 1) / splits;
  void *ws;
  int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * (size_t)(nW + Co), as_stream(s));
  if (rc) return rc;
  dim3 grid((unsigned)splits, (unsigned)cogroups, (unsigned)Ci);
  ProfScope prof(PROF_CONV_WGRAD_F32, 2.0 * N * Co * Ho * Wo * (double)Ci * k * k,
                 4.0 * ((double)N * Ci * Hi * Wi + (double)N * Co * Ho * Wo), as_stream(s));
#define LAUNCH_BW(KK_, COB_)                                                                               \
  hipLaunchKernelGGL((conv_bwd_weight_f32_kernel<KK_, COB_>), grid, dim3(256), 0, as_stream(s), grad_activations, \
                     inputs, (float *)ws, N, Ci, Co, Ho, Wo, Hi, Wi, pad, stride, rps, nW)
  if (k == 1) LAUNCH_BW(1, 16);
  else if (k == 3) LAUNCH_BW(3, 8);
  else if (k == 5) LAUNCH_BW(5, 2);
  else LAUNCH_BW(7, 1);
#undef LAUNCH_BW
  UNET_LAUNCH_CHECK("conv_bwd_weight_f32_kernel");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ew_grid(nW + Co)), dim3(256), 0, as_stream(s), (const float *)ws,
                     grad_weights, nW + Co, splits);
  UNET_LAUNCH_CHECK("reduce_partials_kernel");
  return UNET_OK;
}

int unet_upsample_f32(const float *inputs, const float *weights, float *activations, int scale, int Ci,
                      int Co, int Hi, int Wi, int N, unet_stream_t s) {
  UNET_REQUIRE(inputs && weights && activations, "unet_upsample_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && Ci > 0 && Co > 0 && Hi > 0 && Wi > 0 && scale > 0, "unet_upsample_f32: need positive sizes");
  UNET_REQUIRE(N <= 65535, "unet_upsample_f32: batch above 65535");
  // matrix-core kernel (it loads the four phases of a weight as 16 bytes and stores output pairs as 8)
  if (scale == 2 && Ci >= 16 && Co >= 16 && Wi >= 8 && ((uintptr_t)weights & 15) == 0 && ((uintptr_t)activations & 7) == 0) {
    const bool wide = Wi >= 32;
    const int tw = wide ? 32 : 16, tr = 64 / tw;
    const int tilesX = (Wi + tw - 1) / tw, tilesY = (Hi + tr - 1) / tr, cob = (Co + 63) / 64;
    int splits = 1;
    const i64 wgs = (i64)tilesX * tilesY * cob * N;
    if (wgs < 192) {  // single images, deep levels: divide the input channels over more workgroups
      splits = (int)((384 + wgs - 1) / wgs);
      const int max_splits = Ci / 64 > 0 ? Ci / 64 : 1;
      if (splits > max_splits) splits = max_splits;
      if (splits > 16) splits = 16;
    }
    const int ci_per_split = ((Ci + splits - 1) / splits + UM_CIB - 1) / UM_CIB * UM_CIB;
    splits = (Ci + ci_per_split - 1) / ci_per_split;
    const i64 total = (i64)N * Co * Hi * 2 * Wi * 2;
    float *dst = activations;
    if (splits > 1) {
      void *ws;
      int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * (size_t)total, as_stream(s));
      if (rc) return rc;
      dst = (float *)ws;
    }
    dim3 grid((unsigned)(tilesX * tilesY), (unsigned)cob, (unsigned)(N * splits));
    if (wide)
      hipLaunchKernelGGL(upsample2_fwd_mfma_f32_kernel<32>, grid, dim3(256), 0, as_stream(s), inputs, weights, dst, Ci, Co,
                         Hi, Wi, tilesX, splits, ci_per_split);
    else
      hipLaunchKernelGGL(upsample2_fwd_mfma_f32_kernel<16>, grid, dim3(256), 0, as_stream(s), inputs, weights, dst, Ci, Co,
                         Hi, Wi, tilesX, splits, ci_per_split);
    UNET_LAUNCH_CHECK("upsample2_fwd_mfma_f32_kernel");
    if (splits > 1) {
      hipLaunchKernelGGL(conv_split_reduce_f32_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(s), (const float *)dst,
                         weights + (i64)Ci * Co * 4, activations, total, (i64)Hi * 2 * Wi * 2, Co, splits);
      UNET_LAUNCH_CHECK("conv_split_reduce_f32_kernel");
    }
    return UNET_OK;
  }
  const size_t lds16 = sizeof(float) * (size_t)16 * scale * scale * U_COB, lds64 = 4 * lds16;
  if (lds16 > kMaxDynLds) return unet_set_error(UNET_ERR_UNSUPPORTED, "unet_upsample_f32: scale %d too large", scale);
  const i64 opix = (i64)Hi * scale * Wi * scale;
  dim3 grid(cdiv_u(opix, 256), (unsigned)((Co + U_COB - 1) / U_COB), (unsigned)N);
  // few workgroups (single images, deep levels): every workgroup is alone on its CU and the 64-channel chunks
  // cut its memory round trips (1024 -> 512 channels at 16 x 16: 244 -> 142 us); with more workgroups than that
  // the 64 input registers cost more occupancy than the round trips were worth (measured)
  if (lds64 <= 16384 && Ci >= 64 && (i64)grid.x * grid.y * grid.z <= 512)
    hipLaunchKernelGGL(upsample_fwd_f32_kernel<64>, grid, dim3(256), lds64, as_stream(s), inputs, weights, activations,
                       Ci, Co, Hi, Wi, scale);
  else
    hipLaunchKernelGGL(upsample_fwd_f32_kernel<16>, grid, dim3(256), lds16, as_stream(s), inputs, weights, activations,
                       Ci, Co, Hi, Wi, scale);
  UNET_LAUNCH_CHECK("upsample_fwd_f32_kernel");
  return UNET_OK;
}

int unet_upsample_backward_f32(const float *grad_activations, const float *inputs, const float *weights,
                               float *grad_inputs, float *grad_weights, int scale, int Ci, int Co, int Hi,
                               int Wi, int N, unet_stream_t s) {
  UNET_REQUIRE(grad_activations && inputs && weights && grad_weights, "unet_upsample_backward_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && Ci > 0 && Co > 0 && Hi > 0 && Wi > 0 && scale > 0,
               "unet_upsample_backward_f32: need positive sizes");
  UNET_REQUIRE(N <= 65535 && Ci <= 65535, "unet_upsample_backward_f32: batch or C_in above 65535");
  if (scale < 2 || scale > 4)
    return unet_set_error(UNET_ERR_UNSUPPORTED, "unet_upsample_backward_f32: scale %d (supported: 2,3,4)", scale);
  const int tw_d = Wi >= 32 ? 32 : 16, tiles_d = ((Wi + tw_d - 1) / tw_d) * ((Hi + 64 / tw_d - 1) / (64 / tw_d));
  if (grad_inputs && scale == 2 && Ci >= 16 && Co >= 16 && Wi >= 8 && ((uintptr_t)weights & 15) == 0 &&
      (i64)tiles_d * ((Ci + 63) / 64) * N >= 128) {  // matrix-core kernel, when there are workgroups enough for the device
    dim3 grid((unsigned)tiles_d, (unsigned)((Ci + 63) / 64), (unsigned)N);
    if (tw_d == 32)
      hipLaunchKernelGGL(upsample2_dgrad_mfma_f32_kernel<32>, grid, dim3(256), 0, as_stream(s), grad_activations, weights,
                         grad_inputs, Ci, Co, Hi, Wi, (Wi + 31) / 32);
    else
      hipLaunchKernelGGL(upsample2_dgrad_mfma_f32_kernel<16>, grid, dim3(256), 0, as_stream(s), grad_activations, weights,
                         grad_inputs, Ci, Co, Hi, Wi, (Wi + 15) / 16);
    UNET_LAUNCH_CHECK("upsample2_dgrad_mfma_f32_kernel");
  } else if (grad_inputs) {
    const size_t lds = sizeof(float) * (size_t)UD_COC * scale * scale * UD_CIB;
    dim3 grid(cdiv_u((i64)Hi * Wi, 256), (unsigned)((Ci + UD_CIB - 1) / UD_CIB), (unsigned)N);
    hipLaunchKernelGGL(upsample_bwd_data_f32_kernel, grid, dim3(256), lds, as_stream(s), grad_activations, weights,
                       grad_inputs, Ci, Co, Hi, Wi, scale);
    UNET_LAUNCH_CHECK("upsample_bwd_data_f32_kernel");
  }
  const i64 nW = (i64)Ci * Co * scale * scale;
  if (scale == 2 && Ci >= 16 && Co >= 16 && Wi >= 8) {  // matrix-core kernel
    const bool wide = Wi >= 32;
    const int tw = wide ? 32 : 16, tr = 32 / tw;
    const i64 regions = (i64)N * ((Hi + tr - 1) / tr) * ((Wi + tw - 1) / tw);
    UNET_REQUIRE(regions < (1ll << 30), "unet_upsample_backward_f32: too many pixels for one call");
    const int blocks = ((Co + 63) / 64) * ((Ci + 63) / 64);
    i64 splits = (512 + blocks - 1) / blocks;
    if (splits > regions) splits = regions;
    if (splits > 512) splits = 512;
    const int rps = (int)((regions + splits - 1) / splits);
    splits = (regions + rps - 1) / rps;
    void *ws;
    int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * (size_t)(nW + Co), as_stream(s));
    if (rc) return rc;
    dim3 grid((unsigned)splits, (unsigned)((Co + 63) / 64), (unsigned)((Ci + 63) / 64));
    if (wide)
      hipLaunchKernelGGL(upsample2_wgrad_mfma_f32_kernel<32>, grid, dim3(256), 0, as_stream(s), grad_activations, inputs,
                         (float *)ws, N, Ci, Co, Hi, Wi, rps, nW);
    else
      hipLaunchKernelGGL(upsample2_wgrad_mfma_f32_kernel<16>, grid, dim3(256), 0, as_stream(s), grad_activations, inputs,
                         (float *)ws, N, Ci, Co, Hi, Wi, rps, nW);
    UNET_LAUNCH_CHECK("upsample2_wgrad_mfma_f32_kernel");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ew_grid(nW + Co)), dim3(256), 0, as_stream(s), (const float *)ws,
                       grad_weights, nW + Co, (int)splits);
    UNET_LAUNCH_CHECK("reduce_partials_kernel");
    return UNET_OK;
  }
  const int COB = scale == 2 ? 8 : scale == 3 ? 4 : 2;
  const int cogroups = (Co + COB - 1) / COB;
  const int rows = N * Hi;
  const int splits = pick_splits((i64)Ci * cogroups, rows, Wi);
  const int rps = (rows + splits - 1) / splits;
  void *ws;
  int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * (size_t)(nW + Co), as_stream(s));
  if (rc) return rc;
  dim3 grid((unsigned)splits, (unsigned)cogroups, (unsigned)Ci);
#define LAUNCH_UW(S_, COB_)                                                                                   \
  hipLaunchKernelGGL((upsample_bwd_weight_f32_kernel<S_, COB_>), grid, dim3(256), 0, as_stream(s), grad_activations, \
                     inputs, (float *)ws, N, Ci, Co, Hi, Wi, rps, nW)
  if (scale == 2) LAUNCH_UW(2, 8);
  else if (scale == 3) LAUNCH_UW(3, 4);
  else LAUNCH_UW(4, 2);
#undef LAUNCH_UW
  UNET_LAUNCH_CHECK("upsample_bwd_weight_f32_kernel");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ew_grid(nW + Co)), dim3(256), 0, as_stream(s), (const float *)ws,
                     grad_weights, nW + Co, splits);
  UNET_LAUNCH_CHECK("reduce_partials_kernel");
  return UNET_OK;
}

}  // extern "C"
