// Kernel laboratory for the bf16 3x3 convolution: variants of the implicit-GEMM
// kernel timed on the layer shapes of the 4-level base-64 U-Net (batch 16).
// Not part of the library; build with tools/build_lab.sh and run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/unet_hip.h"

typedef int64_t i64;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

struct ConvArgs {
  const bf16_t *in;
  const bf16_t *wt;
  bf16_t *out;
  int N, H, W, Ci, CoP, in_ld, out_ld, tilesX;
};

constexpr int PITCH = 80;

// ---------------------------------------------------------------------------
// V1: the library's structure (single LDS buffer, register prefetch, two
// barriers per chunk, 2 workgroups per CU).  XROW = halo row stride in bytes.
// ABL: 0 full; 1 no global loads inside the loop; 2 also no LDS stores/barriers
// inside the loop; 3 also no LDS reads (MFMA only).
// ---------------------------------------------------------------------------
template <int BN, int XROW, int ABL>
__global__ __launch_bounds__(256, 2) void conv_v1(const ConvArgs a) {
  constexpr int HALO = 18, NPIX = HALO * HALO;
  constexpr int W_BYTES = 9 * BN * PITCH;
  constexpr int XCH = NPIX * 4, XI = (XCH + 255) / 256;
  constexpr int WCH = 9 * BN * 4, WI = (WCH + 255) / 256;
  constexpr int MF = BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *ldsW = lds, *ldsX = lds + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, co0 = blockIdx.y * BN;
  const i64 n = blockIdx.z;
  const int y0 = (tile / a.tilesX) * 16, x0 = (tile % a.tilesX) * 16;
  const bf16_t *in_n = a.in + n * (i64)a.H * a.W * a.in_ld;
  int xoff[XI], xdst[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int c = tid + 256 * i, pix = c >> 2, part = c & 3;
    const int py = pix / HALO, px = pix - py * HALO;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = c < XCH && y >= 0 && y < a.H && x >= 0 && x < a.W;
    xoff[i] = ok ? (y * a.W + x) * a.in_ld + part * 8 : -1;
    xdst[i] = py * XROW + px * PITCH + part * 16;
  }
  const bf16_t *w_thread = a.wt + (i64)co0 * a.Ci;
  uint4 xr[XI], wr[WI];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xr[i] = make_uint4(0, 0, 0, 0);
      if (xoff[i] >= 0) xr[i] = *reinterpret_cast<const uint4 *>(in_n + xoff[i] + k0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i, row = c >> 2, part = c & 3;
      const int tap = row / BN, co = row % BN;
      wr[i] = make_uint4(0, 0, 0, 0);
      if (WCH % 256 == 0 || c < WCH)
        wr[i] = *reinterpret_cast<const uint4 *>(w_thread + ((i64)tap * a.CoP + co) * a.Ci + k0 + part * 8);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = tid + 256 * i;
      if (c < XCH) *reinterpret_cast<uint4 *>(ldsX + xdst[i]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4 *>(ldsW + (c >> 2) * PITCH + (c & 3) * 16) = wr[i];
    }
  };
  f32x16 acc[MF][2];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
  const int a_lane = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int b_lane = (wave * 4 + ((lane & 31) >> 4)) * XROW + (lane & 15) * PITCH + (lane >> 5) * 16;
  const int nk = a.Ci / 32;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  bf16x8 af[MF], bfr[2];
  if (ABL >= 3) {
#pragma unroll
    for (int m = 0; m < MF; ++m) af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + m * 32 * PITCH + a_lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) bfr[q] = *reinterpret_cast<const bf16x8 *>(ldsX + 2 * q * XROW + b_lane);
  }
  for (int kc = 0; kc < nk; ++kc) {
    if (ABL < 1 && kc + 1 < nk) load_chunk((kc + 1) * 32);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ABL < 3) {
#pragma unroll
          for (int m = 0; m < MF; ++m)
            af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + (tap * BN + m * 32) * PITCH + ks * 32 + a_lane);
#pragma unroll
          for (int q = 0; q < 2; ++q)
            bfr[q] = *reinterpret_cast<const bf16x8 *>(ldsX + (2 * q + kh) * XROW + kw * PITCH + ks * 32 + b_lane);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[q], acc[m][q], 0, 0, 0);
      }
    }
    if (ABL < 2) {
      __syncthreads();
      if (kc + 1 < nk) {
        store_chunk();
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // epilogue: stage in LDS, 16-byte stores
  constexpr int OP = BN * 2 + 16;
  unsigned char *stage = lds;
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pix = wave * 64 + q * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = m * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
        *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = __builtin_convertvector(v, bf16x4);
      }
    }
  __syncthreads();
  constexpr int CH = BN / 8;
  bf16_t *out_n = a.out + n * (i64)a.H * a.W * a.out_ld + co0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + 256 * i, pix = c / CH, part = c % CH;
    const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
    if (y < a.H && x < a.W)
      *reinterpret_cast<uint4 *>(out_n + ((i64)y * a.W + x) * a.out_ld + part * 8) =
          *reinterpret_cast<const uint4 *>(stage + pix * OP + part * 16);
  }
}

// ---------------------------------------------------------------------------
// V2: double-buffered LDS, one barrier per chunk, LDS stores and global loads
// of later chunks interleaved between the MFMA groups, fragments fetched one
// group ahead.  One workgroup (4 waves) per CU.
// ---------------------------------------------------------------------------
template <int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_v2(const ConvArgs a) {
  constexpr int HALO = 18, NPIX = HALO * HALO, XROW = 1536;
  constexpr int W_BYTES = 9 * BN * PITCH, X_BYTES = HALO * XROW, BUF = W_BYTES + X_BYTES;
  constexpr int XCH = NPIX * 4, XI = (XCH + 255) / 256;
  constexpr int WCH = 9 * BN * 4, WI = WCH / 256;
  constexpr int MF = BN / 32;
  constexpr int NG = 18;  // MFMA groups per chunk (9 taps x 2 k-steps)
  constexpr int NP = XI + WI;
  static_assert(NP <= NG, "one staged piece per MFMA group");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, co0 = blockIdx.y * BN;
  const i64 n = blockIdx.z;
  const int y0 = (tile / a.tilesX) * 16, x0 = (tile % a.tilesX) * 16;
  const bf16_t *in_n = a.in + n * (i64)a.H * a.W * a.in_ld;
  int xoff[XI], xdst[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int c = tid + 256 * i, pix = c >> 2, part = c & 3;
    const int py = pix / HALO, px = pix - py * HALO;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = c < XCH && y >= 0 && y < a.H && x >= 0 && x < a.W;
    xoff[i] = ok ? (y * a.W + x) * a.in_ld + part * 8 : -1;
    xdst[i] = (c < XCH) ? W_BYTES + py * XROW + px * PITCH + part * 16 : -1;
  }
  const bf16_t *w_thread = a.wt + (i64)co0 * a.Ci;
  i64 woff[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int c = tid + 256 * i, row = c >> 2, part = c & 3;
    const int tap = row / BN, co = row % BN;
    woff[i] = ((i64)tap * a.CoP + co) * a.Ci + part * 8;
  }
  uint4 pr[NP];  // staged pieces: XI halo pieces then WI weight pieces
  auto load_piece = [&](int i, int k0) {
    if (i < XI) {
      pr[i] = make_uint4(0, 0, 0, 0);
      if (xoff[i] >= 0) pr[i] = *reinterpret_cast<const uint4 *>(in_n + xoff[i] + k0);
    } else {
      pr[i] = *reinterpret_cast<const uint4 *>(w_thread + woff[i - XI] + k0);
    }
  };
  auto store_piece = [&](int i, unsigned char *buf) {
    if (i < XI) {
      if (xdst[i] >= 0) *reinterpret_cast<uint4 *>(buf + xdst[i]) = pr[i];
    } else {
      const int c = tid + 256 * (i - XI);
      *reinterpret_cast<uint4 *>(buf + (c >> 2) * PITCH + (c & 3) * 16) = pr[i];
    }
  };
  f32x16 acc[MF][2];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
  const int a_lane = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int b_lane = W_BYTES + (wave * 4 + ((lane & 31) >> 4)) * XROW + (lane & 15) * PITCH + (lane >> 5) * 16;
  const int nk = a.Ci / 32;
#pragma unroll
  for (int i = 0; i < NP; ++i) load_piece(i, 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) store_piece(i, lds);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < NP; ++i) load_piece(i, 32);
  }
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    unsigned char *cur = lds + (kc & 1) * BUF, *nxt = lds + ((kc & 1) ^ 1) * BUF;
    const bool have_next = kc + 1 < nk, have_next2 = kc + 2 < nk;
    bf16x8 af[2][MF], bfr[2][2];
#pragma unroll
    for (int m = 0; m < MF; ++m) af[0][m] = *reinterpret_cast<const bf16x8 *>(cur + m * 32 * PITCH + a_lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) bfr[0][q] = *reinterpret_cast<const bf16x8 *>(cur + 2 * q * XROW + b_lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
        const int tap = (g + 1) >> 1, ks = (g + 1) & 1, kh = tap / 3, kw = tap % 3;
#pragma unroll
        for (int m = 0; m < MF; ++m)
          af[(g + 1) & 1][m] =
              *reinterpret_cast<const bf16x8 *>(cur + (tap * BN + m * 32) * PITCH + ks * 32 + a_lane);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          bfr[(g + 1) & 1][q] =
              *reinterpret_cast<const bf16x8 *>(cur + (2 * q + kh) * XROW + kw * PITCH + ks * 32 + b_lane);
      }
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][m], bfr[g & 1][q], acc[m][q], 0, 0, 0);
      if (g < NP) {
        if (have_next) store_piece(g, nxt);
        if (have_next2) load_piece(g, (kc + 2) * 32);
      }
    }
    __syncthreads();
  }
  constexpr int OP = BN * 2 + 16;
  unsigned char *stage = lds;
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pix = wave * 64 + q * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = m * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
        *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = __builtin_convertvector(v, bf16x4);
      }
    }
  __syncthreads();
  constexpr int CH = BN / 8;
  bf16_t *out_n = a.out + n * (i64)a.H * a.W * a.out_ld + co0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + 256 * i, pix = c / CH, part = c % CH;
    const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
    if (y < a.H && x < a.W)
      *reinterpret_cast<uint4 *>(out_n + ((i64)y * a.W + x) * a.out_ld + part * 8) =
          *reinterpret_cast<const uint4 *>(stage + pix * OP + part * 16);
  }
}

// ---------------------------------------------------------------------------
// V3: V2's structure (two LDS buffers, one barrier per chunk, staging interleaved with the MFMA
// groups) with lean addressing: buffer loads whose per-lane offsets are fixed for the whole kernel
// (the channel chunk goes into the scalar offset), out-of-image pixels point beyond the descriptor
// and read as zero, LDS store addresses precomputed.
// ---------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_v3(const ConvArgs a) {
  constexpr int HALO = 18, NPIX = HALO * HALO, XROW = 1440;
  constexpr int W_BYTES = 9 * BN * PITCH, X_BYTES = HALO * XROW, BUF = W_BYTES + X_BYTES;
  constexpr int XCH = NPIX * 4, XI = (XCH + 255) / 256;
  constexpr int WCH = 9 * BN * 4, WI = WCH / 256;
  constexpr int MF = BN / 32;
  constexpr int NG = 18;
  constexpr int NP = XI + WI;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NP <= NG, "one staged piece per MFMA group");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, co0 = blockIdx.y * BN;
  const i64 n = blockIdx.z;
  const int y0 = (tile / a.tilesX) * 16, x0 = (tile % a.tilesX) * 16;
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in) + n * (i64)a.H * a.W * a.in_ld, 0,
                                                     (unsigned)a.H * a.W * a.in_ld * 2u, 0x00020000);
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.wt), 0, 9u * a.CoP * a.Ci * 2u, 0x00020000);
  unsigned voff[NP];
  int ldst[NP];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int c = tid + 256 * i, pix = c >> 2, part = c & 3;
    const int py = pix / HALO, px = pix - py * HALO;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = c < XCH && y >= 0 && y < a.H && x >= 0 && x < a.W;
    voff[i] = ok ? (unsigned)(((y * a.W + x) * a.in_ld + part * 8) * 2) : OOB;
    ldst[i] = (c < XCH) ? W_BYTES + py * XROW + px * PITCH + part * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int c = tid + 256 * i, row = c >> 2, part = c & 3;
    const int tap = row / BN, co = row % BN;
    voff[XI + i] = (unsigned)((((tap * a.CoP + co0 + co) * a.Ci) + part * 8) * 2);
    ldst[XI + i] = row * PITCH + part * 16;
  }
  i32x4 pr[NP];
  auto load_piece = [&](int i, int k0) {
    pr[i] = __builtin_amdgcn_raw_buffer_load_b128(i < XI ? xrs : wrs, voff[i], k0 * 2, 0);
  };
  auto store_piece = [&](int i, unsigned char *buf) {
    if (i >= XI || ldst[i] >= 0) *reinterpret_cast<i32x4 *>(buf + ldst[i]) = pr[i];
  };
  f32x16 acc[MF][2];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
  const int a_lane = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int b_lane = W_BYTES + (wave * 4 + ((lane & 31) >> 4)) * XROW + (lane & 15) * PITCH + (lane >> 5) * 16;
  const int nk = a.Ci / 32;
#pragma unroll
  for (int i = 0; i < NP; ++i) load_piece(i, 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) store_piece(i, lds);
#pragma unroll
  for (int i = 0; i < NP; ++i) load_piece(i, min(1, nk - 1) * 32);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    unsigned char *cur = lds + (kc & 1) * BUF, *nxt = lds + ((kc & 1) ^ 1) * BUF;
    const int k2 = min(kc + 2, nk - 1) * 32;
    bf16x8 af[2][MF], bfr[2][2];
#pragma unroll
    for (int m = 0; m < MF; ++m) af[0][m] = *reinterpret_cast<const bf16x8 *>(cur + m * 32 * PITCH + a_lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) bfr[0][q] = *reinterpret_cast<const bf16x8 *>(cur + 2 * q * XROW + b_lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
        const int tap = (g + 1) >> 1, ks = (g + 1) & 1, kh = tap / 3, kw = tap % 3;
#pragma unroll
        for (int m = 0; m < MF; ++m)
          af[(g + 1) & 1][m] =
              *reinterpret_cast<const bf16x8 *>(cur + (tap * BN + m * 32) * PITCH + ks * 32 + a_lane);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          bfr[(g + 1) & 1][q] =
              *reinterpret_cast<const bf16x8 *>(cur + (2 * q + kh) * XROW + kw * PITCH + ks * 32 + b_lane);
      }
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][m], bfr[g & 1][q], acc[m][q], 0, 0, 0);
      // stores of chunk kc+1 (in registers since the previous pass) first, in load order, so that the
      // wait before store g is "all but the NP-1-g youngest loads"; the loads of chunk kc+2 follow
      // in the last groups.  Both are unconditional: past the end they restage the last chunk.
      if (g < NP) store_piece(g, nxt);
      if (g >= NG - 3) {
#pragma unroll
        for (int i = (g - (NG - 3)) * 5; i < (g - (NG - 3)) * 5 + 5 && i < NP; ++i) load_piece(i, k2);
      }
    }
    __syncthreads();
  }
  constexpr int OP = BN * 2 + 16;
  unsigned char *stage = lds;
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pix = wave * 64 + q * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = m * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
        *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = __builtin_convertvector(v, bf16x4);
      }
    }
  __syncthreads();
  constexpr int CH = BN / 8;
  bf16_t *out_n = a.out + n * (i64)a.H * a.W * a.out_ld + co0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + 256 * i, pix = c / CH, part = c % CH;
    const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
    if (y < a.H && x < a.W)
      *reinterpret_cast<uint4 *>(out_n + ((i64)y * a.W + x) * a.out_ld + part * 8) =
          *reinterpret_cast<const uint4 *>(stage + pix * OP + part * 16);
  }
}

// ---------------------------------------------------------------------------
// V4: V1's structure with v_mfma_f32_16x16x32_bf16 (one MFMA eats a whole 32-channel chunk of a
// 16x16 output block) and dense 64-byte LDS rows whose 16-byte pieces are XOR-swizzled by
// (row >> 1) & 3, which makes every ds_read_b128 of a fragment conflict-free.
// ---------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_v4(const ConvArgs a) {
  constexpr int HALO = 18, NPIX = HALO * HALO;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int XCH = NPIX * 4, XI = (XCH + 255) / 256;
  constexpr int WCH = 9 * BN * 4, WI = (WCH + 255) / 256;
  constexpr int MB = BN / 16;  // 16-channel blocks
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *ldsW = lds, *ldsX = lds + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, co0 = blockIdx.y * BN;
  const i64 n = blockIdx.z;
  const int y0 = (tile / a.tilesX) * 16, x0 = (tile % a.tilesX) * 16;
  const bf16_t *in_n = a.in + n * (i64)a.H * a.W * a.in_ld;
  int xoff[XI], xdst[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int c = tid + 256 * i, pix = c >> 2, part = c & 3;
    const int py = pix / HALO, px = pix - py * HALO;
    const int y = y0 + py - 1, x = x0 + px - 1;
    const bool ok = c < XCH && y >= 0 && y < a.H && x >= 0 && x < a.W;
    xoff[i] = ok ? (y * a.W + x) * a.in_ld + part * 8 : -1;
    xdst[i] = pix * 64 + ((part ^ ((pix >> 1) & 3)) * 16);
  }
  const bf16_t *w_thread = a.wt + (i64)co0 * a.Ci;
  uint4 xr[XI], wr[WI];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xr[i] = make_uint4(0, 0, 0, 0);
      if (xoff[i] >= 0) xr[i] = *reinterpret_cast<const uint4 *>(in_n + xoff[i] + k0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i, row = c >> 2, part = c & 3;
      const int tap = row / BN, co = row % BN;
      wr[i] = make_uint4(0, 0, 0, 0);
      if (WCH % 256 == 0 || c < WCH)
        wr[i] = *reinterpret_cast<const uint4 *>(w_thread + ((i64)tap * a.CoP + co) * a.Ci + k0 + part * 8);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = tid + 256 * i;
      if (c < XCH) *reinterpret_cast<uint4 *>(ldsX + xdst[i]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i, row = c >> 2, part = c & 3;
      *reinterpret_cast<uint4 *>(ldsW + row * 64 + ((part ^ ((row >> 1) & 3)) * 16)) = wr[i];
    }
  };
  f32x4v acc[MB][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[m][q] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kp = lane >> 4;
  const int a_lane = l15 * 64 + ((kp ^ ((l15 >> 1) & 3)) * 16);
  // pixel fragments: halo index p = row*18 + col + kw, row = 4*wave + q + kh; (p >> 1) & 3 depends on
  // (row & 3) = (q + kh) & 3 and on (col + kw) >> 1
  int b_off[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) {
      const int pl = rc * 18 + l15 + kw;  // p modulo 72 (= 4 rows), enough for (p >> 1) & 3
      b_off[kw][rc] = (l15 + kw) * 64 + ((kp ^ ((pl >> 1) & 3)) * 16) + wave * 4 * HALO * 64;
    }
  const int nk = a.Ci / 32;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk((kc + 1) * 32);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
      bf16x8 af[MB], bfr[4];
#pragma unroll
      for (int m = 0; m < MB; ++m)
        af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + (tap * BN + m * 16) * 64 + a_lane);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bfr[q] = *reinterpret_cast<const bf16x8 *>(ldsX + (q + kh) * HALO * 64 + b_off[kw][(q + kh) & 3]);
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[q], acc[m][q], 0, 0, 0);
    }
    __syncthreads();
    if (kc + 1 < nk) {
      store_chunk();
      __syncthreads();
    }
  }
  __syncthreads();
  constexpr int OP = BN * 2 + 16;
  unsigned char *stage = lds;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pix = (wave * 4 + q) * 16 + l15;
      const int ch = m * 16 + 4 * kp;
      *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = __builtin_convertvector(acc[m][q], bf16x4);
    }
  __syncthreads();
  constexpr int CH = BN / 8;
  bf16_t *out_n = a.out + n * (i64)a.H * a.W * a.out_ld + co0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + 256 * i, pix = c / CH, part = c % CH;
    const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
    if (y < a.H && x < a.W)
      *reinterpret_cast<uint4 *>(out_n + ((i64)y * a.W + x) * a.out_ld + part * 8) =
          *reinterpret_cast<const uint4 *>(stage + pix * OP + part * 16);
  }
}

// ---------------------------------------------------------------------------
struct Shape {
  int H, Ci, Co;
};

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

template <typename F>
static float time_launch(F launch, int reps) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) {
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
  }
  float best = 1e30f;
  for (int i = 0; i < 5; ++i) launch();
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  CK(hipGetLastError());
  return best * 1e3f / reps;
}

template <typename K>
static void set_lds(K kernel, size_t bytes) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

int main(int argc, char **argv) {
  const int N = 16;
  std::vector<Shape> shapes = {{256, 64, 64}, {256, 128, 64}, {128, 128, 128}, {64, 256, 256},
                               {32, 512, 512}, {16, 1024, 1024}, {32, 1024, 512}, {128, 256, 128}};
  size_t max_in = 0, max_out = 0, max_w = 0;
  for (auto &s : shapes) {
    max_in = std::max(max_in, (size_t)N * s.H * s.H * s.Ci);
    max_out = std::max(max_out, (size_t)N * s.H * s.H * s.Co);
    max_w = std::max(max_w, (size_t)9 * s.Ci * s.Co);
  }
  bf16_t *d_in, *d_w, *d_out, *d_ref;
  CK(hipMalloc(&d_in, max_in * 2));
  CK(hipMalloc(&d_w, max_w * 2));
  CK(hipMalloc(&d_out, max_out * 2));
  CK(hipMalloc(&d_ref, max_out * 2));
  {
    std::vector<bf16_t> h(std::max(max_in, max_w));
    uint32_t s = 12345;
    for (auto &v : h) {
      s = s * 1664525u + 1013904223u;
      v = f2bf(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
    }
    CK(hipMemcpy(d_in, h.data(), max_in * 2, hipMemcpyHostToDevice));
    for (auto &v : h) {
      s = s * 1664525u + 1013904223u;
      v = f2bf(((int)(s >> 8) % 2001 - 1000) * 2e-5f);
    }
    CK(hipMemcpy(d_w, h.data(), max_w * 2, hipMemcpyHostToDevice));
  }
  float *d_bias, *d_stats;
  CK(hipMalloc(&d_bias, 4096 * 4));
  CK(hipMemset(d_bias, 0, 4096 * 4));
  CK(hipMalloc(&d_stats, (size_t)N * 256 * 1024 * 2 * 4));
  constexpr size_t LDS_V1A = 9 * 64 * PITCH + 18 * 1440, LDS_V1B = 9 * 64 * PITCH + 18 * 1536;
  constexpr size_t EPI = 256 * (64 * 2 + 16);
  constexpr size_t LDS_V2 = 2 * (9 * 64 * PITCH + 18 * 1536);
  set_lds(conv_v1<64, 1440, 0>, std::max(LDS_V1A, EPI));
  constexpr size_t LDS_V4 = std::max((size_t)(9 * 64 * 64 + 324 * 64), EPI);
  set_lds(conv_v4<64>, LDS_V4);
  set_lds(conv_v1<64, 1536, 1>, std::max(LDS_V1B, EPI));
  constexpr size_t LDS_V3 = 2 * (9 * 64 * PITCH + 18 * 1440);
  set_lds(conv_v3<64, 1>, LDS_V3);
  set_lds(conv_v1<64, 1536, 3>, std::max(LDS_V1B, EPI));
  set_lds(conv_v2<64, 1>, LDS_V2);
  std::vector<bf16_t> h_ref, h_out;
  printf("%-18s %9s %9s %9s %9s %9s %9s   (us | TFLOP/s)\n", "shape", "v1", "v4", "noGload", "v3", "mfmaOnly",
         "v2");
  for (auto &s : shapes) {
    ConvArgs a{d_in, d_w, d_out, N, s.H, s.H, s.Ci, s.Co, s.Ci, s.Co, (s.H + 15) / 16};
    const int tiles = ((s.H + 15) / 16) * a.tilesX;
    dim3 grid(tiles, s.Co / 64, N);
    const double flops = 2.0 * N * s.H * s.H * (double)s.Ci * s.Co * 9;
    const int reps = 20;
    float t[6];
    ConvArgs ar = a;
    ar.out = d_ref;
    hipLaunchKernelGGL((conv_v1<64, 1440, 0>), grid, dim3(256), std::max(LDS_V1A, EPI), 0, ar);
    t[0] = time_launch([&] { hipLaunchKernelGGL((conv_v1<64, 1440, 0>), grid, dim3(256), std::max(LDS_V1A, EPI), 0, ar); }, reps);
    t[1] = time_launch([&] { hipLaunchKernelGGL((conv_v4<64>), grid, dim3(256), LDS_V4, 0, a); }, reps);
    const size_t out_n = (size_t)N * s.H * s.H * s.Co;
    h_ref.resize(out_n);
    h_out.resize(out_n);
    CK(hipMemcpy(h_ref.data(), d_ref, out_n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_out.data(), d_out, out_n * 2, hipMemcpyDeviceToHost));
    size_t bad1 = 0;
    for (size_t i = 0; i < out_n; ++i) bad1 += h_ref[i] != h_out[i];
    t[2] = time_launch([&] { hipLaunchKernelGGL((conv_v1<64, 1536, 1>), grid, dim3(256), std::max(LDS_V1B, EPI), 0, a); }, reps);
    t[3] = time_launch([&] { hipLaunchKernelGGL((conv_v3<64, 1>), grid, dim3(256), LDS_V3, 0, a); }, reps);
    CK(hipMemcpy(h_out.data(), d_out, out_n * 2, hipMemcpyDeviceToHost));
    size_t bad3 = 0;
    for (size_t i = 0; i < out_n; ++i) bad3 += h_ref[i] != h_out[i];
    t[4] = time_launch([&] { hipLaunchKernelGGL((conv_v1<64, 1536, 3>), grid, dim3(256), std::max(LDS_V1B, EPI), 0, a); }, reps);
    CK(hipMemset(d_out, 0, out_n * 2));
    t[5] = time_launch([&] { hipLaunchKernelGGL((conv_v2<64, 1>), grid, dim3(256), LDS_V2, 0, a); }, reps);
    CK(hipMemcpy(h_out.data(), d_out, out_n * 2, hipMemcpyDeviceToHost));
    size_t bad2 = 0;
    for (size_t i = 0; i < out_n; ++i) bad2 += h_ref[i] != h_out[i];
    float tp[4];
    tp[2] = time_launch([&] { unet_conv3x3_bf16(d_in, s.Ci, d_w, d_bias, s.Co, d_out, s.Co, nullptr, N, s.H, s.H, s.Ci, s.Co, 0, nullptr); }, reps);
    tp[3] = time_launch([&] { unet_conv3x3_bf16(d_in, s.Ci, d_w, nullptr, s.Co, d_out, s.Co, d_stats, N, s.H, s.H, s.Ci, s.Co, 0, nullptr); }, reps);
    tp[0] = time_launch([&] { unet_conv3x3_bf16(d_in, s.Ci, d_w, nullptr, s.Co, d_out, s.Co, nullptr, N, s.H, s.H, s.Ci, s.Co, 0, nullptr); }, reps);
    tp[1] = time_launch([&] { unet_conv3x3_bf16(d_in, s.Ci, d_w, d_bias, s.Co, d_out, s.Co, d_stats, N, s.H, s.H, s.Ci, s.Co, 0, nullptr); }, reps);
    char name[64];
    snprintf(name, sizeof name, "%dx%d %d->%d", s.H, s.H, s.Ci, s.Co);
    printf("%-18s", name);
    for (int i = 0; i < 6; ++i) printf(" %5.0f|%4.0f", t[i], flops / t[i] * 1e-6);
    printf(" | prod %5.0f|%4.0f  prod+bias+stats %5.0f|%4.0f", tp[0], flops / tp[0] * 1e-6, tp[1], flops / tp[1] * 1e-6);
    printf(" bias-only %5.0f stats-only %5.0f", tp[2], tp[3]);
    printf("   mismatches: pad %zu v2 %zu v3 %zu\n", bad1, bad2, bad3);
    fflush(stdout);
  }
  // weight gradient through the library (variant chosen by UNET_WGRAD_VARIANT)
  float *d_grad;
  CK(hipMalloc(&d_grad, max_w * 4 + 4096));
  printf("wgrad (UNET_WGRAD_VARIANT=%s)\n", getenv("UNET_WGRAD_VARIANT") ? getenv("UNET_WGRAD_VARIANT") : "default");
  for (auto &s : shapes) {
    const double flops = 2.0 * N * s.H * s.H * (double)s.Ci * s.Co * 9;
    float t = time_launch([&] {
      if (unet_conv3x3_wgrad_bf16(d_out, s.Co, d_in, s.Ci, d_grad, N, s.H, s.H, s.Ci, s.Co, s.Ci, s.Co, nullptr)) {
        fprintf(stderr, "wgrad failed: %s\n", unet_last_error());
        exit(1);
      }
    }, 20);
    std::vector<float> hg(64);
    CK(hipMemcpy(hg.data(), d_grad, 64 * 4, hipMemcpyDeviceToHost));
    double cs = 0;
    for (float v : hg) cs += v;
    printf("%dx%d %d->%d   %6.0f us  %5.0f TFLOP/s  checksum %.6g\n", s.H, s.H, s.Ci, s.Co, t, flops / t * 1e-6, cs);
  }
  return 0;
}
