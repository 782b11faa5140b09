// 2x2 / stride 2 transposed convolution (forward and data gradient) as ONE matrix product per launch on the
// gfx950 matrix cores, for channel counts in multiples of 64 (conv_bf16.hip keeps the general kernel for the rest):
//
//   forward        D[col][pixel] = sum_ci Wf[col][ci] * X[pixel][ci],            col = phase * CoP + co, phase = 2 i + j,
//                  out[n, 2 y + i, 2 x + j, co] = D[col][(n, y, x)] + bias[co]
//   data gradient  D[ci][pixel]  = sum_phase sum_co Wb[phase][ci][co] * dY[n, 2 y + i, 2 x + j, co]
//
// (reference: upsample.cu:9-312, one thread per output element).  The packed forward weights [phase][CoP][CiP] are
// the (4 CoP) x CiP row-major matrix Wf as they lie; the four phases only differ in where a result is stored
// (forward) or which pixel is read (data gradient).
//
// conv_mfma_kernel<1, ...> runs every phase as a workgroup of its own that re-reads the input tile, with 16 MFMAs
// per wave between two barriers and one chunk of loads in flight: 0.15 of the matrix peak.  Here:
//   - a workgroup of 4 waves owns a 16 x 16 pixel tile x 128 columns; wave w owns ALL 256 pixels x columns
//     32 w .. 32 w + 31: no weight is fetched twice, and the statistics of a column need no exchange between waves;
//   - the weights never touch LDS: a lane loads its own MFMA fragment (16 bytes of one column's row) straight from
//     memory, one 64-channel chunk ahead, two loads per fragment row so that whole 128-byte lines are consumed
//     back to back;
//   - the pixels go through registers into LDS (a direct-to-LDS load moves 16 bytes per clock and CU, a quarter of
//     what this tile needs), one chunk ahead of the ds_write, two stage buffers, one barrier per chunk.  LDS image
//     of a stage: two halves of 32 channels, rows of 64 bytes with the 16-byte pieces XOR-swizzled by
//     (row >> 1) & 3 (conflict-free fragment reads), the second half skewed by 64 bytes (conflict-free writes);
//   - v_mfma_f32_16x16x32_bf16, weights as the A operand: a lane ends up with 4 consecutive channels of one pixel;
//     the epilogue is the one of conv3x3_pipe.h (lanes 16 apart swap halves, 16-byte stores, statistics on the
//     stored layout, no barrier).
// Partial-sum rows keep the layout of the general kernel (one row per image and 16 x 16 tile, per phase in the
// forward pass), so callers size and reduce them the same way whichever kernel ran.
#pragma once
#include "conv3x3_pipe.h"  // swap16, row_sum16, u32x2_t

namespace tgemm {

constexpr int PXT = 256, COLS = 128, KC = 64;
constexpr int SUB = PXT * 64 + 64;    // one 32-channel half of a stage (+ the 64-byte skew)
constexpr int STAGE = 2 * SUB;        // 32896
#ifdef TCONV_LAB_STAMPS
constexpr int LDS_BYTES = 2 * STAGE + 512;
#else
constexpr int LDS_BYTES = 2 * STAGE;  // 65792: two workgroups per CU
#endif

enum Epi {
  EPI_STATS = 0,  // forward in training: + bias, store, (sum, sum of squares) of the rounded values per tile
  EPI_PLAIN = 1,  // + bias (if any), store
  EPI_ACC = 2,    // added to what `out` holds
  EPI_BRED = 3,   // data gradient + the BN-backward pixel sums of the layer that produced the input
};

struct Args {
  const bf16_t *in;     // [N][Hin][Win][in_ld]
  const bf16_t *wt;     // column c, reduction phase p, channel k at (c * w_row + p * wt_phase_stride + k)
  const float *bias;    // or null (forward: indexed by the channel inside a phase)
  bf16_t *out;          // [N][Hout][Wout][out_ld]
  float *stats;
  int N, Hin, Win, Ho, Wo;  // input image size; logical pixel grid the tiles cover
  int Hout, Wout;
  int nk1;              // 64-channel chunks per reduction phase
  int in_s, out_s;      // input / output pixel of logical (y, x): (y * s + i, x * s + j)
  int in_ld, out_ld, w_row;
  i64 wt_phase_stride;
  unsigned wt_bytes;
  int CoP;              // channels of one output phase (forward), all columns (data gradient)
  int col_tiles, tilesX, tiles;  // 128-column tiles; 16 x 16 pixel tiles per image row / per image
  int co_valid, xcd_order;
  i64 stats_phase_stride;
  const bf16_t *bz;     // EPI_BRED: the parent's pre-BN tensor [N][Hout][Wout][bz_ld], its records {a, b, mean, 1/sqrt(var+eps)}
  int bz_ld;
  const float4 *bcf;
};

#ifndef TCONV_LAB_ABLATE
#define TCONV_LAB_ABLATE 0  // laboratory builds only (wrong results, timing only; bits add): 1 = no pixel loads and LDS
                             // writes after the first two chunks, 2 = no weight loads after them, 4 = six pixel
                             // fragments per chunk read from LDS instead of 32
#endif
#ifdef TCONV_LAB_STAMPS  // laboratory build: shader-clock stamps of wave 0 of two workgroups, printed by every 32nd launch
__device__ int tg_lab_launches;
#define TG_STAMP(i_) do { if (tid == 0 && (i_) < 40) st[i_] = __builtin_readcyclecounter(); } while (0)
#define TG_STAMP_NEXT() do { if (tid == 0 && sti < 34) st[sti++] = __builtin_readcyclecounter(); } while (0)
#else
#define TG_STAMP(i_) do { } while (0)
#define TG_STAMP_NEXT() do { } while (0)
#endif
__device__ __forceinline__ void wait_lds_then_barrier() {
  // own LDS reads and writes are complete; vector-memory loads (the next chunk's operands) stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// KPH = reduction phases: 1 forward (the column selects the output phase), 4 data gradient.
template <int KPH, int EPI>
__global__ __launch_bounds__(256, 2) void tconv_gemm_kernel(const Args a) {
  constexpr bool FWD = KPH == 1, BRED = EPI == EPI_BRED, STATS = EPI == EPI_STATS || EPI == EPI_BRED;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kp = lane >> 4;
#ifdef TCONV_LAB_STAMPS
  long long *st = reinterpret_cast<long long *>(lds + 2 * STAGE);
  int sti = 2;
  TG_STAMP(0);
#endif

  // workgroup -> (image, pixel tile, column tile); the column tiles of one pixel tile, which read the same pixels,
  // are neighbours on one XCD (workgroup ids are dealt to the 8 XCDs in turn)
  int bid = blockIdx.x;
  if (a.xcd_order) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int ct = bid % a.col_tiles, t = bid / a.col_tiles;
  const int n = t / a.tiles, tile = t - n * a.tiles;
  const int ty = tile / a.tilesX, tx = tile - ty * a.tilesX;
  const int y0 = ty * 16, x0 = tx * 16, col0 = ct * COLS;

  // ---- pixels: 256 x 8 pieces of 16 bytes per chunk, 8 per thread: pixel (tid >> 3) + 32 i, piece tid & 7 ----
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in) + (i64)n * a.Hin * a.Win * a.in_ld, 0,
                                                     (unsigned)a.Hin * a.Win * a.in_ld * 2u, 0x00020000);
  unsigned x_voff[8];
  {
    const int px = (tid >> 3) & 15, py = tid >> 7, part = tid & 7;
    const unsigned base = (unsigned)((((y0 + py) * a.in_s) * a.Win + (x0 + px) * a.in_s) * a.in_ld + part * 8) * 2u;
    const unsigned row2 = (unsigned)(2 * a.in_s * a.Win * a.in_ld) * 2u;
#pragma unroll
    for (int i = 0; i < 8; ++i) x_voff[i] = (x0 + px < a.Wo && y0 + py + 2 * i < a.Ho) ? base + i * row2 : OOB;
  }
  const int st_off = (tid >> 3) * 64 + (((tid & 3) ^ ((tid >> 4) & 3)) * 16) + ((tid >> 2) & 1) * SUB;  // + i * 2048

  // ---- weights: the lane's fragment rows of the wave's two 16-column blocks ----
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.wt), 0, a.wt_bytes, 0x00020000);
  const unsigned w_base = (unsigned)(((col0 + wave * 32 + l15) * a.w_row + kp * 8) * 2);
  const unsigned w_cf = (unsigned)(16 * a.w_row * 2);

  i32x4 xr[8], wq[2][4];  // staged pixels of a chunk; weight fragments [chunk parity][block * 2 + half]
  // Requests are unconditional -- a chunk past the last one addresses beyond the descriptors, moves nothing and
  // arrives as zeros -- so that every trip of the main loop has the same vector-memory operations in flight and the
  // compiler's counted waits (s_waitcnt vmcnt) leave the younger chunks' loads alone.
  const int nk = KPH * a.nk1;
  int ld_kc = 0, ld_kin = 0, ld_kph = 0;  // the next chunk to request
  auto request = [&](i32x4 *xdst, i32x4 *wdst) __attribute__((always_inline)) {
    const bool live = ld_kc < nk;
    const int kk = ld_kin * KC;
    unsigned xs = (unsigned)kk * 2u, ws = (unsigned)kk * 2u;
    if (KPH > 1) {
      xs += (unsigned)(((ld_kph >> 1) * a.Win + (ld_kph & 1)) * a.in_ld) * 2u;
      ws += (unsigned)(ld_kph * a.wt_phase_stride) * 2u;
    }
#if TCONV_LAB_ABLATE & 1
    if (ld_kc < 2)
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) xdst[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, live ? x_voff[i] : OOB, xs, 0);
#if TCONV_LAB_ABLATE & 2
    if (ld_kc < 2)
#endif
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        wdst[cf * 2 + h] = __builtin_amdgcn_raw_buffer_load_b128(wrs, live ? w_base + cf * w_cf : OOB, ws + h * 64, 0);
    ++ld_kc;
    if (++ld_kin == a.nk1) ld_kin = 0, ++ld_kph;
  };
  auto store_x = [&](int buf, const i32x4 *xsrc) __attribute__((always_inline)) {
#if TCONV_LAB_ABLATE & 1
    if (ld_kc > 2) return;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<i32x4 *>(lds + buf * STAGE + st_off + i * 2048) = xsrc[i];
  };

  f32x4 acc[2][16];  // [16-column block][pixel row of the tile]
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int pf = 0; pf < 16; ++pf) acc[cf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag = l15 * 64 + ((kp ^ ((l15 >> 1) & 3)) * 16);  // pixel l15 of a row, 8-channel piece kp
  // A chunk on the matrix cores: 32 pixel fragments (2 halves x 16 rows), two MFMAs each.  The fragments are read
  // RING steps ahead of their use, in program order (the scheduler left to itself reads a pair and waits for it).
  constexpr int RING = 6;  // (8 measured the same, and puts the fused-reduction variant into spills)
  auto compute = [&](int buf, const i32x4 *w) __attribute__((always_inline)) {
    const unsigned char *st = lds + buf * STAGE + frag;
    bf16x8 b[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) b[i] = *reinterpret_cast<const bf16x8 *>(st + (i >> 4) * SUB + (i & 15) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int h = i >> 4, pf = i & 15;
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
        acc[cf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[cf * 2 + h]), b[i % RING], acc[cf][pf], 0, 0, 0);
      if (i + RING < 32 && !(TCONV_LAB_ABLATE & 4))
        b[i % RING] = *reinterpret_cast<const bf16x8 *>(st + ((i + RING) >> 4) * SUB + ((i + RING) & 15) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop: chunk kc is computed from stage buffer kc & 1 with the weight fragments wq[kc & 1]; behind its
  // MFMAs the pixels of chunk kc + 1 (requested a chunk ago) are written to the other buffer and chunk kc + 2 is
  // requested into the registers that frees.  Measured and not kept: a second set of staging registers with requests
  // two chunks ahead and one workgroup per CU (the same time on the layers with many channels, slower on the
  // others), and eight waves per workgroup on the same tile for the launches of one workgroup per CU (8 % slower).
  // Those launches do not wait for latency: without its pixel loads the 1024 -> 512 layer at 16 x 16 takes 19.9 us
  // instead of 26.7, with nothing but its MFMAs 17.6 -- what is left is the traffic between L2 and the CUs, 16 column
  // tiles fetching every pixel tile (TCONV_LAB_ABLATE, TCONV_LAB_STAMPS builds). ----
  request(xr, wq[0]);
  store_x(0, xr);
  request(xr, wq[1]);
  wait_lds_then_barrier();
  TG_STAMP(1);
  for (int kc = 0; kc < nk; kc += 2) {
    compute(0, wq[0]);
    store_x(1, xr);
    request(xr, wq[0]);
    wait_lds_then_barrier();
    TG_STAMP_NEXT();
    if (kc + 1 < nk) compute(1, wq[1]);
    store_x(0, xr);
    request(xr, wq[1]);
    wait_lds_then_barrier();
    TG_STAMP_NEXT();
  }

  // ---- epilogue: the wave's 256 pixels x 32 columns ----
  const int cwave = col0 + wave * 32;
  int phase = 0, cow = cwave;  // output phase and channel inside it (forward: CoP is a multiple of 32)
  if (FWD) {
    phase = cwave / a.CoP;
    cow = cwave - phase * a.CoP;
  }
  const int ch8 = (kp & 1) * 16 + (kp >> 1) * 8;  // first of the lane's 8 channels after the swap
  const auto ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (i64)n * a.Hout * a.Wout * a.out_ld, 0,
                                                     (unsigned)a.Hout * a.Wout * a.out_ld * 2u, 0x00020000);
  const int x = x0 + l15;
  const unsigned o_base = (unsigned)((((y0 * a.out_s + (phase >> 1)) * a.Wout) + x * a.out_s + (phase & 1)) * a.out_ld + cow + ch8) * 2u;
  const unsigned o_row = (unsigned)(a.out_s * a.Wout * a.out_ld) * 2u;
  f32x4 biasv[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = cow + m * 16 + 4 * kp + j;
      biasv[m][j] = (!BRED && a.bias && co < a.co_valid) ? a.bias[co] : 0.f;
    }
  float4 bcf[BRED ? 8 : 1];
  // (a descriptor of no extent in the variants without the fused reduction: nothing is loaded through it)
  const auto zrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t *>(BRED ? a.bz : a.in) + (BRED ? (i64)n * a.Hout * a.Wout * a.bz_ld : 0), 0,
      BRED ? (unsigned)a.Hout * a.Wout * a.bz_ld * 2u : 0u, 0x00020000);
  unsigned z_base = 0, z_row = 0;
  if (BRED) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bcf[BRED ? j : 0] = a.bcf[cwave + ch8 + j];
    z_base = (unsigned)((y0 * a.Wout + x) * a.bz_ld + cwave + ch8) * 2u;
    z_row = (unsigned)(a.Wout * a.bz_ld) * 2u;
  }
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
  const bool inside = y0 + 16 <= a.Ho && x0 + 16 <= a.Wo;  // no masks in the statistics of a whole tile

  // rows in two batches of 8: the fused reduction's z values (and the accumulate variant's old values) of a
  // batch are requested together, ahead of its conversions
  auto rows = [&](const bool MASKED) __attribute__((always_inline)) {
#pragma unroll
  for (int b0 = 0; b0 < 16; b0 += 8) {
    i32x4 zq[BRED ? 8 : 1], oq[EPI == EPI_ACC ? 8 : 1];
    if (BRED) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = x < a.Wo && y0 + b0 + q < a.Ho;
        zq[BRED ? q : 0] = __builtin_amdgcn_raw_buffer_load_b128(zrs, ok ? z_base + (b0 + q) * z_row : OOB, 0, 0);
      }
    }
    if (EPI == EPI_ACC) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = x < a.Wo && y0 + b0 + q < a.Ho;
        oq[EPI == EPI_ACC ? q : 0] = __builtin_amdgcn_raw_buffer_load_b128(ors, ok ? o_base + (b0 + q) * o_row : OOB, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int pf = b0 + q;
      const bool ok = x < a.Wo && y0 + pf < a.Ho;
      pipe::u32x2_t hw[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 v = acc[m][pf];
        if (!BRED) v += biasv[m];
        hw[m] = __builtin_bit_cast(pipe::u32x2_t, __builtin_convertvector(v, bf16x4));
      }
      i32x4 sw = pipe::swap16(hw);
      if (EPI == EPI_ACC) {  // + what the output holds (same layout), rounded again
        const i32x4 o = oq[EPI == EPI_ACC ? q : 0];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned v0 = (unsigned)sw[d], v1 = (unsigned)o[d];
          const float lo = bf16_to_f32(v0 & 0xffffu) + bf16_to_f32(v1 & 0xffffu);
          const float hi = bf16_to_f32(v0 >> 16) + bf16_to_f32(v1 >> 16);
          sw[d] = (int)((unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16));
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(sw, ors, ok ? o_base + pf * o_row : OOB, 0, 0);
      if (STATS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned w = (unsigned)sw[j >> 1];
          float r = (j & 1) ? bf16_to_f32(w >> 16) : bf16_to_f32(w & 0xffffu);  // statistics of the ROUNDED values
          if (MASKED && !ok) r = 0.f;
          if (BRED) {
            const unsigned zz = (unsigned)zq[BRED ? q : 0][j >> 1];
            const float zv = (j & 1) ? bf16_to_f32(zz >> 16) : bf16_to_f32(zz & 0xffffu);
            const float4 c = bcf[BRED ? j : 0];
            const float g = fmaf(c.x, zv, c.y) > 0.f ? r : 0.f;
            ssum[j] = fmaf(g, zv, ssum[j]);  // sum g*z; turned into sum g*xhat below
            ssq[j] += g;
          } else {
            ssum[j] += r;
            ssq[j] += r * r;
          }
        }
      }
    }
  }
  };
  if (inside)
    rows(false);
  else
    rows(true);
  if (STATS) {
    // over the 16 pixel lanes of a row (DPP rotations); the lane of pixel 0 writes its 8 channels' two sums
    float *row = FWD ? a.stats + phase * a.stats_phase_stride + ((i64)(n * a.tiles + tile) * a.CoP + cow + ch8) * 2
                     : a.stats + ((i64)(n * a.tiles + tile) * a.CoP + cwave + ch8) * 2;
    float res[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = ssum[j], q = ssq[j];
      if (BRED) s = keep_scalar((keep_scalar(s) - bcf[BRED ? j : 0].z * q) * bcf[BRED ? j : 0].w);  // sum g*xhat
      res[2 * j] = pipe::row_sum16(s);
      res[2 * j + 1] = pipe::row_sum16(q);
    }
    if (l15 == 0) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
        *reinterpret_cast<float4 *>(row + 4 * d) = make_float4(res[4 * d], res[4 * d + 1], res[4 * d + 2], res[4 * d + 3]);
    }
  }
#ifdef TCONV_LAB_STAMPS
  TG_STAMP(35);
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101)) {
    int l = 0;
    if (blockIdx.x == 0) l = atomicAdd(&tg_lab_launches, 1);
    if (blockIdx.x != 0 || l % 32 == 8) {
      printf("tg wg %d KPH %d EPI %d nk %d: prologue %lld epilogue %lld total %lld | chunks",
             (int)blockIdx.x, KPH, EPI, nk, st[1] - st[0], st[35] - st[sti - 1], st[35] - st[0]);
      for (int i = 2; i < sti; ++i) printf(" %lld", st[i] - st[i - 1]);
      printf("\n");
    }
  }
#endif
}

#undef TG_STAMP
#undef TG_STAMP_NEXT
inline bool lab_off() {
  static const bool off = getenv("UNET_LAB_NO_TCONV_GEMM") != nullptr;  // laboratory switch: everything on the general kernel
  return off;
}

template <int KPH, int EPI>
int launch(const Args &a, int grid, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&tconv_gemm_kernel<KPH, EPI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set[dev & 63] = true;
  }
  hipLaunchKernelGGL((tconv_gemm_kernel<KPH, EPI>), dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a);
  UNET_LAUNCH_CHECK("tconv_gemm_kernel");
  return UNET_OK;
}

}  // namespace tgemm
