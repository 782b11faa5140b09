// bf16 weight gradient of the 3x3 convolution and of the 2x2 transposed convolution on the gfx950
// matrix cores.
//
// dW[tap][co][ci] = sum_pixel dZ[pixel][co] * X[pixel + off(tap)][ci]
//   - the reduction runs over pixels, which is the strided dimension of both NHWC operands;
//     ds_read_b64_tr_b16 (the transposing LDS read of gfx950) delivers the [pixel][channel] LDS
//     tiles as K-contiguous MFMA fragments;
//   - a workgroup owns a 64(co) x 64(ci) block of all filter taps and a slice of the pixels
//     (split-K); partial blocks go to a workspace and a second kernel adds the slices in a fixed
//     order (deterministic) while converting to the reference's [C_in][C_out][k][k] fp32 layout.
// This file is compiled with -mllvm -amdgpu-mfma-vgpr-form: the 9 accumulator tiles of a wave stay
// in ordinary vector registers.  Without it the compiler copies all 144 accumulator registers
// between the two halves of the register file on every pass of the pixel loop.
#include <stdlib.h>

#include "bf16_util.h"

namespace {

// ---------------------------------------------------------------------------
// Weight gradient.
// ---------------------------------------------------------------------------
struct WgradArgs {
  const bf16_t *dz;  // [N][Hz][Wz][dz_ld], CoP channels used
  const bf16_t *x;   // [N][H][W][x_ld], CiP channels used
  float *partial;    // [splits][TAPS][CoP][CiP] fp32
  int N, H, W;       // logical pixel grid = x image size
  int Hz, Wz;        // dz image size
  int dz_ld, x_ld, CoP, CiP;  // partial-buffer dims (multiples of 64)
  int z_c, x_c;               // channels present in dz / x (multiples of 8)
  int dz_s, dz_oy, dz_ox;  // dz pixel of logical (y,x): (y*dz_s+dz_oy, x*dz_s+dz_ox)
  int chunksX, chunksY;    // 8x16 pixel chunks per image
  int chunks_per_split, total_chunks;
  int phases;  // TAPS == 1 only: 4 = blockIdx.x also selects the (i,j) phase of a 2x2 transposed conv
};

constexpr int WP = 192;  // LDS row pitch (bytes) of a 64-channel pixel row: 4 rows x 64 B fall in 4 distinct bank quarters

// The kernel is written to spend its issue slots on MFMAs, not on address arithmetic:
//   - operands are fetched with buffer loads: one descriptor per image, a 32-bit per-lane byte offset,
//     and out-of-range pixels / channels get an offset beyond the descriptor, which the hardware
//     answers with zeros (no branches, no selects);
//   - the per-lane offsets inside a chunk are computed once; a chunk adds one scalar;
//   - the accumulator tiles stay in ordinary vector registers (see the note on compile flags above).
// PH = 4 (TAPS == 1): the four (i, j) phases of the 2x2 transposed convolution in ONE workgroup -- the x tile of a
// chunk is fetched and read from LDS once and multiplies the dz tiles of the four phases (four accumulator blocks),
// 32 MFMAs per wave between two barriers instead of 8.  One phase per workgroup (the phase taken from blockIdx.x)
// ran at the pace of its loads: 32 KB of operands per 8 MFMAs of a wave.
template <int TAPS, int PH = 1>
__global__ __launch_bounds__(256, 1) void wgrad_mfma_kernel(const WgradArgs a) {
  static_assert(PH == 1 || TAPS == 1, "phases are a variant of the single-tap kernel");
  constexpr int XR = (TAPS == 9) ? 10 : 8, XC = (TAPS == 9) ? 18 : 16;
  constexpr int XPIX = XR * XC, ZPIX = 128;
  constexpr int XCH = XPIX * 8, XI = (XCH + 255) / 256;
  constexpr int ZCH = ZPIX * 8, ZI = ZCH / 256;
  constexpr int NACC = TAPS * PH;
  constexpr unsigned OOB = 0xffffffffu;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *ldsZ = lds, *ldsX = lds + PH * ZPIX * WP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  int split = blockIdx.x, phase = 0, dz_oy = a.dz_oy, dz_ox = a.dz_ox;
  if (TAPS == 1 && PH == 1 && a.phases > 1) {
    phase = split & 3;
    split >>= 2;
    dz_oy = phase >> 1;
    dz_ox = phase & 1;
  }
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = grp >> 1;
  const int z_lane = (8 * h + q) * WP + (wm * 32 + 16 * (grp & 1) + 4 * p) * 2;
  const int x_lane = (8 * h + q) * WP + (wn * 32 + 16 * (grp & 1) + 4 * p) * 2;

  const int c_begin = split * a.chunks_per_split;
  const int c_end = min(a.total_chunks, c_begin + a.chunks_per_split);
  const int per_img = a.chunksX * a.chunksY;

  // per-lane geometry of the staged 16-byte pieces, fixed for the whole kernel
  const int part = tid & 7;
  const bool x_ch_ok = ci0 + part * 8 < a.x_c, z_ch_ok = co0 + part * 8 < a.z_c;
  int x_py[XI], x_px[XI];
  unsigned x_rel[XI], z_rel[ZI];
  int z_py[ZI], z_px[ZI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int pix = (tid + 256 * i) >> 3;
    x_py[i] = pix / XC - (TAPS == 9);
    x_px[i] = pix - (pix / XC) * XC - (TAPS == 9);
    if (!(XCH % 256 == 0 || tid + 256 * i < XCH) || !x_ch_ok) x_py[i] = -100000;  // never in range
    x_rel[i] = (unsigned)(((x_py[i] * a.W + x_px[i]) * a.x_ld + ci0 + part * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int pix = (tid + 256 * i) >> 3;
    z_py[i] = z_ch_ok ? (pix >> 4) : -100000;
    z_px[i] = pix & 15;
    z_rel[i] = (unsigned)((((z_py[i] * a.dz_s + dz_oy) * a.Wz + z_px[i] * a.dz_s + dz_ox) * a.dz_ld + co0 + part * 8) * 2);
  }
  const unsigned x_img_bytes = (unsigned)a.H * a.W * a.x_ld * 2u, z_img_bytes = (unsigned)a.Hz * a.Wz * a.dz_ld * 2u;
  const int st_off = (tid >> 3) * WP + (tid & 7) * 16;  // piece i sits 32 pixel rows further

  // PH > 1: phase p = 2 i + j reads the dz pixel (i, j) further (a scalar offset of the load)
  unsigned ph_off[PH];
#pragma unroll
  for (int p = 0; p < PH; ++p) ph_off[p] = (unsigned)((((p >> 1) * a.Wz + (p & 1)) * a.dz_ld) * 2);
  i32x4 xr[XI], zr[PH][ZI];
  auto load_chunk = [&](int chunk) {
    const int n = chunk / per_img, r = chunk - n * per_img;
    const int cy = r / a.chunksX;
    const int y0 = cy * 8, x0 = (r - cy * a.chunksX) * 16;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.x) + (i64)n * a.H * a.W * a.x_ld, 0,
                                                       x_img_bytes, 0x00020000);
    const auto zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.dz) + (i64)n * a.Hz * a.Wz * a.dz_ld, 0,
                                                       z_img_bytes, 0x00020000);
    const unsigned x_org = (unsigned)((y0 * a.W + x0) * a.x_ld * 2);
    const unsigned z_org = (unsigned)((y0 * a.dz_s * a.Wz + x0 * a.dz_s) * a.dz_ld * 2);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const bool ok = (unsigned)(y0 + x_py[i]) < (unsigned)a.H && (unsigned)(x0 + x_px[i]) < (unsigned)a.W;
      xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? x_org + x_rel[i] : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const bool ok = (unsigned)(y0 + z_py[i]) < (unsigned)a.H && (unsigned)(x0 + z_px[i]) < (unsigned)a.W;
#pragma unroll
      for (int p = 0; p < PH; ++p)
        zr[p][i] = __builtin_amdgcn_raw_buffer_load_b128(zrs, ok ? z_org + z_rel[i] : OOB, PH > 1 ? ph_off[p] : 0, 0);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (XCH % 256 == 0 || tid + 256 * i < XCH) *reinterpret_cast<i32x4 *>(ldsX + st_off + i * 32 * WP) = xr[i];
#pragma unroll
    for (int p = 0; p < PH; ++p)
#pragma unroll
      for (int i = 0; i < ZI; ++i) *reinterpret_cast<i32x4 *>(ldsZ + p * ZPIX * WP + st_off + i * 32 * WP) = zr[p][i];
  };

  load_chunk(c_begin);
  store_chunk();
  __syncthreads();
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    if (chunk + 1 < c_end) load_chunk(chunk + 1);
    if (TAPS == 9) {
      // Walk the 10 halo rows of the chunk: the x fragment of (row r, filter column kw) is read once and
      // multiplies the dz fragments of pixel rows r, r-1, r-2 (filter rows 0, 1, 2) -- 30 + 8 fragment
      // reads per chunk instead of 72 + 8.  Each accumulator still sums its pixel rows in ascending order.
      bf16x8 aw[3];
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        if (r < 8) {
          const unsigned char *zp = ldsZ + r * 16 * WP + z_lane;
          aw[r % 3] = tr_read_pair(zp, zp + 4 * WP);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const unsigned char *xp = ldsX + (r * XC + kw) * WP + x_lane;
          const bf16x8 bfr = tr_read_pair(xp, xp + 4 * WP);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int ks = r - kh;
            if (ks >= 0 && ks < 8) {
#ifdef WGRAD_LAB_NO_MFMA  // tools/corun_lab.sh: the same operands through VALU instead of the matrix cores
              acc[(TAPS == 9) ? kh * 3 + kw : 0][0] += (float)aw[ks % 3][0] * (float)bfr[0];
#else
              acc[(TAPS == 9) ? kh * 3 + kw : 0] =
                  __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[ks % 3], bfr, acc[(TAPS == 9) ? kh * 3 + kw : 0], 0, 0, 0);
#endif
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const unsigned char *xp = ldsX + ks * XC * WP + x_lane;
        const bf16x8 bfr = tr_read_pair(xp, xp + 4 * WP);
#pragma unroll
        for (int p = 0; p < PH; ++p) {
          const unsigned char *zp = ldsZ + p * ZPIX * WP + ks * 16 * WP + z_lane;
          const bf16x8 af = tr_read_pair(zp, zp + 4 * WP);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[p], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (chunk + 1 < c_end) {
      store_chunk();
      __syncthreads();
    }
  }
  // partial[split][tap][co][ci] through one descriptor per split slice and 32-bit offsets
  const int slices = (TAPS == 1 && a.phases > 1) ? a.phases : TAPS;  // tap slices per split in the partial buffer
  const auto prs = __builtin_amdgcn_make_buffer_rsrc(a.partial + ((i64)split * slices + phase) * a.CoP * a.CiP, 0,
                                                     (unsigned)NACC * a.CoP * a.CiP * 4u, 0x00020000);
  const unsigned row_bytes = (unsigned)a.CiP * 4u;
  unsigned off = ((unsigned)(co0 + wm * 32 + 4 * (lane >> 5)) * a.CiP + ci0 + wn * 32 + (lane & 31)) * 4u;
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    unsigned o = off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // register r holds row (r&3) + 8*(r>>2) (+ 4*(lane>>5))
      const float v = acc[t][r];
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), prs, o, 0, 0);
      o += ((r & 3) == 3) ? 5 * row_bytes : row_bytes;
    }
    off += (unsigned)a.CoP * row_bytes;
  }
}

// ---------------------------------------------------------------------------
// The same weight gradient with the operands going memory -> LDS directly (buffer_load ... lds) two chunks
// ahead of the MFMAs, one barrier per chunk.  wgrad_mfma_kernel above stages a chunk through registers
// (load, MFMA loop, barrier, ds_write, barrier) with the next chunk's loads in flight only while one chunk is
// computed -- about a microsecond, less than a memory round trip under load, with one wave per SIMD to hide it.
// Same MFMA loop, same fragment values, same order of additions: the partial blocks are bit-identical.
//
// LDS image of a stage: pixel rows of 128 bytes (64 channels), dense; the 64-byte halves of a row are swapped
// when bit 1 of the row index is set, so that the 4 rows x 64 bytes a transposing read touches per half wave
// cover the 64 banks once (the register-staged kernel pads rows to 192 bytes instead, which a lane-linear
// LDS load cannot produce).  The swap is applied to the SOURCE address: wave instruction i fills bytes
// [1024 i, 1024 i + 1024), lane l's 16 bytes are row 8 i + l / 8, slot l % 8, and hold source piece
// slot ^ (4 * ((row >> 1) & 1)).  Out-of-image pixels and absent channels address beyond the descriptor and
// arrive as zeros.
constexpr int DROW = 128;
typedef __attribute__((address_space(3))) void *wg_lds_ptr_t;

// One LDS load instruction: 64 lanes x 16 bytes from the buffer `rsrc` (lane offset `voffset`, beyond the
// descriptor = zeros) to the 1 KiB of LDS at byte address `lds_addr`, lane-linear.  Issued as assembly and not
// through the builtin: the compiler then does not know of an outstanding LDS write and does not put
// s_waitcnt vmcnt(0) in front of the transposing LDS reads of the chunk being computed (it cannot tell that
// they read another stage buffer) -- which would serialise the loads two chunks ahead with the MFMAs.  The
// kernel counts its vector-memory operations itself (wg_wait_vmem_then_barrier).
__device__ __forceinline__ void wg_lds_load_b128(i32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voffset), "s"(rsrc));
}
__device__ __forceinline__ i32x4 wg_make_rsrc(const void *base, unsigned bytes) {
  const uint64_t b = (uint64_t)base;
  return i32x4{(int)(unsigned)b, (int)(unsigned)(b >> 32), (int)bytes, 0x00020000};
}

template <int N>
__device__ __forceinline__ void wg_wait_vmem_then_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

#ifdef WGRAD_LAB_STAMPS  // laboratory build: shader-clock stamps of wave 0, printed by two workgroups of every 64th launch
__device__ int wg_lab_launches;
#define WG_STAMP(i_) do { if (tid == 0 && (i_) < 36 + 12) st[i_] = __builtin_readcyclecounter(); } while (0)
#else
#define WG_STAMP(i_) do { } while (0)
#endif
template <int TAPS, int STAGES>
__global__ __launch_bounds__(256, 1) void wgrad_dma_kernel(const WgradArgs a) {
  constexpr int XR = (TAPS == 9) ? 10 : 8, XC = (TAPS == 9) ? 18 : 16;
  constexpr int XPIX = XR * XC;
  constexpr int Z_INSTR = 16, X_INSTR = (TAPS == 9) ? 24 : 16;  // wave instructions of 8 pixel rows each
  constexpr int PER_WAVE = (Z_INSTR + X_INSTR) / 4;               // every wave issues the same number
  constexpr int Z_BYTES = Z_INSTR * 1024, STAGE = (Z_INSTR + X_INSTR) * 1024;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // three stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WGRAD_LAB_STAMPS
  unsigned long long *st = reinterpret_cast<unsigned long long *>(lds + 3 * 40 * 1024);
  WG_STAMP(0);
#endif
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  int split = blockIdx.x, phase = 0, dz_oy = a.dz_oy, dz_ox = a.dz_ox;
  if (TAPS == 1 && a.phases > 1) {
    phase = split & 3;
    split >>= 2;
    dz_oy = phase >> 1;
    dz_ox = phase & 1;
  }
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment addresses: logical (row, byte column) of wgrad_mfma_kernel, physical = row * 128 + (column ^ swap)
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = grp >> 1;
  const int z_col = (wm * 32 + 16 * (grp & 1) + 4 * p) * 2, x_col = (wn * 32 + 16 * (grp & 1) + 4 * p) * 2;
  const int z_lane = (8 * h + q) * DROW + (z_col ^ (((q >> 1) & 1) << 6));  // base rows of dz are multiples of 16
  int x_lane[4];  // by (first row of the fragment) & 3
#pragma unroll
  for (int v = 0; v < 4; ++v) x_lane[v] = (8 * h + q) * DROW + (x_col ^ ((((v + q) >> 1) & 1) << 6));

  const int c_begin = split * a.chunks_per_split;
  const int c_end = min(a.total_chunks, c_begin + a.chunks_per_split);
  const int per_img = a.chunksX * a.chunksY;

  // per-lane source plan of this wave's PER_WAVE instructions (fixed for the whole kernel): byte offset
  // relative to the chunk origin and pixel column relative to the chunk's first.  A piece that does not exist
  // (row beyond the halo box, channel beyond the tensor) gets a column that is never inside an image.
  int rel[PER_WAVE], pxk[PER_WAVE];
#pragma unroll
  for (int k = 0; k < PER_WAVE; ++k) {
    const int i = wave + 4 * k;
    const bool is_z = k < Z_INSTR / 4;  // the same for all four waves: known at compile time
    const int row = 8 * (is_z ? i : i - Z_INSTR) + (lane >> 3);
    const int piece = (lane & 7) ^ (((row >> 1) & 1) << 2);
    if (is_z) {
      const int py = row >> 4, px = row & 15;
      const bool ok = co0 + piece * 8 < a.z_c;
      rel[k] = (((py * a.dz_s + dz_oy) * a.Wz + px * a.dz_s + dz_ox) * a.dz_ld + co0 + piece * 8) * 2;
      pxk[k] = ok ? px : -(1 << 24);
    } else {
      const int py = row / XC - (TAPS == 9), px = row - (row / XC) * XC - (TAPS == 9);
      const bool ok = row < XPIX && ci0 + piece * 8 < a.x_c;
      rel[k] = ((py * a.W + px) * a.x_ld + ci0 + piece * 8) * 2;
      pxk[k] = ok ? px : -(1 << 24);
    }
  }
  const unsigned x_img_bytes = (unsigned)a.H * a.W * a.x_ld * 2u, z_img_bytes = (unsigned)a.Hz * a.Wz * a.dz_ld * 2u;

  // The chunk whose loads are issued next: its image and chunk coordinates advance by one per call (no
  // divisions in the loop).  Past the end of the slice the loads still go out, addressed beyond the
  // descriptor (zeros into a stage buffer nobody reads): every trip of the main loop is the same straight
  // line of code, which lets the compiler spread the address arithmetic between the MFMAs.
  int i_chunk = c_begin, i_n = c_begin / per_img, i_cy = (c_begin - i_n * per_img) / a.chunksX,
      i_cx = c_begin - i_n * per_img - i_cy * a.chunksX;
  // plan_chunk() works out the addresses of the chunk to be issued next, issue_chunk() is then ten load
  // instructions: the prologue's form (the main loop has its own, in pieces, below).
  unsigned vo[PER_WAVE];
  i32x4 xrs, zrs;
  // The same in two steps for the main loop, which spreads the arithmetic over a chunk's rows of MFMAs like the
  // loads themselves: plan_begin() (scalar: descriptors, origins, the walk), then plan_piece(k) per load into
  // vo_next; plan_commit() makes it the plan of the next chunk to be issued.
  unsigned vo_next[PER_WAVE];
  i32x4 xrs_next, zrs_next;
  int p_x_org = 0, p_z_org = 0, p_x0_test = 0;
  auto plan_begin = [&]() {
    const int y0 = i_cy * 8, x0 = i_cx * 16;
    p_x0_test = (i_chunk < c_end) ? x0 : -(1 << 24);
    xrs_next = wg_make_rsrc(a.x + (i64)i_n * a.H * a.W * a.x_ld, x_img_bytes);
    zrs_next = wg_make_rsrc(a.dz + (i64)i_n * a.Hz * a.Wz * a.dz_ld, z_img_bytes);
    p_x_org = (y0 * a.W + x0) * a.x_ld * 2;
    p_z_org = (y0 * a.dz_s * a.Wz + x0 * a.dz_s) * a.dz_ld * 2;
    ++i_chunk;
    const bool row_end = ++i_cx == a.chunksX;
    i_cx = row_end ? 0 : i_cx;
    i_cy += row_end;
    const bool img_end = i_cy == a.chunksY;
    i_cy = img_end ? 0 : i_cy;
    i_n += img_end;
  };
  auto plan_piece = [&](const int k) {
    const unsigned off = (unsigned)((k < Z_INSTR / 4 ? p_z_org : p_x_org) + rel[k]);
    vo_next[k] = ((unsigned)(p_x0_test + pxk[k]) < (unsigned)a.W) ? off : OOB;
  };
  auto plan_commit = [&]() {
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) vo[k] = vo_next[k];
    xrs = xrs_next;
    zrs = zrs_next;
  };
  auto plan_chunk = [&]() {
    // Only the pixel column is tested here.  A pixel row above the image makes the offset negative, one below
    // it puts the offset past the image: both are beyond the descriptor (which covers one image of less than
    // 2^30 bytes) and read as zeros.  A chunk past the end of the slice gets a column origin far outside.
    const int y0 = i_cy * 8, x0 = i_cx * 16;
    const int x0_test = (i_chunk < c_end) ? x0 : -(1 << 24);
    xrs = wg_make_rsrc(a.x + (i64)i_n * a.H * a.W * a.x_ld, x_img_bytes);
    zrs = wg_make_rsrc(a.dz + (i64)i_n * a.Hz * a.Wz * a.dz_ld, z_img_bytes);
    const int x_org = (y0 * a.W + x0) * a.x_ld * 2;
    const int z_org = (y0 * a.dz_s * a.Wz + x0 * a.dz_s) * a.dz_ld * 2;
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
      const unsigned off = (unsigned)((k < Z_INSTR / 4 ? z_org : x_org) + rel[k]);
      vo[k] = ((unsigned)(x0_test + pxk[k]) < (unsigned)a.W) ? off : OOB;
    }
    ++i_chunk;
    const bool row_end = ++i_cx == a.chunksX;
    i_cx = row_end ? 0 : i_cx;
    i_cy += row_end;
    const bool img_end = i_cy == a.chunksY;
    i_cy = img_end ? 0 : i_cy;
    i_n += img_end;
  };
  // one of the chunk's PER_WAVE load instructions (the main loop spreads them over a chunk's rows of MFMAs: issued
  // in one block they cost the wave, which is alone on its SIMD, some 600 cycles with nothing on the matrix pipe)
  auto issue_piece = [&](unsigned char *dst, const int k) {
#ifdef WGRAD_LAB_NO_LOADS
    if (i_chunk > c_begin + 3) return;
#endif
    const unsigned dst_addr = (unsigned)(uintptr_t)(wg_lds_ptr_t)dst + wave * 1024;
    wg_lds_load_b128(k < Z_INSTR / 4 ? zrs : xrs, dst_addr + k * 4096, vo[k]);
  };
  auto issue_chunk = [&](unsigned char *dst) {
#ifdef WGRAD_LAB_NO_LOADS  // laboratory build: only the first three chunks are fetched (what the loads cost the loop)
    if (i_chunk > c_begin + 3) return;
#endif
    const unsigned dst_addr = (unsigned)(uintptr_t)(wg_lds_ptr_t)dst + wave * 1024;
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k)  // instruction wave + 4 k: dz for k < 4 whatever the wave
      wg_lds_load_b128(k < Z_INSTR / 4 ? zrs : xrs, dst_addr + k * 4096, vo[k]);
  };

  static_assert(STAGES == 3 && TAPS == 9, "the loop below keeps two chunks of a 3x3 filter in flight");
  // One wave per SIMD: nobody else's MFMAs cover this wave's LDS latency, so the fragments of halo row r + 1
  // are read while the MFMAs of row r run (four dz fragments and two sets of x fragments live), and that
  // pipeline does not stop at a chunk's end.  The one barrier of a chunk stands in front of its last halo
  // row: by then a wave has read all of the chunk's fragments and its own pieces of the next chunk have
  // landed (the chunk after that may be in flight); behind the barrier that holds for every wave, so the
  // next chunk's first row can be read, and this chunk's buffer is free for the loads of the chunk three
  // ahead -- which the next trip issues one per row of MFMAs, with their address arithmetic.
  plan_chunk();
  issue_chunk(lds);
  plan_chunk();
  issue_chunk(lds + STAGE);
  plan_chunk();
  bf16x8 aw[4], bq[2][3];
  auto read_row = [&](const unsigned char *stage, int rn) {
    if (rn < 8) {
      const unsigned char *zp = stage + rn * 16 * DROW + z_lane;
      aw[rn & 3] = tr_read_pair(zp, zp + 4 * DROW);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const unsigned char *xp = stage + Z_BYTES + (rn * XC + kw) * DROW + x_lane[(rn * XC + kw) & 3];
      bq[rn & 1][kw] = tr_read_pair(xp, xp + 4 * DROW);
    }
  };
  wg_wait_vmem_then_barrier<PER_WAVE>();
  WG_STAMP(1);
  read_row(lds, 0);
  int buf = 0;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    if (chunk - c_begin < 36) WG_STAMP(4 + (chunk - c_begin));
    const unsigned char *cur = lds + buf * STAGE;
    const int nb = buf == 2 ? 0 : buf + 1;
    auto mfma_row = [&](int r) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ks = r - kh;
          if (ks >= 0 && ks < 8)
            acc[kh * 3 + kw] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[ks & 3], bq[r & 1][kw], acc[kh * 3 + kw], 0, 0, 0);
        }
    };
    // the loads of the chunk after the next go out one per row, into the buffer the previous chunk was read from
    // (free since that chunk's barrier)
    unsigned char *const dst2 = lds + (buf == 0 ? 2 : buf - 1) * STAGE;
    plan_begin();
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      read_row(cur, r + 1);
      issue_piece(dst2, r);
      if (r == 8) issue_piece(dst2, 9);
      plan_piece(r);
      if (r == 8) plan_piece(9);
      mfma_row(r);
    }
    // the order above, pinned for the scheduler: reads of row r + 1, then the MFMAs of row r
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      if (r + 1 >= 8)
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      else
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      if (r == 0) __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      if (r == 1 || r == 8) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      if (r >= 2 && r < 8) __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // the barrier stays behind the MFMAs of row 8, which cover the reads of row 9
#ifdef WGRAD_LAB_STAMPS
    if (chunk == c_begin + 10) WG_STAMP(40);
#endif
    wg_wait_vmem_then_barrier<PER_WAVE>();
#ifdef WGRAD_LAB_STAMPS
    if (chunk == c_begin + 10) WG_STAMP(41);
#endif
    read_row(lds + nb * STAGE, 0);
#ifdef WGRAD_LAB_STAMPS
    if (chunk == c_begin + 10) WG_STAMP(42);
#endif
    mfma_row(9);
    plan_commit();
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    buf = nb;
  }
  WG_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads issued past the end of the slice
  const int slices = (TAPS == 1 && a.phases > 1) ? a.phases : TAPS;
  const auto prs = __builtin_amdgcn_make_buffer_rsrc(a.partial + ((i64)split * slices + phase) * a.CoP * a.CiP, 0,
                                                     (unsigned)TAPS * a.CoP * a.CiP * 4u, 0x00020000);
  const unsigned row_bytes = (unsigned)a.CiP * 4u;
  unsigned off = ((unsigned)(co0 + wm * 32 + 4 * (lane >> 5)) * a.CiP + ci0 + wn * 32 + (lane & 31)) * 4u;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    unsigned o = off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[t][r];
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), prs, o, 0, 0);
      o += ((r & 3) == 3) ? 5 * row_bytes : row_bytes;
    }
    off += (unsigned)a.CoP * row_bytes;
  }
#ifdef WGRAD_LAB_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WG_STAMP(3);
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 3) && blockIdx.y == 0 && blockIdx.z == 0) {
    const int launch = blockIdx.x == 0 ? atomicAdd(&wg_lab_launches, 1) : wg_lab_launches - 1;
    if ((launch & 63) == 5) {
      const int nc = c_end - c_begin;
      printf("wgrad stamps H %d Ci %d Co %d wg %d: %d chunks; start->first barrier %llu, loop %llu (%llu per chunk), end of loop->loads drained+partial stored %llu; chunk lengths:",
             a.H, a.CiP, a.CoP, (int)blockIdx.x, nc, st[1] - st[0], st[2] - st[1], (st[2] - st[1]) / (nc > 0 ? nc : 1), st[3] - st[2]);
      printf(" [chunk 10: rows 0-8 %llu, barrier %llu, reads+loads issued %llu, row 9+plan %llu]", st[40] - st[14], st[41] - st[40], st[42] - st[41], st[15] - st[42]);
      for (int i = 1; i < nc && i < 12; ++i) printf(" %llu", st[4 + i] - st[4 + i - 1]);
      printf("\n");
    }
  }
#endif
}

// grad[(ci*Co + co)*taps_total + tap_base + t] = sum_split partial[split][t][co][ci]
// A workgroup owns 32 consecutive ci of one (t, co) row; its 8 groups of 32 lanes each add every
// eighth split slice (128-byte coalesced reads), then the 8 partial sums are added in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial,
                                                           float *__restrict__ grad, int splits, int taps,
                                                           int CoP, int CiP, int Co, int Ci, int taps_total,
                                                           int tap_base) {
  __shared__ float red[8][32];
  const int ci_blocks = (Ci + 31) / 32;
  const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const i64 rows = (i64)taps * Co * ci_blocks;
  const i64 slice = (i64)taps * CoP * CiP;
  for (i64 b = blockIdx.x; b < rows; b += gridDim.x) {
    const int cb = (int)(b % ci_blocks);
    const i64 r = b / ci_blocks;
    const int co = (int)(r % Co), t = (int)(r / Co);
    const int ci = cb * 32 + lane32;
    float s = 0.f;
    if (ci < Ci) {
      const float *p = partial + ((i64)t * CoP + co) * CiP + ci;
      int k = grp;
      // eight slices per trip: their loads are in flight together (the order of the additions stays k ascending)
      for (; k + 56 < splits; k += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(k + 8 * u) * slice];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < splits; k += 8) s += p[k * slice];
    }
    red[grp][lane32] = s;
    __syncthreads();
    if (grp == 0 && ci < Ci) {
      float tot = red[0][lane32];
#pragma unroll
      for (int g = 1; g < 8; ++g) tot += red[g][lane32];
      grad[((i64)ci * Co + co) * taps_total + tap_base + t] = tot;
    }
    __syncthreads();
  }
}

// Same reduction for layers with many outputs and few split slices: a workgroup of 32 x COB threads
// (COB = blockDim.x / 32 = 4 or 8) owns a 32(ci) x COB(co) x TAPS box, reads it with 128-byte coalesced rows --
// all taps of four slices requested before the first addition, so a thread has 4 * TAPS loads in flight and a
// layer with four slices costs one memory round trip instead of nine -- transposes through LDS and writes runs
// of COB * TAPS consecutive floats of the reference layout.  The additions run in slice order.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float *__restrict__ partial,
                                                                float *__restrict__ grad, int splits, int CoP,
                                                                int CiP, int Co, int Ci) {
  __shared__ float tile[32][8 * TAPS + 1];
  const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5, COB = blockDim.x >> 5;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * COB;
  const int ci = ci0 + lane32, co = co0 + grp;
  const i64 tap_stride = (i64)CoP * CiP, slice = (i64)TAPS * tap_stride;
  float s[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
  if (ci < Ci && co < Co) {
    const float *p = partial + (i64)co * CiP + ci;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      float v[4][TAPS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) v[u][t] = p[(k + u) * slice + t * tap_stride];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s[t] += v[u][t];
    }
    for (; k < splits; ++k) {
      float v[TAPS];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) v[t] = p[k * slice + t * tap_stride];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) s[t] += v[t];
    }
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t) tile[lane32][grp * TAPS + t] = s[t];
  __syncthreads();
  const int run = COB * TAPS;
  for (int idx = threadIdx.x; idx < 32 * run; idx += blockDim.x) {
    const int cl = idx / run, rem = idx - cl * run;
    const int col = rem / TAPS, t = rem - col * TAPS;
    if (ci0 + cl < Ci && co0 + col < Co) grad[((i64)(ci0 + cl) * Co + co0 + col) * TAPS + t] = tile[cl][rem];
  }
}

}  // namespace

extern "C" {

// taps = 9: 3x3 convolution.  taps = 1 with phases = 4: the four (i,j) phases of the 2x2 transposed
// convolution, one phase per workgroup, reduced together as a 4-tap filter.
static int run_wgrad(int taps, int phases, const WgradArgs &base, float *grad, int Ci, int Co, double flops,
                     hipStream_t s) {
  WgradArgs a = base;
  a.phases = phases;
  const int slices = taps * phases;  // tap slices per split in the partial buffer
  a.chunksX = (a.W + 15) / 16;
  a.chunksY = (a.H + 7) / 8;
  a.total_chunks = a.N * a.chunksX * a.chunksY;
  const int tiles = (a.CoP / 64) * (a.CiP / 64);
  // every workgroup writes a 64 x 64 x taps fp32 partial block, so the number of workgroups sets the
  // partial-sum traffic: one per CU for the 3x3 filters (147 KB each; measured faster in sum with the
  // reduction than two per CU), two per CU for the 16 KB blocks of the transposed convolution
  // -- unless a workgroup's pixel slice is long anyway (large images): then the partials are negligible
  // and two workgroups per CU hide each other's staging
  const bool long_slices = (i64)a.total_chunks * tiles >= (i64)48 * 512;
  const int target_blocks = (taps == 9 && !long_slices) ? 256 : 512;
  // (transposed convolution: all four phases in one workgroup, one workgroup per CU -- see wgrad_mfma_kernel's PH)
  static const bool lab_phase_wgs = getenv("UNET_LAB_TCONV_WGRAD_PHASE_WGS") != nullptr;  // laboratory switch: one phase per workgroup
  const bool joint = phases == 4 && !lab_phase_wgs;
  int splits = joint ? (256 + tiles - 1) / tiles : (target_blocks / phases + tiles - 1) / tiles;
  if (splits > a.total_chunks) splits = a.total_chunks;
  if (splits < 1) splits = 1;
  a.chunks_per_split = (a.total_chunks + splits - 1) / splits;
  splits = (a.total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
  void *ws;
  int rc = unet_workspace(&ws, sizeof(float) * (size_t)splits * slices * a.CoP * a.CiP, s);
  if (rc) return rc;
  a.partial = (float *)ws;
  const size_t lds = (size_t)128 * WP + (size_t)(taps == 9 ? 180 : 128) * WP;
  // laboratory switch: 0 = register-staged kernel, 1 = direct-to-LDS, 2 = direct-to-LDS with prefetched fragments
  static const int dma_env = getenv("UNET_LAB_WGRAD_DMA") ? atoi(getenv("UNET_LAB_WGRAD_DMA")) : 2;
#ifdef WGRAD_LAB_STAMPS
  const size_t lds_dma = (size_t)3 * 40 * 1024 + 512;
#else
  const size_t lds_dma = (size_t)3 * 40 * 1024;
#endif
  static bool attr_set[64] = {};
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mfma_kernel<9>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)((128 + 180) * WP)));
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mfma_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)((128 + 128) * WP)));
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mfma_kernel<1, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4 * 128 + 128) * WP)));
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_dma_kernel<9, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma));
    attr_set[dev & 63] = true;
  }
  dim3 grid((unsigned)(joint ? splits : splits * phases), (unsigned)(a.CoP / 64), (unsigned)(a.CiP / 64));
  const double partial_bytes = 4.0 * splits * slices * a.CoP * a.CiP;
  {
    ProfScope prof(taps == 9 ? PROF_CONV_WGRAD_BF16 : PROF_TCONV_WGRAD_BF16, flops,
                   2.0 * a.N * a.H * a.W * ((double)a.x_c * (joint ? 1 : phases) + (double)a.z_c * phases) + partial_bytes, s);
    // the direct-to-LDS kernel relies on one image staying below 2^30 bytes (see plan_chunk)
    const bool small_images = (i64)a.H * a.W * a.x_ld * 2 < (1ll << 30) && (i64)a.Hz * a.Wz * a.dz_ld * 2 < (1ll << 30);
    const int dma = small_images ? dma_env : 0;
    if (taps == 9 && dma)
      hipLaunchKernelGGL((wgrad_dma_kernel<9, 3>), grid, dim3(256), lds_dma, s, a);
    else if (taps == 9)
      hipLaunchKernelGGL(wgrad_mfma_kernel<9>, grid, dim3(256), lds, s, a);
    else if (joint)
      hipLaunchKernelGGL((wgrad_mfma_kernel<1, 4>), grid, dim3(256), (size_t)(4 * 128 + 128) * WP, s, a);
    else
      hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), lds, s, a);
  }
  UNET_LAUNCH_CHECK("wgrad_mfma_kernel");
  ProfScope prof(PROF_WGRAD_REDUCE, 0, partial_bytes + 4.0 * slices * Co * Ci, s);
  // few slices: the box kernel, with as many workgroups as the layer allows (8 or 4 output channels per box);
  // many slices (small layers): one workgroup per (tap, co, 32 ci) row, the slices spread over its 8 groups
  const i64 boxes8 = (i64)((Ci + 31) / 32) * ((Co + 7) / 8), boxes4 = (i64)((Ci + 31) / 32) * ((Co + 3) / 4);
  static const int tile_max_splits = getenv("UNET_LAB_REDUCE_TILE_SPLITS") ? atoi(getenv("UNET_LAB_REDUCE_TILE_SPLITS")) : 32;
  const int cob = boxes8 >= 512 ? 8 : 4;
  if (splits <= tile_max_splits && (cob == 8 ? boxes8 : boxes4) >= 256) {
    const dim3 tgrid((unsigned)((Ci + 31) / 32), (unsigned)((Co + cob - 1) / cob));
    if (slices == 9)
      hipLaunchKernelGGL(wgrad_reduce_tile_kernel<9>, tgrid, dim3(32 * cob), 0, s, (const float *)ws, grad, splits, a.CoP,
                         a.CiP, Co, Ci);
    else
      hipLaunchKernelGGL(wgrad_reduce_tile_kernel<4>, tgrid, dim3(32 * cob), 0, s, (const float *)ws, grad, splits, a.CoP,
                         a.CiP, Co, Ci);
  } else {
    const i64 reduce_rows = (i64)slices * Co * ((Ci + 31) / 32);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(reduce_rows < 65536 ? reduce_rows : 65536)), dim3(256), 0,
                       s, (const float *)ws, grad, splits, slices, a.CoP, a.CiP, Co, Ci, slices, 0);
  }
  UNET_LAUNCH_CHECK("wgrad_reduce_kernel");
  return UNET_OK;
}

// grad_w [Ci][Co][3][3] fp32 (reference layout, overwritten)
int unet_conv3x3_wgrad_bf16(const uint16_t *dz, int dz_ld, const uint16_t *x, int x_ld, float *grad_w, int N, int H,
                            int W, int CiP, int CoP, int Ci, int Co, unet_stream_t s) {
  UNET_REQUIRE(dz && x && grad_w, "unet_conv3x3_wgrad_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && Ci <= CiP && Co <= CoP,
               "unet_conv3x3_wgrad_bf16: bad sizes");
  UNET_REQUIRE(CiP % 8 == 0 && CoP % 8 == 0 && dz_ld >= CoP && x_ld >= CiP && dz_ld % 8 == 0 && x_ld % 8 == 0,
               "unet_conv3x3_wgrad_bf16: padded channel counts and pixel strides must be multiples of 8");
  UNET_REQUIRE((i64)H * W * x_ld < (1ll << 31) && (i64)H * W * dz_ld < (1ll << 31),
               "unet_conv3x3_wgrad_bf16: one image must stay below 2^31 elements");
  WgradArgs a{};
  a.dz = dz; a.x = x; a.N = N; a.H = H; a.W = W; a.Hz = H; a.Wz = W; a.dz_ld = dz_ld; a.x_ld = x_ld;
  a.CoP = (CoP + 63) / 64 * 64; a.CiP = (CiP + 63) / 64 * 64; a.z_c = CoP; a.x_c = CiP;
  a.dz_s = 1; a.dz_oy = a.dz_ox = 0;
  return run_wgrad(9, 1, a, grad_w, Ci, Co, 2.0 * N * H * W * (double)CiP * CoP * 9, as_stream(s));
}

// grad_w [Ci][Co][2][2] fp32 (overwritten); x = the transposed conv's input, dy its output gradient
int unet_tconv2x2_wgrad_bf16(const uint16_t *dy, int dy_ld, const uint16_t *x, int x_ld, float *grad_w, int N,
                             int H_in, int W_in, int CiP, int CoP, int Ci, int Co, unet_stream_t s) {
  UNET_REQUIRE(dy && x && grad_w, "unet_tconv2x2_wgrad_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && H_in > 0 && W_in > 0 && Ci > 0 && Co > 0 && Ci <= CiP && Co <= CoP,
               "unet_tconv2x2_wgrad_bf16: bad sizes");
  UNET_REQUIRE(CiP % 8 == 0 && CoP % 8 == 0 && dy_ld >= CoP && x_ld >= CiP && dy_ld % 8 == 0 && x_ld % 8 == 0,
               "unet_tconv2x2_wgrad_bf16: padded channel counts and pixel strides must be multiples of 8");
  UNET_REQUIRE((i64)H_in * W_in * x_ld < (1ll << 31) && (i64)4 * H_in * W_in * dy_ld < (1ll << 31),
               "unet_tconv2x2_wgrad_bf16: one image must stay below 2^31 elements");
  WgradArgs a{};
  a.dz = dy; a.x = x; a.N = N; a.H = H_in; a.W = W_in; a.Hz = 2 * H_in; a.Wz = 2 * W_in;
  a.dz_ld = dy_ld; a.x_ld = x_ld; a.CoP = (CoP + 63) / 64 * 64; a.CiP = (CiP + 63) / 64 * 64;
  a.z_c = CoP; a.x_c = CiP; a.dz_s = 2; a.dz_oy = a.dz_ox = 0;
  return run_wgrad(1, 4, a, grad_w, Ci, Co, 2.0 * N * H_in * W_in * 4.0 * CiP * CoP, as_stream(s));
}

}  // extern "C"
