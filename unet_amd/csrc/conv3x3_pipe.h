// Pipelined 3x3 / pad 1 / stride 1 bf16 convolution for gfx950: the kernel the convolution entry points use
// for images of at least 32 rows and channel counts in multiples of 64 (conv_bf16.hip keeps the general one).
//
// What it changes against conv_mfma_kernel<9, 64, ...> (same MFMA loop, same LDS image, same summation order,
// so the two kernels return bit-identical outputs):
//   - a workgroup is PERSISTENT: 8 wavefronts own one block of 64 output channels and walk over many
//     32 x 16 pixel tiles.  The stages of all its tiles (stage = one 32-channel K chunk of one tile) form one
//     software pipeline: the operands of stage s+1 travel while stage s is on the matrix cores, including
//     across tile boundaries, so a tile's first chunk is no longer fetched with nothing to overlap it -- the
//     cost that dominated the full-resolution layers (two chunks per tile);
//   - operands go from memory straight into LDS (buffer_load ... lds, 1 KiB per wave instruction, no staging
//     registers, no ds_write pass): two stage buffers, ONE barrier per stage.  The XOR swizzle of the LDS
//     image is applied to the per-lane source address, the LDS side of such a load being lane-linear;
//     out-of-image halo pixels address beyond the buffer descriptor and arrive as zeros;
//   - twice the pixels per weight slab (512 instead of 256), and with C_in <= 64 the workgroup's whole
//     weight block (72 KiB) stays in LDS for all of its tiles: per tile only the halo (38 KiB a chunk) moves;
//   - nothing is computed between a stage's barrier and its first MFMA: behind the barrier come the twelve
//     fragment reads of the first tap, back to back, then the MFMAs.  The per-lane source offsets of a tile's
//     halo loads are set up once per tile (the K chunk goes into the loads' scalar offset), the waves that own
//     one piece fewer than the others send that load out of range and into a spare KiB instead of branching
//     round it, and the tile's last stage (the one that prefetches the next tile and, in the fused variants,
//     requests the epilogue's operands) is a second copy of the stage body rather than a branch inside it;
//   - inside a stage a wave never waits for LDS: the weight fragments of the next tap are requested before the
//     16 MFMAs of the current one, the pixel fragments of the next tap column behind the last MFMAs that read the
//     registers they replace; and the two halves of the workgroup request the next stage's operands at
//     different times (see `late` in the kernel);
//   - the epilogue has no barrier and no LDS round trip: every wave converts, stores and reduces its own
//     8 x 16 pixel x 32 channel block (8-byte stores; batch-norm sums over the 16 pixel lanes by DPP row
//     rotations), and writes its own row of partial sums.
#pragma once
#include "bf16_util.h"
#ifndef PIPE_LAB_ABLATE
#define PIPE_LAB_ABLATE 0  // laboratory builds only: 1 = no output stores, 2 = no MFMAs, 3 = no halo loads,
                            // 4 = no operand loads at all after the first stage (matrix cores + LDS reads only)
#endif

#ifndef PIPE_LAB_NORM_ON_LOAD
#define PIPE_LAB_NORM_ON_LOAD 0  // laboratory builds only: every pixel fragment goes through relu(a * x + b) after its LDS read
                                  // (what folding the previous layer's normalisation pass into this kernel would cost;
                                  // constant a, b: the results are not those of the library)
#endif
#ifndef PIPE_LAB_STAMPS
#define PIPE_LAB_STAMPS 0  // laboratory builds only: clock stamps of the first 16 stages of every wave (Args::lab_stamps)
#endif

namespace pipe {

constexpr int TW = 16;                             // pixel tile: TH_ rows (32, or 16 for images of fewer than 32 rows) x 16
template <int TH_>
struct Geo {
  static constexpr int TH = TH_, QW = TH_ / 4;            // tile rows; rows per wave (4 row groups)
  static constexpr int HR = TH_ + 2, HC = TW + 2;         // halo rows / columns
  static constexpr int HALO_PIECES = HR * HC * 4;         // 16-byte pieces of one halo stage (64 B per pixel)
  static constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;  // wave instructions (1 KiB each) that fill it
  static constexpr int HALO_BYTES = HALO_INSTR * 1024;    // 39936 (TH 32): the last instruction's tail is padding
  static constexpr int KH = (HALO_INSTR + 7) / 8;         // halo instructions per wave, at most
  static constexpr int STAGE_BYTES = HALO_BYTES + 9 * 64 * 64;
  static constexpr int STAGES_BYTES = 2 * STAGE_BYTES;    // (resident weights: 2 * HALO_BYTES + 2 * W_BYTES)
  static constexpr int DUMP_OFF = STAGES_BYTES + 64 * 16;  // behind the batch-norm records of the 64 channels (EPI_BRED)
  static constexpr int LDS_BYTES = DUMP_OFF + 1024 + (PIPE_LAB_STAMPS ? 8192 : 0);  // + 1 KiB nobody reads (see the stage's loads)
};
constexpr int TH = 32;                             // the tile height of the main variant
constexpr int W_INSTR = 9 * 64 * 4 / 64;           // 36
constexpr int W_BYTES = W_INSTR * 1024;            // 36864: 9 taps x 64 channels x 64 B
constexpr int MAX_WS_CHUNKS = 2;                   // weights resident in LDS up to C_in = 64
constexpr int LDS_BYTES = Geo<32>::LDS_BYTES;      // 155648

struct Args {
  const bf16_t *in;
  const bf16_t *wt;     // packed [C_in/32][9][CoP][32]
  const float *bias;    // or null
  bf16_t *out;
  float *stats;         // [N][rows_per_image][CoP][2] or null; rows_per_image = 4 * tilesY * tilesX
  int N, H, W, nk, CoP; // nk = K chunks of 32 channels
  int in_ld, out_ld;
  int in_c;             // > 0: channels present in memory (< 32, multiple of 8, nk == 1); the rest reads as zero
  int tilesX, tilesY;
  int units;            // N * tilesY * tilesX
  int co_blocks;        // CoP / 64
  int xcd_order;        // 1: renumber the workgroups per XCD (needs gridDim.x % (8 * co_blocks) == 0)
  int co_valid;
  int accumulate;
  int nt_store;         // 1: the output is stored non-temporal (large tensors that no cache holds until they are read)
  int reverse;          // 1: the units are walked from the last to the first (the input's most recently written end first)
  // BRED: the output is the gradient of a BN + ReLU layer's output; bz = that layer's pre-BN tensor,
  // bcf its per-channel records {a, b, mean, 1/sqrt(var+eps)}; stats receives (sum g*xhat, sum g),
  // g = out * relu'(a*z + b).  Channels below bred_c0 take no part; bz, bcf and stats are indexed by
  // (channel - bred_c0) and the stats rows hold CoP - bred_c0 channels.
  const bf16_t *bz;
  int bz_ld;
  const float4 *bcf;
  int bred_c0;
  const float4 *affine;  // AFF: per-channel {a, b, ., .}: out = relu(a * (conv + bias) + b)
#if PIPE_LAB_STAMPS
  unsigned long long *lab_stamps;  // [workgroup][wave][16 stages][8]: 0 behind the barrier, 1 epilogue done, 2 MFMAs issued, 3 operands landed, 4 before the barrier
#endif
};

__device__ __forceinline__ float row_sum16(float v) {  // sum over the 16 lanes of a DPP row, in every lane
  v = keep_scalar(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
  v = keep_scalar(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));
  v = keep_scalar(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true));
  v = keep_scalar(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true));
  return v;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// What the epilogue does with a finished tile (compile-time, so that each variant is straight-line code):
enum Epi {
  EPI_STATS = 0,  // forward in training: + bias, bf16 store, per-strip (sum, sum of squares) of the rounded values
  EPI_PLAIN = 1,  // + bias (if any), bf16 store: data gradient without fusion, forward without statistics
  EPI_ACC = 2,    // the result is added to what `out` holds (gradient fan-in)
  EPI_BRED = 3,   // data gradient + the BN-backward pixel sums of the layer that produced the input
  EPI_AFF = 4,    // inference: relu(a * (conv + bias) + b) with per-channel {a, b}
};
constexpr int epi_vmem_ops(int epi, int qw) {  // vector-memory instructions a wave issues per tile epilogue
  return (epi == EPI_ACC || epi == EPI_BRED) ? 2 * qw : qw;
}

// Lanes 16 apart exchange halves (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of
// the second).  In: hw[m] = the lane's 4 channels (2 dwords) of 16-channel block m, lane row kp holding
// channels 4 kp .. 4 kp + 3.  Out: 16 bytes = 8 consecutive channels of block (kp & 1), from channel
// 8 (kp >> 1).  The operands are tied to registers of their own first (see keep_scalar in bf16_util.h).
// unswap16 is the inverse (the exchange is an involution).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x4 swap16(const u32x2_t (&hw)[2]) {
  unsigned a0 = hw[0].x, a1 = hw[0].y, b0 = hw[1].x, b1 = hw[1].y;
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
  const u32x2_t lo = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const u32x2_t hi = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  return i32x4{(int)lo.x, (int)hi.x, (int)lo.y, (int)hi.y};
}
__device__ __forceinline__ void unswap16(const i32x4 &v, u32x2_t (&hw)[2]) {
  unsigned p0 = (unsigned)v.x, p1 = (unsigned)v.y, q0 = (unsigned)v.z, q1 = (unsigned)v.w;
  asm volatile("" : "+v"(p0), "+v"(p1), "+v"(q0), "+v"(q1));
  const u32x2_t lo = __builtin_amdgcn_permlane16_swap(p0, q0, false, false);
  const u32x2_t hi = __builtin_amdgcn_permlane16_swap(p1, q1, false, false);
  hw[0] = u32x2_t{lo.x, hi.x};
  hw[1] = u32x2_t{lo.y, hi.y};
}

template <int N>
__device__ __forceinline__ void wait_vmem_then_barrier() {
  // all but this wave's N youngest vector-memory operations are complete (the LDS loads of the stage about to
  // be read are older than the N stores of a tile epilogue); own LDS reads are complete; then the barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <bool WS, int EPI, int TH_ = 32>
__global__ __launch_bounds__(512, 2) void conv3x3_pipe_kernel(const Args a) {
  constexpr bool BRED = EPI == EPI_BRED, AFF = EPI == EPI_AFF, STATS = EPI == EPI_STATS || EPI == EPI_BRED;
  typedef Geo<TH_> G;
  constexpr int TH = G::TH, QW = G::QW, HR = G::HR, HC = G::HC, HALO_INSTR = G::HALO_INSTR, HALO_BYTES = G::HALO_BYTES,
                KH = G::KH, STAGE_BYTES = G::STAGE_BYTES, STAGES_BYTES = G::STAGES_BYTES;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  // WS: [halo 0][halo 1][weights chunk 0][weights chunk 1];  else: [halo 0][weights 0][halo 1][weights 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave & 3, cw = wave >> 2;
  const int l15 = lane & 15, kp = lane >> 4;

  // Workgroup ids are dealt to the 8 XCDs (each with an L2 of its own) in turn.  Renumbered so that an XCD's
  // workgroups are consecutive, the channel blocks of one tile sequence -- which read the same input -- and the
  // neighbouring sequences -- which share halos -- sit on one XCD and fetch that input into one L2 once.
  int bid = blockIdx.x;
  if (a.xcd_order) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int cob = bid % a.co_blocks, wgi = bid / a.co_blocks, nwg = gridDim.x / a.co_blocks;
  const int co0 = cob * 64;
  const int my_units = wgi < a.units ? (a.units - wgi + nwg - 1) / nwg : 0;
  if (my_units == 0) return;
  const int tpi = a.tilesX * a.tilesY;

  // ---- per-lane source plan of the LDS loads (fixed for the whole kernel) ----
  // wave instruction i fills LDS bytes [1024 i, 1024 i + 1024) of its region; lane l's 16 bytes are LDS row
  // r = 16 i + l / 4, slot l % 4, which the swizzled image assigns to piece slot ^ ((r >> 1) & 3) of row r
  constexpr unsigned OOB = 0x80000000u;
  unsigned w_rel[5];         // weights: byte offset within one chunk's slab
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = (7 - wave) + 8 * k;  // the waves with one halo instruction fewer take one weight instruction more
    const int rw = 16 * j + (lane >> 2), pw = (lane & 3) ^ ((rw >> 1) & 3);
    const int tap = rw >> 6, co = rw & 63;
    w_rel[k] = j < W_INSTR ? (unsigned)(((tap * a.CoP + co0 + co) * 32 + pw * 8) * 2) : OOB;
  }
  const unsigned img_bytes = (unsigned)a.H * a.W * a.in_ld * 2u;
  const unsigned slab_bytes = 9u * a.CoP * 64u;  // one K chunk of the packed weights
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.wt), 0, slab_bytes * a.nk, 0x00020000);

  // The workgroup's tiles: unit u = wgi + it * nwg (counted from the last unit if a.reverse) = (image n, tile row
  // ty, tile column tx), tx fastest.  One pair of divisions here; the walk then steps by nwg units with carries.
  const int step_n = nwg / tpi, step_r = nwg - step_n * tpi, step_ty = step_r / a.tilesX, step_tx = step_r - step_ty * a.tilesX;
  auto first_unit = [&](int &n, int &ty, int &tx) __attribute__((always_inline)) {
    const int u = a.reverse ? a.units - 1 - wgi : wgi;
    n = u / tpi;
    const int t = u - n * tpi;
    ty = t / a.tilesX;
    tx = t - ty * a.tilesX;
  };
  auto next_unit = [&](int &n, int &ty, int &tx) __attribute__((always_inline)) {
    if (a.reverse) {
      tx -= step_tx;
      if (tx < 0) tx += a.tilesX, --ty;
      ty -= step_ty;
      if (ty < 0) ty += a.tilesY, --n;
      n -= step_n;
    } else {
      tx += step_tx;
      if (tx >= a.tilesX) tx -= a.tilesX, ++ty;
      ty += step_ty;
      if (ty >= a.tilesY) ty -= a.tilesY, ++n;
      n += step_n;
    }
  };
  // A tile's halo loads: per-lane byte offsets into its image for K chunk 0 (beyond the descriptor for pixels
  // outside the image, for pieces the layer does not have, and -- live = false -- for a tile past the
  // workgroup's last).  (Arrays go into these lambdas as pointers: with array-reference parameters the host pass
  // silently drops the kernel's stub.)
  // plan_piece: halo load k of this wave (KH per tile); plan_tile: all of them.  hk[k] = the piece's halo row, halo
  // column and 16-byte slot, packed (or -1 for a piece this wave does not own or the layer does not have): one
  // register per piece instead of the offsets they expand to.  No short-circuit evaluation: a handful of vector
  // instructions without branches.
  int hk[KH];
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    const int i = wave + 8 * k;
    const int r = 16 * i + (lane >> 2), piece = (lane & 3) ^ ((r >> 1) & 3);
    const int py = r / HC, px = r - py * HC;
    const bool ok = i < HALO_INSTR && r < HR * HC && (a.in_c == 0 || piece * 8 < a.in_c);
    hk[k] = ok ? ((py << 16) | (px << 8) | piece) : -1;
  }
  auto plan_piece = [&](const int k, bool live, int y0, int x0, unsigned *voff) __attribute__((always_inline)) {
    const int org = (((y0 - 1) * a.W + (x0 - 1)) * a.in_ld) * 2;
    int h = hk[k];
    asm volatile("" : "+v"(h));  // expanded per tile: hoisted out of the tile loop, the terms below would sit in
                                 // registers of their own through every stage
    const int py = h >> 16, px = (h >> 8) & 255, piece = h & 255;
    const int ok = (int)live & (int)(h >= 0) & (int)((unsigned)(y0 + py - 1) < (unsigned)a.H) &
                   (int)((unsigned)(x0 + px - 1) < (unsigned)a.W);
    voff[k] = ok ? (unsigned)(org + ((py * a.W + px) * a.in_ld + piece * 8) * 2) : OOB;
  };
  auto plan_tile = [&](bool live, int y0, int x0, unsigned *voff) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KH; ++k) plan_piece(k, live, y0, x0, voff);
  };
  typedef __attribute__((address_space(3))) unsigned char *lds_bytes_t;
  const lds_bytes_t lds3 = (lds_bytes_t)lds;  // destinations as offsets from here: a select between two generic
                                              // LDS pointers does not compile
#if PIPE_LAB_STAMPS
  unsigned long long *lds_stamps = reinterpret_cast<unsigned long long *>(lds + G::DUMP_OFF + 1024) + wave * 128;
  for (int i = lane; i < 128; i += 64) lds_stamps[i] = 0;
#define PIPE_STAMP(stage_, slot_)                                                                   \
  do {                                                                                              \
    if ((stage_) < 16 && lane == 0) lds_stamps[(stage_) * 8 + (slot_)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PIPE_STAMP(stage_, slot_) \
  do {                            \
  } while (0)
#endif
#if PIPE_LAB_ABLATE == 3 || PIPE_LAB_ABLATE == 4
  int s_lab = 0;  // 0 until the first stage's operands have been requested
#endif
  // The loads of one stage: every wave issues KH halo pieces (and 5 weight pieces); the ones a wave does not own
  // address beyond their descriptor and land (as zeros) in the spare KiB.  soff = the K chunk's byte offset,
  // dst = the stage buffer's byte offset in LDS.
  auto issue_halo = [&](int n, const unsigned *voff, int soff, int dst) __attribute__((always_inline)) {
#if PIPE_LAB_ABLATE == 3 || PIPE_LAB_ABLATE == 4
    if (s_lab > 0 && a.N > 0) return;
#endif
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in) + (i64)n * a.H * a.W * a.in_ld, 0,
                                                       img_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int i = wave + 8 * k;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(lds3 + (i < HALO_INSTR ? dst + i * 1024 : G::DUMP_OFF)), 16,
                                               voff[k], soff, 0, 0);
    }
  };
  auto issue_weights = [&](const unsigned *voff, int soff, int dst) __attribute__((always_inline)) {
#if PIPE_LAB_ABLATE == 4
    if (s_lab > 0 && a.N > 0) return;
#endif
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int j = (7 - wave) + 8 * k;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(lds3 + (j < W_INSTR ? dst + j * 1024 : G::DUMP_OFF)), 16,
                                               voff[k], soff, 0, 0);
    }
  };
  auto halo_off = [&](int s) __attribute__((always_inline)) { return WS ? (s & 1) * HALO_BYTES : (s & 1) * STAGE_BYTES; };  // byte offsets in LDS
  auto weight_off = [&](int s, int kc) __attribute__((always_inline)) {
    return WS ? 2 * HALO_BYTES + kc * W_BYTES : (s & 1) * STAGE_BYTES + HALO_BYTES;
  };

  // ---- fragment addresses (the LDS image and the loop of conv_mfma_kernel's 16x16x32 path) ----
  const int a4_lane = l15 * 64 + ((kp ^ ((l15 >> 1) & 3)) * 16);
  int b4_off[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) {
      const int pl = rc * HC + l15 + kw;
      b4_off[kw][rc] = (l15 + kw) * 64 + ((kp ^ ((pl >> 1) & 3)) * 16) + ph * QW * HC * 64;
    }

  // ---- per-lane epilogue constants: the lane's 2 x 4 channels ----
  f32x4 biasv[2], scalev[2];
  const bool red_here = BRED && co0 >= a.bred_c0;  // uniform over the workgroup
  const int rc0 = co0 - a.bred_c0;
  // BRED: the 64 channels' records {a, b, mean, 1/sqrt(var+eps)} wait in LDS and are read per tile epilogue;
  // kept in registers across the MFMA loop they (with the prefetched z values) push the kernel into spills
  float4 *lds_cf = reinterpret_cast<float4 *>(lds + STAGES_BYTES);
  if (BRED && tid < 64) lds_cf[tid] = red_here ? a.bcf[rc0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ch = (cw * 2 + m) * 16 + 4 * kp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = (!BRED && a.bias && co0 + ch + j < a.co_valid) ? a.bias[co0 + ch + j] : 0.f;
      if (AFF) {
        const float4 cf = a.affine[co0 + ch + j];
        scalev[m][j] = cf.x;
        biasv[m][j] = fmaf(cf.x, b, cf.y);
      } else {
        scalev[m][j] = 1.f;
        biasv[m][j] = b;
      }
    }
  }
  const unsigned out_img_bytes = (unsigned)a.H * a.W * a.out_ld * 2u, z_img_bytes = (unsigned)a.H * a.W * a.bz_ld * 2u;
  const int stat_width = BRED ? a.CoP - a.bred_c0 : a.CoP;  // channels per row of partial sums

  const int nk = a.nk;
  int n, ty, tx;  // the current tile
  unsigned hoff[KH], hoff_nxt[KH];
  first_unit(n, ty, tx);
  plan_tile(true, ty * TH, tx * TW, hoff);
  if (WS)
    for (int kc = 0; kc < nk; ++kc) issue_weights(w_rel, kc * slab_bytes, weight_off(0, kc));
  else
    issue_weights(w_rel, 0, weight_off(0, 0));
  issue_halo(n, hoff, 0, halo_off(0));
#if PIPE_LAB_ABLATE == 3 || PIPE_LAB_ABLATE == 4
  s_lab = 1;
#endif

  // batch-norm sums over all tiles of the workgroup, kept in the layout the results are STORED in: the lane's 8
  // consecutive channels (from ch16, below) of its pixel column
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;

  const int ch16 = cw * 32 + (kp & 1) * 16 + (kp >> 1) * 8;  // first of the lane's 8 channels after the swap
  // Who requests the next stage's operands when.  The two waves of a SIMD (w and w + 4) share its matrix pipe, and
  // the arbitration goes to the older one (w < 4) whenever it has an MFMA ready: it runs through its stage first
  // and then waits at the barrier, while the younger one does most of its MFMAs afterwards.  So the younger half
  // issues its loads at the head of the stage, under the older half's MFMAs, and the older half -- which would
  // otherwise spend the head of every stage issuing loads with both waves of the SIMD off the matrix pipe --
  // issues its own after LATE_G of the nine MFMA groups: after all nine, when it has nothing left to do in this
  // stage but wait for its partner (measured over LATE_G = 2 .. 9: the later the better, 2 slower than the head,
  // 9 another 1 - 3 % faster than 4; the loads still land before the partner is through its MFMAs).  Not so
  // in the fused-reduction variant of the layers with resident weights (few channels, full-resolution images: the
  // memory-bound ones, whose last stage also requests the epilogue's operands): measured 2 - 6 % faster with every
  // load out at the head of the stage.
  // (Not done, because measured slower: the younger half writing a finished tile behind the next tile's first
  // barrier instead of with the SIMD waiting for it -- under its partner's MFMAs that epilogue takes twice as
  // long, and the stage it sits in grows by more than the tile's end saves.)
#ifndef PIPE_LATE_G
#define PIPE_LATE_G 9  // laboratory builds: 0 = every wave at the head of the stage, 1 .. 8 = the older half between groups
#endif
  constexpr int LATE_G = (WS && BRED) ? 0 : PIPE_LATE_G;
  const bool late = wave < 4 && LATE_G > 0;
#if defined(PIPE_LAB_YOUNG_PRIO)  // laboratory builds: the younger half at raised priority for the whole kernel
  if (wave >= 4) __builtin_amdgcn_s_setprio(PIPE_LAB_YOUNG_PRIO);
#endif

  i32x4 zq[BRED ? QW : 1], oq[EPI == EPI_ACC ? QW : 1];  // what the epilogue reads, requested in a tile's last stage
  f32x4 acc[2][QW];
  // Output addressing of tile (en, ety, etx): the lane's 16 bytes of row q, or out of range
  auto out_offsets = [&](int ety, int etx, unsigned *ooff, bool *okq, int ld, int c0) __attribute__((always_inline)) {
    int lx = l15;
    asm volatile("" : "+v"(lx));  // keeps the eight per-row offsets from being hoisted out of the tile loop
    const int x = etx * TW + lx;
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      const int y = ety * TH + ph * QW + q;
      okq[q] = y < a.H && x < a.W;
      ooff[q] = okq[q] ? (unsigned)(((y * a.W + x) * ld + c0 + ch16) * 2) : OOB;
    }
  };
  // ---- epilogue of tile (en, ety, etx): every wave on its own 8 rows x 16 pixels x 32 channels.  Buffer loads
  // and stores whose out-of-image lanes address beyond the descriptor: no branches, and a fixed number of
  // vector-memory instructions per tile (epi_vmem_ops).  The matrix cores leave a lane with 4 channels of
  // each of its two 16-channel blocks; v_permlane16_swap between lanes 16 apart turns that into 8
  // consecutive channels of one block, so a lane moves 16 bytes and a pixel's 64 bytes leave in one piece.
  auto epilogue = [&](const int en, const int ety, const int etx, const int stamp_stage) __attribute__((always_inline)) {
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (i64)en * a.H * a.W * a.out_ld, 0, out_img_bytes, 0x00020000);
    unsigned ooff[QW];
    bool okq[QW];
    out_offsets(ety, etx, ooff, okq, a.out_ld, co0);
    float4 bcf[BRED ? 8 : 1];
    if (BRED) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bcf[BRED ? j : 0] = lds_cf[ch16 + j];
    }
    // a tile that lies inside the image (every tile of an image whose sides are multiples of the tile's) needs no
    // per-pixel masks in its statistics
    const bool inside = ety * TH + TH <= a.H && etx * TW + TW <= a.W;
    // Statistics of one stored row (sw = the lane's 8 rounded channels, z = BRED: the parent's pre-BN values of
    // the same 8 channels).  They run on what is stored, after the exchange between lanes: no second copy of the
    // results in the matrix cores' layout, and the fused reduction's operands need no exchange at all.
    auto row_stats = [&](const i32x4 &sw, const i32x4 &z, const bool masked, const bool ok) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned w = (unsigned)sw[j >> 1];
        float r = (j & 1) ? bf16_to_f32(w >> 16) : bf16_to_f32(w & 0xffffu);  // statistics of the ROUNDED values
        if (masked && !ok) r = 0.f;
        if (BRED) {
          const unsigned zz = (unsigned)z[j >> 1];
          const float zv = (j & 1) ? bf16_to_f32(zz >> 16) : bf16_to_f32(zz & 0xffffu);
          const float4 c = bcf[BRED ? j : 0];
          const float g = fmaf(c.x, zv, c.y) > 0.f ? r : 0.f;
          ssum[j] = fmaf(g, zv, ssum[j]);  // sum g*z; turned into sum g*xhat at the end
          ssq[j] += g;
        } else {
          ssum[j] += r;
          ssq[j] += r * r;
        }
      }
    };
    i32x4 stored[STATS ? QW : 1];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      typedef u32x2_t u32x2;
      u32x2 hw[2];  // the rounded result, matrix-core layout: hw[m] = 4 channels of block m
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 v = acc[m][q];
        if (AFF) {
          v = v * scalev[m] + biasv[m];
          v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
        } else if (!BRED) {
          v += biasv[m];
        }
        hw[m] = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
      }
      i32x4 sw = swap16(hw);
      if (EPI == EPI_ACC) {  // + what the output holds (same layout), rounded again
        const i32x4 o = oq[EPI == EPI_ACC ? q : 0];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned x0_ = (unsigned)sw[d], x1_ = (unsigned)o[d];
          const float lo = bf16_to_f32(x0_ & 0xffffu) + bf16_to_f32(x1_ & 0xffffu);
          const float hi = bf16_to_f32(x0_ >> 16) + bf16_to_f32(x1_ >> 16);
          sw[d] = (int)((unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16));
        }
      }
#if PIPE_LAB_ABLATE == 1
      if (a.N < 0)
#endif
      if (a.nt_store)
        __builtin_amdgcn_raw_buffer_store_b128(sw, ors, ooff[q], 0, 2);  // cache policy bit 1: nt
      else
        __builtin_amdgcn_raw_buffer_store_b128(sw, ors, ooff[q], 0, 0);
      if (STATS) stored[STATS ? q : 0] = sw;
    }
    if (STATS) {
      if (inside) {
#pragma unroll
        for (int q = 0; q < QW; ++q) row_stats(stored[STATS ? q : 0], zq[BRED ? q : 0], false, true);
      } else {
#pragma unroll
        for (int q = 0; q < QW; ++q) row_stats(stored[STATS ? q : 0], zq[BRED ? q : 0], true, okq[q]);
      }
    }
    PIPE_STAMP(stamp_stage, 1);
  };

  // A stage: barrier, fragment reads, the loads of the next stage, nine groups of 16 MFMAs (one per tap; tap
  // column kw outermost: the order every variant of this convolution adds in).
  // head_reads: the first tap's fragments (ten pixel rows, two weight blocks), requested back to back: left to
  // itself the scheduler pairs each read with the MFMA that needs it and the head of the stage waits for one
  // LDS round trip per MFMA.
#if PIPE_LAB_NORM_ON_LOAD
  auto lab_norm = [&](bf16x8 &f) __attribute__((always_inline)) {
    const f32x4 lo = __builtin_convertvector(__builtin_shufflevector(f, f, 0, 1, 2, 3), f32x4);
    const f32x4 hi = __builtin_convertvector(__builtin_shufflevector(f, f, 4, 5, 6, 7), f32x4);
    const f32x4 a4 = {1.0f, 0.5f, 2.0f, 1.5f}, b4 = {0.01f, -0.02f, 0.03f, 0.0f}, z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x4 l = __builtin_convertvector(__builtin_elementwise_max(lo * a4 + b4, z4), bf16x4);
    const bf16x4 h = __builtin_convertvector(__builtin_elementwise_max(hi * a4 + b4, z4), bf16x4);
    f = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
  };
#define PIPE_LAB_NORM(f_) lab_norm(f_)
#else
#define PIPE_LAB_NORM(f_) \
  do {                    \
  } while (0)
#endif
  auto head_reads = [&](const unsigned char *ldsX, const unsigned char *ldsW, bf16x8 *brow, bf16x8 *af) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < QW + 2; ++r)
      brow[r] = *reinterpret_cast<const bf16x8 *>(ldsX + r * HC * 64 + b4_off[0][r & 3]);
#pragma unroll
    for (int m = 0; m < 2; ++m)
      af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + ((cw * 2 + m) * 16) * 64 + a4_lane);
    __builtin_amdgcn_sched_barrier(0);
#if PIPE_LAB_NORM_ON_LOAD
#pragma unroll
    for (int r = 0; r < QW + 2; ++r) PIPE_LAB_NORM(brow[r]);
#endif
  };
  // stage_mfma: groups g0 .. g1-1.  A wave's own stream must not wait for LDS: once the older wave of the SIMD
  // is through its stage, the younger one runs alone with nobody to cover its waits.  So the weight fragments of
  // group g+1 are requested before the MFMAs of group g, and the pixel fragments of the next tap column move in
  // behind the last MFMAs that read the registers they replace.
  // zero_start: group 0 starts the tile's sums (the MFMA's addend is the constant zero instead of the accumulator:
  // no accumulator is cleared anywhere).
  auto stage_mfma = [&](const int g0, const int g1, const bool zero_start, const unsigned char *ldsX, const unsigned char *ldsW,
                        bf16x8 *brow, bf16x8 *af, bf16x8 *af2) __attribute__((always_inline)) {
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      const int kw = g / 3, kh = g % 3;
      bf16x8 *cur = (g & 1) ? af2 : af, *nxt = (g & 1) ? af : af2;
      if (g < 8) {
        const int kw2 = (g + 1) / 3, kh2 = (g + 1) % 3;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          nxt[m] = *reinterpret_cast<const bf16x8 *>(ldsW + ((kh2 * 3 + kw2) * 64 + (cw * 2 + m) * 16) * 64 + a4_lane);
      }
#pragma unroll
      for (int q = 0; q < QW; ++q) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#if PIPE_LAB_ABLATE == 2
          if (a.N < 0)
#endif
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m], brow[q + kh],
                                                              (zero_start && g == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[m][q], 0, 0, 0);
        if (kh == 2 && kw < 2) {
          brow[q + 2] = *reinterpret_cast<const bf16x8 *>(ldsX + (q + 2) * HC * 64 + b4_off[kw + 1][(q + 2) & 3]);
          PIPE_LAB_NORM(brow[q + 2]);
        }
      }
      if (kh < 2 && kw < 2) {
        brow[kh] = *reinterpret_cast<const bf16x8 *>(ldsX + kh * HC * 64 + b4_off[kw + 1][kh & 3]);
        PIPE_LAB_NORM(brow[kh]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int s = 0;
  for (int it = 0; it < my_units; ++it) {
    // the next tile's halo plan (every offset out of range behind the workgroup's last tile), and the weight
    // offsets of its first chunk
    const bool more = it + 1 < my_units;
    int n_nxt = n, ty_nxt = ty, tx_nxt = tx;
    if (more) next_unit(n_nxt, ty_nxt, tx_nxt);
    plan_tile(more, ty_nxt * TH, tx_nxt * TW, hoff_nxt);
    unsigned woff_nxt[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) woff_nxt[k] = more ? w_rel[k] : OOB;

    // One stage of the tile (first / last: literal at every call, so each combination is straight-line code).
    // Opening: this wave's pieces of stage s have landed and its LDS reads of stage s-1 are complete; after the
    // barrier everybody's are, so stage s may be read and the other buffer overwritten.  The first stage of a
    // tile follows the previous tile's epilogue, whose stores were issued after these loads and need not be
    // waited for.  The tile's last stage prefetches the next tile's first chunk and requests what the epilogue
    // reads.
    auto stage = [&](const bool first, const bool last, const int kc) __attribute__((always_inline)) {
      const unsigned char *ldsX = lds + halo_off(s), *ldsW = lds + weight_off(s, kc);
      bf16x8 brow[QW + 2], af[2], af2[2];
      const int hn = last ? n_nxt : n, hs = last ? 0 : (kc + 1) * 64, ws_ = last ? 0 : (kc + 1) * slab_bytes;
      const unsigned *hv = last ? hoff_nxt : hoff, *wv = last ? woff_nxt : w_rel;
      const int dst_h = halo_off(s + 1), dst_w = weight_off(s + 1, last ? 0 : kc + 1);
      PIPE_STAMP(s, 4);
      if (first && it > 0)
        wait_vmem_then_barrier<epi_vmem_ops(EPI, QW)>();
      else
        wait_vmem_then_barrier<0>();
      PIPE_STAMP(s, 0);
      head_reads(ldsX, ldsW, brow, af);
      if (!late) {
        issue_halo(hn, hv, hs, dst_h);
        if (!WS) issue_weights(wv, ws_, dst_w);
      }
      if (last && (BRED || EPI == EPI_ACC)) {
        unsigned eoff[QW];
        bool okq[QW];
        if (BRED) {  // the parent's pre-BN values (zeros for the channel half that takes no part)
          const auto zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.bz) + (i64)n * a.H * a.W * a.bz_ld, 0,
                                                             red_here ? z_img_bytes : 0u, 0x00020000);
          out_offsets(ty, tx, eoff, okq, a.bz_ld, rc0);
#pragma unroll
          for (int q = 0; q < QW; ++q) zq[BRED ? q : 0] = __builtin_amdgcn_raw_buffer_load_b128(zrs, eoff[q], 0, 0);
        } else {
          const auto ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (i64)n * a.H * a.W * a.out_ld, 0, out_img_bytes, 0x00020000);
          out_offsets(ty, tx, eoff, okq, a.out_ld, co0);
#pragma unroll
          for (int q = 0; q < QW; ++q) oq[EPI == EPI_ACC ? q : 0] = __builtin_amdgcn_raw_buffer_load_b128(ors, eoff[q], 0, 0);
        }
      }
      stage_mfma(0, LATE_G, first, ldsX, ldsW, brow, af, af2);
      if (late) {
        issue_halo(hn, hv, hs, dst_h);
        if (!WS) issue_weights(wv, ws_, dst_w);
      }
      stage_mfma(LATE_G, 9, first && LATE_G == 0, ldsX, ldsW, brow, af, af2);
#if PIPE_LAB_STAMPS
      PIPE_STAMP(s, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PIPE_STAMP(s, 3);
#endif
      ++s;
    };
    if (nk == 1) {
      stage(true, true, 0);
    } else {
      stage(true, false, 0);
      for (int kc = 1; kc + 1 < nk; ++kc) stage(false, false, kc);
      stage(false, true, nk - 1);
    }
    epilogue(n, ty, tx, s - 1);
    n = n_nxt, ty = ty_nxt, tx = tx_nxt;
#pragma unroll
    for (int k = 0; k < KH; ++k) hoff[k] = hoff_nxt[k];
  }
#if PIPE_LAB_STAMPS
  if (a.lab_stamps) {
    __syncthreads();
    for (int i = tid; i < 8 * 128; i += 512)
      a.lab_stamps[(size_t)blockIdx.x * 1024 + i] = reinterpret_cast<unsigned long long *>(lds + G::DUMP_OFF + 1024)[i];
  }
#endif
  if (STATS && (!BRED || red_here)) {
    // One row of partial sums per (workgroup, 8-row strip position): the sums ran over all tiles of the
    // workgroup.  rows = 4 * (workgroups per channel block); lane 0 of each 16-lane row stores.
    float *row = a.stats + ((i64)(wgi * 4 + ph) * stat_width + (BRED ? rc0 : co0)) * 2;
    float o[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sv = row_sum16(ssum[j]), qv = row_sum16(ssq[j]);
      if (BRED) {
        const float4 c = lds_cf[ch16 + j];
        sv = (sv - c.z * qv) * c.w;
      }
      o[2 * j] = sv;
      o[2 * j + 1] = qv;
    }
    if (l15 == 0) {
      float *dst = row + ch16 * 2;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        *reinterpret_cast<float4 *>(dst + 4 * d) = make_float4(o[4 * d], o[4 * d + 1], o[4 * d + 2], o[4 * d + 3]);
    }
  }
}

// th = tile_rows(...): 32-row tiles, except 16-row tiles
//  - for images of fewer than 32 rows;
//  - where 32-row tiles are fewer than the 256 workgroups a launch can have (small batches, deep levels): twice the
//    tiles, twice the workgroups;
//  - for layers whose weights stay resident in LDS (C_in <= 64), where 32-row tiles would leave a workgroup fewer
//    than small_tiles_below() of them: a persistent workgroup with two tiles spends as long filling and draining
//    its pipeline as computing, and with resident weights smaller tiles cost only their halo.  (Layers that stream
//    their weights per tile lose more to the doubled weight traffic than they gain: measured.)
// Exactly one 32-row tile per workgroup stays as it is.  Both sizes add in the same order.
inline int small_tiles_below() {  // UNET_LAB_PIPE_SMALL_TILES_BELOW: laboratory switch (tiles per workgroup)
  static const int v = getenv("UNET_LAB_PIPE_SMALL_TILES_BELOW") ? atoi(getenv("UNET_LAB_PIPE_SMALL_TILES_BELOW")) : 3;
  return v;
}
inline int tile_rows(int N, int H, int W, int CiP, int CoP) {
  static const int forced = getenv("UNET_LAB_PIPE_TILE_ROWS") ? atoi(getenv("UNET_LAB_PIPE_TILE_ROWS")) : 0;  // laboratory switch
  if (forced == 16 || (forced == 32 && H >= 32)) return forced;
  if (H < 32) return 16;
  const long long units32 = (long long)N * ((H + 31) / 32) * ((W + TW - 1) / TW) * (CoP / 64);
  if (units32 < 256) return 16;
  if (CiP <= 32 * MAX_WS_CHUNKS && units32 < 256LL * small_tiles_below()) return 16;
  return 32;
}
// Launch plan: persistent workgroups, a multiple of the channel-block count, at most one per tile and channel block.
inline int plan_grid(int N, int H, int W, int CoP, int th, int max_wgs = 256) {
  const int co_blocks = CoP / 64, units = N * ((H + th - 1) / th) * ((W + TW - 1) / TW);
  int per_block = max_wgs / co_blocks;
  if (per_block < 1) per_block = 1;
  if (per_block > units) per_block = units;
  return per_block * co_blocks;
}
// Rows of partial sums the kernel writes (in all, not per image): four per workgroup of a channel block.
inline int stat_rows(int N, int H, int W, int CoP, int th, int max_wgs = 256) {
  return 4 * (plan_grid(N, H, W, CoP, th, max_wgs) / (CoP / 64));
}

// Whether the pipelined kernel serves this call.  It needs images of 16 rows and more, 64-channel output
// blocks, and enough (tile, channel block) units to give every CU a workgroup: below that (single images at
// inference, the deepest levels) the general kernel's smaller workgroups spread the work better.  The two
// kernels return bit-identical outputs, so the choice may depend on the batch size.
inline int &min_units() {  // UNET_LAB_PIPE_MIN_UNITS: laboratory switch
  static int v = getenv("UNET_LAB_PIPE_MIN_UNITS") ? atoi(getenv("UNET_LAB_PIPE_MIN_UNITS")) : 128;
  return v;
}
inline int &min_rows() {  // laboratory switch: 32 keeps the 16-row variant off
  static int v = getenv("UNET_LAB_PIPE_MIN_ROWS") ? atoi(getenv("UNET_LAB_PIPE_MIN_ROWS")) : 16;
  return v;
}
inline bool eligible(int N, int H, int W, int CiP, int CoP) {
  if (!(H >= min_rows() && H >= 16 && W >= 16 && CoP % 64 == 0 && CiP % 32 == 0)) return false;
  const int th = tile_rows(N, H, W, CiP, CoP);
  return (i64)N * ((H + th - 1) / th) * ((W + TW - 1) / TW) * (CoP / 64) >= min_units();
}

}  // namespace pipe
