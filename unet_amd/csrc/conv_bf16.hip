// bf16 implicit-GEMM convolution on the gfx950 matrix cores.
//
// Activations are NHWC bf16 (channels innermost, pixel stride `ld` elements so
// that a tensor can be a channel slice of a wider buffer -- that is how the
// skip-connection concat costs no copy).  Weights are packed per filter tap as
// [tap][C_out][C_in] bf16 (C_in innermost) by unet_pack_weights_bf16.
//
// conv_mfma_kernel<TAPS,BN>:   D[co][pixel] = sum_tap sum_ci W[tap][co][ci] * X[pixel + off(tap)][ci]
//   - one workgroup (4 waves) = one 16x16 pixel tile of one image x BN output
//     channels; wave w owns pixel rows 4w..4w+3 (two 32-pixel MFMA column
//     blocks) and all BN channels (BN/32 MFMA row blocks);
//   - v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (rows =
//     channels) and the PIXELS as the B operand (columns = pixels), so each lane
//     ends up with 4 consecutive channels of one pixel per accumulator quad:
//     8-byte packed bf16 writes into the LDS staging tile, then whole 16-byte
//     row pieces go to HBM;
//   - K loop over C_in in chunks of 32 channels: the (16+2)^2 input halo patch
//     and the 9 x BN x 32 weight slice are staged in LDS (rows padded to 80
//     bytes: 16-byte pieces of consecutive rows fall in distinct bank groups),
//     the next chunk's global loads are issued before the MFMAs of the current
//     one and written to LDS after them (register prefetch);
//   - fp32 accumulation; epilogue adds the bias, rounds to bf16 and (optionally)
//     emits per-tile per-channel sum / sum-of-squares for the batch norm that
//     follows, so the statistics pass needs no second read of the tensor.
// TAPS = 9: 3x3, pad 1, stride 1 (forward, and data gradient with flipped
// taps / transposed weights).  TAPS = 1: 1x1 with strided input / output pixel
// maps -- the four (i,j) phases of the 2x2 stride-2 transposed convolution.
//
#include "bf16_util.h"
#include "conv3x3_pipe.h"
#include "tconv_gemm.h"

namespace {

struct ConvArgs {
  const bf16_t *in;
  const bf16_t *wt;
  const float *bias;
  bf16_t *out;
  float *stats;  // [N][tiles][CoP][2] or null
  int N, Hin, Win;       // input image size (pixels that may be addressed)
  int Ho, Wo;            // logical pixel grid the tiles cover
  int Ci, CoP;           // reduction channels (multiple of 32), output channels (multiple of BN)
  int in_ld, out_ld;     // pixel strides in elements
  int in_s, in_oy, in_ox;      // input pixel of logical (y,x): (y*in_s+in_oy, x*in_s+in_ox)  [TAPS==1]
  int out_s, out_oy, out_ox;   // output pixel of logical (y,x)
  int Hout, Wout;        // output image size
  int accumulate;
  int tilesX;
  int co_valid;          // bias has this many entries
  // TAPS == 1 only: the four (i,j) phases of a 2x2 stride-2 transposed convolution in one launch.
  int out_phases;        // 4: blockIdx.y also selects the output phase (forward); else 1
  int k_phases;          // 4: the reduction also runs over the input phases (data gradient); else 1
  i64 wt_phase_stride;   // elements between the packed weights of consecutive phases
  i64 stats_phase_stride;
  // BRED kernels only (data gradient whose output is the gradient of a BN + ReLU layer's output):
  // that layer's pre-BN tensor and per-channel records {a, b, mean, 1/sqrt(var+eps)}; `stats` then
  // receives per-tile (sum g*xhat, sum g), g = dy * relu'(a*z + b).
  const bf16_t *bz;
  int bz_ld;
  const float4 *bcf;
  const float4 *affine;  // forward only, or null: per-channel records {a, b, .., ..}; the epilogue then writes
                         // relu(a * (conv + bias) + b) -- inference-time batch norm + ReLU folded into the conv
  int in_c;     // > 0: input channels present in memory (multiple of 8, < 32, single K chunk); pieces beyond
                // it are taken as zero without touching memory (first layer: 3 image channels stored as 8)
  int bred_c0;  // output channels below this one take no part (the other half of a concat's gradient);
                // bz, bcf and stats are indexed by (channel - bred_c0), stats rows hold CoP - bred_c0 channels
};

// KC = channels staged per main-loop trip: 32 for the 3x3 kernels (the 9-tap weight slab sets the LDS
// budget); 64 for the 1x1 kernels when the channel count allows, whose trips are otherwise too short to
// cover a load's latency
// MMA = 0: v_mfma_f32_32x32x16_bf16, LDS rows padded by 16 bytes.  MMA = 1 (3x3 kernels, KC = 32):
// v_mfma_f32_16x16x32_bf16 -- one instruction consumes a whole 32-channel chunk of a 16x16 output block --
// over dense 64-byte LDS rows whose 16-byte pieces are XOR-swizzled by (row >> 1) & 3, which makes every
// fragment read conflict-free and the staging buffers 20 % smaller; the chip also holds a higher clock on
// this MFMA shape.  3-5 % faster on every layer shape but the 16x16-pixel images (one workgroup per CU).
// AFF: forward-only variant that folds inference-time batch norm + ReLU into the epilogue (a.affine); a
// template parameter so that the training kernels carry none of it.
template <int TAPS, int BN, bool BRED, int KC, int MMA = 0, bool AFF = false>
__global__ __launch_bounds__(256, (TAPS == 1 && BN == 64 && KC == 64) ? 3 : 2) void conv_mfma_kernel(const ConvArgs a) {
  static_assert(!(AFF && BRED), "the folded BN + ReLU epilogue is a forward variant");
  static_assert(MMA == 0 || (TAPS == 9 && KC == 32), "the 16x16x32 path is written for the 3x3 kernels");
  constexpr int PITCH = MMA ? 64 : KC * 2 + 16;  // bytes per LDS row of KC bf16 (MMA 0: + 16-byte pad)
  constexpr int PPR = KC / 8;         // 16-byte pieces per row
  constexpr int HALO = (TAPS == 9) ? 18 : 16;
  constexpr int NPIX = HALO * HALO;
  constexpr int W_BYTES = TAPS * BN * PITCH;
  constexpr int XCH = NPIX * PPR, XI = (XCH + 255) / 256;       // 16-byte pieces of the halo patch
  constexpr int WCH = TAPS * BN * PPR, WI = (WCH + 255) / 256;  // 16-byte pieces of the weight slice
  constexpr int MF = BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *ldsW = lds, *ldsX = lds + W_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x;
  int co_blk = blockIdx.y, out_oy = a.out_oy, out_ox = a.out_ox;
  const bf16_t *wt = a.wt;
  float *stats = a.stats;
  if (TAPS == 1 && a.out_phases > 1) {
    const int cob = a.CoP / BN, ph = co_blk / cob;
    co_blk -= ph * cob;
    out_oy = ph >> 1;
    out_ox = ph & 1;
    wt += ph * a.wt_phase_stride;
    if (stats) stats += ph * a.stats_phase_stride;
  }
  const int co0 = co_blk * BN;
  const i64 n = blockIdx.z;
  const int y0 = (tile / a.tilesX) * 16, x0 = (tile % a.tilesX) * 16;
  const bf16_t *in_n = a.in + n * (i64)a.Hin * a.Win * a.in_ld;

  // per-thread staging plan (independent of the K chunk)
  int xoff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int c = tid + 256 * i, pix = c / PPR, part = c % PPR;
    const int py = pix / HALO, px = pix - py * HALO;
    const int y = y0 + py - (TAPS == 9), x = x0 + px - (TAPS == 9);
    const bool ok = c < XCH && y >= 0 && y < a.Ho && x >= 0 && x < a.Wo && (a.in_c == 0 || part * 8 < a.in_c);
    const int iy = (TAPS == 9) ? y : y * a.in_s + a.in_oy, ix = (TAPS == 9) ? x : x * a.in_s + a.in_ox;
    xoff[i] = ok ? (iy * a.Win + ix) * a.in_ld + part * 8 : -1;
  }
  const bf16_t *w_thread = wt + (i64)co0 * a.Ci;  // + (tap*CoP + co)*Ci + part*8 + k0

  const int nk1 = a.Ci / KC;  // chunks per reduction phase
  uint4 xr[XI], wr[WI];
  auto load_chunk = [&](int kc) {
    int k0 = kc * KC;
    i64 xo = k0, wo = k0;
    if (TAPS == 1 && a.k_phases > 1) {
      const int ph = kc / nk1;
      k0 = (kc - ph * nk1) * KC;
      xo = k0 + (i64)((ph >> 1) * a.Win + (ph & 1)) * a.in_ld;
      wo = k0 + ph * a.wt_phase_stride;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xr[i] = make_uint4(0, 0, 0, 0);
      if (xoff[i] >= 0) xr[i] = *reinterpret_cast<const uint4 *>(in_n + xoff[i] + xo);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i, row = c / PPR, part = c % PPR;
      const int tap = row / BN, co = row % BN;
      wr[i] = make_uint4(0, 0, 0, 0);
      if (WCH % 256 == 0 || c < WCH) {
        if (TAPS == 9)  // chunk-major operand: the whole 9 x BN x 32 slab of chunk kc is contiguous per channel block
          wr[i] = *reinterpret_cast<const uint4 *>(wt + (((i64)kc * TAPS + tap) * a.CoP + co0 + co) * 32 + part * 8);
        else
          wr[i] = *reinterpret_cast<const uint4 *>(w_thread + ((i64)tap * a.CoP + co) * a.Ci + wo + part * 8);
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = tid + 256 * i;
      const int row = c / PPR, slot = MMA ? ((c % PPR) ^ ((row >> 1) & 3)) : (c % PPR);
      if (XCH % 256 == 0 || c < XCH) *reinterpret_cast<uint4 *>(ldsX + row * PITCH + slot * 16) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int c = tid + 256 * i;
      const int row = c / PPR, slot = MMA ? ((c % PPR) ^ ((row >> 1) & 3)) : (c % PPR);
      if (WCH % 256 == 0 || c < WCH) *reinterpret_cast<uint4 *>(ldsW + row * PITCH + slot * 16) = wr[i];
    }
  };

  constexpr int MB = BN / 16;  // MMA 1: 16-channel blocks
  f32x16 acc[MMA ? 1 : MF][MMA ? 1 : 2];  // MMA 0: [32-channel block][32-pixel block]
  // MMA 1: wave = (pixel half ph: tile rows 8ph..8ph+7, channel half cw: 16-channel blocks 2cw, 2cw+1);
  // acc4[16-channel block of the wave][pixel row of the wave].  A 2 x 2 split reads 18 weight + 30 pixel
  // fragments per chunk where 1 x 4 (every wave all channels) read 36 + 18.
  constexpr int MW = MB / 2, QW = 8;
  f32x4 acc4[MMA ? MW : 1][MMA ? QW : 1];
#pragma unroll
  for (int m = 0; m < (MMA ? 1 : MF); ++m)
#pragma unroll
    for (int q = 0; q < (MMA ? 1 : 2); ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
#pragma unroll
  for (int m = 0; m < (MMA ? MW : 1); ++m)
#pragma unroll
    for (int q = 0; q < (MMA ? QW : 1); ++q) acc4[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ph = wave & 1, cw = wave >> 1;

  const int a_lane = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int b_lane = ((wave * 4 + ((lane & 31) >> 4)) * HALO + (lane & 15)) * PITCH + (lane >> 5) * 16;
  // MMA 1 fragments: lane = (row or pixel l15, 8-channel part kp) of a 16 x 32 block
  const int l15 = lane & 15, kp = lane >> 4;
  const int a4_lane = l15 * 64 + ((kp ^ ((l15 >> 1) & 3)) * 16);
  // pixel fragments: halo index p = row*18 + col + kw with row = 4*wave + q + kh; the swizzle term
  // (p >> 1) & 3 depends only on (row & 3) and on (col + kw) -- wave*4*18 is a multiple of 8
  int b4_off[MMA ? 3 : 1][MMA ? 4 : 1];
  if (MMA) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int rc = 0; rc < 4; ++rc) {
        const int pl = rc * HALO + l15 + kw;
        b4_off[MMA ? kw : 0][MMA ? rc : 0] = (l15 + kw) * 64 + ((kp ^ ((pl >> 1) & 3)) * 16) + ph * 8 * HALO * 64;
      }
  }

  // bias of this workgroup's channels: one global load per thread into LDS (behind the staging buffers),
  // read back as broadcast 16-byte pieces in the epilogue
  constexpr int MAIN_BYTES = W_BYTES + NPIX * PITCH, EPI_BYTES = 256 * (BN * 2 + 16) + 2048;
  static_assert(4 * BN * 2 * sizeof(float) <= 2048, "the 2048-byte slack holds the 4 waves x BN x 2 floats of the epilogue reduction");
  float *ldsB = reinterpret_cast<float *>(lds + (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES));
  float *ldsS = ldsB + BN;  // per-channel scale when a.affine (ldsB then holds the folded shift)
  if (!BRED && tid < BN) {
    const float bias = (a.bias && co0 + tid < a.co_valid) ? a.bias[co0 + tid] : 0.f;
    if (AFF) {
      const float4 cf = a.affine[co0 + tid];
      ldsS[tid] = cf.x;
      ldsB[tid] = fmaf(cf.x, bias, cf.y);
    } else {
      ldsB[tid] = bias;
    }
  }
  constexpr bool affine = AFF;

  const int nk = (TAPS == 1) ? nk1 * a.k_phases : nk1;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk(kc + 1);
    if (MMA) {
      // Per filter column kw the wave needs 10 halo rows (its 8 pixel rows + 2): each is read once and serves
      // the up to three (kh, q) pairs with q + kh = row.
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        bf16x8 brow[QW + 2];
#pragma unroll
        for (int r = 0; r < QW + 2; ++r)
          brow[r] = *reinterpret_cast<const bf16x8 *>(ldsX + r * HALO * 64 + b4_off[MMA ? kw : 0][MMA ? (r & 3) : 0]);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          bf16x8 af[MW];
#pragma unroll
          for (int m = 0; m < MW; ++m)
            af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + ((kh * 3 + kw) * BN + (cw * MW + m) * 16) * 64 + a4_lane);
#pragma unroll
          for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int q = 0; q < QW; ++q)
              acc4[MMA ? m : 0][MMA ? q : 0] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], brow[q + kh], acc4[MMA ? m : 0][MMA ? q : 0], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int tap = 0; tap < (MMA ? 0 : TAPS); ++tap) {
      const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8 af[MF], bfr[2];
#pragma unroll
        for (int m = 0; m < MF; ++m)
          af[m] = *reinterpret_cast<const bf16x8 *>(ldsW + (tap * BN + m * 32) * PITCH + ks * 32 + a_lane);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          bfr[q] = *reinterpret_cast<const bf16x8 *>(ldsX + ((2 * q + kh) * HALO + kw) * PITCH + ks * 32 + b_lane);
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[q], acc[m][q], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kc + 1 < nk) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> bf16 staging tile [256 pixels][BN] in LDS ----
  constexpr int OP = BN * 2 + 16;  // staging row pitch in bytes
  unsigned char *stage = lds;
  // BRED: this thread's pieces of the parent's pre-BN tensor, requested now so that they arrive
  // while the accumulators are staged
  constexpr int CHB = BN / 8;
  uint4 zq[BRED ? CHB : 1];
  float4 bcf[BRED ? 8 : 1];  // BN records of the 8 channels this thread handles in the store pass
  const bool red_here = BRED && co0 >= a.bred_c0;  // uniform over the workgroup
  const int rc0 = co0 - a.bred_c0;                 // first channel of this block in the BN layer's numbering
  if (red_here) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bcf[j] = a.bcf[rc0 + (tid % CHB) * 8 + j];
#pragma unroll
    for (int i = 0; i < CHB; ++i) {
      const int pix = tid / CHB + i * (256 / CHB);
      const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
      zq[i] = make_uint4(0, 0, 0, 0);
      if (y < a.Ho && x < a.Wo)
        zq[i] = *reinterpret_cast<const uint4 *>(
            a.bz + (n * (i64)a.Hout * a.Wout + (i64)(y * a.out_s + out_oy) * a.Wout + (x * a.out_s + out_ox)) * a.bz_ld +
            rc0 + (tid % CHB) * 8);
    }
  }
  if (MMA) {
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        const int pix = (ph * QW + q) * 16 + l15, ch = (cw * MW + m) * 16 + 4 * kp;
        f32x4 v = acc4[MMA ? m : 0][MMA ? q : 0];
        if (affine) {
          v = v * *reinterpret_cast<const f32x4 *>(ldsS + ch) + *reinterpret_cast<const f32x4 *>(ldsB + ch);
          v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
        } else if (!BRED) {
          v += *reinterpret_cast<const f32x4 *>(ldsB + ch);
        }
        *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = __builtin_convertvector(v, bf16x4);
      }
  } else {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pix = wave * 64 + q * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = m * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
          if (affine) {
            v = v * *reinterpret_cast<const f32x4 *>(ldsS + ch) + *reinterpret_cast<const f32x4 *>(ldsB + ch);
            v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
          } else if (!BRED) {
            v += *reinterpret_cast<const f32x4 *>(ldsB + ch);
          }
          const bf16x4 h = __builtin_convertvector(v, bf16x4);
          *reinterpret_cast<bf16x4 *>(stage + pix * OP + ch * 2) = h;
        }
      }
  }
  __syncthreads();

  // ---- staging tile -> HBM, 16 bytes per lane, whole channel rows; the same pass
  // accumulates the per-tile batch-norm partial sums of the rounded values ----
  constexpr int CH = BN / 8;  // 16-byte pieces per pixel; a thread keeps one piece index
  const int part = tid % CH;
  bf16_t *out_n = a.out + n * (i64)a.Hout * a.Wout * a.out_ld + co0;
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int pix = tid / CH + i * (256 / CH);
    const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
    if (y < a.Ho && x < a.Wo) {
      uint4 v = *reinterpret_cast<const uint4 *>(stage + pix * OP + part * 16);
      if (BRED && red_here) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, zz[4] = {zq[i].x, zq[i].y, zq[i].z, zq[i].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t vw = vv[j >> 1], zw = zz[j >> 1];
          const float dy = (j & 1) ? bf16_to_f32(vw >> 16) : bf16_to_f32(vw & 0xffffu);
          const float zv = (j & 1) ? bf16_to_f32(zw >> 16) : bf16_to_f32(zw & 0xffffu);
          // g and the two sums stay scalar arithmetic (see keep_scalar in bf16_util.h: the packed form went
          // wrong beside waves of another MFMA kernel)
          const float g = keep_scalar(fmaf(bcf[j].x, zv, bcf[j].y) > 0.f ? dy : 0.f);
          ssum[j] = keep_scalar(fmaf(g, zv, ssum[j]));  // sum g*z; turned into sum g*xhat after the loop
          ssq[j] = keep_scalar(ssq[j] + g);
        }
      } else if (stats) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = bf16_to_f32(vv[j] & 0xffffu), hi = bf16_to_f32(vv[j] >> 16);
          ssum[2 * j] += lo; ssq[2 * j] += lo * lo;
          ssum[2 * j + 1] += hi; ssq[2 * j + 1] += hi * hi;
        }
      }
      bf16_t *dst = out_n + ((i64)(y * a.out_s + out_oy) * a.Wout + (x * a.out_s + out_ox)) * a.out_ld + part * 8;
      if (a.accumulate) {
        const uint4 o = *reinterpret_cast<const uint4 *>(dst);
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, oo[4] = {o.x, o.y, o.z, o.w};
        uint32_t rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = bf16_to_f32(vv[j] & 0xffffu) + bf16_to_f32(oo[j] & 0xffffu);
          const float hi = bf16_to_f32(vv[j] >> 16) + bf16_to_f32(oo[j] >> 16);
          rr[j] = f32_to_bf16(lo) | (f32_to_bf16(hi) << 16);
        }
        v = make_uint4(rr[0], rr[1], rr[2], rr[3]);
      }
      *reinterpret_cast<uint4 *>(dst) = v;
    }
  }
  if (BRED && red_here) {  // sum g*xhat = (sum g*z - mean * sum g) / sqrt(var + eps)
#ifdef UNET_LAB_BRED_FENCE  // tools/dpp_lab.sh: an instruction pinned here in laboratory builds
    __builtin_amdgcn_sched_barrier(0);
    asm volatile(UNET_LAB_BRED_FENCE ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[j] = (ssum[j] - bcf[j].z * ssq[j]) * bcf[j].w;
  }
  if (BRED ? red_here : (stats != nullptr)) {
    // lanes with equal (lane % CH) hold the same 8 channels: sum over them (VALU cross-lane ops),
    // then one partial per wave through LDS, summed in wave order (deterministic)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ssum[j] = lane_class_sum<CH>(ssum[j]);
      ssq[j] = lane_class_sum<CH>(ssq[j]);
    }
    float *red = reinterpret_cast<float *>(lds + 256 * OP);  // [wave][BN][2]
    if (lane < CH) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * BN + lane * 8 + j) * 2] = ssum[j];
        red[(wave * BN + lane * 8 + j) * 2 + 1] = ssq[j];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const float t = (red[tid] + red[2 * BN + tid]) + (red[4 * BN + tid] + red[6 * BN + tid]);
      if (BRED)
        stats[((n * gridDim.x + tile) * (i64)(a.CoP - a.bred_c0) + rc0) * 2 + tid] = t;
      else
        stats[((n * gridDim.x + tile) * (i64)a.CoP + co0) * 2 + tid] = t;
    }
  }
}

template <int TAPS, int BN, int KC, int MMA = 0>
constexpr size_t conv_lds_bytes() {
  constexpr int HALO = (TAPS == 9) ? 18 : 16;
  constexpr int PITCH = MMA ? 64 : KC * 2 + 16;
  constexpr size_t main_loop = (size_t)TAPS * BN * PITCH + (size_t)HALO * HALO * PITCH;
  constexpr size_t epilogue = (size_t)256 * (BN * 2 + 16) + 2048;
  return (main_loop > epilogue ? main_loop : epilogue) + 2 * BN * sizeof(float);  // + bias and scale vectors
}

template <int TAPS, int BN, bool BRED, int KC = 32, int MMA = 0, bool AFF = false>
int launch_conv(const ConvArgs &a, hipStream_t s) {
  static bool attr_set[64] = {};
  constexpr size_t lds = conv_lds_bytes<TAPS, BN, KC, MMA>();
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<TAPS, BN, BRED, KC, MMA, AFF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev & 63] = true;
  }
  const int tilesY = (a.Ho + 15) / 16;
  dim3 grid((unsigned)(tilesY * a.tilesX), (unsigned)(a.CoP / BN * (a.out_phases > 1 ? a.out_phases : 1)),
            (unsigned)a.N);
  hipLaunchKernelGGL((conv_mfma_kernel<TAPS, BN, BRED, KC, MMA, AFF>), grid, dim3(256), lds, s, a);
  UNET_LAUNCH_CHECK("conv_mfma_kernel");
  return UNET_OK;
}

int launch_pipe_from(const ConvArgs &c, hipStream_t s);

// Launches of at most this many 64-channel workgroups go out as 32-channel blocks (inference forward).
int narrow_block_workgroups() {
  static const int v = [] {
    const char *e = getenv("UNET_LAB_CONV_NARROW_WGS");
    return e ? atoi(e) : 128;
  }();
  return v;
}

// The transposed convolution as one matrix product (tconv_gemm.h) for the shapes it serves.  Chosen by SHAPE only,
// never by batch size: its sums run in another order than the general kernel's, and one image must give the same
// bits alone and inside a batch.
// Inference (the folded BN + ReLU epilogue) stays with the general kernel: measured with that epilogue added to the
// matrix product, batch 16 gains 1.2 % and one image loses 1 % (16 to 128 workgroups where the general kernel's
// 32-channel blocks make twice as many), and the two cannot be mixed by batch size for the reason above.
bool tconv_gemm_serves(const ConvArgs &c) {
  const bool fwd = c.out_phases == 4 && c.k_phases == 1 && c.in_s == 1 && c.out_s == 2;
  const bool dgrad = c.out_phases == 1 && c.k_phases == 4 && c.in_s == 2 && c.out_s == 1;
  if (tgemm::lab_off() || !(fwd || dgrad) || c.affine || c.in_c != 0 || c.bred_c0 != 0) return false;
  if (c.in_oy | c.in_ox | c.out_oy | c.out_ox) return false;
  if (fwd && c.accumulate) return false;
  const int cols = fwd ? 4 * c.CoP : c.CoP;
  if (c.Ci % tgemm::KC != 0 || c.CoP % 32 != 0 || cols % tgemm::COLS != 0) return false;
  if ((((uintptr_t)c.in | (uintptr_t)c.wt | (uintptr_t)c.out | (uintptr_t)c.bz | (uintptr_t)c.bcf) & 15) != 0) return false;
  if (c.stats && (((uintptr_t)c.stats) & 15) != 0) return false;
  // lanes outside the image address 2^31 bytes into their descriptor: every image must end below that
  const i64 lim = 1ll << 31;
  if ((i64)c.Hin * c.Win * c.in_ld * 2 >= lim || (i64)c.Hout * c.Wout * c.out_ld * 2 >= lim ||
      (c.bz && (i64)c.Hout * c.Wout * c.bz_ld * 2 >= lim))
    return false;
  return (i64)4 * c.CoP * c.Ci * 2 < lim;
}
int launch_tconv_gemm(const ConvArgs &c, hipStream_t s) {
  const bool fwd = c.out_phases == 4;
  tgemm::Args a{};
  a.in = c.in; a.wt = c.wt; a.bias = c.bias; a.out = c.out; a.stats = c.stats;
  a.N = c.N; a.Hin = c.Hin; a.Win = c.Win; a.Ho = c.Ho; a.Wo = c.Wo; a.Hout = c.Hout; a.Wout = c.Wout;
  a.nk1 = c.Ci / tgemm::KC; a.in_s = c.in_s; a.out_s = c.out_s; a.in_ld = c.in_ld; a.out_ld = c.out_ld;
  a.w_row = c.Ci; a.wt_phase_stride = c.wt_phase_stride; a.wt_bytes = (unsigned)((i64)4 * c.CoP * c.Ci * 2);
  a.CoP = c.CoP; a.col_tiles = (fwd ? 4 * c.CoP : c.CoP) / tgemm::COLS;
  a.tilesX = c.tilesX; a.tiles = c.tilesX * ((c.Ho + 15) / 16);
  a.co_valid = c.co_valid; a.stats_phase_stride = c.stats_phase_stride;
  a.bz = c.bz; a.bz_ld = c.bz_ld; a.bcf = c.bcf;
  const i64 grid = (i64)c.N * a.tiles * a.col_tiles;
  if (grid >= (1ll << 31)) return -1;
  a.xcd_order = grid % 8 == 0 ? 1 : 0;
  if (fwd) return c.stats ? tgemm::launch<1, tgemm::EPI_STATS>(a, (int)grid, s) : tgemm::launch<1, tgemm::EPI_PLAIN>(a, (int)grid, s);
  if (c.bz) return tgemm::launch<4, tgemm::EPI_BRED>(a, (int)grid, s);
  return c.accumulate ? tgemm::launch<4, tgemm::EPI_ACC>(a, (int)grid, s) : tgemm::launch<4, tgemm::EPI_PLAIN>(a, (int)grid, s);
}

int launch_conv_any(int taps, const ConvArgs &a, hipStream_t s) {
  if (taps == 9 && pipe::eligible(a.N, a.Ho, a.Wo, a.Ci, a.CoP)) return launch_pipe_from(a, s);
  if (taps == 1 && tconv_gemm_serves(a)) {
    const int rc = launch_tconv_gemm(a, s);
    if (rc >= 0) return rc;
  }
  const bool wide = a.CoP % 64 == 0, deep = a.Ci % 64 == 0;
  // the 16x16x32 path for every image the pipelined kernel could take at some batch size (16 x 16 pixels and
  // more), and for smaller ones with many tiles.  The choice must not depend on the batch size: the two loops
  // add in different orders, and inference on one image has to give bit for bit what the same image gives
  // inside a batch -- whichever of the two kernels the batch size selects.
  const bool mma16 = taps == 9 && wide &&
                     ((a.Ho >= 16 && a.Wo >= 16) || (i64)a.tilesX * ((a.Ho + 15) / 16) * (a.CoP / 64) > 16);
  if (a.affine) {  // inference forward
    // Few images: a launch of 64-channel blocks would leave most of the device idle (one image of 16 x 16 pixels
    // and 1024 channels is 16 workgroups), and a workgroup alone on its CU runs at the pace of its own LDS reads
    // and MFMAs.  Blocks of 32 channels double the workgroups; every output is still added up in the same order.
    // (Two 32-channel chunks per main-loop trip on top of that measured nothing: the trips are not waiting for
    // their loads.)
    if (mma16 && (i64)a.N * a.tilesX * ((a.Ho + 15) / 16) * (a.CoP / 64) <= narrow_block_workgroups())
      return launch_conv<9, 32, false, 32, 1, true>(a, s);
    if (mma16) return launch_conv<9, 64, false, 32, 1, true>(a, s);
    if (taps == 9) return wide ? launch_conv<9, 64, false, 32, 0, true>(a, s) : launch_conv<9, 32, false, 32, 0, true>(a, s);
    if (wide && deep && (i64)a.N * a.tilesX * ((a.Ho + 15) / 16) * (a.CoP / 64) * (a.out_phases > 1 ? a.out_phases : 1) <=
                            narrow_block_workgroups())
      return launch_conv<1, 32, false, 64, 0, true>(a, s);  // the same ascending sum over the input channels
    if (wide) return deep ? launch_conv<1, 64, false, 64, 0, true>(a, s) : launch_conv<1, 64, false, 32, 0, true>(a, s);
    return launch_conv<1, 32, false, 32, 0, true>(a, s);
  }
  if (mma16 && (i64)a.N * a.tilesX * ((a.Ho + 15) / 16) * (a.CoP / 64) <= narrow_block_workgroups())  // as above
    return a.bz ? launch_conv<9, 32, true, 32, 1>(a, s) : launch_conv<9, 32, false, 32, 1>(a, s);
  if (mma16) return a.bz ? launch_conv<9, 64, true, 32, 1>(a, s) : launch_conv<9, 64, false, 32, 1>(a, s);
  if (a.bz) {
    if (taps == 9) return wide ? launch_conv<9, 64, true>(a, s) : launch_conv<9, 32, true>(a, s);
    if (wide) return deep ? launch_conv<1, 64, true, 64>(a, s) : launch_conv<1, 64, true>(a, s);
    return launch_conv<1, 32, true>(a, s);
  }
  if (taps == 9) return wide ? launch_conv<9, 64, false>(a, s) : launch_conv<9, 32, false>(a, s);
  if (wide) return deep ? launch_conv<1, 64, false, 64>(a, s) : launch_conv<1, 64, false>(a, s);
  return launch_conv<1, 32, false>(a, s);
}

// The pipelined persistent kernel (conv3x3_pipe.h) for the shapes it serves; returns false when the shape
// stays with conv_mfma_kernel.  Both add in the same order: their outputs are bit-identical, so one image gives
// the same result alone (general kernel) and inside a batch (pipelined kernel).
template <bool WS, int EPI, int TH>
int launch_pipe(const pipe::Args &a, int grid, hipStream_t s) {
  static bool attr_set[64] = {};
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<WS, EPI, TH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, pipe::Geo<TH>::LDS_BYTES));
    attr_set[dev & 63] = true;
  }
  hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<WS, EPI, TH>), dim3((unsigned)grid), dim3(512), pipe::Geo<TH>::LDS_BYTES, s,
                     a);
  UNET_LAUNCH_CHECK("conv3x3_pipe_kernel");
  return UNET_OK;
}
template <int EPI>
int launch_pipe_ws(const pipe::Args &a, int grid, hipStream_t s) {
  if (pipe::tile_rows(a.N, a.H, a.W, a.nk * 32, a.CoP) == 16)
    return a.nk <= pipe::MAX_WS_CHUNKS ? launch_pipe<true, EPI, 16>(a, grid, s) : launch_pipe<false, EPI, 16>(a, grid, s);
  return a.nk <= pipe::MAX_WS_CHUNKS ? launch_pipe<true, EPI, 32>(a, grid, s) : launch_pipe<false, EPI, 32>(a, grid, s);
}
int launch_pipe_from(const ConvArgs &c, hipStream_t s) {
  pipe::Args a{};
  a.in = c.in; a.wt = c.wt; a.bias = c.bias; a.out = c.out; a.stats = c.stats;
  a.N = c.N; a.H = c.Ho; a.W = c.Wo; a.nk = c.Ci / 32; a.CoP = c.CoP;
  a.in_ld = c.in_ld; a.out_ld = c.out_ld; a.in_c = c.in_c;
  const int th = pipe::tile_rows(c.N, c.Ho, c.Wo, c.Ci, c.CoP);
  a.tilesX = (c.Wo + pipe::TW - 1) / pipe::TW; a.tilesY = (c.Ho + th - 1) / th;
  a.units = c.N * a.tilesX * a.tilesY; a.co_blocks = c.CoP / 64; a.co_valid = c.co_valid;
  a.accumulate = c.accumulate;
  {
    // Data gradients of 60 MB and more are stored non-temporal: -3 % on that family (same-box A/B), and the
    // batch-norm backward pass that reads them next runs as before.  Forward outputs stay cached: the
    // normalisation pass that reads them right after the convolution was 3-6 % slower behind non-temporal stores.
    // UNET_LAB_CONV_NT_BYTES: laboratory switch (threshold in bytes for every launch; 0 = never).
    static const double lab_bytes = getenv("UNET_LAB_CONV_NT_BYTES") ? atof(getenv("UNET_LAB_CONV_NT_BYTES")) : -1;
    static const double dgrad_bytes = getenv("UNET_LAB_CONV_NT_DGRAD_BYTES") ? atof(getenv("UNET_LAB_CONV_NT_DGRAD_BYTES")) : 60e6;
    const double out_bytes = 2.0 * c.N * c.Ho * c.Wo * c.CoP;
    const bool data_gradient = c.bz || (!c.stats && !c.affine);
    // The tiles are walked from the last image to the first: the pass that produced the input wrote it in
    // ascending order, so its end is what the caches still hold, and the pass that reads the output next
    // (ascending again) starts with what this kernel wrote last.  Step -0.4 %.  UNET_LAB_CONV_REVERSE=0: laboratory switch.
    static const int lab_reverse = getenv("UNET_LAB_CONV_REVERSE") ? atoi(getenv("UNET_LAB_CONV_REVERSE")) : 1;
    a.reverse = lab_reverse != 0;
    a.nt_store = lab_bytes >= 0 ? (lab_bytes > 0 && out_bytes >= lab_bytes) : (data_gradient && out_bytes >= dgrad_bytes);
  }
  a.bz = c.bz; a.bz_ld = c.bz_ld; a.bcf = c.bcf; a.bred_c0 = c.bred_c0; a.affine = c.affine;
  const int grid = pipe::plan_grid(c.N, c.Ho, c.Wo, c.CoP, th);
  static const bool xcd_off = getenv("UNET_LAB_NO_XCD_ORDER") != nullptr;  // laboratory switch (same results either way)
  a.xcd_order = (!xcd_off && grid % (8 * a.co_blocks) == 0) ? 1 : 0;
  if (c.affine) return launch_pipe_ws<pipe::EPI_AFF>(a, grid, s);
  if (c.bz) return launch_pipe_ws<pipe::EPI_BRED>(a, grid, s);
  if (c.accumulate) return launch_pipe_ws<pipe::EPI_ACC>(a, grid, s);
  if (c.stats) return launch_pipe_ws<pipe::EPI_STATS>(a, grid, s);
  return launch_pipe_ws<pipe::EPI_PLAIN>(a, grid, s);
}

// ---------------------------------------------------------------------------
// Weight packing and layout conversion.
// ---------------------------------------------------------------------------
// w_ref [Ci][Co][K][K] fp32 ->
//   K=3: fwd [CiP/32][tap][CoP][32] bf16 (tap = kh*3+kw, zero padded): the 32-channel K chunks of the
//        conv kernel's main loop are the outer index, so the slab a workgroup stages per chunk is
//        contiguous runs of whole cache lines;
//        bwd [CoP/32][tap][CiP][32] with flipped taps (data gradient as a forward conv over dZ)
//   K=2: fwd [phase][CoP][CiP], bwd [phase][CiP][CoP] (the (i,j) phases of the transposed conv)
// A workgroup moves a 32(ci) x 32(co) x taps box: contiguous 16-byte reads of the reference layout
// into LDS, then each lane packs 8 consecutive bf16 and stores 16 bytes: 64-byte runs along ci (fwd)
// and along co (bwd).
template <int K>
__device__ __forceinline__ void pack_box(const float *__restrict__ w, bf16_t *__restrict__ fwd, bf16_t *__restrict__ bwd,
                                         int Ci, int Co, int CiP, int CoP, int box_x, int box_y, float *tile) {
  constexpr int taps = K * K, run = 32 * taps, pitch = run + 1;
  const int ci0 = box_x * 32, co0 = box_y * 32;
  const bool full = ci0 + 32 <= Ci && co0 + 32 <= Co && (Co * taps) % 4 == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
  if (full) {
    for (int idx = threadIdx.x; idx < 32 * (run / 4); idx += 256) {
      const int cl = idx / (run / 4), r4 = idx - cl * (run / 4);
      const float4 v = *reinterpret_cast<const float4 *>(w + ((i64)(ci0 + cl) * Co + co0) * taps + 4 * r4);
      float *d = tile + cl * pitch + 4 * r4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
      const int cl = idx / run, rem = idx - cl * run;
      const int ci = ci0 + cl, co = co0 + rem / taps;
      tile[cl * pitch + rem] = (ci < Ci && co < Co) ? w[((i64)ci * Co + co0) * taps + rem] : 0.f;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < taps * 128; idx += 256) {
    const int a = (idx & 3) * 8, b = (idx >> 2) & 31, t = idx >> 7;
    if (fwd) {  // a = first of 8 consecutive ci, b = co
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = f32_to_bf16(tile[(a + 2 * j) * pitch + b * taps + t]) |
               (f32_to_bf16(tile[(a + 2 * j + 1) * pitch + b * taps + t]) << 16);
      bf16_t *dst = (K == 3) ? fwd + (((i64)box_x * taps + t) * CoP + co0 + b) * 32 + a   // [ci chunk][tap][co][32]
                             : fwd + ((i64)t * CoP + co0 + b) * CiP + ci0 + a;            // [phase][co][ci]
      *reinterpret_cast<uint4 *>(dst) = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (bwd) {  // a = first of 8 consecutive co, b = ci
      const int tb = (K == 3) ? (taps - 1 - t) : t;
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = f32_to_bf16(tile[b * pitch + (a + 2 * j) * taps + t]) |
               (f32_to_bf16(tile[b * pitch + (a + 2 * j + 1) * taps + t]) << 16);
      bf16_t *dst = (K == 3) ? bwd + (((i64)box_y * taps + tb) * CiP + ci0 + b) * 32 + a  // [co chunk][tap][ci][32]
                             : bwd + ((i64)tb * CiP + ci0 + b) * CoP + co0 + a;           // [phase][ci][co]
      *reinterpret_cast<uint4 *>(dst) = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, bf16_t *__restrict__ fwd,
                                                           bf16_t *__restrict__ bwd, int Ci, int Co, int CiP,
                                                           int CoP) {
  __shared__ float tile[32 * (32 * K * K + 1)];
  pack_box<K>(w, fwd, bwd, Ci, Co, CiP, CoP, blockIdx.x, blockIdx.y, tile);
}

// All layers of a network in one launch.  table[l] = {w_ref, w_fwd, w_bwd, Ci, Co, CiP, CoP, K, first_box, 0}
// (pointers as 64-bit integers); workgroup b serves the layer with first_box <= b < next first_box.
constexpr int PACK_ROW = 10;
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const int64_t *__restrict__ table, int layers) {
  __shared__ float tile[32 * (32 * 9 + 1)];
  const int b = blockIdx.x;
  int lo = 0, hi = layers - 1;  // last layer with first_box <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * PACK_ROW + 8] <= b) lo = mid; else hi = mid - 1;
  }
  const int64_t *r = table + lo * PACK_ROW;
  const float *w = reinterpret_cast<const float *>(r[0]);
  bf16_t *fwd = reinterpret_cast<bf16_t *>(r[1]), *bwd = reinterpret_cast<bf16_t *>(r[2]);
  const int Ci = (int)r[3], Co = (int)r[4], CiP = (int)r[5], CoP = (int)r[6], K = (int)r[7];
  const int box = b - (int)r[8], bx = box % (CiP / 32), by = box / (CiP / 32);
  if (K == 3)
    pack_box<3>(w, fwd, bwd, Ci, Co, CiP, CoP, bx, by, tile);
  else
    pack_box<2>(w, fwd, bwd, Ci, Co, CiP, CoP, bx, by, tile);
}

// NCHW fp32 -> NHWC bf16 with pixel stride ld; channels [C, Cpad) are zero-filled.
// One block transposes a 32-pixel x 32-channel tile through LDS.  ACC: the converted values are added to what
// `out` holds (channels [0, C) only).
template <bool ACC>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ in, bf16_t *__restrict__ out,
                                                           int C, i64 HW, int ld, int Cpad) {
  __shared__ float t[32][33];
  const i64 n = blockIdx.z;
  const i64 p0 = (i64)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows of 32
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const i64 p = p0 + tx;
    t[r][tx] = (c < C && p < HW) ? in[(n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const i64 p = p0 + r;
    const int c = c0 + tx;
    if (p < HW && c < Cpad) {
      bf16_t *o = out + (n * HW + p) * ld + c;
      *o = (bf16_t)f32_to_bf16(ACC ? t[tx][r] + bf16_to_f32(*o) : t[tx][r]);
    }
  }
}

// The same for at most 8 channels stored as exactly 8 (ld = 8): one thread per pixel, coalesced plane reads,
// one 16-byte store.
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float *__restrict__ in, bf16_t *__restrict__ out, int C,
                                                            i64 HW) {
  const i64 n = blockIdx.y;
  const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  uint32_t h[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) h[c] = c < C ? f32_to_bf16(in[(n * C + c) * HW + p]) : 0u;
  *reinterpret_cast<uint4 *>(out + (n * HW + p) * 8) =
      make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

// NHWC bf16 (pixel stride ld) -> NCHW fp32, channels [0, C)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16_t *__restrict__ in, float *__restrict__ out,
                                                           int C, i64 HW, int ld) {
  __shared__ float t[32][33];
  const i64 n = blockIdx.z;
  const i64 p0 = (i64)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const i64 p = p0 + r;
    const int c = c0 + tx;
    t[r][tx] = (p < HW && c < C) ? bf16_to_f32(in[(n * HW + p) * ld + c]) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const i64 p = p0 + tx;
    if (c < C && p < HW) out[(n * C + c) * HW + p] = t[tx][r];
  }
}

}  // namespace

extern "C" {

int unet_pack_weights_bf16(const float *w_ref, uint16_t *w_fwd, uint16_t *w_bwd, int Ci, int Co, int CiP,
                           int CoP, int K, unet_stream_t s) {
  UNET_REQUIRE(w_ref && (w_fwd || w_bwd), "unet_pack_weights_bf16: NULL pointer");
  UNET_REQUIRE(Ci > 0 && Co > 0 && CiP >= Ci && CoP >= Co && (K == 3 || K == 2) && CiP % 32 == 0 && CoP % 32 == 0,
               "unet_pack_weights_bf16: need K in {2,3} and padded channel counts that are multiples of 32");
  const dim3 grid((unsigned)(CiP / 32), (unsigned)(CoP / 32));
  if (K == 3)
    hipLaunchKernelGGL(pack_weights_kernel<3>, grid, dim3(256), 0, as_stream(s), w_ref, w_fwd, w_bwd, Ci, Co, CiP, CoP);
  else
    hipLaunchKernelGGL(pack_weights_kernel<2>, grid, dim3(256), 0, as_stream(s), w_ref, w_fwd, w_bwd, Ci, Co, CiP, CoP);
  UNET_LAUNCH_CHECK("pack_weights_kernel");
  return UNET_OK;
}

int unet_pack_weights_batch_bf16(const int64_t *table_device, int layers, int total_boxes, unet_stream_t s) {
  UNET_REQUIRE(table_device && layers > 0 && total_boxes > 0, "unet_pack_weights_batch_bf16: bad arguments");
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)total_boxes), dim3(256), 0, as_stream(s), table_device, layers);
  UNET_LAUNCH_CHECK("pack_weights_batch_kernel");
  return UNET_OK;
}

int unet_nchw_f32_to_nhwc_bf16(const float *in, uint16_t *out, int N, int C, int H, int W, int ld, int Cpad,
                               unet_stream_t s) {
  UNET_REQUIRE(in && out, "unet_nchw_f32_to_nhwc_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && ld >= Cpad && N <= 65535,
               "unet_nchw_f32_to_nhwc_bf16: bad sizes");
  const i64 HW = (i64)H * W;
  if (Cpad == 8 && ld == 8 && ((uintptr_t)out & 15) == 0) {
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(cdiv_u(HW, 256), (unsigned)N), dim3(256), 0, as_stream(s), in, out, C, HW);
    UNET_LAUNCH_CHECK("nchw_to_nhwc8_kernel");
    return UNET_OK;
  }
  dim3 grid(cdiv_u(HW, 32), cdiv_u(Cpad, 32), (unsigned)N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<false>, grid, dim3(256), 0, as_stream(s), in, out, C, HW, ld, Cpad);
  UNET_LAUNCH_CHECK("nchw_to_nhwc_kernel");
  return UNET_OK;
}

int unet_nchw_f32_add_to_nhwc_bf16(const float *in, uint16_t *out, int N, int C, int H, int W, int ld,
                                   unet_stream_t s) {
  UNET_REQUIRE(in && out, "unet_nchw_f32_add_to_nhwc_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && N <= 65535, "unet_nchw_f32_add_to_nhwc_bf16: bad sizes");
  const i64 HW = (i64)H * W;
  dim3 grid(cdiv_u(HW, 32), cdiv_u(C, 32), (unsigned)N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<true>, grid, dim3(256), 0, as_stream(s), in, out, C, HW, ld, C);
  UNET_LAUNCH_CHECK("nchw_to_nhwc_kernel");
  return UNET_OK;
}

int unet_nhwc_bf16_to_nchw_f32(const uint16_t *in, float *out, int N, int C, int H, int W, int ld,
                               unet_stream_t s) {
  UNET_REQUIRE(in && out, "unet_nhwc_bf16_to_nchw_f32: NULL pointer");
  UNET_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && N <= 65535, "unet_nhwc_bf16_to_nchw_f32: bad sizes");
  const i64 HW = (i64)H * W;
  dim3 grid(cdiv_u(HW, 32), cdiv_u(C, 32), (unsigned)N);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, as_stream(s), in, out, C, HW, ld);
  UNET_LAUNCH_CHECK("nhwc_to_nchw_kernel");
  return UNET_OK;
}

static int conv3x3_impl(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                        uint16_t *out, int out_ld, float *stat_partials, int N, int H, int W, int CiP, int CoP,
                        int accumulate, const float4 *affine, unet_stream_t s) {
  UNET_REQUIRE(in && w_packed && out, "unet_conv3x3_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "unet_conv3x3_bf16: bad sizes");
  // (the statistics describe what this call computed, not the sum it leaves in `out`; and the row count
  // unet_conv3x3_stat_rows promises holds for plain calls only)
  UNET_REQUIRE(!(accumulate && stat_partials), "unet_conv3x3_bf16: stat_partials cannot be combined with accumulate");
  UNET_REQUIRE(CiP > 0 && (CiP % 32 == 0 || (CiP < 32 && CiP % 8 == 0)) && CoP > 0 && CoP % 32 == 0 && in_ld >= CiP &&
                   out_ld >= CoP && in_ld % 8 == 0 && out_ld % 8 == 0,
               "unet_conv3x3_bf16: CiP must be a multiple of 32 (or 8, 16, 24 with weights packed for 32), CoP a "
               "multiple of 32, pixel strides multiples of 8");
  UNET_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)w_packed) & 15) == 0,
               "unet_conv3x3_bf16: pointers must be 16-byte aligned");
  const int in_present = CiP < 32 ? CiP : 0;  // fewer than 32 input channels in memory: one K chunk, tail pieces zero
  if (CiP < 32) CiP = 32;
  UNET_REQUIRE((i64)H * W * in_ld < (1ll << 31) && (i64)H * W * out_ld < (1ll << 31),
               "unet_conv3x3_bf16: one image must stay below 2^31 elements");
  ConvArgs a{};
  a.in = in; a.wt = w_packed; a.bias = bias; a.out = out; a.stats = stat_partials;
  a.N = N; a.Hin = H; a.Win = W; a.Ho = H; a.Wo = W; a.Ci = CiP; a.CoP = CoP;
  a.in_ld = in_ld; a.out_ld = out_ld; a.in_s = 1; a.in_oy = a.in_ox = 0;
  a.out_s = 1; a.out_oy = a.out_ox = 0; a.Hout = H; a.Wout = W; a.accumulate = accumulate;
  a.tilesX = (W + 15) / 16;
  a.co_valid = co_valid;
  a.out_phases = a.k_phases = 1;
  a.in_c = in_present;
  a.affine = affine;
  const double flops = 2.0 * N * H * W * (double)CiP * CoP * 9;
  ProfScope prof(stat_partials || bias ? PROF_CONV_FWD_BF16 : PROF_CONV_DGRAD_BF16, flops,
                 2.0 * N * H * W * ((double)CiP + CoP), as_stream(s));
  return launch_conv_any(9, a, as_stream(s));
}

int unet_conv3x3_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                      uint16_t *out, int out_ld, float *stat_partials, int N, int H, int W, int CiP, int CoP,
                      int accumulate, unet_stream_t s) {
  return conv3x3_impl(in, in_ld, w_packed, bias, co_valid, out, out_ld, stat_partials, N, H, W, CiP, CoP, accumulate,
                      nullptr, s);
}

int unet_conv3x3_bn_relu_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                              const float *bn_coef, uint16_t *out, int out_ld, int N, int H, int W, int CiP, int CoP,
                              unet_stream_t s) {
  UNET_REQUIRE(bn_coef && (((uintptr_t)bn_coef) & 15) == 0,
               "unet_conv3x3_bn_relu_bf16: bn_coef must be a 16-byte aligned pointer");
  return conv3x3_impl(in, in_ld, w_packed, bias, co_valid, out, out_ld, nullptr, N, H, W, CiP, CoP, 0,
                      reinterpret_cast<const float4 *>(bn_coef), s);
}

int unet_conv3x3_stat_tiles(int H, int W) { return ((H + 15) / 16) * ((W + 15) / 16); }

int unet_conv3x3_pipeline_min_units(int min_units) {
  const int old = pipe::min_units();
  if (min_units >= 0) pipe::min_units() = min_units;
  return old;
}

int unet_conv3x3_stat_rows(int N, int H, int W, int CiP, int CoP) {
  if (CiP < 32) CiP = 32;
  if (pipe::eligible(N, H, W, CiP, CoP)) return pipe::stat_rows(N, H, W, CoP, pipe::tile_rows(N, H, W, CiP, CoP));
  return N * unet_conv3x3_stat_tiles(H, W);
}

int unet_conv3x3_dgrad_bnred_bf16(const uint16_t *dz, int dz_ld, const uint16_t *w_packed_bwd, uint16_t *dx,
                                  int dx_ld, const uint16_t *bn_z, int bn_z_ld, const float *bn_coef,
                                  float *red_partials, int bn_first_channel, int N, int H, int W, int CiP, int CoP,
                                  unet_stream_t s) {
  UNET_REQUIRE(dz && w_packed_bwd && dx && bn_z && bn_coef && red_partials, "unet_conv3x3_dgrad_bnred_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "unet_conv3x3_dgrad_bnred_bf16: bad sizes");
  UNET_REQUIRE(CiP > 0 && CiP % 32 == 0 && CoP > 0 && CoP % 32 == 0 && dz_ld >= CiP && dx_ld >= CoP &&
                   dz_ld % 8 == 0 && dx_ld % 8 == 0 && bn_z_ld % 8 == 0,
               "unet_conv3x3_dgrad_bnred_bf16: channel counts must be multiples of 32 and pixel strides multiples of 8");
  UNET_REQUIRE(bn_first_channel >= 0 && bn_first_channel < CoP && bn_first_channel % (CoP % 64 == 0 ? 64 : 32) == 0 &&
                   bn_z_ld >= CoP - bn_first_channel,
               "unet_conv3x3_dgrad_bnred_bf16: bn_first_channel must be a multiple of the 64-channel block (32 when "
               "CoP is not a multiple of 64) and bn_z must hold CoP - bn_first_channel channels");
  UNET_REQUIRE((((uintptr_t)dz | (uintptr_t)dx | (uintptr_t)w_packed_bwd | (uintptr_t)bn_z | (uintptr_t)bn_coef) & 15) == 0,
               "unet_conv3x3_dgrad_bnred_bf16: pointers must be 16-byte aligned");
  UNET_REQUIRE((i64)H * W * dz_ld < (1ll << 31) && (i64)H * W * dx_ld < (1ll << 31) && (i64)H * W * bn_z_ld < (1ll << 31),
               "unet_conv3x3_dgrad_bnred_bf16: one image must stay below 2^31 elements");
  ConvArgs a{};
  a.in = dz; a.wt = w_packed_bwd; a.bias = nullptr; a.out = dx; a.stats = red_partials;
  a.N = N; a.Hin = H; a.Win = W; a.Ho = H; a.Wo = W; a.Ci = CiP; a.CoP = CoP;
  a.in_ld = dz_ld; a.out_ld = dx_ld; a.in_s = 1; a.out_s = 1; a.Hout = H; a.Wout = W;
  a.tilesX = (W + 15) / 16;
  a.out_phases = a.k_phases = 1;
  a.bz = bn_z; a.bz_ld = bn_z_ld; a.bcf = reinterpret_cast<const float4 *>(bn_coef);
  a.bred_c0 = bn_first_channel;
  ProfScope prof(PROF_CONV_DGRAD_BF16, 2.0 * N * H * W * (double)CiP * CoP * 9,
                 2.0 * N * H * W * ((double)CiP + 2.0 * CoP), as_stream(s));
  return launch_conv_any(9, a, as_stream(s));
}

// One (i,j) phase launch per call is an implementation detail: these entry
// points run all four phases of the 2x2 stride-2 transposed convolution.
static int tconv2x2_impl(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                         uint16_t *out, int out_ld, float *stat_partials, int N, int H_in, int W_in, int CiP, int CoP,
                         const float4 *affine, unet_stream_t s) {
  UNET_REQUIRE(in && w_packed && out, "unet_tconv2x2_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H_in > 0 && W_in > 0, "unet_tconv2x2_bf16: bad sizes");
  UNET_REQUIRE(CiP > 0 && CiP % 32 == 0 && CoP > 0 && CoP % 32 == 0 && in_ld >= CiP && out_ld >= CoP &&
                   in_ld % 8 == 0 && out_ld % 8 == 0,
               "unet_tconv2x2_bf16: channel counts must be multiples of 32 and pixel strides multiples of 8");
  UNET_REQUIRE((i64)H_in * W_in * in_ld < (1ll << 31) && (i64)4 * H_in * W_in * out_ld < (1ll << 31),
               "unet_tconv2x2_bf16: one image must stay below 2^31 elements");
  const int tiles = unet_conv3x3_stat_tiles(H_in, W_in);
  ProfScope prof(PROF_TCONV_BF16, 2.0 * N * H_in * W_in * 4.0 * CiP * CoP,
                 2.0 * N * H_in * W_in * ((double)CiP + 4.0 * CoP), as_stream(s));
  ConvArgs a{};
  a.in = in; a.wt = w_packed; a.bias = bias; a.out = out; a.stats = stat_partials;
  a.N = N; a.Hin = H_in; a.Win = W_in; a.Ho = H_in; a.Wo = W_in; a.Ci = CiP; a.CoP = CoP;
  a.in_ld = in_ld; a.out_ld = out_ld; a.in_s = 1; a.in_oy = a.in_ox = 0;
  a.out_s = 2; a.out_oy = a.out_ox = 0; a.Hout = 2 * H_in; a.Wout = 2 * W_in; a.accumulate = 0;
  a.tilesX = (W_in + 15) / 16;
  a.co_valid = co_valid;
  a.out_phases = 4; a.k_phases = 1;
  a.wt_phase_stride = (i64)CoP * CiP;
  a.stats_phase_stride = (i64)N * tiles * CoP * 2;
  a.affine = affine;
  return launch_conv_any(1, a, as_stream(s));
}

int unet_tconv2x2_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                       uint16_t *out, int out_ld, float *stat_partials, int N, int H_in, int W_in, int CiP, int CoP,
                       unet_stream_t s) {
  return tconv2x2_impl(in, in_ld, w_packed, bias, co_valid, out, out_ld, stat_partials, N, H_in, W_in, CiP, CoP, nullptr, s);
}

int unet_tconv2x2_bn_relu_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                               const float *bn_coef, uint16_t *out, int out_ld, int N, int H_in, int W_in, int CiP,
                               int CoP, unet_stream_t s) {
  UNET_REQUIRE(bn_coef && (((uintptr_t)bn_coef) & 15) == 0,
               "unet_tconv2x2_bn_relu_bf16: bn_coef must be a 16-byte aligned pointer");
  return tconv2x2_impl(in, in_ld, w_packed, bias, co_valid, out, out_ld, nullptr, N, H_in, W_in, CiP, CoP,
                       reinterpret_cast<const float4 *>(bn_coef), s);
}

// dX[n,h,w,ci] = sum_{i,j} sum_co dY[n,2h+i,2w+j,co] * W[ci,co,i,j];  w_packed_bwd = [phase][CiP][CoP]
int unet_tconv2x2_dgrad_bf16(const uint16_t *dy, int dy_ld, const uint16_t *w_packed_bwd, uint16_t *dx, int dx_ld,
                             int N, int H_in, int W_in, int CiP, int CoP, int accumulate, unet_stream_t s) {
  UNET_REQUIRE(dy && w_packed_bwd && dx, "unet_tconv2x2_dgrad_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H_in > 0 && W_in > 0, "unet_tconv2x2_dgrad_bf16: bad sizes");
  UNET_REQUIRE(CiP > 0 && CiP % 32 == 0 && CoP > 0 && CoP % 32 == 0 && dy_ld >= CoP && dx_ld >= CiP &&
                   dy_ld % 8 == 0 && dx_ld % 8 == 0,
               "unet_tconv2x2_dgrad_bf16: channel counts must be multiples of 32 and pixel strides multiples of 8");
  UNET_REQUIRE((i64)4 * H_in * W_in * dy_ld < (1ll << 31) && (i64)H_in * W_in * dx_ld < (1ll << 31),
               "unet_tconv2x2_dgrad_bf16: one image must stay below 2^31 elements");
  ProfScope prof(PROF_TCONV_BF16, 2.0 * N * H_in * W_in * 4.0 * CiP * CoP,
                 2.0 * N * H_in * W_in * ((double)CiP + 4.0 * CoP), as_stream(s));
  ConvArgs a{};
  a.in = dy; a.wt = w_packed_bwd; a.bias = nullptr; a.out = dx; a.stats = nullptr;
  a.N = N; a.Hin = 2 * H_in; a.Win = 2 * W_in; a.Ho = H_in; a.Wo = W_in; a.Ci = CoP; a.CoP = CiP;
  a.in_ld = dy_ld; a.out_ld = dx_ld; a.in_s = 2; a.in_oy = a.in_ox = 0;
  a.out_s = 1; a.out_oy = a.out_ox = 0; a.Hout = H_in; a.Wout = W_in; a.accumulate = accumulate;
  a.tilesX = (W_in + 15) / 16;
  a.out_phases = 1; a.k_phases = 4;
  a.wt_phase_stride = (i64)CiP * CoP;
  return launch_conv_any(1, a, as_stream(s));
}

int unet_tconv2x2_dgrad_bnred_bf16(const uint16_t *dy, int dy_ld, const uint16_t *w_packed_bwd, uint16_t *dx,
                                   int dx_ld, const uint16_t *bn_z, int bn_z_ld, const float *bn_coef,
                                   float *red_partials, int N, int H_in, int W_in, int CiP, int CoP,
                                   unet_stream_t s) {
  UNET_REQUIRE(dy && w_packed_bwd && dx && bn_z && bn_coef && red_partials,
               "unet_tconv2x2_dgrad_bnred_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H_in > 0 && W_in > 0, "unet_tconv2x2_dgrad_bnred_bf16: bad sizes");
  UNET_REQUIRE(CiP > 0 && CiP % 32 == 0 && CoP > 0 && CoP % 32 == 0 && dy_ld >= CoP && dx_ld >= CiP &&
                   bn_z_ld >= CiP && dy_ld % 8 == 0 && dx_ld % 8 == 0 && bn_z_ld % 8 == 0,
               "unet_tconv2x2_dgrad_bnred_bf16: channel counts must be multiples of 32 and pixel strides multiples of 8");
  UNET_REQUIRE((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w_packed_bwd | (uintptr_t)bn_z | (uintptr_t)bn_coef) & 15) == 0,
               "unet_tconv2x2_dgrad_bnred_bf16: pointers must be 16-byte aligned");
  UNET_REQUIRE((i64)4 * H_in * W_in * dy_ld < (1ll << 31) && (i64)H_in * W_in * dx_ld < (1ll << 31) &&
                   (i64)H_in * W_in * bn_z_ld < (1ll << 31),
               "unet_tconv2x2_dgrad_bnred_bf16: one image must stay below 2^31 elements");
  ProfScope prof(PROF_TCONV_BF16, 2.0 * N * H_in * W_in * 4.0 * CiP * CoP,
                 2.0 * N * H_in * W_in * (2.0 * CiP + 4.0 * CoP), as_stream(s));
  ConvArgs a{};
  a.in = dy; a.wt = w_packed_bwd; a.bias = nullptr; a.out = dx; a.stats = red_partials;
  a.N = N; a.Hin = 2 * H_in; a.Win = 2 * W_in; a.Ho = H_in; a.Wo = W_in; a.Ci = CoP; a.CoP = CiP;
  a.in_ld = dy_ld; a.out_ld = dx_ld; a.in_s = 2; a.in_oy = a.in_ox = 0;
  a.out_s = 1; a.out_oy = a.out_ox = 0; a.Hout = H_in; a.Wout = W_in; a.accumulate = 0;
  a.tilesX = (W_in + 15) / 16;
  a.out_phases = 1; a.k_phases = 4;
  a.wt_phase_stride = (i64)CiP * CoP;
  a.bz = bn_z; a.bz_ld = bn_z_ld; a.bcf = reinterpret_cast<const float4 *>(bn_coef);
  return launch_conv_any(1, a, as_stream(s));
}

}  // extern "C"
