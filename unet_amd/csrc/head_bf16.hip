// The segmentation head: a 3x3 / pad 1 / stride 1 convolution with ONE output channel
// (reference: the last CONV_LAYER of main.py:187-224, computed by convolution.cu:10-136 forward and
// convolution.cu:153-414 backward).  With one output channel the matrix cores have nothing to tile:
// all three passes are bound by HBM (one read of the 64-channel activation, or one write of its
// gradient), so they run on the vector ALUs with the single-channel tensor in fp32 [N][H][W] --
// exactly the layout the fp32 batch norm / sigmoid / loss kernels of the head use, which removes the
// layout conversions and the 32-channel padding of the generic bf16 path.
//
//   head_fwd_kernel    z[n,y,x]      = b + sum_{tap,ci} x[n,y+kh-1,x+kw-1,ci] * w[ci][tap]     (v_dot2c_f32_bf16)
//   head_dgrad_kernel  dx[n,y,x,ci]  = sum_tap dz[n,y+1-kh,x+1-kw] * w[ci][tap]   (+ the parent's BN-backward
//                                       pixel reduction, as conv_mfma_kernel<.., BRED> does)
//   head_wgrad_kernel  dw[ci][tap]   = sum_{n,y,x} x[n,y+kh-1,x+kw-1,ci] * dz[n,y,x],  db = sum dz
//                      (per-workgroup partial sums, then head_wgrad_reduce_kernel in a fixed order)
// Every workgroup owns 16x16 pixel tiles and stages the tile's halo in LDS.
#include "bf16_util.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int HALO = 18, NPIX = HALO * HALO;

// ---------------------------------------------------------------------------------------------
template <int CI>  // padded input channels
__global__ __launch_bounds__(256) void head_fwd_kernel(const bf16_t *__restrict__ in, int in_ld,
                                                       const uint32_t *__restrict__ wt,  // [CI/32][tap][CoP][32] bf16, rows co = 0 used
                                                       int w_tap_stride_dw, const float *__restrict__ bias,
                                                       float *__restrict__ out, int H, int W, int tilesX) {
  // The halo is staged at most 64 channels at a time: a 128-channel halo (88 KB) would leave one workgroup
  // per CU and nothing to cover its load phase.
  constexpr int CK = CI > 64 ? 64 : CI;
  constexpr int PPR = CK / 8, PITCH = CK * 2 + 16, PIECES = NPIX * PPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const i64 n = blockIdx.y;
  const int y0 = (blockIdx.x / tilesX) * 16, x0 = (blockIdx.x % tilesX) * 16;
  const bf16_t *in_n = in + n * (i64)H * W * in_ld;
  const int ty = tid >> 4, tx = tid & 15;
  float acc0 = bias ? bias[0] : 0.f, acc1 = 0.f;  // two chains for the dependent dot-product accumulations
#pragma unroll 1
  for (int cc = 0; cc < CI; cc += CK) {
    if (cc) __syncthreads();  // the previous chunk's readers are done
    {
      // all of a thread's pieces are requested before the first one is stored: one memory latency per chunk
      constexpr int TRIPS = (PIECES + 255) / 256;
      uint4 v[TRIPS];
#pragma unroll
      for (int i = 0; i < TRIPS; ++i) {
        const int c = tid + i * 256;
        const int pix = c / PPR, part = c % PPR;
        const int py = pix / HALO, px = pix - py * HALO;
        const int y = y0 + py - 1, x = x0 + px - 1;
        v[i] = make_uint4(0, 0, 0, 0);
        if (c < PIECES && y >= 0 && y < H && x >= 0 && x < W)
          v[i] = *reinterpret_cast<const uint4 *>(in_n + ((i64)y * W + x) * in_ld + cc + part * 8);
      }
#pragma unroll
      for (int i = 0; i < TRIPS; ++i) {
        const int c = tid + i * 256;
        if (c < PIECES) *reinterpret_cast<uint4 *>(lds + (c / PPR) * PITCH + (c % PPR) * 16) = v[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const unsigned char *row = lds + ((ty + tap / 3) * HALO + tx + tap % 3) * PITCH;
#pragma unroll
      for (int p = 0; p < PPR; ++p) {
        // operand layout [ci chunk of 32][tap][CoP][32]: row (chunk, tap, co = 0); uniform -> scalar loads
        const uint32_t *wrow = wt + ((i64)(cc / 32 + p / 4) * 9 + tap) * w_tap_stride_dw + (p % 4) * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(row + p * 16);
        acc0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v.x), __builtin_bit_cast(bf16x2, wrow[0]), acc0, false);
        acc1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v.y), __builtin_bit_cast(bf16x2, wrow[1]), acc1, false);
        acc0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v.z), __builtin_bit_cast(bf16x2, wrow[2]), acc0, false);
        acc1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v.w), __builtin_bit_cast(bf16x2, wrow[3]), acc1, false);
      }
    }
  }
  const int y = y0 + ty, x = x0 + tx;
  if (y < H && x < W) out[(n * H + y) * (i64)W + x] = acc0 + acc1;
}

// ---------------------------------------------------------------------------------------------
// BRED: the output is all the gradient a BN + ReLU layer's output receives; bz = that layer's
// pre-BN tensor, bcf its per-channel records {a, b, mean, 1/sqrt(var+eps)}; stats receives per tile
// (sum g*xhat, sum g) per channel, g = dx * relu'(a*z + b), in the layout of the generic kernel:
// stats[((n*tiles + tile)*CP + c)*2 + {0,1}].
template <int CP, bool BRED>  // padded channels of dx
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float *__restrict__ dz, const float *__restrict__ w,
                                                         int c_valid, bf16_t *__restrict__ dx, int dx_ld,
                                                         const bf16_t *__restrict__ bz, int bz_ld,
                                                         const float4 *__restrict__ bcf, float *__restrict__ stats,
                                                         int H, int W, int tilesX, int accumulate) {
  // A thread keeps 4 channels (one 8-byte piece index) and walks over pixels: 36 weights in registers,
  // which leaves room for 4-5 workgroups per CU to cover each other's memory phases.
  constexpr int CH = CP / 4;       // 8-byte pieces per pixel
  constexpr int PPI = 256 / CH;    // pixels per trip of the workgroup
  __shared__ float sdz[NPIX];
  __shared__ float sw[CP * 9];
  __shared__ float red[4 * CP * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const i64 n = blockIdx.y;
  const int tile = blockIdx.x;
  const int y0 = (tile / tilesX) * 16, x0 = (tile % tilesX) * 16;
  const int part = tid % CH;
  // BRED: the parent's pre-BN values of this thread's pieces, requested before anything waits on memory
  uint2 zq[BRED ? CH : 1];
  float4 cf[BRED ? 4 : 1];
  if (BRED) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int pix = tid / CH + i * PPI;
      const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
      zq[i] = make_uint2(0, 0);
      if (y < H && x < W) zq[i] = *reinterpret_cast<const uint2 *>(bz + ((n * H + y) * (i64)W + x) * bz_ld + part * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) cf[j] = bcf[part * 4 + j];
  }
  const float *dz_n = dz + n * (i64)H * W;
  for (int c = tid; c < NPIX; c += 256) {
    const int py = c / HALO, px = c - py * HALO;
    const int y = y0 + py - 1, x = x0 + px - 1;
    sdz[c] = (y >= 0 && y < H && x >= 0 && x < W) ? dz_n[(i64)y * W + x] : 0.f;
  }
  // the weights, rounded to bf16 like the packed operands of the MFMA kernels, staged once per workgroup
  for (int c = tid; c < CP * 9; c += 256) sw[c] = c < c_valid * 9 ? bf16_to_f32(f32_to_bf16(w[c])) : 0.f;
  __syncthreads();
  // wk[j][t] multiplies dz at halo offset (2 - kh, 2 - kw) of tap t = kh*3 + kw
  float wk[4][9];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[j][t] = sw[(part * 4 + j) * 9 + t];
  float ssum[4], ssq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ssum[j] = ssq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int pix = tid / CH + i * PPI;
    const int ty = pix >> 4, tx = pix & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) {
      float d[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) d[t] = sdz[(ty + 2 - t / 3) * HALO + tx + 2 - t % 3];
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(d[t], wk[j][t], a);
        v[j] = a;
      }
      bf16_t *dst = dx + ((n * H + y) * (i64)W + x) * dx_ld + part * 4;
      if (accumulate) {
        const uint2 o = *reinterpret_cast<const uint2 *>(dst);
        v[0] += bf16_to_f32(o.x & 0xffffu); v[1] += bf16_to_f32(o.x >> 16);
        v[2] += bf16_to_f32(o.y & 0xffffu); v[3] += bf16_to_f32(o.y >> 16);
      }
      const uint32_t r0 = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16), r1 = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
      if (BRED) {
        const float dy[4] = {bf16_to_f32(r0 & 0xffffu), bf16_to_f32(r0 >> 16), bf16_to_f32(r1 & 0xffffu), bf16_to_f32(r1 >> 16)};
        const float zv[4] = {bf16_to_f32(zq[i].x & 0xffffu), bf16_to_f32(zq[i].x >> 16), bf16_to_f32(zq[i].y & 0xffffu),
                             bf16_to_f32(zq[i].y >> 16)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = fmaf(cf[j].x, zv[j], cf[j].y) > 0.f ? dy[j] : 0.f;
          ssum[j] = fmaf(g, zv[j], ssum[j]);  // sum g*z; turned into sum g*xhat below
          ssq[j] += g;
        }
      }
      *reinterpret_cast<uint2 *>(dst) = make_uint2(r0, r1);
    }
  }
  if (BRED) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ssum[j] = (ssum[j] - cf[j].z * ssq[j]) * cf[j].w;
    // lanes with equal (lane % CH) hold the same channels
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ssum[j] = lane_class_sum<CH>(ssum[j]);
      ssq[j] = lane_class_sum<CH>(ssq[j]);
    }
    if (lane < CH) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wave * CP + lane * 4 + j) * 2] = ssum[j];
        red[(wave * CP + lane * 4 + j) * 2 + 1] = ssq[j];
      }
    }
    __syncthreads();
    for (int k = tid; k < 2 * CP; k += 256) {
      const float t = (red[k] + red[2 * CP + k]) + (red[4 * CP + k] + red[6 * CP + k]);
      stats[((n * gridDim.x + tile) * (i64)CP) * 2 + k] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Workgroup b walks tiles b, b + gridDim.x, ...  Thread = (channel pair cp, pixel-row pair rp): it
// accumulates 9 taps x 2 channels over its 32 pixels of every tile; at the end the 8 row pairs are
// summed through LDS and the workgroup writes partial[b][ci*9 + tap] and partial[b][CI*9] (sum dz).
template <int CI>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const bf16_t *__restrict__ x, int x_ld,
                                                         const float *__restrict__ dz, float *__restrict__ partial,
                                                         int H, int W, int tilesX, int tiles_per_image, int total_tiles) {
  // the halo is staged at most 64 channels at a time (see head_fwd_kernel); a thread keeps the sums of its
  // channel pair in every chunk
  constexpr int CK = CI > 64 ? 64 : CI, NCH = CI / CK;
  constexpr int PPR = CK / 8, PITCH = CK * 2 + 16, PIECES = NPIX * PPR;
  constexpr int CPAIRS = CK / 2, RG = 256 / CPAIRS, ROWS = 16 / RG;  // row groups, tile rows per thread
  static_assert(256 % CPAIRS == 0 && 16 % RG == 0, "thread split");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *sdz = reinterpret_cast<float *>(lds + NPIX * PITCH);  // [256] dz of the tile (0 outside the image)
  const int tid = threadIdx.x;
  const int cp = tid % CPAIRS, rg = tid / CPAIRS;
  float acc[NCH][9][2];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ch][t][0] = acc[ch][t][1] = 0.f;
  float dzsum = 0.f;
  for (int tt = blockIdx.x; tt < total_tiles; tt += gridDim.x) {
    const i64 n = tt / tiles_per_image;
    const int r = tt - (int)n * tiles_per_image;
    const int y0 = (r / tilesX) * 16, x0 = (r % tilesX) * 16;
    const bf16_t *x_n = x + n * (i64)H * W * x_ld;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      __syncthreads();  // the previous tile's / chunk's readers are done
      {
        // all of a thread's pieces are requested before the first one is stored
        constexpr int TRIPS = (PIECES + 255) / 256;
        uint4 v[TRIPS];
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
          const int c = tid + i * 256;
          const int pix = c / PPR, part = c % PPR;
          const int py = pix / HALO, px = pix - py * HALO;
          const int y = y0 + py - 1, xx = x0 + px - 1;
          v[i] = make_uint4(0, 0, 0, 0);
          if (c < PIECES && y >= 0 && y < H && xx >= 0 && xx < W)
            v[i] = *reinterpret_cast<const uint4 *>(x_n + ((i64)y * W + xx) * x_ld + ch * CK + part * 8);
        }
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
          const int c = tid + i * 256;
          if (c < PIECES) *reinterpret_cast<uint4 *>(lds + (c / PPR) * PITCH + (c % PPR) * 16) = v[i];
        }
      }
      if (ch == 0) {
        const int y = y0 + (tid >> 4), xx = x0 + (tid & 15);
        const float d = (y < H && xx < W) ? dz[(n * H + y) * (i64)W + xx] : 0.f;
        sdz[tid] = d;
        dzsum += d;
      }
      __syncthreads();
#pragma unroll 1
      for (int ry = 0; ry < ROWS; ++ry) {
        const int ty = rg * ROWS + ry;
#pragma unroll 4
        for (int tx = 0; tx < 16; ++tx) {
          const float d = sdz[ty * 16 + tx];
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const uint32_t xv = *reinterpret_cast<const uint32_t *>(lds + ((ty + t / 3) * HALO + tx + t % 3) * PITCH + cp * 4);
            acc[ch][t][0] = fmaf(bf16_to_f32(xv & 0xffffu), d, acc[ch][t][0]);
            acc[ch][t][1] = fmaf(bf16_to_f32(xv >> 16), d, acc[ch][t][1]);
          }
        }
      }
    }
  }
  // ---- sum over the row groups (fixed order), write this workgroup's partial row ----
  __syncthreads();
  float *red = reinterpret_cast<float *>(lds);  // [RG][CI*9]
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      red[rg * (CI * 9) + (ch * CK + cp * 2) * 9 + t] = acc[ch][t][0];
      red[rg * (CI * 9) + (ch * CK + cp * 2 + 1) * 9 + t] = acc[ch][t][1];
    }
  __shared__ float wred[4];
  float s1[1] = {dzsum};
  block_sum<1>(s1, wred);  // contains the barrier that publishes `red`
  float *prow = partial + (i64)blockIdx.x * (CI * 9 + 1);
  for (int k = tid; k < CI * 9; k += 256) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) s += red[g * (CI * 9) + k];
    prow[k] = s;
  }
  if (tid == 0) prow[CI * 9] = s1[0];
}

// grad_w[ci*9 + tap] for ci < c_valid, grad_b[0]: one wavefront per output sums the workgroup partials in
// double -- lanes over strided rows, then a fixed-order butterfly
__global__ __launch_bounds__(64) void head_wgrad_reduce_kernel(const float *__restrict__ partial, int rows, int row_len,
                                                               int valid, float *__restrict__ grad_w,
                                                               float *__restrict__ grad_b) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const bool is_bias = k == row_len - 1;
  if (!is_bias && k >= valid) return;
  double s = 0;
  for (int r = lane; r < rows; r += 64) s += partial[(i64)r * row_len + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane != 0) return;
  if (is_bias) {
    if (grad_b) grad_b[0] = (float)s;
  } else {
    grad_w[k] = (float)s;
  }
}

template <typename K>
int set_lds_limit(K kernel, size_t bytes, bool *done) {
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  if (!done[dev & 63]) {
    UNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes));
    done[dev & 63] = true;
  }
  return UNET_OK;
}

}  // namespace

extern "C" {

int unet_conv3x3_head_supported(int CiP) { return CiP == 32 || CiP == 64 || CiP == 128; }

int unet_conv3x3_head_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, float *out,
                           int N, int H, int W, int CiP, int CoP, unet_stream_t s) {
  UNET_REQUIRE(in && w_packed && out, "unet_conv3x3_head_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "unet_conv3x3_head_bf16: bad sizes");
  UNET_REQUIRE(unet_conv3x3_head_supported(CiP) && CoP > 0 && CoP % 32 == 0 && in_ld >= CiP && in_ld % 8 == 0,
               "unet_conv3x3_head_bf16: CiP must be 32, 64 or 128, CoP a multiple of 32, in_ld a multiple of 8 >= CiP");
  UNET_REQUIRE((((uintptr_t)in | (uintptr_t)w_packed) & 15) == 0, "unet_conv3x3_head_bf16: pointers must be 16-byte aligned");
  UNET_REQUIRE((i64)H * W * in_ld < (1ll << 31), "unet_conv3x3_head_bf16: one image must stay below 2^31 elements");
  const int tilesX = (W + 15) / 16, tiles = tilesX * ((H + 15) / 16);
  const size_t lds = (size_t)NPIX * ((CiP > 64 ? 64 : CiP) * 2 + 16);
  const uint32_t *wt = reinterpret_cast<const uint32_t *>(w_packed);
  const int tap_stride = CoP * 32 / 2;  // dwords between consecutive (chunk, tap) rows
  ProfScope prof(PROF_CONV_HEAD_BF16, 2.0 * N * H * W * (double)CiP * 9, (double)N * H * W * (2.0 * CiP + 4.0), as_stream(s));
  dim3 grid((unsigned)tiles, (unsigned)N);
  static bool done32[64], done64[64], done128[64];
  if (CiP == 32) {
    if (int rc = set_lds_limit(&head_fwd_kernel<32>, lds, done32)) return rc;
    hipLaunchKernelGGL(head_fwd_kernel<32>, grid, dim3(256), lds, as_stream(s), in, in_ld, wt, tap_stride, bias, out, H, W, tilesX);
  } else if (CiP == 64) {
    if (int rc = set_lds_limit(&head_fwd_kernel<64>, lds, done64)) return rc;
    hipLaunchKernelGGL(head_fwd_kernel<64>, grid, dim3(256), lds, as_stream(s), in, in_ld, wt, tap_stride, bias, out, H, W, tilesX);
  } else {
    if (int rc = set_lds_limit(&head_fwd_kernel<128>, lds, done128)) return rc;
    hipLaunchKernelGGL(head_fwd_kernel<128>, grid, dim3(256), lds, as_stream(s), in, in_ld, wt, tap_stride, bias, out, H, W, tilesX);
  }
  UNET_LAUNCH_CHECK("head_fwd_kernel");
  return UNET_OK;
}

int unet_conv3x3_head_dgrad_bf16(const float *dz, const float *w, int Ci, uint16_t *dx, int dx_ld, const uint16_t *bn_z,
                                 int bn_z_ld, const float *bn_coef, float *red_partials, int N, int H, int W, int CiP,
                                 int accumulate, unet_stream_t s) {
  UNET_REQUIRE(dz && w && dx, "unet_conv3x3_head_dgrad_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && Ci > 0 && Ci <= CiP, "unet_conv3x3_head_dgrad_bf16: bad sizes");
  UNET_REQUIRE(unet_conv3x3_head_supported(CiP) && dx_ld >= CiP && dx_ld % 8 == 0 && ((uintptr_t)dx & 15) == 0,
               "unet_conv3x3_head_dgrad_bf16: CiP must be 32, 64 or 128; dx 16-byte aligned with a stride multiple of 8");
  const bool bred = bn_z != nullptr;
  UNET_REQUIRE(!bred || (bn_coef && red_partials && bn_z_ld >= CiP && bn_z_ld % 8 == 0 && !accumulate &&
                         (((uintptr_t)bn_z | (uintptr_t)bn_coef) & 15) == 0),
               "unet_conv3x3_head_dgrad_bf16: the fused BN reduction needs bn_coef, red_partials, an aligned bn_z and "
               "accumulate = 0");
  UNET_REQUIRE((i64)H * W * dx_ld < (1ll << 31), "unet_conv3x3_head_dgrad_bf16: one image must stay below 2^31 elements");
  const int tilesX = (W + 15) / 16, tiles = tilesX * ((H + 15) / 16);
  ProfScope prof(PROF_CONV_HEAD_BF16, 2.0 * N * H * W * (double)CiP * 9,
                 (double)N * H * W * (2.0 * CiP * (bred ? 2 : 1) + 4.0), as_stream(s));
  dim3 grid((unsigned)tiles, (unsigned)N);
  const float4 *cf = reinterpret_cast<const float4 *>(bn_coef);
#define UNET_HEAD_DGRAD(CP)                                                                                              \
  if (bred)                                                                                                               \
    hipLaunchKernelGGL((head_dgrad_kernel<CP, true>), grid, dim3(256), 0, as_stream(s), dz, w, Ci, dx, dx_ld, bn_z,      \
                       bn_z_ld, cf, red_partials, H, W, tilesX, 0);                                                      \
  else                                                                                                                    \
    hipLaunchKernelGGL((head_dgrad_kernel<CP, false>), grid, dim3(256), 0, as_stream(s), dz, w, Ci, dx, dx_ld, nullptr, \
                       0, nullptr, nullptr, H, W, tilesX, accumulate)
  if (CiP == 32) {
    UNET_HEAD_DGRAD(32);
  } else if (CiP == 64) {
    UNET_HEAD_DGRAD(64);
  } else {
    UNET_HEAD_DGRAD(128);
  }
#undef UNET_HEAD_DGRAD
  UNET_LAUNCH_CHECK("head_dgrad_kernel");
  return UNET_OK;
}

int unet_conv3x3_head_wgrad_bf16(const float *dz, const uint16_t *x, int x_ld, float *grad_w, float *grad_b, int N,
                                 int H, int W, int CiP, int Ci, unet_stream_t s) {
  UNET_REQUIRE(dz && x && grad_w, "unet_conv3x3_head_wgrad_bf16: NULL pointer");
  UNET_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Ci <= CiP, "unet_conv3x3_head_wgrad_bf16: bad sizes");
  UNET_REQUIRE(unet_conv3x3_head_supported(CiP) && x_ld >= CiP && x_ld % 8 == 0 && ((uintptr_t)x & 15) == 0,
               "unet_conv3x3_head_wgrad_bf16: CiP must be 32, 64 or 128; x 16-byte aligned with a stride multiple of 8");
  UNET_REQUIRE((i64)H * W * x_ld < (1ll << 31), "unet_conv3x3_head_wgrad_bf16: one image must stay below 2^31 elements");
  const int tilesX = (W + 15) / 16, tpi = tilesX * ((H + 15) / 16);
  const i64 total64 = (i64)tpi * N;
  UNET_REQUIRE(total64 < (1ll << 31), "unet_conv3x3_head_wgrad_bf16: too many tiles");
  const int total = (int)total64;
  const int blocks = total < 1024 ? total : 1024;
  const int row_len = CiP * 9 + 1;
  void *ws;
  int rc = unet_workspace(&ws, sizeof(float) * (size_t)blocks * row_len, as_stream(s));
  if (rc) return rc;
  float *partial = (float *)ws;
  const int ck = CiP > 64 ? 64 : CiP;  // channels staged at a time
  const size_t lds_main = (size_t)NPIX * (ck * 2 + 16) + 256 * sizeof(float);
  const size_t lds_red = sizeof(float) * (size_t)(256 / (ck / 2)) * CiP * 9;
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  ProfScope prof(PROF_CONV_HEAD_BF16, 2.0 * N * H * W * (double)CiP * 9, (double)N * H * W * (2.0 * CiP + 4.0), as_stream(s));
  static bool done32[64], done64[64], done128[64];
  if (CiP == 32) {
    if ((rc = set_lds_limit(&head_wgrad_kernel<32>, lds, done32))) return rc;
    hipLaunchKernelGGL(head_wgrad_kernel<32>, dim3(blocks), dim3(256), lds, as_stream(s), x, x_ld, dz, partial, H, W, tilesX, tpi, total);
  } else if (CiP == 64) {
    if ((rc = set_lds_limit(&head_wgrad_kernel<64>, lds, done64))) return rc;
    hipLaunchKernelGGL(head_wgrad_kernel<64>, dim3(blocks), dim3(256), lds, as_stream(s), x, x_ld, dz, partial, H, W, tilesX, tpi, total);
  } else {
    if ((rc = set_lds_limit(&head_wgrad_kernel<128>, lds, done128))) return rc;
    hipLaunchKernelGGL(head_wgrad_kernel<128>, dim3(blocks), dim3(256), lds, as_stream(s), x, x_ld, dz, partial, H, W, tilesX, tpi, total);
  }
  UNET_LAUNCH_CHECK("head_wgrad_kernel");
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((unsigned)row_len), dim3(64), 0, as_stream(s),
                     (const float *)partial, blocks, row_len, Ci * 9, grad_w, grad_b);
  UNET_LAUNCH_CHECK("head_wgrad_reduce_kernel");
  return UNET_OK;
}

}  // extern "C"
