/*
 * CPU oracle for the U-Net layer hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It restates, in plain
 * single-threaded C, the algorithms the reference states in its own CPU test
 * loops (the headers under src/cuda_library/test) and, for ops the reference never
 * tests, in its kernels (src/cuda_library/cuda_ops.cu).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped library never links or calls anything in here.
 *
 * Pinning: the reference holds no golden vectors for this path and its CPU
 * headers cannot be compiled in this image (they include <curand.h> and
 * <curand_kernel.h>, which are absent), so this restatement is pinned by
 * golden vectors produced with PyTorch-CPU -- the oracle that the reference's
 * test/torch_compare.py:109-146 itself names -- committed under tests/golden/
 * together with the script that made them.
 *
 * Deliberate deviations from the reference loops (all documented per function):
 *   - long reductions accumulate in double and round once to float, so the
 *     oracle is the more accurate side of every comparison;
 *   - 64-bit indexing (the reference's int indices overflow at 2^31 elements);
 *   - H and W may differ (the reference loops are square-only);
 *   - batch-norm inference handles batch > 1 (reference: batch index 0 only).
 *
 * Layouts (src/cuda_library/cuda_lib.h:17-20 and the indexing of the kernels):
 *   tensors NCHW contiguous fp32; conv / transposed-conv weights
 *   [C_in][C_out][kh][kw] followed by C_out biases; BN params interleaved
 *   [gamma0,beta0,...]; BN batch stats interleaved raw sums [sum x, sum x^2];
 *   running stats interleaved [mean,var]; Adam state interleaved [m,v].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* src/cuda_library/cuda_lib.h:17-20 (flattenIndex), widened to 64 bit. */
static inline i64 flat(i64 n, i64 c, i64 C, i64 h, i64 H, i64 w, i64 W) {
  return n * (C * H * W) + c * (H * W) + h * W + w;
}

/* ------------------------------------------------------------------ conv -- */

/* Follows cpuConvolve, src/cuda_library/test/convolution_test.h:12-67:
 * Y[n,co,h,w] = b[co] + sum_ci sum_kh,kw W[ci,co,kh,kw] * X[n,ci,h*s-p+kh,w*s-p+kw]
 * with zero padding.  Accumulates in double (deviation). */
void oracle_convolve(const float *input, const float *weights, float *out,
                     int H_in, int W_in, int padding, int stride, int C_in,
                     int C_out, int k, int batch) {
  const int H_out = (H_in + 2 * padding - k) / stride + 1;
  const int W_out = (W_in + 2 * padding - k) / stride + 1;
  const i64 bias_off = (i64)C_in * C_out * k * k;
  for (int b = 0; b < batch; ++b)
    for (int co = 0; co < C_out; ++co)
      for (int h = 0; h < H_out; ++h)
        for (int w = 0; w < W_out; ++w) {
          double acc = weights[bias_off + co];
          for (int ci = 0; ci < C_in; ++ci)
            for (int kh = 0; kh < k; ++kh)
              for (int kw = 0; kw < k; ++kw) {
                const int ih = h * stride - padding + kh;
                const int iw = w * stride - padding + kw;
                if (ih >= 0 && ih < H_in && iw >= 0 && iw < W_in)
                  acc += (double)input[flat(b, ci, C_in, ih, H_in, iw, W_in)] *
                         (double)weights[flat(ci, co, C_out, kh, k, kw, k)];
              }
          out[flat(b, co, C_out, h, H_out, w, W_out)] = (float)acc;
        }
}

/* Follows cpuConvolutionBackwardWeightsAndInputs + cpuConvolutionBackwardBiases,
 * src/cuda_library/test/convolution_test.h:69-155.  grad_weights has
 * C_in*C_out*k*k + C_out entries (weights then biases) and is overwritten, as
 * cudaConvolveBackward memsets it first (convolution.cu:316-320).
 * dX[n,ci,hi,wi] = sum W[ci,co,kh,kw] * dY[n,co,ho,wo],
 * dW[ci,co,kh,kw] = sum dY[n,co,ho,wo] * X[n,ci,ho*s-p+kh,wo*s-p+kw],
 * db[co] = sum dY.  Double accumulators (deviation). */
void oracle_convolve_backward(const float *input, const float *grad_out,
                              const float *weights, float *grad_input,
                              float *grad_weights, int H_in, int W_in,
                              int padding, int stride, int C_in, int C_out,
                              int k, int batch) {
  const int H_out = (H_in + 2 * padding - k) / stride + 1;
  const int W_out = (W_in + 2 * padding - k) / stride + 1;
  const i64 n_in = (i64)batch * C_in * H_in * W_in;
  const i64 n_w = (i64)C_in * C_out * k * k;
  double *dx = (double *)calloc((size_t)n_in, sizeof(double));
  double *dw = (double *)calloc((size_t)(n_w + C_out), sizeof(double));
  for (int b = 0; b < batch; ++b)
    for (int co = 0; co < C_out; ++co)
      for (int ho = 0; ho < H_out; ++ho)
        for (int wo = 0; wo < W_out; ++wo) {
          const double g = grad_out[flat(b, co, C_out, ho, H_out, wo, W_out)];
          dw[n_w + co] += g;
          for (int ci = 0; ci < C_in; ++ci)
            for (int kh = 0; kh < k; ++kh)
              for (int kw = 0; kw < k; ++kw) {
                const int ih = ho * stride - padding + kh;
                const int iw = wo * stride - padding + kw;
                if (ih < 0 || ih >= H_in || iw < 0 || iw >= W_in) continue;
                const i64 xi = flat(b, ci, C_in, ih, H_in, iw, W_in);
                const i64 wi = flat(ci, co, C_out, kh, k, kw, k);
                dx[xi] += (double)weights[wi] * g;
                dw[wi] += g * (double)input[xi];
              }
        }
  for (i64 i = 0; i < n_in; ++i) grad_input[i] = (float)dx[i];
  for (i64 i = 0; i < n_w + C_out; ++i) grad_weights[i] = (float)dw[i];
  free(dx);
  free(dw);
}

/* ------------------------------------------------------------ batch norm -- */

#define BN_EPS 1e-8f
#define BN_MOMENTUM 0.9f

/* Follows cpuBatchNormTrain, src/cuda_library/test/batch_norm_test.h:11-52,
 * plus what BatchNormTrainNormalizeKernel adds (batch_norm.cu:121-140):
 * batch_stats[2c] = sum x, [2c+1] = sum x^2 (raw sums);
 * mean = sum/M, var = sumsq/M - mean^2 (biased);
 * y = gamma*(x-mean)/(sqrt(var)+eps)+beta   (eps OUTSIDE the sqrt);
 * running[2c] = 0.9*running[2c] + 0.1*mean, running[2c+1] likewise with var.
 * Sums in double (deviation); var clamped at 0 (the reference can go negative
 * and NaN, batch_norm.cu:121-123). batch_stats / running may be NULL. */
void oracle_bn_train(const float *x, const float *gamma_beta, float *y,
                     float *batch_stats, float *running, int C, int H, int W,
                     int batch) {
  const double M = (double)H * W * batch;
  for (int c = 0; c < C; ++c) {
    double s = 0, ss = 0;
    for (int b = 0; b < batch; ++b)
      for (i64 i = 0; i < (i64)H * W; ++i) {
        const double v = x[flat(b, c, C, 0, H, 0, W) + i];
        s += v;
        ss += v * v;
      }
    const float mean = (float)(s / M);
    float var = (float)(ss / M - (s / M) * (s / M));
    if (var < 0) var = 0;
    if (batch_stats) {
      batch_stats[2 * c] = (float)s;
      batch_stats[2 * c + 1] = (float)ss;
    }
    const float g = gamma_beta[2 * c], be = gamma_beta[2 * c + 1];
    const float denom = sqrtf(var) + BN_EPS;
    for (int b = 0; b < batch; ++b)
      for (i64 i = 0; i < (i64)H * W; ++i) {
        const i64 idx = flat(b, c, C, 0, H, 0, W) + i;
        y[idx] = g * ((x[idx] - mean) / denom) + be;
      }
    if (running) {
      running[2 * c] = running[2 * c] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * mean;
      running[2 * c + 1] =
          running[2 * c + 1] * BN_MOMENTUM + (1.0f - BN_MOMENTUM) * var;
    }
  }
}

/* Follows cpuBatchNormInference, batch_norm_test.h:54-74:
 * y = gamma*(x-mean)/sqrt(var+eps)+beta with running [mean,var]
 * (eps INSIDE the sqrt).  Handles batch >= 1 (deviation: reference is
 * batch-index 0 only, batch_norm.cu:32). */
void oracle_bn_inference(const float *x, float *y, const float *gamma_beta,
                         const float *running, int C, int H, int W, int batch) {
  for (int b = 0; b < batch; ++b)
    for (int c = 0; c < C; ++c) {
      const float mean = running[2 * c];
      const float sd = sqrtf(running[2 * c + 1] + BN_EPS);
      for (i64 i = 0; i < (i64)H * W; ++i) {
        const i64 idx = flat(b, c, C, 0, H, 0, W) + i;
        y[idx] = gamma_beta[2 * c] * ((x[idx] - mean) / sd) + gamma_beta[2 * c + 1];
      }
    }
}

/* Follows cpuBatchNormBackwards, batch_norm_test.h:76-129:
 * mean/var from the raw sums; xhat = (x-mean)/sqrt(var+eps) (eps INSIDE);
 * dgamma = sum dy*xhat -> grad_gb[2c]; dbeta = sum dy -> grad_gb[2c+1];
 * dx = gamma/sqrt(var+eps) * (dy - (dbeta + xhat*dgamma)/M).
 * Double accumulators (deviation). */
void oracle_bn_backward(const float *x, const float *dy, const float *gamma_beta,
                        const float *batch_stats, float *dx, float *grad_gb,
                        int C, int H, int W, int batch) {
  const double M = (double)H * W * batch;
  for (int c = 0; c < C; ++c) {
    const float mean = (float)(batch_stats[2 * c] / M);
    float var = (float)(batch_stats[2 * c + 1] / M - (double)mean * mean);
    if (var < 0) var = 0;
    const float inv = 1.0f / sqrtf(var + BN_EPS);
    double dg = 0, db = 0;
    for (int b = 0; b < batch; ++b)
      for (i64 i = 0; i < (i64)H * W; ++i) {
        const i64 idx = flat(b, c, C, 0, H, 0, W) + i;
        const double xh = (x[idx] - mean) * inv;
        dg += dy[idx] * xh;
        db += dy[idx];
      }
    grad_gb[2 * c] = (float)dg;
    grad_gb[2 * c + 1] = (float)db;
    const float g = gamma_beta[2 * c];
    for (int b = 0; b < batch; ++b)
      for (i64 i = 0; i < (i64)H * W; ++i) {
        const i64 idx = flat(b, c, C, 0, H, 0, W) + i;
        const float xh = (x[idx] - mean) * inv;
        dx[idx] = g * inv *
                  (dy[idx] - (float)(1.0 / M) * ((float)db + xh * (float)dg));
      }
  }
}

/* -------------------------------------------------------------- upsample -- */

/* Follows cpuUpsample, src/cuda_library/test/upsample_test.h:7-36: transposed
 * convolution with kernel = stride = scale (no overlap):
 * Y[n,co,h,w] = b[co] + sum_ci W[ci,co,h%s,w%s] * X[n,ci,h/s,w/s]. */
void oracle_upsample(const float *x, float *y, const float *weights, int batch,
                     int C_in, int C_out, int scale, int H_in, int W_in) {
  const i64 woff = (i64)C_in * C_out * scale * scale;
  const int H_out = H_in * scale, W_out = W_in * scale;
  for (int b = 0; b < batch; ++b)
    for (int co = 0; co < C_out; ++co)
      for (int h = 0; h < H_out; ++h)
        for (int w = 0; w < W_out; ++w) {
          double acc = weights[woff + co];
          for (int ci = 0; ci < C_in; ++ci)
            acc += (double)x[flat(b, ci, C_in, h / scale, H_in, w / scale, W_in)] *
                   (double)weights[flat(ci, co, C_out, h % scale, scale, w % scale, scale)];
          y[flat(b, co, C_out, h, H_out, w, W_out)] = (float)acc;
        }
}

/* Follows cpuUpsampleBackward, upsample_test.h:38-72.  grad_weights
 * (weights then biases) is overwritten (upsample.cu:249-252 memsets it). */
void oracle_upsample_backward(const float *x, const float *weights,
                              const float *dy, float *dx, float *grad_weights,
                              int batch, int C_out, int C_in, int H_in, int W_in,
                              int scale) {
  const int H_out = H_in * scale, W_out = W_in * scale;
  const i64 n_w = (i64)C_in * C_out * scale * scale;
  const i64 n_in = (i64)batch * C_in * H_in * W_in;
  double *dxa = (double *)calloc((size_t)n_in, sizeof(double));
  double *dwa = (double *)calloc((size_t)(n_w + C_out), sizeof(double));
  for (int b = 0; b < batch; ++b)
    for (int co = 0; co < C_out; ++co)
      for (int h = 0; h < H_out; ++h)
        for (int w = 0; w < W_out; ++w) {
          const double g = dy[flat(b, co, C_out, h, H_out, w, W_out)];
          dwa[n_w + co] += g;
          for (int ci = 0; ci < C_in; ++ci) {
            const i64 xi = flat(b, ci, C_in, h / scale, H_in, w / scale, W_in);
            const i64 wi = flat(ci, co, C_out, h % scale, scale, w % scale, scale);
            dxa[xi] += g * (double)weights[wi];
            dwa[wi] += g * (double)x[xi];
          }
        }
  for (i64 i = 0; i < n_in; ++i) dx[i] = (float)dxa[i];
  for (i64 i = 0; i < n_w + C_out; ++i) grad_weights[i] = (float)dwa[i];
  free(dxa);
  free(dwa);
}

/* -------------------------------------------------------------- max pool -- */

/* Follows cpuForwardMaxPool, src/cuda_library/test/max_pool_test.h:10-39, and
 * maxPoolKernel, max_pool/max_pool.cu:11-56: k x k window starting at
 * (ho*s, wo*s), no padding, H_out = H/s, W_out = W/s (floor), positions past
 * the edge skipped.  Row stride is W (the reference loop uses in_H, identical
 * for its square shapes).  Optional extra outputs (either may be NULL):
 *   indices[n,c,ho,wo] = flat index hi*W+wi of the FIRST maximum in row-major
 *     window order (what torch.nn.functional.max_pool2d(return_indices=True)
 *     returns);
 *   tiemask[n,c,ho,wo] = bit (i*k+j) set when window element (i,j) equals the
 *     maximum (k*k <= 8 bits used when k <= 2; for larger windows only the
 *     first 8 positions are recorded). */
void oracle_maxpool(const float *x, float *y, int32_t *indices, uint8_t *tiemask,
                    int batch, int C, int H, int W, int k, int stride) {
  const int H_out = H / stride, W_out = W / stride;
  for (int b = 0; b < batch; ++b)
    for (int c = 0; c < C; ++c)
      for (int ho = 0; ho < H_out; ++ho)
        for (int wo = 0; wo < W_out; ++wo) {
          float m = -INFINITY;
          int32_t arg = -1;
          for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
              const int hi = ho * stride + i, wi = wo * stride + j;
              if (hi >= H || wi >= W) continue;
              const float v = x[flat(b, c, C, hi, H, wi, W)];
              if (arg < 0 || v > m) { m = v; arg = hi * W + wi; }
            }
          const i64 o = flat(b, c, C, ho, H_out, wo, W_out);
          y[o] = m;
          if (indices) indices[o] = arg;
          if (tiemask) {
            uint8_t bits = 0;
            for (int i = 0; i < k; ++i)
              for (int j = 0; j < k; ++j) {
                const int hi = ho * stride + i, wi = wo * stride + j;
                const int bit = i * k + j;
                if (hi >= H || wi >= W || bit >= 8) continue;
                if (x[flat(b, c, C, hi, H, wi, W)] == m) bits |= (uint8_t)(1u << bit);
              }
            tiemask[o] = bits;
          }
        }
}

/* Follows cpuMaxPoolBackwards, max_pool_test.h:103-134, and
 * maxPoolBackwardKernel (max_pool.cu:172): dX = (X == Y[window]) ? dY : 0, so
 * EVERY tied maximum receives the gradient.  Input rows/cols past the last
 * full window get 0. */
void oracle_maxpool_backward(const float *dy, const float *x, const float *y,
                             float *dx, int batch, int stride, int C, int H,
                             int W) {
  const int H_out = H / stride, W_out = W / stride;
  for (int b = 0; b < batch; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          const i64 i = flat(b, c, C, h, H, w, W);
          const int ho = h / stride, wo = w / stride;
          if (ho >= H_out || wo >= W_out) { dx[i] = 0; continue; }
          const i64 o = flat(b, c, C, ho, H_out, wo, W_out);
          dx[i] = (x[i] == y[o]) ? dy[o] : 0.0f;
        }
}

/* Index-based backward (no counterpart in the reference; PyTorch semantics):
 * dX is zero except dX[n,c,indices] += dY. */
void oracle_maxpool_backward_indices(const float *dy, const int32_t *indices,
                                     float *dx, int batch, int C, int H, int W,
                                     int H_out, int W_out) {
  memset(dx, 0, sizeof(float) * (size_t)batch * C * H * W);
  for (int b = 0; b < batch; ++b)
    for (int c = 0; c < C; ++c)
      for (i64 o = 0; o < (i64)H_out * W_out; ++o) {
        const i64 oo = flat(b, c, C, 0, H_out, 0, W_out) + o;
        dx[flat(b, c, C, 0, H, 0, W) + indices[oo]] += dy[oo];
      }
}

/* ---------------------------------------------------------------- concat -- */

/* Follows cudaConcat, src/cuda_library/cuda_ops.cu:64-91:
 * out[n, 0:C1] = A[n], out[n, C1:C1+C2] = B[n]. */
void oracle_concat(const float *a, const float *b, float *out, int batch,
                   i64 HxW, int C1, int C2) {
  const i64 Ct = C1 + C2;
  for (int n = 0; n < batch; ++n) {
    memcpy(out + n * Ct * HxW, a + n * C1 * HxW, sizeof(float) * (size_t)(C1 * HxW));
    memcpy(out + n * Ct * HxW + C1 * HxW, b + n * C2 * HxW,
           sizeof(float) * (size_t)(C2 * HxW));
  }
}

/* Inverse of oracle_concat: dA[n] = dY[n,0:C1], dB[n] = dY[n,C1:].  The
 * reference's cudaConcatBackward (cuda_ops.cu:375-401) copies dY unchanged and
 * lets Layer::backward slice it as [all-n C1 block][all-n C2 block]
 * (layer.h:150-165), which is only right for batch 1; this is the split that
 * makes concat backward the adjoint of concat forward for any batch. */
void oracle_concat_backward(const float *dy, float *da, float *db, int batch,
                            i64 HxW, int C1, int C2) {
  const i64 Ct = C1 + C2;
  for (int n = 0; n < batch; ++n) {
    memcpy(da + n * C1 * HxW, dy + n * Ct * HxW, sizeof(float) * (size_t)(C1 * HxW));
    memcpy(db + n * C2 * HxW, dy + n * Ct * HxW + C1 * HxW,
           sizeof(float) * (size_t)(C2 * HxW));
  }
}

/* ----------------------------------------------------------- activations -- */

/* Concat of a skip tensor that is LARGER than the tensor from below (U-Nets built from unpadded
 * convolutions): A [N,C1,Ha,Wa] is centre-cropped to B's [Hb,Wb] (offsets (Ha-Hb)/2, (Wa-Wb)/2) and laid out
 * like cudaConcat (cuda_ops.cu:64-91): out[n,0:C1] = crop(A[n]), out[n,C1:] = B[n].  The reference rejects
 * unequal sizes (layer.h:523-533); this is the extension that makes its valid-convolution descriptors usable.
 * Backward: dA is zero outside the window. */
void oracle_concat_crop(const float *a, const float *b, float *out, int N, int C1, int C2, int Ha, int Wa, int Hb,
                        int Wb) {
  const int oy = (Ha - Hb) / 2, ox = (Wa - Wb) / 2;
  for (i64 n = 0; n < N; ++n) {
    for (i64 c = 0; c < C1; ++c)
      for (i64 y = 0; y < Hb; ++y)
        for (i64 x = 0; x < Wb; ++x)
          out[((n * (C1 + C2) + c) * Hb + y) * Wb + x] = a[((n * C1 + c) * Ha + y + oy) * Wa + x + ox];
    for (i64 c = 0; c < C2; ++c)
      for (i64 i = 0; i < (i64)Hb * Wb; ++i) out[(n * (C1 + C2) + C1 + c) * Hb * Wb + i] = b[(n * C2 + c) * Hb * Wb + i];
  }
}
void oracle_concat_crop_backward(const float *dy, float *da, float *db, int N, int C1, int C2, int Ha, int Wa,
                                 int Hb, int Wb) {
  const int oy = (Ha - Hb) / 2, ox = (Wa - Wb) / 2;
  for (i64 i = 0; i < (i64)N * C1 * Ha * Wa; ++i) da[i] = 0.f;
  for (i64 n = 0; n < N; ++n) {
    for (i64 c = 0; c < C1; ++c)
      for (i64 y = 0; y < Hb; ++y)
        for (i64 x = 0; x < Wb; ++x)
          da[((n * C1 + c) * Ha + y + oy) * Wa + x + ox] = dy[((n * (C1 + C2) + c) * Hb + y) * Wb + x];
    for (i64 c = 0; c < C2; ++c)
      for (i64 i = 0; i < (i64)Hb * Wb; ++i) db[(n * C2 + c) * Hb * Wb + i] = dy[(n * (C1 + C2) + C1 + c) * Hb * Wb + i];
  }
}

/* reluKernel / reluBackwardKernel, cuda_ops.cu:93-123, 188-221.  Backward
 * masks on the PRE-activation input. */
void oracle_relu(const float *x, float *y, i64 n) {
  for (i64 i = 0; i < n; ++i) y[i] = x[i] > 0.0f ? x[i] : 0.0f;
}
void oracle_relu_backward(const float *dy, const float *x, float *dx, i64 n) {
  for (i64 i = 0; i < n; ++i) dx[i] = x[i] > 0.0f ? dy[i] : 0.0f;
}
/* LEAKY_ReLU / TANH of enum Activation_type (layer.h:31): declared by the reference, run there as ReLU
 * (convolution_layer_functions.h:59-65); restated from their definitions.  The leaky backward masks on the
 * pre-activation like reluBackwardKernel, the tanh backward uses the output like sigmoidBackwardKernel. */
void oracle_leaky_relu(const float *x, float *y, i64 n, float slope) {
  for (i64 i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : slope * x[i];
}
void oracle_leaky_relu_backward(const float *dy, const float *x, float *dx, i64 n, float slope) {
  for (i64 i = 0; i < n; ++i) dx[i] = x[i] > 0.f ? dy[i] : slope * dy[i];
}
void oracle_tanh(const float *x, float *y, i64 n) {
  for (i64 i = 0; i < n; ++i) y[i] = (float)tanh((double)x[i]);
}
void oracle_tanh_backward(const float *dy, const float *y, float *dx, i64 n) {
  for (i64 i = 0; i < n; ++i) dx[i] = (float)((double)dy[i] * (1.0 - (double)y[i] * (double)y[i]));
}

/* sigmoidKernel / sigmoidBackwardKernel, cuda_ops.cu:125-186 and
 * cuda_lib.h:22-24.  Backward takes the OUTPUT y. */
void oracle_sigmoid(const float *x, float *y, i64 n) {
  for (i64 i = 0; i < n; ++i) y[i] = 1.0f / (1.0f + expf(-x[i]));
}
void oracle_sigmoid_backward(const float *dy, const float *y, float *dx, i64 n) {
  for (i64 i = 0; i < n; ++i) dx[i] = dy[i] * y[i] * (1.0f - y[i]);
}

/* ---------------------------------------------------------------- losses -- */

/* dice_score_forward_kernel_safe + cudaDiceScore, cuda_ops.cu:223-294:
 * loss_values[3n+0..2] = [sum m*p, sum m, sum p] per sample; returns
 * mean_n 2*I/(sum m + sum p + 1e-5).  Double accumulators (deviation). */
float oracle_dice_score(const float *mask, const float *pred, int H, int W,
                        int batch, float *loss_values) {
  double dice = 0;
  for (int n = 0; n < batch; ++n) {
    double I = 0, sm = 0, sp = 0;
    for (i64 i = 0; i < (i64)H * W; ++i) {
      const double m = mask[(i64)n * H * W + i], p = pred[(i64)n * H * W + i];
      I += m * p;
      sm += m;
      sp += p;
    }
    loss_values[3 * n] = (float)I;
    loss_values[3 * n + 1] = (float)sm;
    loss_values[3 * n + 2] = (float)sp;
    dice += 2.0f * (float)I / ((float)sm + (float)sp + 1e-5f);
  }
  return (float)(dice / batch);
}

/* diceLossBackwardKernel, cuda_ops.cu:296-320: with D = sum m + sum p + 1e-5,
 * dLdY[n,i] = 2*(m*D - I)/D^2 (the derivative of the Dice score itself; the
 * reference does not negate it). */
void oracle_dice_backward(const float *mask, const float *loss_values,
                          float *dLdY, int H, int W, int batch) {
  for (int n = 0; n < batch; ++n) {
    const float I = loss_values[3 * n];
    const float D = loss_values[3 * n + 1] + loss_values[3 * n + 2] + 1e-5f;
    for (i64 i = 0; i < (i64)H * W; ++i) {
      const i64 idx = (i64)n * H * W + i;
      dLdY[idx] = 2.0f * (mask[idx] * D - I) / (D * D);
    }
  }
}

/* differenceLossBackwardKernel, cuda_ops.cu:615-625: dLdY = (y - m)/n. */
void oracle_difference_backward(const float *mask, const float *y, float *dLdY,
                                i64 n) {
  for (i64 i = 0; i < n; ++i) dLdY[i] = (y[i] - mask[i]) / (float)n;
}

/* Softmax cross-entropy over the channel axis (not in the reference: SOFTMAX
 * is only an enum value, layer.h:31; semantics follow
 * torch.nn.functional.cross_entropy with mean reduction): logits [N,C,H,W],
 * target class per pixel [N,H,W] (int32); a target outside [0, C) marks an
 * ignored pixel (torch's ignore_index, for every non-class value).  Returns the
 * mean loss over the V pixels that count; writes dlogits = (softmax - onehot)/V
 * (zeros at ignored pixels) when dlogits != NULL. */
float oracle_softmax_xent(const float *logits, const int32_t *target,
                          float *dlogits, int batch, int C, int H, int W) {
  const i64 HW = (i64)H * W;
  i64 valid = 0;
  for (i64 p = 0; p < (i64)batch * HW; ++p) valid += target[p] >= 0 && target[p] < C;
  const double inv = valid > 0 ? 1.0 / (double)valid : 0.0;
  double loss = 0;
  for (int n = 0; n < batch; ++n)
    for (i64 i = 0; i < HW; ++i) {
      const int t = target[(i64)n * HW + i];
      if (t < 0 || t >= C) {
        if (dlogits)
          for (int c = 0; c < C; ++c) dlogits[((i64)n * C + c) * HW + i] = 0.f;
        continue;
      }
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) {
        const float v = logits[((i64)n * C + c) * HW + i];
        if (v > mx) mx = v;
      }
      double se = 0;
      for (int c = 0; c < C; ++c) se += exp((double)logits[((i64)n * C + c) * HW + i] - mx);
      loss += log(se) + mx - logits[((i64)n * C + t) * HW + i];
      if (dlogits)
        for (int c = 0; c < C; ++c) {
          const double p = exp((double)logits[((i64)n * C + c) * HW + i] - mx) / se;
          dlogits[((i64)n * C + c) * HW + i] = (float)((p - (c == t ? 1.0 : 0.0)) * inv);
        }
    }
  return (float)(loss * inv);
}

/* ------------------------------------------------------------- optimiser -- */

/* adamUpdateKernel + cudaUpdateWeights(6 args), cuda_ops.cu:405-471:
 * beta1 .9, beta2 .999, eps 1e-8, decoupled weight decay 1e-4,
 * w -= lr*(wd*w + mhat/(sqrt(vhat)+eps)); state [m,v] interleaved. */
void oracle_adam(float *w, const float *g, float *state, float lr, i64 n, int t) {
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, wd = 1e-4f;
  const float c1 = (float)(1.0 - pow((double)b1, (double)t));
  const float c2 = (float)(1.0 - pow((double)b2, (double)t));
  for (i64 i = 0; i < n; ++i) {
    float m = state[2 * i], v = state[2 * i + 1];
    m = b1 * m + (1.0f - b1) * g[i];
    v = b2 * v + (1.0f - b2) * (g[i] * g[i]);
    state[2 * i] = m;
    state[2 * i + 1] = v;
    const float mh = m / c1, vh = v / c2;
    w[i] -= lr * (wd * w[i] + mh / (sqrtf(vh) + eps));
  }
}

/* genericUpdateWeightsKernel, cuda_ops.cu:480-490: w -= lr*g. */
void oracle_sgd(float *w, const float *g, float lr, i64 n) {
  for (i64 i = 0; i < n; ++i) w[i] -= g[i] * lr;
}

/* addArrayBtoAKernel, cuda_ops.cu:599-605. */
void oracle_add_b_to_a(float *a, const float *b, i64 n) {
  for (i64 i = 0; i < n; ++i) a[i] += b[i];
}

/* thresholdKernel, cuda_ops.cu:511-518. */
void oracle_threshold(float *x, float thr, i64 n) {
  for (i64 i = 0; i < n; ++i) x[i] = x[i] >= thr ? 1.0f : 0.0f;
}

/* -------------------------------------------------------- attention gate -- */

/* The reference's additive attention gate, src/cuda_library/attention/attention.cu:9-203
 * (the reference has no CPU loop for it: src/cuda_library/test/attention_test.h:12-22 is empty,
 * so this follows the kernels).  x [N,Cx,Hx,Wx] is the skip tensor, g [N,Cg,Hx/2,Wx/2] the gating
 * tensor; Ci intermediate channels (the layer passes Cx/2, attention_layer_functions.h:18).
 * params (layer.h:558-566): W[(Cx+Cg)][Ci] (x rows, then g rows) | psi[Ci] | b1[Ci] | bpsi.
 *   int1[n,o,h,w] = relu(b1[o] + sum_i x[n,i,2h,2w] W[i][o] + sum_j g[n,j,h,w] W[Cx+j][o])   (:84-132)
 *   alpha[n,h,w]  = sigmoid(bpsi + sum_o int1[n,o,h,w] psi[o])                              (:9-43)
 *   out[n,c,h,w]  = x[n,c,h,w] * interp(alpha)(h,w)                                         (:45-81)
 * interp: the 4x4 block of output pixels that contains (h,w) blends the 2x2 alpha cells whose
 * top-left is (2*(h/4), 2*(w/4)) with fractions (h%4)/3 and (w%4)/3.
 * Deviations: double accumulators; the "+1" cells are clamped to the last row/column -- the
 * reference computes the clamped indices (:65-66) but indexes with the unclamped ones, which only
 * differs (reads out of bounds) when Hx/2 or Wx/2 is odd; int1/alpha are overwritten, not
 * accumulated into. */
static inline void attn_corner(int h, int H_g, int *a0, int *a1, float *frac) {
  const int small = h / 2;
  *a0 = small & ~1;
  *a1 = (*a0 + 1 < H_g) ? *a0 + 1 : H_g - 1;
  *frac = (float)(h % 4) / 3.0f;
}

static double attn_interp(const float *alpha, i64 n, int h, int w, int H_g, int W_g) {
  int h0, h1, w0, w1;
  float fh, fw;
  attn_corner(h, H_g, &h0, &h1, &fh);
  attn_corner(w, W_g, &w0, &w1, &fw);
  const float *a = alpha + n * (i64)H_g * W_g;
  const double A = a[(i64)h0 * W_g + w0], B = a[(i64)h0 * W_g + w1];
  const double C = a[(i64)h1 * W_g + w0], D = a[(i64)h1 * W_g + w1];
  const double r1 = B * fw + A * (1.0 - fw), r2 = D * fw + C * (1.0 - fw);
  return r2 * fh + r1 * (1.0 - fh);
}

void oracle_attention(const float *x, const float *g, const float *params, float *int1,
                      float *alpha, float *out, int batch, int H_x, int W_x, int C_x, int C_g,
                      int C_i) {
  const int H_g = H_x / 2, W_g = W_x / 2;
  const float *psi = params + (i64)(C_x + C_g) * C_i;
  const float *b1 = psi + C_i;
  const float bpsi = b1[C_i];
  for (i64 n = 0; n < batch; ++n)
    for (int h = 0; h < H_g; ++h)
      for (int w = 0; w < W_g; ++w) {
        double z = bpsi;
        for (int o = 0; o < C_i; ++o) {
          double acc = b1[o];
          for (int i = 0; i < C_x; ++i)
            acc += (double)x[flat(n, i, C_x, 2 * h, H_x, 2 * w, W_x)] * (double)params[(i64)i * C_i + o];
          for (int j = 0; j < C_g; ++j)
            acc += (double)g[flat(n, j, C_g, h, H_g, w, W_g)] * (double)params[(i64)(C_x + j) * C_i + o];
          const float r = acc > 0 ? (float)acc : 0.0f;
          int1[flat(n, o, C_i, h, H_g, w, W_g)] = r;
          z += (double)r * (double)psi[o];
        }
        alpha[(n * H_g + h) * W_g + w] = (float)(1.0 / (1.0 + exp(-z)));
      }
  for (i64 n = 0; n < batch; ++n)
    for (int h = 0; h < H_x; ++h)
      for (int w = 0; w < W_x; ++w) {
        const double a = attn_interp(alpha, n, h, w, H_g, W_g);
        for (int c = 0; c < C_x; ++c) {
          const i64 idx = flat(n, c, C_x, h, H_x, w, W_x);
          out[idx] = (float)((double)x[idx] * a);
        }
      }
}

/* attention.cu:205-480 (cudaAttentionBackward): resample backward (grad_x = grad_out * interp,
 * grad_alpha scattered to the four cells with their blend weights), sigmoid + psi backward,
 * ReLU + gate backward (weight, bias and input gradients; the x gradient of the stride-2 1x1
 * projection adds onto grad_x at the even pixels).  All outputs overwritten; double accumulators. */
void oracle_attention_backward(const float *grad_out, const float *int1, const float *alpha,
                               const float *x, const float *g, const float *params,
                               float *grad_x, float *grad_g, float *grad_params, int batch,
                               int H_x, int W_x, int C_x, int C_g, int C_i) {
  const int H_g = H_x / 2, W_g = W_x / 2;
  const i64 cells = (i64)batch * H_g * W_g;
  const i64 n_params = (i64)(C_x + C_g + 2) * C_i + 1;
  const float *psi = params + (i64)(C_x + C_g) * C_i;
  double *ga = (double *)calloc((size_t)cells, sizeof(double));
  double *gp = (double *)calloc((size_t)n_params, sizeof(double));
  double *gx = (double *)calloc((size_t)batch * C_x * H_x * W_x, sizeof(double));
  for (i64 n = 0; n < batch; ++n)
    for (int h = 0; h < H_x; ++h)
      for (int w = 0; w < W_x; ++w) {
        int h0, h1, w0, w1;
        float fh, fw;
        attn_corner(h, H_g, &h0, &h1, &fh);
        attn_corner(w, W_g, &w0, &w1, &fw);
        const double a = attn_interp(alpha, n, h, w, H_g, W_g);
        double s = 0;
        for (int c = 0; c < C_x; ++c) {
          const i64 idx = flat(n, c, C_x, h, H_x, w, W_x);
          gx[idx] = (double)grad_out[idx] * a;
          s += (double)grad_out[idx] * (double)x[idx];
        }
        double *cell = ga + n * (i64)H_g * W_g;
        cell[(i64)h0 * W_g + w0] += s * (1.0 - fw) * (1.0 - fh);
        cell[(i64)h0 * W_g + w1] += s * fw * (1.0 - fh);
        cell[(i64)h1 * W_g + w0] += s * (1.0 - fw) * fh;
        cell[(i64)h1 * W_g + w1] += s * fw * fh;
      }
  for (i64 n = 0; n < batch; ++n)
    for (int h = 0; h < H_g; ++h)
      for (int w = 0; w < W_g; ++w) {
        const double al = alpha[(n * H_g + h) * W_g + w];
        const double gz = ga[(n * H_g + h) * W_g + w] * al * (1.0 - al);
        gp[(i64)(C_x + C_g + 2) * C_i] += gz;  /* bpsi */
        for (int o = 0; o < C_i; ++o) {
          const double act = int1[flat(n, o, C_i, h, H_g, w, W_g)];
          gp[(i64)(C_x + C_g) * C_i + o] += gz * act;  /* psi */
          const double gpre = act > 0 ? gz * (double)psi[o] : 0.0;
          if (gpre == 0.0) continue;
          gp[(i64)(C_x + C_g + 1) * C_i + o] += gpre;  /* b1 */
          for (int i = 0; i < C_x; ++i) {
            const i64 xi = flat(n, i, C_x, 2 * h, H_x, 2 * w, W_x);
            gp[(i64)i * C_i + o] += gpre * (double)x[xi];
            gx[xi] += gpre * (double)params[(i64)i * C_i + o];
          }
        }
        for (int j = 0; j < C_g; ++j) {
          double acc = 0;
          const double gv = g[flat(n, j, C_g, h, H_g, w, W_g)];
          for (int o = 0; o < C_i; ++o) {
            const double act = int1[flat(n, o, C_i, h, H_g, w, W_g)];
            const double gpre = act > 0 ? gz * (double)psi[o] : 0.0;
            gp[(i64)(C_x + j) * C_i + o] += gpre * gv;
            acc += gpre * (double)params[(i64)(C_x + j) * C_i + o];
          }
          grad_g[flat(n, j, C_g, h, H_g, w, W_g)] = (float)acc;
        }
      }
  for (i64 k = 0; k < (i64)batch * C_x * H_x * W_x; ++k) grad_x[k] = (float)gx[k];
  for (i64 k = 0; k < n_params; ++k) grad_params[k] = (float)gp[k];
  free(ga);
  free(gp);
  free(gx);
}
