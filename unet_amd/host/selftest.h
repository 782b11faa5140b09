// `neural_net.UNetTest`: the module's built-in operator self-test.  Each test
// runs one HIP operator through the C ABI on random data and compares it with a
// straightforward host loop of the same definition, at the shapes and with the
// closeness rule of the reference's tests (src/cuda_library/test/test.h:10-17:
// |a-b| <= 1e-6 or relative difference <= 1 %; 2 % for batch-norm training,
// batch_norm_test.h:217-220), with one change: the absolute floor is 1e-6 times
// the largest magnitude in the expected tensor (at least 1e-6), because an fp32
// sum of several hundred terms of size ~0.1 carries ~1e-6 of absolute error and
// a few of millions of outputs land within 1e-4 of zero.  Host sums accumulate
// in double.  A mismatch raises std::runtime_error (Python RuntimeError);
// success prints the same green line as the reference.
//
// The host loops here exist only for this self-test; no operator falls back
// to them.
#pragma once
#include <cmath>
#include <cstdint>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "net.h"

class UNetTest {
 public:
  void testConvolution() {
    const int N = 16, Ci = 3, Co = 64, S = 256, k = 3, pad = 1, stride = 1;  // convolution_test.h:157-165
    const int So = (S + 2 * pad - k) / stride + 1;
    const size_t nin = (size_t)N * Ci * S * S, nout = (size_t)N * Co * So * So, nw = (size_t)Ci * Co * k * k + Co;
    std::vector<float> x = rnd(nin), w = rnd(nw), dy = rnd(nout);
    // host: forward, input gradient, weight gradient
    std::vector<float> y(nout);
    std::vector<double> dx(nin, 0.0), dw(nw, 0.0);
    for (int n = 0; n < N; ++n)
      for (int co = 0; co < Co; ++co)
        for (int h = 0; h < So; ++h)
          for (int q = 0; q < So; ++q) {
            const size_t o = (((size_t)n * Co + co) * So + h) * So + q;
            const double g = dy[o];
            double acc = w[nw - Co + co];
            dw[nw - Co + co] += g;
            for (int ci = 0; ci < Ci; ++ci)
              for (int a = 0; a < k; ++a) {
                const int ih = h * stride - pad + a;
                if (ih < 0 || ih >= S) continue;
                for (int b = 0; b < k; ++b) {
                  const int iw = q * stride - pad + b;
                  if (iw < 0 || iw >= S) continue;
                  const size_t xi = (((size_t)n * Ci + ci) * S + ih) * S + iw;
                  const size_t wi = (((size_t)ci * Co + co) * k + a) * k + b;
                  acc += (double)x[xi] * w[wi];
                  dx[xi] += w[wi] * g;
                  dw[wi] += g * x[xi];
                }
              }
            y[o] = (float)acc;
          }
    Dev dX(x), dW(w), dDY(dy), dY(nout), dDX(nin), dDW(nw);
    UNET_CALL(unet_convolve_f32(dW.p, dX.p, dY.p, (int64_t)nout, N, k, Ci, Co, So, So, S, S, pad, stride, nullptr));
    UNET_CALL(unet_convolve_backward_f32(dDY.p, dX.p, dW.p, dDX.p, dDW.p, N, k, Ci, Co, So, So, S, S, pad, stride,
                                         nullptr));
    expect_close(dY.get(), y, 1.f, "convolution forward");
    expect_close(dDX.get(), dx, 1.f, "convolution input gradient");
    expect_close(dDW.get(), dw, 1.f, "convolution weight gradient");
    std::cout << "\033[32mConvolution tests pass!\033[0m" << std::endl;
  }

  void testUpsample() {
    const int N = 16, Ci = 64, Co = 32, S = 128, s = 2;  // upsample_test.h:74-79
    const int So = S * s;
    const size_t nin = (size_t)N * Ci * S * S, nout = (size_t)N * Co * So * So, nw = (size_t)Ci * Co * s * s + Co;
    std::vector<float> x = rnd(nin), w = rnd(nw), dy = rnd(nout);
    std::vector<float> y(nout);
    std::vector<double> dx(nin, 0.0), dw(nw, 0.0);
    for (int n = 0; n < N; ++n)
      for (int co = 0; co < Co; ++co)
        for (int h = 0; h < So; ++h)
          for (int q = 0; q < So; ++q) {
            const size_t o = (((size_t)n * Co + co) * So + h) * So + q;
            const double g = dy[o];
            double acc = w[nw - Co + co];
            dw[nw - Co + co] += g;
            for (int ci = 0; ci < Ci; ++ci) {
              const size_t xi = (((size_t)n * Ci + ci) * S + h / s) * S + q / s;
              const size_t wi = (((size_t)ci * Co + co) * s + h % s) * s + q % s;
              acc += (double)x[xi] * w[wi];
              dx[xi] += g * w[wi];
              dw[wi] += g * x[xi];
            }
            y[o] = (float)acc;
          }
    Dev dX(x), dW(w), dDY(dy), dY(nout), dDX(nin), dDW(nw);
    UNET_CALL(unet_upsample_f32(dX.p, dW.p, dY.p, s, Ci, Co, S, S, N, nullptr));
    UNET_CALL(unet_upsample_backward_f32(dDY.p, dX.p, dW.p, dDX.p, dDW.p, s, Ci, Co, S, S, N, nullptr));
    expect_close(dY.get(), y, 1.f, "upsample forward");
    expect_close(dDX.get(), dx, 1.f, "upsample input gradient");
    expect_close(dDW.get(), dw, 1.f, "upsample weight gradient");
    std::cout << "\033[32mUpsample tests pass!\033[0m" << std::endl;
  }

  void testBatchNorm() {
    const int N = 16, C = 64, S = 256;  // batch_norm_test.h:178-185, 233-238
    const size_t HW = (size_t)S * S, n = (size_t)N * C * HW;
    const double M = (double)N * HW;
    std::vector<float> x = rnd(n), gb = rnd(2 * C), dy = rnd(n), run = rnd(2 * C);
    for (int c = 0; c < C; ++c) run[2 * c + 1] = std::fabs(run[2 * c + 1]);
    std::vector<float> y(n), yi(n), dx(n), dgb(2 * C);
    for (int c = 0; c < C; ++c) {
      double s1 = 0, s2 = 0;
      for (int b = 0; b < N; ++b)
        for (size_t i = 0; i < HW; ++i) {
          const double v = x[((size_t)b * C + c) * HW + i];
          s1 += v;
          s2 += v * v;
        }
      const float mean = (float)(s1 / M);
      const float var = std::max(0.f, (float)(s2 / M - (s1 / M) * (s1 / M)));
      const float inv_t = 1.f / (std::sqrt(var) + 1e-8f), inv_b = 1.f / std::sqrt(var + 1e-8f);
      const float rm = run[2 * c], rinv = 1.f / std::sqrt(run[2 * c + 1] + 1e-8f);
      double dg = 0, db = 0;
      for (int b = 0; b < N; ++b)
        for (size_t i = 0; i < HW; ++i) {
          const size_t j = ((size_t)b * C + c) * HW + i;
          y[j] = gb[2 * c] * ((x[j] - mean) * inv_t) + gb[2 * c + 1];
          yi[j] = gb[2 * c] * ((x[j] - rm) * rinv) + gb[2 * c + 1];
          dg += (double)dy[j] * ((x[j] - mean) * inv_b);
          db += dy[j];
        }
      dgb[2 * c] = (float)dg;
      dgb[2 * c + 1] = (float)db;
      for (int b = 0; b < N; ++b)
        for (size_t i = 0; i < HW; ++i) {
          const size_t j = ((size_t)b * C + c) * HW + i;
          const float xh = (x[j] - mean) * inv_b;
          dx[j] = gb[2 * c] * inv_b * (dy[j] - (float)(1.0 / M) * ((float)db + xh * (float)dg));
        }
    }
    Dev dX(x), dGB(gb), dDY(dy), dRun(run), dRun2(run), dY(n), dYi(n), dDX(n), dStats(2 * C), dDGB(2 * C);
    UNET_CALL(unet_batch_norm_train_f32(dX.p, dY.p, dGB.p, dStats.p, dRun.p, C, S, S, N, nullptr));
    UNET_CALL(unet_batch_norm_backward_f32(dDY.p, dX.p, dStats.p, dGB.p, dDX.p, dDGB.p, C, S, S, N, nullptr));
    // inference: reference shape is one image (batch_norm_test.h:131-137)
    UNET_CALL(unet_batch_norm_inference_f32(dX.p, dYi.p, dGB.p, dRun2.p, S, S, C, 1, nullptr));
    expect_close(dDX.get(), dx, 1.f, "batch norm input gradient");
    expect_close(dDGB.get(), dgb, 1.f, "batch norm weight gradient");
    std::vector<float> got_i = dYi.get();
    got_i.resize((size_t)C * HW);
    yi.resize((size_t)C * HW);
    expect_close(got_i, yi, 1.f, "batch norm inference");
    expect_close(dY.get(), y, 2.f, "batch norm training forward");
    std::cout << "\033[32mBatch Norm tests pass!\033[0m" << std::endl;
  }

  void testMaxPool() {
    {  // forward: max_pool_test.h:41-47
      const int N = 8, C = 64, S = 256, s = 2;
      const size_t n = (size_t)N * C * S * S;
      std::vector<float> x = rnd(n), y(n / 4);
      host_pool(x, y, N * C, S, s);
      Dev dX(x), dY(n / 4);
      UNET_CALL(unet_max_pool_f32(dX.p, dY.p, N, s, s, C, S, S, nullptr));
      expect_close(dY.get(), y, 1.f, "max pool forward");
    }
    {  // backward: max_pool_test.h:136-141
      const int N = 16, C = 64, S = 128, s = 2, So = S / s;
      const size_t n = (size_t)N * C * S * S;
      std::vector<float> x = rnd(n), y(n / 4), dy = rnd(n / 4), dx(n);
      host_pool(x, y, N * C, S, s);
      for (size_t pl = 0; pl < (size_t)N * C; ++pl)
        for (int h = 0; h < S; ++h)
          for (int q = 0; q < S; ++q) {
            const size_t i = (pl * S + h) * S + q, o = (pl * So + h / s) * So + q / s;
            dx[i] = (x[i] == y[o]) ? dy[o] : 0.f;
          }
      Dev dX(x), dY(y), dDY(dy), dDX(n);
      UNET_CALL(unet_max_pool_backward_f32(dDY.p, dX.p, dY.p, dDX.p, N, s, C, S, S, nullptr));
      expect_close(dDX.get(), dx, 1.f, "max pool backward");
    }
    std::cout << "\033[32mMax pool tests pass!\033[0m" << std::endl;
  }

 private:
  struct Dev {  // device array, optionally filled from a host vector
    float *p = nullptr;
    size_t n = 0;
    explicit Dev(size_t count) : n(count) { alloc(); }
    explicit Dev(const std::vector<float> &h) : n(h.size()) {
      alloc();
      UNET_CALL(unet_memcpy_h2d(p, h.data(), sizeof(float) * n, nullptr));
    }
    Dev(const Dev &) = delete;
    ~Dev() { unet_free(p); }
    void alloc() {
      void *q = nullptr;
      UNET_CALL(unet_malloc(&q, sizeof(float) * n));
      p = static_cast<float *>(q);
    }
    std::vector<float> get() const {
      std::vector<float> h(n);
      UNET_CALL(unet_memcpy_d2h(h.data(), p, sizeof(float) * n, 1, nullptr));
      return h;
    }
  };

  std::mt19937 gen_{12345};
  std::vector<float> rnd(size_t n) {  // uniform in [-0.5, 0.5) like test.h:52-56
    std::vector<float> v(n);
    std::uniform_real_distribution<float> d(-0.5f, 0.5f);
    for (float &f : v) f = d(gen_);
    return v;
  }

  static void host_pool(const std::vector<float> &x, std::vector<float> &y, size_t planes, int S, int s) {
    const int So = S / s;
    for (size_t pl = 0; pl < planes; ++pl)
      for (int h = 0; h < So; ++h)
        for (int q = 0; q < So; ++q) {
          float m = -INFINITY;
          for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) m = std::max(m, x[(pl * S + h * s + a) * S + q * s + b]);
          y[(pl * So + h) * So + q] = m;
        }
  }

  static void expect_close(const std::vector<float> &got, const std::vector<double> &want, float percent,
                           const char *what) {
    std::vector<float> w(want.begin(), want.end());
    expect_close(got, w, percent, what);
  }
  static void expect_close(const std::vector<float> &got, const std::vector<float> &want, float percent,
                           const char *what) {
    if (got.size() != want.size()) throw std::runtime_error(std::string(what) + ": size mismatch");
    float big = 1.f;
    for (float v : want) big = std::max(big, std::fabs(v));
    const float floor = 1e-6f * big;
    for (size_t i = 0; i < got.size(); ++i) {
      const float a = got[i], b = want[i];
      const float diff = std::fabs(a - b);
      if (diff <= floor) continue;
      const float denom = std::max(std::fabs(a), std::fabs(b));
      if (!(diff / denom * 100.f <= percent)) {
        std::stringstream ss;
        ss << what << " mismatch at " << i << ": device " << a << " host " << b;
        throw std::runtime_error(ss.str());
      }
    }
  }
};
