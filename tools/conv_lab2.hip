// Laboratory for the pipelined 3x3 convolution kernel (unet_amd/csrc/conv3x3_pipe.h): runs it next to the
// library's entry point on the layer shapes of the 4-level base-64 U-Net (batch 16), compares outputs bit for
// bit and statistics by their totals, and times both.  Build with tools/build_lab.sh, run on a GPU box.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

// the laboratory's copy of the kernel gets symbols of its own: with the library's names the loader binds both
// to one definition and the build switches of this file (PIPE_LAB_ABLATE) would not be what runs
#define pipe pipe_lab
#include "../unet_amd/csrc/conv3x3_pipe.h"

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

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
  for (int i = 0; i < 3; ++i) launch();
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

int main(int argc, char **argv) {
  const int N = 16;
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  std::vector<Shape> shapes = {{256, 64, 64}, {256, 128, 64}, {256, 64, 128}, {128, 128, 128}, {128, 64, 128},
                               {64, 256, 256}, {32, 512, 512}, {32, 1024, 512}, {128, 256, 128}, {72, 64, 64},
                               {16, 512, 1024}, {16, 1024, 1024}, {16, 1024, 512}, {256, 32, 64}};
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
  if (getenv("LAB_SWAP_OUT")) std::swap(d_out, d_ref);  // does the placement of the output buffer matter?
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
  float *d_bias, *d_stats, *d_stats_ref;
  CK(hipMalloc(&d_bias, 4096 * 4));
  {
    std::vector<float> hb(4096);
    for (int i = 0; i < 4096; ++i) hb[i] = 0.01f * (i % 17 - 8);
    CK(hipMemcpy(d_bias, hb.data(), 4096 * 4, hipMemcpyHostToDevice));
  }
  float *d_bcf;
  CK(hipMalloc(&d_bcf, 4096 * 16));
  {
    std::vector<float> hb(4096 * 4);
    for (int i = 0; i < 4096; ++i) { hb[4 * i] = 1.0f + 0.01f * (i % 7); hb[4 * i + 1] = 0.001f * (i % 5 - 2); hb[4 * i + 2] = 0.002f * (i % 3); hb[4 * i + 3] = 1.1f; }
    CK(hipMemcpy(d_bcf, hb.data(), 4096 * 16, hipMemcpyHostToDevice));
  }
  const size_t stat_floats = (size_t)N * 2048 * 1024 * 2;
  CK(hipMalloc(&d_stats, stat_floats * 4));
  CK(hipMalloc(&d_stats_ref, stat_floats * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<false, pipe::EPI_STATS>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<true, pipe::EPI_BRED>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<false, pipe::EPI_BRED>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<true, pipe::EPI_PLAIN>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<false, pipe::EPI_PLAIN>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe::conv3x3_pipe_kernel<true, pipe::EPI_STATS>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_BYTES));
  std::vector<bf16_t> h_ref, h_out;
  for (auto &s : shapes) {
    const double flops = 2.0 * N * s.H * s.H * (double)s.Ci * s.Co * 9;
    pipe::Args a{};
    a.in = d_in; a.wt = d_w; a.bias = d_bias; a.out = d_out; a.stats = d_stats;
    a.N = N; a.H = s.H; a.W = s.H; a.nk = s.Ci / 32; a.CoP = s.Co; a.in_ld = s.Ci; a.out_ld = s.Co;
    a.tilesX = (s.H + pipe::TW - 1) / pipe::TW; a.tilesY = (s.H + pipe::TH - 1) / pipe::TH;
    a.units = N * a.tilesX * a.tilesY; a.co_blocks = s.Co / 64; a.co_valid = s.Co;
    const int grid = pipe::plan_grid(N, s.H, s.H, s.Co, s.H >= 32 ? 32 : 16, wgs);
    a.xcd_order = (grid % (8 * a.co_blocks) == 0 && !getenv("LAB_NO_XCD")) ? 1 : 0;
    const bool ws = a.nk <= pipe::MAX_WS_CHUNKS;
    auto launch = [&] {
      if (ws)
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<true, pipe::EPI_STATS>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, a);
      else
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<false, pipe::EPI_STATS>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, a);
    };
    const int tiles16 = unet_conv3x3_stat_tiles(s.H, s.H), rows = pipe::stat_rows(N, s.H, s.H, s.Co, s.H >= 32 ? 32 : 16, wgs);
    const size_t out_n = (size_t)N * s.H * s.H * s.Co;
    CK(hipMemset(d_out, 0xff, out_n * 2));
    CK(hipMemset(d_stats, 0xff, (size_t)rows * s.Co * 8));
    if (unet_conv3x3_bf16(d_in, s.Ci, d_w, d_bias, s.Co, d_ref, s.Co, d_stats_ref, N, s.H, s.H, s.Ci, s.Co, 0, nullptr)) {
      fprintf(stderr, "library conv failed: %s\n", unet_last_error());
      return 1;
    }
    launch();
    CK(hipDeviceSynchronize());
    h_ref.resize(out_n);
    h_out.resize(out_n);
    CK(hipMemcpy(h_ref.data(), d_ref, out_n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_out.data(), d_out, out_n * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < out_n; ++i) bad += h_ref[i] != h_out[i];
    std::vector<float> sr((size_t)N * tiles16 * s.Co * 2), sn((size_t)rows * s.Co * 2);
    CK(hipMemcpy(sr.data(), d_stats_ref, sr.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sn.data(), d_stats, sn.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int c = 0; c < 2 * s.Co; ++c) {
      double x = 0, y = 0;
      for (size_t r = 0; r < (size_t)N * tiles16; ++r) x += sr[r * 2 * s.Co + c];
      for (size_t r = 0; r < (size_t)rows; ++r) y += sn[r * 2 * s.Co + c];
      worst = std::max(worst, fabs(x - y) / (fabs(x) + 1e-3));
    }
    auto launch_lib = [&] {
      unet_conv3x3_bf16(d_in, s.Ci, d_w, d_bias, s.Co, d_ref, s.Co, d_stats_ref, N, s.H, s.H, s.Ci, s.Co, 0, nullptr);
    };
    float t_lib, t_new;
    if (getenv("LAB_PIPE_FIRST")) {  // does the order of the two measurements matter?
      t_new = time_launch(launch, 20);
      t_lib = time_launch(launch_lib, 20);
    } else {
      t_lib = time_launch(launch_lib, 20);
      t_new = time_launch(launch, 20);
    }
    // the data-gradient variant with the fused batch-norm reduction (timing only; operands: z = the reference output)
    pipe::Args b = a;
    b.bias = nullptr; b.bz = d_ref; b.bz_ld = s.Co; b.bcf = reinterpret_cast<const float4 *>(d_bcf); b.bred_c0 = 0;
    auto launch_bred = [&] {
      if (ws)
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<true, pipe::EPI_BRED>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, b);
      else
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<false, pipe::EPI_BRED>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, b);
    };
    pipe::Args c = a;
    c.stats = nullptr;
    auto launch_plain = [&] {
      if (ws)
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<true, pipe::EPI_PLAIN>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, c);
      else
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<false, pipe::EPI_PLAIN>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, c);
    };
    const float t_bred = time_launch(launch_bred, 20), t_plain = time_launch(launch_plain, 20);
    // checksums of the fused-reduction variant's output and partial sums (compared between builds of this file)
    CK(hipMemset(d_out, 0xff, out_n * 2));
    CK(hipMemset(d_stats, 0xff, (size_t)rows * s.Co * 8));
    launch_bred();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_out.data(), d_out, out_n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sn.data(), d_stats, sn.size() * 4, hipMemcpyDeviceToHost));
    uint64_t ck_out = 0, ck_stats = 0;
    for (size_t i = 0; i < out_n; ++i) ck_out = ck_out * 1099511628211ull + h_out[i];
    for (size_t i = 0; i < sn.size(); ++i) {
      uint32_t u;
      memcpy(&u, &sn[i], 4);
      ck_stats = ck_stats * 1099511628211ull + u;
    }
#if PIPE_LAB_STAMPS
    {  // where a stage's time goes: clock stamps of every wave's first 32 stages (statistics variant)
      static unsigned long long *d_st = nullptr;
      if (!d_st) CK(hipMalloc(&d_st, (size_t)2048 * 1024 * 8));
      CK(hipMemset(d_st, 0, (size_t)grid * 1024 * 8));
      pipe::Args t = a;
      t.lab_stamps = d_st;
      if (ws)
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<true, pipe::EPI_STATS>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, t);
      else
        hipLaunchKernelGGL((pipe::conv3x3_pipe_kernel<false, pipe::EPI_STATS>), dim3(grid), dim3(512), pipe::LDS_BYTES, 0, t);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> st((size_t)grid * 1024);
      CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
      // per stage s (1 .. min(stages, 32) - 2): length = t0[s+1] - t0[s]; mfma = t2 - t0; landed = t3 - t2; barrier = t0[s+1] - t3
      double len = 0, mf = 0, land = 0, bar = 0, first = 0, total = 0;
      long cnt = 0, wgs_seen = 0;
      for (int g = 0; g < grid; ++g)
        for (int w = 0; w < 8; ++w) {
          const unsigned long long *q = &st[(size_t)g * 1024 + w * 128];
          int ns = 0;
          while (ns < 16 && q[ns * 8] != 0) ++ns;
          if (ns < 4) continue;
          ++wgs_seen;
          total += (double)(q[(ns - 1) * 8 + 3] - q[0]);
          for (int k = 1; k + 1 < ns; ++k) {
            len += (double)(q[(k + 1) * 8] - q[k * 8]);
            mf += (double)(q[k * 8 + 2] - q[k * 8]);
            land += (double)(q[k * 8 + 3] - q[k * 8 + 2]);
            bar += (double)(q[(k + 1) * 8] - q[k * 8 + 3]);
            ++cnt;
          }
        }
      // LAB_STAMP_DETAIL=H,Ci: per wave [stage start relative to the workgroup's first stamp (time spent at the barrier
      // before it), + MFMA phase, + wait for the operands, (+ epilogue, on a tile's last stage)]
      int dh = 0, dci = 0;
      if (getenv("LAB_STAMP_DETAIL") && sscanf(getenv("LAB_STAMP_DETAIL"), "%d,%d", &dh, &dci) == 2 && s.H == dh && s.Ci == dci) {
        for (int g : {0, 100}) {
          const unsigned long long base = st[(size_t)g * 1024];
          for (int k = 8; k < 14; ++k) {
            printf("    wg %3d stage %2d:", g, k);
            for (int w = 0; w < 8; ++w) {
              const unsigned long long *q = &st[(size_t)g * 1024 + w * 128 + k * 8];
              printf(" w%d[%6lld(bar %4lld) +%5lld +%4lld", w, (long long)(q[0] - base), (long long)(q[0] - q[4]), (long long)(q[2] - q[0]), (long long)(q[3] - q[2]));
              if (q[1]) printf(" +%5lld", (long long)(q[1] - q[3]));
              printf("]");
            }
            printf("\n");
          }
        }
      }
      if (cnt)
        printf("  stamps %4dx%-4d %4d->%-4d: stage %7.0f = mfma phase %7.0f + operands landed %6.0f + barrier %6.0f ticks (%ld samples); stamped span per wave %.0f\n",
               s.H, s.H, s.Ci, s.Co, len / cnt, mf / cnt, land / cnt, bar / cnt, cnt, total / wgs_seen);
    }
#endif
    printf("  [plain %6.1f us %5.0f TF | bred %6.1f us %5.0f TF] ", t_plain, flops / t_plain * 1e-6, t_bred, flops / t_bred * 1e-6);
    printf("%4dx%-4d %4d->%-4d %s grid %4d | library %6.1f us %5.0f TF | pipe %6.1f us %5.0f TF | mismatches %zu, stats rel %.2e | bred sums %016llx %016llx\n",
           s.H, s.H, s.Ci, s.Co, ws ? "WS" : "  ", grid, t_lib, flops / t_lib * 1e-6, t_new, flops / t_new * 1e-6, bad, worst,
           (unsigned long long)ck_out, (unsigned long long)ck_stats);
    fflush(stdout);
  }
  return 0;
}
