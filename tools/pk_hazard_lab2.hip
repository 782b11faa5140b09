// Second laboratory for the packed-fp32 sequence of the transposed-convolution data-gradient epilogue: the 34
// instructions hipcc emitted between the end of the store loop and the first DPP row rotation, replayed verbatim
// (same physical registers, same unused upper halves) after the same kind of producer block (v_mov_b64 copies
// and a 16-byte global store under a saved EXEC mask).  Each lane checks the eight products against scalar
// arithmetic on the same operands.  JUNK selects what the unused halves (v3, v5, v7) hold before the replay.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_lab2.hip -o tools/bin/pk_hazard_lab2
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rnd(unsigned &state, float lo, float span) {
  state = state * 1664525u + 1013904223u;
  return lo + span * (float)(state >> 8) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256, 3) void lab_kernel(unsigned long long *bad, unsigned *bad_j, uint4 *sink, int iters,
                                                    unsigned junk_bits) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long n_bad = 0;
  unsigned state = tid * 2654435761u + 12345u;
  uint4 *my_sink = sink + tid;
  for (int it = 0; it < iters; ++it) {
    float sum[8], sq[8], z[8], w[8];
    for (int j = 0; j < 8; ++j) {
      sum[j] = rnd(state, -20.f, 40.f);
      sq[j] = rnd(state, -5.f, 10.f);
      z[j] = rnd(state, -0.3f, 0.6f);
      w[j] = rnd(state, 0.7f, 1.0f);
    }
    // register pairs as the kernel holds them: (odd j, even j)
    f32x2 s01 = {sum[1], sum[0]}, s23 = {sum[3], sum[2]}, s45 = {sum[5], sum[4]}, s67 = {sum[7], sum[6]};
    f32x2 q01 = {sq[1], sq[0]}, q23 = {sq[3], sq[2]}, q45 = {sq[5], sq[4]}, q67 = {sq[7], sq[6]};
    f32x2 c0 = {z[0], w[0]}, c1 = {z[1], w[1]}, c2 = {z[2], w[2]}, c3 = {z[3], w[3]};
    f32x2 c4 = {z[4], w[4]}, c5 = {z[5], w[5]}, c6 = {z[6], w[6]}, c7 = {z[7], w[7]};
    float o0, o1, o2, o3, o4, o5, o6, o7;
    const unsigned long long addr = (unsigned long long)my_sink;
    asm volatile(
        "v_mov_b64 v[96:97], %[c0]\n\t"
        "v_mov_b64 v[80:81], %[c1]\n\t"
        "v_mov_b64 v[88:89], %[c2]\n\t"
        "v_mov_b64 v[72:73], %[c3]\n\t"
        "v_mov_b64 v[92:93], %[c4]\n\t"
        "v_mov_b64 v[76:77], %[c5]\n\t"
        "v_mov_b64 v[84:85], %[c6]\n\t"
        "v_mov_b64 v[68:69], %[c7]\n\t"
        "v_mov_b64 v[6:7], %[addr]\n\t"
        "v_mov_b32 v2, %[junk]\n\t"
        "v_mov_b32 v3, %[junk]\n\t"
        "v_mov_b32 v4, %[junk]\n\t"
        "v_mov_b32 v5, %[junk]\n\t"
        "s_nop 7\n\t"
        "s_mov_b64 s[6:7], exec\n\t"
        "v_mov_b64 v[14:15], %[q01]\n\t"
        "v_mov_b64 v[12:13], %[q23]\n\t"
        "v_mov_b64 v[10:11], %[q45]\n\t"
        "v_mov_b64 v[8:9], %[q67]\n\t"
        "v_mov_b64 v[16:17], %[s01]\n\t"
        "v_mov_b64 v[18:19], %[s23]\n\t"
        "v_mov_b64 v[20:21], %[s45]\n\t"
        "v_mov_b64 v[136:137], %[s67]\n\t"
        "global_store_dwordx4 v[6:7], v[2:5], off\n\t"
        "s_or_b64 exec, exec, s[6:7]\n\t"
        "s_and_b64 vcc, exec, s[6:7]\n\t"
        "s_nop 0\n\t"
        "v_pk_fma_f32 v[2:3], v[96:97], v[14:15], v[16:17] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v4, v97\n\t"
        "v_pk_mul_f32 v[2:3], v[4:5], v[2:3]\n\t"
        "v_pk_fma_f32 v[4:5], v[80:81], v[14:15], v[16:17] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v6, v81\n\t"
        "v_mbcnt_lo_u32_b32 v1, -1, 0\n\t"
        "v_pk_mul_f32 v[4:5], v[6:7], v[4:5]\n\t"
        "v_mbcnt_hi_u32_b32 v1, -1, v1\n\t"
        "v_pk_fma_f32 v[6:7], v[88:89], v[12:13], v[18:19] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v16, v89\n\t"
        "v_and_b32_e32 v5, 64, v1\n\t"
        "v_pk_mul_f32 v[16:17], v[16:17], v[6:7]\n\t"
        "v_pk_fma_f32 v[6:7], v[72:73], v[12:13], v[18:19] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v18, v73\n\t"
        "v_xor_b32_e32 v3, 16, v1\n\t"
        "v_add_u32_e32 v5, 64, v5\n\t"
        "v_pk_mul_f32 v[18:19], v[18:19], v[6:7]\n\t"
        "v_pk_fma_f32 v[6:7], v[92:93], v[10:11], v[20:21] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v22, v93\n\t"
        "v_cmp_lt_i32_e32 vcc, v3, v5\n\t"
        "v_pk_mul_f32 v[22:23], v[22:23], v[6:7]\n\t"
        "v_pk_fma_f32 v[6:7], v[76:77], v[10:11], v[20:21] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v20, v77\n\t"
        "v_cndmask_b32_e32 v3, v1, v3, vcc\n\t"
        "v_pk_mul_f32 v[24:25], v[20:21], v[6:7]\n\t"
        "v_pk_fma_f32 v[6:7], v[84:85], v[8:9], v[136:137] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v20, v85\n\t"
        "v_lshlrev_b32_e32 v36, 2, v3\n\t"
        "v_xor_b32_e32 v3, 32, v1\n\t"
        "v_pk_mul_f32 v[28:29], v[20:21], v[6:7]\n\t"
        "v_pk_fma_f32 v[6:7], v[68:69], v[8:9], v[136:137] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_mov_b32_e32 v20, v69\n\t"
        "v_cmp_lt_i32_e32 vcc, v3, v5\n\t"
        "v_pk_mul_f32 v[30:31], v[20:21], v[6:7]\n\t"
        "s_nop 7\n\t"
        "v_mov_b32 %[o0], v2\n\t"
        "v_mov_b32 %[o1], v4\n\t"
        "v_mov_b32 %[o2], v16\n\t"
        "v_mov_b32 %[o3], v18\n\t"
        "v_mov_b32 %[o4], v22\n\t"
        "v_mov_b32 %[o5], v24\n\t"
        "v_mov_b32 %[o6], v28\n\t"
        "v_mov_b32 %[o7], v30\n\t"
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5), [o6] "=&v"(o6),
          [o7] "=&v"(o7)
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [c6] "v"(c6), [c7] "v"(c7),
          [s01] "v"(s01), [s23] "v"(s23), [s45] "v"(s45), [s67] "v"(s67), [q01] "v"(q01), [q23] "v"(q23), [q45] "v"(q45),
          [q67] "v"(q67), [addr] "v"(addr), [junk] "v"(junk_bits ^ (unsigned)it)
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
          "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v28", "v29", "v30", "v31", "v36", "v68", "v69", "v72", "v73", "v76",
          "v77", "v80", "v81", "v84", "v85", "v88", "v89", "v92", "v93", "v96", "v97", "v136", "v137", "s6", "s7", "vcc",
          "memory");
    const float got[8] = {o0, o1, o2, o3, o4, o5, o6, o7};
    for (int j = 0; j < 8; ++j) {
      const float want = w[j] * __builtin_fmaf(-z[j], sq[j], sum[j]);
      if (got[j] != want) {
        ++n_bad;
        atomicOr(bad_j, 1u << j);
      }
    }
  }
  if (n_bad) atomicAdd(bad, n_bad);
}

// A co-runner for the second stream: every wave issues MFMAs back to back for `iters` rounds.
typedef __bf16 lab_bf16x8 __attribute__((ext_vector_type(8)));
typedef float lab_f32x16 __attribute__((ext_vector_type(16)));
template <bool VGPR_FORM>
__global__ __launch_bounds__(256, 1) void mfma_spin_kernel(float *out, int iters) {
  lab_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(float)(threadIdx.x % 7);
    b[i] = (__bf16)(float)(threadIdx.x % 5);
  }
  lab_f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  for (int it = 0; it < iters; ++it) {
    if (VGPR_FORM) {  // accumulators in ordinary vector registers
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
    } else {          // accumulators in the accumulation registers
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc3) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// Co-runners that isolate one ingredient of the library's weight-gradient kernel each.
typedef short lab_s16x4 __attribute__((ext_vector_type(4)));
template <int KIND>  // 0: transposing LDS reads (ds_read_b64_tr_b16)  1: plain 8-byte LDS reads  2: VALU only (v_pk_fma)
__global__ __launch_bounds__(256, 1) void spin2_kernel(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
  __syncthreads();
  unsigned acc = 0;
  float f0 = 1.0f, f1 = 2.0f;
  const unsigned char *p = lds + (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 2048;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      const lab_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) lab_s16x4 *)(__attribute__((address_space(3))) void *)(p + (it & 3) * 512));
      acc += (unsigned)v[0] + (unsigned)v[3];
    } else if (KIND == 1) {
      const uint2 v = *reinterpret_cast<const uint2 *>(p + (it & 3) * 512);
      acc += v.x + v.y;
    } else {
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double *>(&f0)));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc + f0 + f1;
}

#define CHECK(x)                                              \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

int main(int argc, char **argv) {
  // few iterations and many workgroups: workgroups retire and launch on every CU all the time
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int blocks = 256 * 3 * (argc > 2 ? atoi(argv[2]) : 2), threads = 256;
  unsigned long long *bad;
  unsigned *bad_j;
  uint4 *sink;
  CHECK(hipMalloc(&bad, sizeof(*bad)));
  CHECK(hipMalloc(&bad_j, sizeof(*bad_j)));
  CHECK(hipMalloc(&sink, sizeof(uint4) * blocks * threads));
  // third argument: rounds of the MFMA co-runner on a second stream (0 = none); it runs one 256-lane workgroup
  // per CU beside the replay's workgroups
  const int mfma_iters = argc > 3 ? atoi(argv[3]) : 0;
  hipStream_t s_main, s_side;
  CHECK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
  float *spin_out;
  CHECK(hipMalloc(&spin_out, sizeof(float) * 256 * 256));
  typedef int (*wgrad_fn)(const void *, int, const void *, int, float *, int, int, int, int, int, int, int, void *);
  wgrad_fn wgrad = nullptr;
  void *wg_dz = nullptr, *wg_x = nullptr;
  float *wg_g = nullptr;
  if (mfma_iters < 0) {
    const char *libname = getenv("UNET_LAB_LIB") ? getenv("UNET_LAB_LIB") : "unet_amd/lib/libunet_hip.so";
    void *h = dlopen(libname, RTLD_NOW);
    if (!h) {
      fprintf(stderr, "dlopen: %s\n", dlerror());
      return 1;
    }
    wgrad = (wgrad_fn)dlsym(h, "unet_conv3x3_wgrad_bf16");
    CHECK(hipMalloc(&wg_dz, 2 * 16 * 16 * 16 * 1024));
    CHECK(hipMalloc(&wg_x, 2 * 16 * 16 * 16 * 512));
    CHECK(hipMalloc(&wg_g, sizeof(float) * 512 * 1024 * 9));
    CHECK(hipMemset(wg_dz, 0x3c, 2 * 16 * 16 * 16 * 1024));
    CHECK(hipMemset(wg_x, 0x3c, 2 * 16 * 16 * 16 * 512));
  }
  const unsigned junks[] = {0x00007f12u, 0x7fc00000u, 0x3f800000u};
  for (unsigned junk : junks) {
    CHECK(hipMemset(bad, 0, sizeof(*bad)));
    CHECK(hipMemset(bad_j, 0, sizeof(*bad_j)));
    CHECK(hipDeviceSynchronize());
    const int kind = argc > 4 ? atoi(argv[4]) : -1;  // fourth argument: which co-runner
    if (mfma_iters > 0 && kind == -1)  // MFMA loop, accumulators in the accumulation registers
      hipLaunchKernelGGL(mfma_spin_kernel<false>, dim3(256), dim3(256), 0, s_side, spin_out, mfma_iters);
    if (mfma_iters > 0 && kind == -2)  // MFMA loop, accumulators in ordinary vector registers
      hipLaunchKernelGGL(mfma_spin_kernel<true>, dim3(256), dim3(256), 0, s_side, spin_out, mfma_iters);
    if (mfma_iters > 0 && kind == 0) hipLaunchKernelGGL(spin2_kernel<0>, dim3(256), dim3(256), 0, s_side, spin_out, mfma_iters);
    if (mfma_iters > 0 && kind == 1) hipLaunchKernelGGL(spin2_kernel<1>, dim3(256), dim3(256), 0, s_side, spin_out, mfma_iters);
    if (mfma_iters < 0)  // the library's 3x3 weight-gradient kernel as the co-runner, -mfma_iters launches
      for (int i = 0; i < -mfma_iters; ++i)
        if (wgrad(wg_dz, 1024, wg_x, 512, wg_g, 16, 16, 16, 512, 1024, 512, 1024, s_side)) {
          fprintf(stderr, "wgrad call failed\n");
          return 1;
        }
    hipLaunchKernelGGL(lab_kernel, dim3(blocks), dim3(threads), 0, s_main, bad, bad_j, sink, iters, junk);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    unsigned hj = 0;
    CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hj, bad_j, sizeof(hj), hipMemcpyDeviceToHost));
    printf("unused halves 0x%08x: %llu wrong products of %llu (mask of j: 0x%02x)\n", junk, h,
           (unsigned long long)blocks * threads * iters * 8, hj);
  }
  return 0;
}
