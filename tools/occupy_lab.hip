// Laboratory: a kernel that holds `wgs` workgroup slots for `usec` microseconds without touching memory --
// a stand-in for the resident kernels of a collective library -- to see what a training step costs when some CUs
// are not available to its persistent kernels.  Built as a shared library for tools/lab_occupied_cus.py:
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/occupy_lab.hip -o tools/bin/liboccupy_lab.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, int *sink) {
  const long long t0 = wall_clock64();  // 100 MHz
  int spins = 0;
  // both a time limit and a trip limit: every wave leaves
  while (wall_clock64() - t0 < ticks && spins < (1 << 26)) {
    __builtin_amdgcn_s_sleep(32);
    ++spins;
  }
  if (sink && threadIdx.x == 0 && spins < 0) *sink = spins;
}

extern "C" int occupy(void *stream, int wgs, int usec, int lds_bytes) {
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), lds_bytes, (hipStream_t)stream, (long long)usec * 100, (int *)nullptr);
  return (int)hipGetLastError();
}
