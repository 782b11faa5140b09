// Data-parallel collectives of the C ABI on RCCL (one process per GPU, one communicator rank per process).
//
// The reference trains on one device and has no collective at all; these entry points carry the only two
// exchanges batch-sharded training needs: the mean of the parameter gradients (bucket by bucket, on a
// communication stream of the communicator's own so that it overlaps with the rest of backward) and small
// in-stream sums (batch-norm statistics across ranks, parameter broadcast).
//
// librccl.so.1 is opened on the first unet_comm_* call, not at load time: single-GPU users of the library do
// not pay for (or depend on) it, and a process that already holds an RCCL (e.g. through torch) shares that
// copy, because the loader resolves by soname.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return UNET_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return unet_set_error(UNET_ERR_UNSUPPORTED, "cannot open librccl.so.1: %s", dlerror());
  Rccl r;
  r.handle = h;
#define SYM(field, name)                                                                         \
  do {                                                                                           \
    *(void **)(&r.field) = dlsym(h, name);                                                       \
    if (!r.field) return unet_set_error(UNET_ERR_UNSUPPORTED, "librccl has no symbol %s", name); \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Broadcast, "ncclBroadcast");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl = r;
  return UNET_OK;
}

#define UNET_NCCL_TRY(expr)                                                                                  \
  do {                                                                                                       \
    ncclResult_t r_ = (expr);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return unet_set_error(UNET_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, \
                            __LINE__);                                                                       \
  } while (0)

struct Comm {
  ncclComm_t nccl = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;  // the communication stream
  hipEvent_t done = nullptr;     // recorded after the last collective enqueued on it
  bool pending = false;
  bool bf16_gradients = false;   // unet_comm_set_bf16_gradients: buckets travel as bf16
  uint16_t *stage = nullptr;     // bf16 copy of the bucket in flight (grow-only; the stream serialises its uses)
  size_t stage_elems = 0;
};

// fp32 <-> bf16 (round to nearest even) over a gradient bucket, grid-stride
__global__ __launch_bounds__(256) void grad_to_bf16_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const __bf16 h = (__bf16)in[i];
    out[i] = __builtin_bit_cast(unsigned short, h);
  }
}
__global__ __launch_bounds__(256) void grad_from_bf16_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
    out[i] = __uint_as_float((uint32_t)in[i] << 16);
}

// Grows the bf16 staging buffer to `elems` elements (earlier buckets may still use the old one: drain first).
int reserve_stage(Comm *c, size_t elems) {
  if (c->stage_elems >= elems) return UNET_OK;
  UNET_HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->stage) UNET_HIP_TRY(hipFree(c->stage));
  c->stage = nullptr;
  c->stage_elems = 0;
  UNET_HIP_TRY(hipMalloc((void **)&c->stage, sizeof(uint16_t) * elems));
  c->stage_elems = elems;
  return UNET_OK;
}

// ---- laboratory: what the collective's resident kernels cost the kernels they overlap --------------------------
// UNET_LAB_COMM_EMULATE="<workgroups>,<LDS bytes>,<GB/s>": on a ONE-rank communicator (a one-GPU box), the bucket
// all-reduce of unet_allreduce_begin is replaced by a kernel that holds that many workgroup slots (256 threads and
// that much LDS each, no memory traffic) for the time the bucket would take at that rate -- a stand-in for RCCL's
// device kernels on an 8-GPU node, issued from the same place, on the same stream, behind the same event.  Used
// by tools/lab_comm_overlap.py; never active unless the variable is set.
struct CommEmulation {
  int wgs = 0, lds = 0;
  double gbps = 0;
};
const CommEmulation &comm_emulation() {
  static const CommEmulation e = [] {
    CommEmulation v;
    if (const char *t = getenv("UNET_LAB_COMM_EMULATE")) {
      if (sscanf(t, "%d,%d,%lf", &v.wgs, &v.lds, &v.gbps) != 3 || v.wgs < 1 || v.wgs > 256 || v.lds < 0 || v.lds > 65536 ||
          v.gbps <= 0)
        v = CommEmulation();
    }
    return v;
  }();
  return e;
}
__global__ __launch_bounds__(256) void occupy_slots_kernel(long long ticks) {  // ticks of the 100 MHz wall clock
  extern __shared__ unsigned char held[];
  const long long t0 = wall_clock64();
  int spins = 0;
  // a time limit and a trip limit: every wave leaves
  while (wall_clock64() - t0 < ticks && spins < (1 << 22)) {
    __builtin_amdgcn_s_sleep(32);
    ++spins;
  }
  if (spins < 0) held[threadIdx.x] = 0;
}

// The communicator batch-norm launchers synchronise their statistics over (NULL = per-rank statistics).
Comm *g_bn_sync = nullptr;

}  // namespace

// Used by the batch-norm launchers (ops_f32.hip, fused_bf16.hip): sums `count` doubles over the ranks of the
// communicator selected with unet_bn_set_sync, in place, on stream `s`.  *world receives the number of
// ranks (1 and no work when no communicator is selected).
int unet_bn_sync_sum_f64(double *data, i64 count, int *world, hipStream_t s) {
  *world = 1;
  Comm *c = g_bn_sync;
  if (!c) return UNET_OK;
  *world = c->world;
  if (count <= 0 || (c->world == 1 && !getenv("UNET_COMM_FORCE"))) return UNET_OK;
  UNET_NCCL_TRY(g_rccl.AllReduce(data, data, (size_t)count, ncclFloat64, ncclSum, c->nccl, s));
  return UNET_OK;
}
bool unet_bn_sync_on() { return g_bn_sync != nullptr; }

extern "C" {

int unet_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

int unet_comm_get_unique_id(void *id_host) {
  UNET_REQUIRE(id_host, "unet_comm_get_unique_id: NULL pointer");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  UNET_NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_host, &id, NCCL_UNIQUE_ID_BYTES);
  return UNET_OK;
}

int unet_comm_init_rank(unet_comm_t *comm, const void *id_host, int world_size, int rank) {
  UNET_REQUIRE(comm && id_host, "unet_comm_init_rank: NULL pointer");
  UNET_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "unet_comm_init_rank: need 0 <= rank < world_size");
  *comm = nullptr;
  int rc = load_rccl();
  if (rc) return rc;
  Comm *c = new Comm();
  c->world = world_size;
  c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, id_host, NCCL_UNIQUE_ID_BYTES);
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (e != hipSuccess) {
    delete c;
    return unet_set_error(UNET_ERR_HIP, "unet_comm_init_rank: %s", hipGetErrorString(e));
  }
  ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world_size, id, rank);
  if (r != ncclSuccess) {
    (void)hipStreamDestroy(c->stream);
    (void)hipEventDestroy(c->done);
    delete c;
    return unet_set_error(UNET_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  *comm = c;
  return UNET_OK;
}

int unet_comm_destroy(unet_comm_t comm) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c) return UNET_OK;
  if (g_bn_sync == c) g_bn_sync = nullptr;
  (void)hipStreamSynchronize(c->stream);
  if (c->stage) (void)hipFree(c->stage);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  (void)hipStreamDestroy(c->stream);
  (void)hipEventDestroy(c->done);
  delete c;
  return UNET_OK;
}

int unet_comm_size(unet_comm_t comm, int *world_size, int *rank) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c, "unet_comm_size: NULL communicator");
  if (world_size) *world_size = c->world;
  if (rank) *rank = c->rank;
  return UNET_OK;
}

int unet_allreduce_begin(unet_comm_t comm, float *data, int64_t count, unet_event_t ready) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c && data && count > 0, "unet_allreduce_begin: bad argument");
  if (ready) UNET_HIP_TRY(hipStreamWaitEvent(c->stream, (hipEvent_t)ready, 0));
  const CommEmulation &emu = comm_emulation();
  if (emu.wgs > 0 && c->world == 1) {
    const double bytes = (double)count * (c->bf16_gradients ? 2.0 : 4.0);
    long long ticks = (long long)(bytes / (emu.gbps * 1e9) * 1e8);
    if (ticks > 100000000ll) ticks = 100000000ll;  // one second at most
    hipLaunchKernelGGL(occupy_slots_kernel, dim3(emu.wgs), dim3(256), emu.lds, c->stream, ticks);
    UNET_LAUNCH_CHECK("occupy_slots_kernel");
    c->pending = true;
    return UNET_OK;
  }
  if (c->bf16_gradients) {
    // half the bytes on the links and half the time the collective's kernels stay resident: the bucket is
    // rounded to bf16, averaged in bf16 and widened again, all on the communication stream
    // (unet_comm_reserve_bf16_stage sizes the buffer ahead of time, so that nothing allocates or synchronises
    // inside a backward pass; a bucket that still outgrows it takes the slow path once)
    if (c->stage_elems < (size_t)count) {
      int rc = reserve_stage(c, (size_t)count);
      if (rc) return rc;
    }
    const unsigned blocks = (unsigned)((count + 1023) / 1024 < 4096 ? (count + 1023) / 1024 : 4096);
    hipLaunchKernelGGL(grad_to_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, (const float *)data, c->stage, (i64)count);
    UNET_LAUNCH_CHECK("grad_to_bf16_kernel");
    UNET_NCCL_TRY(g_rccl.AllReduce(c->stage, c->stage, (size_t)count, ncclBfloat16, ncclAvg, c->nccl, c->stream));
    hipLaunchKernelGGL(grad_from_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint16_t *)c->stage, data, (i64)count);
    UNET_LAUNCH_CHECK("grad_from_bf16_kernel");
  } else {
    UNET_NCCL_TRY(g_rccl.AllReduce(data, data, (size_t)count, ncclFloat32, ncclAvg, c->nccl, c->stream));
  }
  c->pending = true;
  return UNET_OK;
}

int unet_comm_set_bf16_gradients(unet_comm_t comm, int on) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c, "unet_comm_set_bf16_gradients: NULL communicator");
  c->bf16_gradients = on != 0;
  return UNET_OK;
}

int unet_comm_reserve_bf16_stage(unet_comm_t comm, int64_t max_count) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c && max_count > 0, "unet_comm_reserve_bf16_stage: bad argument");
  return reserve_stage(c, (size_t)max_count);
}

int unet_allreduce_wait(unet_comm_t comm, unet_stream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c, "unet_allreduce_wait: NULL communicator");
  if (!c->pending) return UNET_OK;
  UNET_HIP_TRY(hipEventRecord(c->done, c->stream));
  UNET_HIP_TRY(hipStreamWaitEvent(as_stream(stream), c->done, 0));
  c->pending = false;
  return UNET_OK;
}

int unet_allreduce_sum_f32(unet_comm_t comm, float *data, int64_t count, unet_stream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c && data && count > 0, "unet_allreduce_sum_f32: bad argument");
  UNET_NCCL_TRY(g_rccl.AllReduce(data, data, (size_t)count, ncclFloat32, ncclSum, c->nccl, as_stream(stream)));
  return UNET_OK;
}

int unet_broadcast_f32(unet_comm_t comm, float *data, int64_t count, int root, unet_stream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  UNET_REQUIRE(c && data && count > 0 && root >= 0 && root < c->world, "unet_broadcast_f32: bad argument");
  UNET_NCCL_TRY(g_rccl.Broadcast(data, data, (size_t)count, ncclFloat32, root, c->nccl, as_stream(stream)));
  return UNET_OK;
}

int unet_bn_set_sync(unet_comm_t comm) {
  g_bn_sync = static_cast<Comm *>(comm);
  return UNET_OK;
}

}  // extern "C"
