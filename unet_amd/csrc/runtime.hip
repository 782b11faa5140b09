// Runtime half of the C ABI: errors, devices, streams, events, memory.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

int unet_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- workspace -------------------------------------------------------------
namespace {
struct Workspace {
  int dev = -1;
  hipStream_t stream = nullptr;
  void *ptr = nullptr;
  size_t bytes = 0;
};
std::vector<Workspace> g_ws;  // one block per (device, stream) that ever asked
std::mutex g_ws_mu;
}  // namespace

int unet_workspace(void **ptr, size_t bytes, hipStream_t stream) {
  int dev = 0;
  UNET_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ws_mu);
  Workspace *w = nullptr;
  for (Workspace &e : g_ws)
    if (e.dev == dev && e.stream == stream) w = &e;
  if (!w) {
    g_ws.push_back(Workspace());
    w = &g_ws.back();
    w->dev = dev;
    w->stream = stream;
  }
  if (w->bytes < bytes) {
    // Kernels still reading the old block must finish before it is freed.
    if (w->ptr) {
      UNET_HIP_TRY(hipDeviceSynchronize());
      UNET_HIP_TRY(hipFree(w->ptr));
      w->ptr = nullptr;
      w->bytes = 0;
    }
    size_t want = bytes < (size_t)(8u << 20) ? (size_t)(8u << 20) : bytes + bytes / 2;
    UNET_HIP_TRY(hipMalloc(&w->ptr, want));
    w->bytes = want;
  }
  *ptr = w->ptr;
  return UNET_OK;
}

void unet_workspace_release_all() {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return;
  for (Workspace &e : g_ws)
    if (e.ptr) {
      (void)hipSetDevice(e.dev);
      (void)hipFree(e.ptr);
    }
  g_ws.clear();
  (void)hipSetDevice(cur);
}

// ---- profiling ---------------------------------------------------------------
namespace {
struct ProfRec {
  int kind;
  double flops, bytes;
  hipEvent_t a, b;
};
std::mutex g_prof_mu;
bool g_prof_enabled = false;
uint64_t g_prof_mask = ~0ull;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
const char *kProfNames[PROF_KINDS] = {
    "conv_fwd_f32",      "conv_dgrad_f32",     "conv_wgrad_f32",     "tconv_f32",
    "batch_norm_f32",    "conv3x3_fwd_bf16",   "conv3x3_dgrad_bf16", "conv3x3_wgrad_bf16",
    "wgrad_reduce",      "tconv2x2_bf16",      "tconv2x2_wgrad_bf16", "bn_act_bf16",
    "bn_bwd_reduce_bf16", "bn_bwd_apply_bf16", "pool_bf16",         "conv3x3_head_bf16"};
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  // no system-scope release at the record: the timers bracket kernels of one stream, nothing on the host reads
  // what those kernels wrote through these events
  (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  return e;
}
}  // namespace

bool unet_prof_on() { return g_prof_enabled; }
void unet_prof_mark(int kind, double flops, double bytes, hipStream_t s, bool begin) {
  if (!((g_prof_mask >> kind) & 1ull)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (begin) {
    ProfRec r{kind, flops, bytes, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, s);
    g_prof_recs.push_back(r);
  } else {
    for (size_t i = g_prof_recs.size(); i-- > 0;)
      if (g_prof_recs[i].kind == kind) {
        (void)hipEventRecord(g_prof_recs[i].b, s);
        break;
      }
  }
}

extern "C" {

int unet_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec &r : g_prof_recs) {
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_recs.clear();
  g_prof_enabled = on != 0;
  g_prof_mask = ~0ull;
  return UNET_OK;
}
int unet_prof_enable_kinds(uint64_t kind_mask) {
  int rc = unet_prof_enable(kind_mask != 0);
  g_prof_mask = kind_mask;
  return rc;
}
int unet_prof_kinds(int *count) {
  UNET_REQUIRE(count, "count is NULL");
  *count = PROF_KINDS;
  return UNET_OK;
}
const char *unet_prof_name(int kind) { return (kind >= 0 && kind < PROF_KINDS) ? kProfNames[kind] : ""; }
int unet_prof_get(int kind, int64_t *launches, double *total_ms, double *total_flops, double *total_bytes) {
  UNET_REQUIRE(kind >= 0 && kind < PROF_KINDS && launches && total_ms && total_flops && total_bytes,
               "unet_prof_get: bad argument");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *launches = 0;
  *total_ms = *total_flops = *total_bytes = 0;
  for (const ProfRec &r : g_prof_recs) {
    if (r.kind != kind) continue;
    float ms = 0.f;
    UNET_HIP_TRY(hipEventSynchronize(r.b));
    UNET_HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    *launches += 1;
    *total_ms += ms;
    *total_flops += r.flops;
    *total_bytes += r.bytes;
  }
  return UNET_OK;
}

const char *unet_last_error(void) { return g_err; }
int unet_version(void) { return 100; }

int unet_device_count(int *count) {
  UNET_REQUIRE(count, "count is NULL");
  UNET_HIP_TRY(hipGetDeviceCount(count));
  return UNET_OK;
}
int unet_set_device(int device) {
  UNET_HIP_TRY(hipSetDevice(device));
  return UNET_OK;
}
int unet_get_device(int *device) {
  UNET_REQUIRE(device, "device is NULL");
  UNET_HIP_TRY(hipGetDevice(device));
  return UNET_OK;
}
int unet_device_synchronize(void) {
  UNET_HIP_TRY(hipGetLastError());
  UNET_HIP_TRY(hipDeviceSynchronize());
  return UNET_OK;
}
int unet_device_reset(void) {
  UNET_HIP_TRY(hipDeviceSynchronize());
  unet_workspace_release_all();
  (void)hipGetLastError();
  return UNET_OK;
}

int unet_stream_create(unet_stream_t *stream) {
  UNET_REQUIRE(stream, "stream is NULL");
  hipStream_t s;
  UNET_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (unet_stream_t)s;
  return UNET_OK;
}
int unet_stream_destroy(unet_stream_t stream) {
  UNET_HIP_TRY(hipStreamDestroy(as_stream(stream)));
  return UNET_OK;
}
int unet_stream_synchronize(unet_stream_t stream) {
  UNET_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return UNET_OK;
}
int unet_event_create(unet_event_t *event) {
  UNET_REQUIRE(event, "event is NULL");
  hipEvent_t e;
  UNET_HIP_TRY(hipEventCreate(&e));
  *event = (unet_event_t)e;
  return UNET_OK;
}
int unet_event_destroy(unet_event_t event) {
  UNET_HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return UNET_OK;
}
int unet_event_record(unet_event_t event, unet_stream_t stream) {
  UNET_HIP_TRY(hipEventRecord((hipEvent_t)event, as_stream(stream)));
  return UNET_OK;
}
int unet_event_synchronize(unet_event_t event) {
  UNET_HIP_TRY(hipEventSynchronize((hipEvent_t)event));
  return UNET_OK;
}
int unet_stream_wait_event(unet_stream_t stream, unet_event_t event) {
  UNET_HIP_TRY(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
  return UNET_OK;
}
int unet_event_elapsed_ms(float *ms, unet_event_t start, unet_event_t stop) {
  UNET_REQUIRE(ms, "ms is NULL");
  UNET_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return UNET_OK;
}

int unet_malloc(void **ptr, size_t bytes) {
  UNET_REQUIRE(ptr, "ptr is NULL");
  *ptr = nullptr;
  if (bytes == 0) bytes = 4;
  UNET_HIP_TRY(hipMalloc(ptr, bytes));
  UNET_HIP_TRY(hipMemset(*ptr, 0, bytes));
  // The fill runs on the null stream and may still be queued when hipMemset returns (behind work of the
  // blocking streams).  A non-blocking stream does not wait for it: without this wait, a copy enqueued on
  // one right after the allocation could land first and then be zeroed.
  UNET_HIP_TRY(hipStreamSynchronize(nullptr));
  return UNET_OK;
}
int unet_free(void *ptr) {
  if (ptr) UNET_HIP_TRY(hipFree(ptr));
  return UNET_OK;
}
int unet_host_malloc(void **ptr, size_t bytes) {
  UNET_REQUIRE(ptr, "ptr is NULL");
  UNET_HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 4, hipHostMallocDefault));
  return UNET_OK;
}
int unet_host_free(void *ptr) {
  if (ptr) UNET_HIP_TRY(hipHostFree(ptr));
  return UNET_OK;
}
int unet_memset(void *ptr, int value, size_t bytes, unet_stream_t stream) {
  if (bytes) UNET_HIP_TRY(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
  return UNET_OK;
}
int unet_memcpy_h2d(void *dst, const void *src_host, size_t bytes, unet_stream_t stream) {
  if (bytes) UNET_HIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  return UNET_OK;
}
int unet_memcpy_d2h(void *dst_host, const void *src, size_t bytes, int blocking, unet_stream_t stream) {
  if (bytes) UNET_HIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  if (blocking) UNET_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return UNET_OK;
}
int unet_memcpy_d2d(void *dst, const void *src, size_t bytes, unet_stream_t stream) {
  if (bytes) UNET_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return UNET_OK;
}

}  // extern "C"
