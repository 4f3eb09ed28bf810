// Context, memory and communicator entry points of libcxschur
// (ContextImpl of the reference: context_impl.h:60-150, context_impl.cc:125-203).
#include <dlfcn.h>

#include <chrono>

#include "cx_internal.h"

static thread_local char g_error[512] = "";

void cx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------ RCCL
// librccl is resolved at run time so that a process that already carries a copy
// (torch's) keeps exactly one instance: dlopen by soname returns the loaded one.
namespace {
struct UniqueId { char internal[128]; };
using GetUniqueIdFn = int (*)(UniqueId*);
using CommInitRankFn = int (*)(void**, int, UniqueId, int);
using AllReduceFn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using CommDestroyFn = int (*)(void*);
using GetErrorStringFn = const char* (*)(int);
struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
} g_rccl;
constexpr int kNcclFloat64 = 8;  // rccl.h:467
constexpr int kNcclSum = 0;      // rccl.h:448

int LoadRccl() {
  if (g_rccl.handle) return CX_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    cx_set_error("cannot load librccl: %s", dlerror());
    return CX_ERR_COMM;
  }
  g_rccl.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
  g_rccl.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
  g_rccl.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  g_rccl.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
    cx_set_error("librccl lacks a required symbol");
    return CX_ERR_COMM;
  }
  g_rccl.handle = h;
  return CX_OK;
}
int RcclCheck(int rc, const char* what) {
  if (rc == 0) return CX_OK;
  cx_set_error("%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "rccl error");
  return CX_ERR_COMM;
}
}  // namespace

static int AllReduce(cx_context* ctx, double* p, int64_t n, bool even_single_rank) {
  if (n == 0) return CX_OK;
  // with one rank the sum is the identity: skipped inside the solvers, but cx_allreduce_sum still goes through
  // RCCL when a communicator exists, so that the call path can be exercised on a one-GPU box
  if (ctx->nranks <= 1 && !(even_single_rank && ctx->comm)) return CX_OK;
  ctx->ar_calls += 1;
  ctx->ar_bytes += n * int64_t(sizeof(double));
  if (ctx->allreduce_cb) {
    CX_HIP(hipStreamSynchronize(ctx->stream));
    auto t0 = std::chrono::steady_clock::now();
    if (ctx->allreduce_cb(p, n, ctx->allreduce_cb_user) != 0) {
      cx_set_error("all-reduce callback failed");
      return CX_ERR_COMM;
    }
    ctx->allreduce_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CX_OK;
  }
  if (!ctx->comm) {
    cx_set_error("context has %d ranks but no communicator", ctx->nranks);
    return CX_ERR_COMM;
  }
  // device time of the collective: an event pair on the stream it runs on (host enqueue time says nothing about it)
  const bool timed = ctx->ar_timed < cx_context::kTimedCollectives;
  if (timed) {
    if (ctx->ar_events.empty()) {
      ctx->ar_events.resize(2 * cx_context::kTimedCollectives);
      for (auto& ev : ctx->ar_events) CX_HIP(hipEventCreate(&ev));
    }
    CX_HIP(hipEventRecord(ctx->ar_events[size_t(2 * ctx->ar_timed)], ctx->stream));
  }
  auto t0 = std::chrono::steady_clock::now();
  int rc = g_rccl.all_reduce(p, p, size_t(n), kNcclFloat64, kNcclSum, ctx->comm, ctx->stream);
  ctx->allreduce_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (timed) {
    CX_HIP(hipEventRecord(ctx->ar_events[size_t(2 * ctx->ar_timed + 1)], ctx->stream));
    ++ctx->ar_timed;
  }
  return RcclCheck(rc, "ncclAllReduce");
}

void cx_allreduce_reset(cx_context* ctx) {
  ctx->allreduce_host_ms = 0.0;
  ctx->ar_timed = 0;
  ctx->ar_calls = 0;
  ctx->ar_bytes = 0;
}

int cx_allreduce_collect(cx_context* ctx, double* device_ms, double* host_ms, double* calls, double* bytes) {
  double sum = 0.0;
  for (int i = 0; i < ctx->ar_timed; ++i) {
    float ms = 0.f;
    CX_HIP(hipEventSynchronize(ctx->ar_events[size_t(2 * i + 1)]));
    CX_HIP(hipEventElapsedTime(&ms, ctx->ar_events[size_t(2 * i)], ctx->ar_events[size_t(2 * i + 1)]));
    sum += ms;
  }
  // more collectives than event pairs (a long CG run): the untimed ones are priced at the timed ones' mean
  if (ctx->ar_timed > 0 && ctx->ar_calls > ctx->ar_timed) sum *= double(ctx->ar_calls) / double(ctx->ar_timed);
  *device_ms = sum;
  *host_ms = ctx->allreduce_host_ms;
  *calls = double(ctx->ar_calls);
  *bytes = double(ctx->ar_bytes);
  return CX_OK;
}

int cx_allreduce_device(cx_context* ctx, double* p, int64_t n) { return AllReduce(ctx, p, n, false); }

extern "C" {

const char* cx_last_error(void) { return g_error; }

int cx_context_create(int device_id, cx_context** out) {
  CX_CHECK_ARG(out != nullptr);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    cx_set_error("no HIP device available (%s); libcxschur has no CPU path", hipGetErrorString(e));
    return CX_ERR_NO_DEVICE;
  }
  CX_CHECK_ARG(device_id >= 0 && device_id < count);
  CX_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  CX_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    cx_set_error("device %d is %s; libcxschur carries gfx950 code objects only", device_id, prop.gcnArchName);
    return CX_ERR_NO_DEVICE;
  }
  auto* ctx = new cx_context;
  ctx->device = device_id;
  ctx->num_cus = prop.multiProcessorCount;
  std::snprintf(ctx->name, sizeof(ctx->name), "%s %s (%d CUs, %.0f GiB)", prop.gcnArchName, prop.name,
                prop.multiProcessorCount, double(prop.totalGlobalMem) / (1 << 30));
  CX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto& ev : ctx->ev) CX_HIP(hipEventCreate(&ev));
  *out = ctx;
  return CX_OK;
}

void cx_context_destroy(cx_context* ctx) {
  if (!ctx) return;
  if (cxm_is_front(ctx) || ctx->group) cxm_context_destroy_shards(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  cx_xfer_destroy(ctx);
  if (ctx->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(ctx->comm);
  for (auto& ev : ctx->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ar_events)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int cx_comm_unique_id(void* out) {
  CX_CHECK_ARG(out != nullptr);
  CX_TRY(LoadRccl());
  UniqueId id;
  CX_TRY(RcclCheck(g_rccl.get_unique_id(&id), "ncclGetUniqueId"));
  std::memcpy(out, &id, sizeof(id));
  return CX_OK;
}

int cx_context_set_comm(cx_context* ctx, int rank, int nranks, const void* unique_id) {
  CX_CHECK_ARG(ctx != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
  if (nranks == 1 && unique_id == nullptr) {
    ctx->rank = 0;
    ctx->nranks = 1;
    return CX_OK;
  }
  CX_CHECK_ARG(unique_id != nullptr);
  CX_TRY(LoadRccl());
  CX_HIP(hipSetDevice(ctx->device));
  UniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  void* comm = nullptr;
  CX_TRY(RcclCheck(g_rccl.comm_init_rank(&comm, nranks, id, rank), "ncclCommInitRank"));
  ctx->comm = comm;
  ctx->rank = rank;
  ctx->nranks = nranks;
  return CX_OK;
}

int cx_context_set_comm_callback(cx_context* ctx, int rank, int nranks, cx_allreduce_fn fn, void* user) {
  CX_CHECK_ARG(ctx != nullptr && nranks >= 1 && rank >= 0 && rank < nranks && fn != nullptr);
  ctx->allreduce_cb = fn;
  ctx->allreduce_cb_user = user;
  ctx->rank = rank;
  ctx->nranks = nranks;
  return CX_OK;
}

int cx_context_rank(const cx_context* ctx) { return ctx ? ctx->rank : 0; }
int cx_context_num_ranks(const cx_context* ctx) { return ctx ? ctx->nranks : 1; }

int cx_allreduce_sum(cx_context* ctx, double* device_ptr, int64_t n) {
  CX_CHECK_ARG(ctx != nullptr && (device_ptr != nullptr || n == 0));
  return AllReduce(ctx, device_ptr, n, true);
}

int cx_malloc(cx_context* ctx, size_t bytes, void** device_ptr) {
  CX_CHECK_ARG(ctx != nullptr && device_ptr != nullptr);
  CX_HIP(hipMalloc(device_ptr, bytes ? bytes : 1));
  return CX_OK;
}
int cx_free(cx_context* ctx, void* device_ptr) {
  CX_CHECK_ARG(ctx != nullptr);
  if (device_ptr) CX_HIP(hipFree(device_ptr));
  return CX_OK;
}
int cx_memcpy_h2d(cx_context* ctx, void* dst, const void* src, size_t bytes) {
  CX_CHECK_ARG(ctx != nullptr);
  if (bytes == 0) return CX_OK;
  CX_HIP(hipSetDevice(ctx->device));
  CX_TRY(cx_copy_h2d(ctx, dst, src, bytes));
  CX_HIP(hipStreamSynchronize(ctx->stream));
  return CX_OK;
}
int cx_memcpy_d2h(cx_context* ctx, void* dst, const void* src, size_t bytes) {
  CX_CHECK_ARG(ctx != nullptr);
  if (bytes == 0) return CX_OK;
  CX_HIP(hipSetDevice(ctx->device));
  CX_TRY(cx_copy_d2h(ctx, dst, src, bytes));
  CX_HIP(hipStreamSynchronize(ctx->stream));
  return CX_OK;
}
int cx_memset_zero(cx_context* ctx, void* p, size_t bytes) {
  CX_CHECK_ARG(ctx != nullptr);
  if (bytes) CX_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream));
  return CX_OK;
}
int cx_synchronize(cx_context* ctx) {
  CX_CHECK_ARG(ctx != nullptr);
  CX_HIP(hipStreamSynchronize(ctx->stream));
  return CX_OK;
}
void* cx_context_stream(cx_context* ctx) { return ctx ? ctx->stream : nullptr; }

int cx_device_name(cx_context* ctx, char* out, size_t n) {
  CX_CHECK_ARG(ctx != nullptr && out != nullptr && n > 0);
  std::snprintf(out, n, "%s", ctx->name);
  return CX_OK;
}

}  // extern "C"

