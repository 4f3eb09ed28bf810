// Context, memory and communicator entry points of libcxschur
// (ContextImpl of the reference: context_impl.h:60-150, context_impl.cc:125-203).
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <thread>

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
using ReduceScatterFn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using CommDestroyFn = int (*)(void*);
using CommAbortFn = int (*)(void*);
using GetErrorStringFn = const char* (*)(int);
struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  ReduceScatterFn reduce_scatter = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  CommAbortFn comm_abort = nullptr;
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
  g_rccl.reduce_scatter = reinterpret_cast<ReduceScatterFn>(dlsym(h, "ncclReduceScatter"));
  g_rccl.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  g_rccl.comm_abort = reinterpret_cast<CommAbortFn>(dlsym(h, "ncclCommAbort"));
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

// ---------------------------------------------------------------- a collective must not be able to hang the caller
// RCCL collectives are enqueued; a rank whose peer never arrives (the peer failed locally and returned, or died) shows it
// at the next wait on the stream, which would block for good.  On a context with an RCCL communicator of several ranks
// every such wait polls with a deadline instead (cx_stream_sync / cx_event_sync, used by all code a sharded solve runs
// through); when the deadline passes the communicator is aborted -- ncclCommAbort makes the stuck collective's kernel
// leave -- the context is marked broken and the call returns CX_ERR_COMM, which the callers turn into FATAL_ERROR.
// Nothing is retried and no process is re-executed.  CX_COMM_TIMEOUT_S (default 120) or cx_context_set_comm_timeout.
static double CommTimeout(const cx_context* ctx) {
  if (ctx->comm_timeout_s > 0.0) return ctx->comm_timeout_s;
  static const double from_env = [] {
    const char* e = std::getenv("CX_COMM_TIMEOUT_S");
    const double v = e ? std::atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
  }();
  return from_env;
}

double cx_comm_timeout(const cx_context* ctx) { return CommTimeout(ctx); }

int cx_read_back(cx_context* ctx, void* host, const void* dev, size_t bytes, hipStream_t st) {
  if (!st) st = ctx->stream;
  if (ctx->comm && ctx->nranks > 1) CX_TRY(cx_stream_sync(ctx, st));
  CX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
  CX_HIP(hipStreamSynchronize(st));
  return CX_OK;
}

int cx_comm_abort(cx_context* ctx) {
  if (!ctx) return CX_OK;
  void* comm = ctx->comm;
  ctx->comm = nullptr;
  ctx->comm_broken = ctx->comm_broken || ctx->nranks > 1;
  if (comm) {
    if (g_rccl.comm_abort) g_rccl.comm_abort(comm);
    else if (g_rccl.comm_destroy) g_rccl.comm_destroy(comm);
  }
  return CX_OK;
}

template <typename Ready>
static int BoundedWait(cx_context* ctx, hipStream_t drain, Ready ready, const char* what) {
  const double limit = CommTimeout(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {
    const hipError_t q = ready();
    if (q == hipSuccess) return CX_OK;
    if (q != hipErrorNotReady) {
      cx_set_error("%s failed: %s", what, hipGetErrorString(q));
      return CX_ERR_HIP;
    }
    if (spins > 4000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 127) == 127 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
  }
  // no progress for `limit` seconds: the exchange step is waiting for a rank that will not come
  cx_comm_abort(ctx);
  // the aborted collective leaves its kernel; give the stream a bounded time to drain so that later frees are safe
  const auto t1 = std::chrono::steady_clock::now();
  while (hipStreamQuery(drain) == hipErrorNotReady && std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 10.0)
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  cx_set_error("rank %d of %d: no progress on the stream for %.3g s -- a peer never reached the exchange step (it failed or died); "
               "communicator aborted, this context cannot run sharded work any more", ctx->rank, ctx->nranks, limit);
  return CX_ERR_COMM;
}

int cx_stream_sync(cx_context* ctx, hipStream_t st) {
  if (!st) st = ctx->stream;
  if (!(ctx->comm && ctx->nranks > 1)) {
    CX_HIP(hipStreamSynchronize(st));
    return CX_OK;
  }
  return BoundedWait(ctx, st, [&] { return hipStreamQuery(st); }, "hipStreamQuery");
}

int cx_event_sync(cx_context* ctx, hipEvent_t ev) {
  if (!(ctx->comm && ctx->nranks > 1)) {
    CX_HIP(hipEventSynchronize(ev));
    return CX_OK;
  }
  return BoundedWait(ctx, ctx->stream, [&] { return hipEventQuery(ev); }, "hipEventQuery");
}

static int AllReduce(cx_context* ctx, double* p, int64_t n, bool even_single_rank);

// Every rank says whether it is still healthy; all of them learn whether any is not.  Called where a sharded phase is
// about to start exchanging (after argument checks, allocations and input copies -- the places a rank fails on its own):
// a rank that failed still takes part, with its flag set, and ALL ranks leave the call with an error together instead of
// the healthy ones waiting inside the first collective.
int cx_comm_agree(cx_context* ctx, int local_rc) {
  if (ctx->nranks <= 1) return local_rc;
  if (ctx->comm_broken) {
    if (local_rc == CX_OK) cx_set_error("the communicator of this context was aborted after a lost rank");
    return local_rc != CX_OK ? local_rc : CX_ERR_COMM;
  }
  std::string keep = local_rc != CX_OK ? cx_last_error() : "";
  int rc = ctx->agree.alloc(1);
  double flag = local_rc != CX_OK ? 1.0 : 0.0;
  if (rc == CX_OK && hipMemcpyAsync(ctx->agree.p, &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = CX_ERR_HIP;
  if (rc == CX_OK) rc = AllReduce(ctx, ctx->agree.p, 1, false);
  if (rc == CX_OK && hipMemcpyAsync(&flag, ctx->agree.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = CX_ERR_HIP;
  if (rc == CX_OK) rc = cx_stream_sync(ctx, ctx->stream);
  if (local_rc != CX_OK) {
    cx_set_error("%s", keep.c_str());
    return local_rc;
  }
  if (rc != CX_OK) return rc;
  if (flag != 0.0) {
    cx_set_error("rank %d of %d: another rank failed before the exchange step of this call; all ranks return together", ctx->rank, ctx->nranks);
    return CX_ERR_COMM;
  }
  return CX_OK;
}

static int AllReduce(cx_context* ctx, double* p, int64_t n, bool even_single_rank) {
  if (n == 0) return CX_OK;
  // with one rank the sum is the identity: skipped inside the solvers, but cx_allreduce_sum still goes through
  // RCCL when a communicator exists, so that the call path can be exercised on a one-GPU box
  if (ctx->nranks <= 1 && !(even_single_rank && ctx->comm)) return CX_OK;
  if (ctx->comm_broken) {
    cx_set_error("the communicator of this context was aborted after a lost rank");
    return CX_ERR_COMM;
  }
  // failure injection (cx_debug_inject_failure): this rank "fails locally" right before its n-th collective from now
  if (ctx->fail_countdown >= 0 && ctx->fail_countdown-- == 0) {
    ctx->fail_countdown = -1;
    cx_set_error("injected failure before a collective (cx_debug_inject_failure)");
    return CX_ERR_HIP;
  }
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

int cx_reduce_scatter_device(cx_context* ctx, double* send, double* recv, int64_t count) {
  if (count == 0) return CX_OK;
  if (ctx->nranks <= 1) {
    CX_HIP(hipMemcpyAsync(recv, send, size_t(count) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return CX_OK;
  }
  if (ctx->comm_broken) {
    cx_set_error("the communicator of this context was aborted after a lost rank");
    return CX_ERR_COMM;
  }
  const int64_t total = count * ctx->nranks;
  const bool native = ctx->reduce_scatter_cb != nullptr || (ctx->allreduce_cb == nullptr && ctx->comm != nullptr && g_rccl.reduce_scatter != nullptr);
  if (!native) {  // a transport that only sums whole buffers: sum, keep the own range
    CX_TRY(AllReduce(ctx, send, total, false));
    CX_HIP(hipMemcpyAsync(recv, send + int64_t(ctx->rank) * count, size_t(count) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return CX_OK;
  }
  if (ctx->fail_countdown >= 0 && ctx->fail_countdown-- == 0) {
    ctx->fail_countdown = -1;
    cx_set_error("injected failure before a collective (cx_debug_inject_failure)");
    return CX_ERR_HIP;
  }
  ctx->ar_calls += 1;
  ctx->ar_bytes += total * int64_t(sizeof(double)) / 2;  // what it moves, in all-reduce terms
  if (ctx->reduce_scatter_cb) {
    CX_HIP(hipStreamSynchronize(ctx->stream));
    auto t0 = std::chrono::steady_clock::now();
    if (ctx->reduce_scatter_cb(send, recv, count, ctx->allreduce_cb_user) != 0) {
      cx_set_error("reduce-scatter callback failed");
      return CX_ERR_COMM;
    }
    ctx->allreduce_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CX_OK;
  }
  auto t0 = std::chrono::steady_clock::now();
  const int rc = g_rccl.reduce_scatter(send, recv, size_t(count), kNcclFloat64, kNcclSum, ctx->comm, ctx->stream);
  ctx->allreduce_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return RcclCheck(rc, "ncclReduceScatter");
}

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

int cx_context_set_comm_timeout(cx_context* ctx, double seconds) {
  CX_CHECK_ARG(ctx != nullptr && seconds >= 0.0);
  ctx->comm_timeout_s = seconds;
  for (cx_context* shard : ctx->shards)
    if (shard) shard->comm_timeout_s = seconds;
  return CX_OK;
}

int cx_debug_inject_failure(cx_context* ctx, int32_t shard, int64_t nth_collective) {
  CX_CHECK_ARG(ctx != nullptr && nth_collective >= -1);
  cx_context* target = ctx;
  if (!ctx->shards.empty()) {
    CX_CHECK_ARG(shard >= 0 && shard < int32_t(ctx->shards.size()));
    target = ctx->shards[size_t(shard)];
  }
  target->fail_countdown = nth_collective;
  return CX_OK;
}

int cx_debug_force_rank_count(cx_context* ctx, int32_t nranks) {
  CX_CHECK_ARG(ctx != nullptr && nranks >= 1);
  ctx->nranks = nranks;
  return CX_OK;
}

static void StallCallback(void* user) {
  std::this_thread::sleep_for(std::chrono::milliseconds(reinterpret_cast<intptr_t>(user)));
}
int cx_debug_stall_stream(cx_context* ctx, int32_t milliseconds) {
  CX_CHECK_ARG(ctx != nullptr && milliseconds >= 0);
  CX_HIP(hipSetDevice(ctx->device));
  CX_HIP(hipLaunchHostFunc(ctx->stream, StallCallback, reinterpret_cast<void*>(intptr_t(milliseconds))));
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
  return cx_stream_sync(ctx, ctx->stream);
}
void* cx_context_stream(cx_context* ctx) { return ctx ? ctx->stream : nullptr; }

int cx_device_name(cx_context* ctx, char* out, size_t n) {
  CX_CHECK_ARG(ctx != nullptr && out != nullptr && n > 0);
  std::snprintf(out, n, "%s", ctx->name);
  return CX_OK;
}

}  // extern "C"

