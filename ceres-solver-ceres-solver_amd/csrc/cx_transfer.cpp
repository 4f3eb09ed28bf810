// Host vectors at the boundary: how the arrays a LinearSolver::Solve / Evaluator::Evaluate caller passes in HOST memory
// (linear_solver.h:363-390, evaluator.h:120-150) reach HBM and come back.
//
// TrustRegionMinimizer allocates its vectors once (trust_region_minimizer.cc:181-203: x_, candidate_x_, gradient_,
// residuals_, model_residuals_, trust_region_step_, delta_, jacobian_scaling_) and LevenbergMarquardtStrategy its
// diagonal_ / lm_diagonal_ once (levenberg_marquardt_strategy.cc:77-99), then hands the same arrays to every call of an
// LM iteration.  This file keeps a grow-only pool of device staging buffers per context (no hipMalloc / hipFree per
// call), per-context counters of what crossed PCIe, and -- OPT-IN -- a process-wide registry of caller arrays registered
// with the HIP runtime (hipHostRegister: the pages are pinned and mapped once, later copies are plain DMA that do not
// block the calling thread).
//
// Why opt-in (measured, round 4, Final-13682 vectors on the MI355X boxes of this pool): pageable copies of arrays this
// size already run at the PCIe rate (55 GB/s either way; the runtime pins the pages of a large copy in place), so
// registration saves host time only (about 3 ms of a 31 ms iteration's boundary calls).  And a registration cannot see
// its array die: a registered range that the owner frees (munmap) and whose addresses come back with the next
// allocation still looks registered to the runtime, and the DMA into it faults the GPU.  So nothing is registered unless
// the caller vouches for the lifetime of what it hands in (cx_host_registration_policy / cx_host_register) and calls
// cx_host_registrations_release before those arrays are freed.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>

#include "cx_internal.h"

// ------------------------------------------------------------------ registry
namespace {

struct PinRegistry {
  std::mutex m;
  struct Entry {
    char* base;
    size_t bytes;
    uint64_t stamp;
  };
  std::vector<Entry> entries;
  struct Sighting {
    const void* p;
    size_t bytes;
    int count;
  };
  std::vector<Sighting> sightings;  // arrays seen but not (yet) registered, most recent last
  std::vector<std::pair<const void*, size_t>> refused;  // hipHostRegister said no: do not ask again
  uint64_t clock = 0;
  size_t total_bytes = 0;
  // policy (cx_host_registration_policy / CX_PIN_* in the environment)
  int sightings_needed = 0;           // register an array the n-th time it is handed in; 0 = never (the default)
  size_t min_bytes = size_t(256) << 10;
  size_t max_total_bytes = size_t(16) << 30;
  size_t max_entries = 64;
  bool env_read = false;
  // statistics
  double register_ms = 0.0;
  int64_t num_register_calls = 0, num_evictions = 0;

  void read_env() {
    if (env_read) return;
    env_read = true;
    if (const char* e = std::getenv("CX_PIN")) sightings_needed = std::atoi(e);  // 0 off, 1 first sight, 2 second sight
    if (const char* e = std::getenv("CX_PIN_MIN_KB")) min_bytes = size_t(std::atoll(e)) << 10;
    if (const char* e = std::getenv("CX_PIN_MAX_MB")) max_total_bytes = size_t(std::atoll(e)) << 20;
  }

  Entry* find(const char* p, size_t bytes) {
    for (auto& e : entries)
      if (p >= e.base && p + bytes <= e.base + e.bytes) return &e;
    return nullptr;
  }

  void unregister_at(size_t i) {
    (void)hipHostUnregister(entries[i].base);
    total_bytes -= entries[i].bytes;
    entries.erase(entries.begin() + long(i));
  }

  // drop every registration that overlaps [p, p + bytes): the caller's allocator has reused those addresses
  void drop_overlapping(const char* p, size_t bytes) {
    for (size_t i = entries.size(); i-- > 0;)
      if (entries[i].base < p + bytes && p < entries[i].base + entries[i].bytes) unregister_at(i);
  }

  void release_all() {
    // copies from / to these arrays were enqueued by calls that have returned, i.e. they are complete (every entry
    // point waits for its own copies), so nothing is in flight on them
    while (!entries.empty()) unregister_at(entries.size() - 1);
    sightings.clear();
    refused.clear();
  }

  bool pin(const void* vp, size_t bytes) {
    read_env();
    if (vp == nullptr || bytes == 0) return false;
    const char* p = static_cast<const char*>(vp);
    if (Entry* e = find(p, bytes)) {
      e->stamp = ++clock;
      return true;
    }
    // A range that overlaps a registration without lying inside it: the registered array is gone and its addresses
    // belong to something else now.  The runtime refuses a copy whose host range is only partly registered, so the
    // stale registration goes before anything else happens -- whatever the policy says about registering the new one.
    drop_overlapping(p, bytes);
    if (sightings_needed <= 0 || bytes < min_bytes) return false;
    for (const auto& r : refused)
      if (r.first == vp && r.second == bytes) return false;
    if (bytes > max_total_bytes) return false;
    if (sightings_needed > 1) {
      auto it = std::find_if(sightings.begin(), sightings.end(), [&](const Sighting& s) { return s.p == vp && s.bytes == bytes; });
      if (it == sightings.end()) {
        if (sightings.size() >= 256) sightings.erase(sightings.begin());
        sightings.push_back({vp, bytes, 1});
        return false;
      }
      if (++it->count < sightings_needed) return false;
      sightings.erase(it);
    }
    // room: least recently used registrations go first
    // (never one used within the last few calls: a copy of another shard's thread may still be running on it)
    while (!entries.empty() && (total_bytes + bytes > max_total_bytes || entries.size() >= max_entries)) {
      size_t lru = 0;
      for (size_t i = 1; i < entries.size(); ++i)
        if (entries[i].stamp < entries[lru].stamp) lru = i;
      if (clock - entries[lru].stamp < 64) return false;
      unregister_at(lru);
      ++num_evictions;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t rc = hipHostRegister(const_cast<char*>(p), bytes, hipHostRegisterDefault);
    if (rc == hipErrorHostMemoryAlreadyRegistered) {
      // a stale registration of ours (the array it covered was freed and the addresses came back in another shape)
      (void)hipGetLastError();
      drop_overlapping(p, bytes);
      rc = hipHostRegister(const_cast<char*>(p), bytes, hipHostRegisterDefault);
    }
    register_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++num_register_calls;
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      if (refused.size() >= 64) refused.erase(refused.begin());
      refused.push_back({vp, bytes});
      return false;
    }
    entries.push_back({const_cast<char*>(p), bytes, ++clock});
    total_bytes += bytes;
    return true;
  }
};

PinRegistry& Registry() {
  static PinRegistry* r = new PinRegistry;  // never destroyed: contexts may outlive static destruction order
  return *r;
}

}  // namespace

bool cx_pin_range(const void* p, size_t bytes) {
  PinRegistry& r = Registry();
  std::lock_guard<std::mutex> lk(r.m);
  return r.pin(p, bytes);
}

// ------------------------------------------------------------------ per-context state
struct cx_xfer_state {
  std::mutex m;
  hipStream_t copy_stream = nullptr;
  // counters since the last reset
  int64_t bytes[2] = {0, 0}, copies[2] = {0, 0}, pinned_bytes[2] = {0, 0};  // [0] H2D, [1] D2H
  double ms[2] = {0.0, 0.0};
  struct Timed { hipEvent_t a, b; int dir; };
  std::vector<Timed> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t h2d_done = nullptr;
  bool h2d_outstanding = false;
  std::vector<DevBuf<double>*> stage_free, stage_all;

  int event(hipEvent_t* out) {
    if (!free_events.empty()) {
      *out = free_events.back();
      free_events.pop_back();
      return CX_OK;
    }
    CX_HIP(hipEventCreate(out));
    return CX_OK;
  }
  // device-side durations of the recorded copies (waits for them)
  void collect() {
    for (auto& t : pending) {
      float f = 0.f;
      if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&f, t.a, t.b) == hipSuccess) ms[t.dir] += f;
      free_events.push_back(t.a);
      free_events.push_back(t.b);
    }
    pending.clear();
  }
};

static cx_xfer_state* Xfer(cx_context* ctx) {
  if (!ctx->xfer) ctx->xfer = new cx_xfer_state;
  return ctx->xfer;
}

void cx_xfer_destroy(cx_context* ctx) {
  cx_xfer_state* x = ctx->xfer;
  if (!x) return;
  x->collect();
  for (auto ev : x->free_events) (void)hipEventDestroy(ev);
  if (x->h2d_done) (void)hipEventDestroy(x->h2d_done);
  if (x->copy_stream) (void)hipStreamDestroy(x->copy_stream);
  for (auto* b : x->stage_all) delete b;
  delete x;
  ctx->xfer = nullptr;
}

hipStream_t cx_copy_stream(cx_context* ctx) {
  cx_xfer_state* x = Xfer(ctx);
  if (!x->copy_stream && hipStreamCreateWithFlags(&x->copy_stream, hipStreamNonBlocking) != hipSuccess) return ctx->stream;
  return x->copy_stream;
}

static int Copy(cx_context* ctx, void* dst, const void* src, size_t bytes, int dir, hipStream_t st) {
  if (bytes == 0) return CX_OK;
  cx_xfer_state* x = Xfer(ctx);
  if (!st) st = ctx->stream;
  const bool pinned = cx_pin_range(dir == 0 ? src : dst, bytes);
  std::lock_guard<std::mutex> lk(x->m);
  if (x->pending.size() >= 512) x->collect();
  cx_xfer_state::Timed t{nullptr, nullptr, dir};
  CX_TRY(x->event(&t.a));
  CX_TRY(x->event(&t.b));
  CX_HIP(hipEventRecord(t.a, st));
  CX_HIP(hipMemcpyAsync(dst, src, bytes, dir == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
  CX_HIP(hipEventRecord(t.b, st));
  x->pending.push_back(t);
  x->bytes[dir] += int64_t(bytes);
  x->copies[dir] += 1;
  if (pinned) x->pinned_bytes[dir] += int64_t(bytes);
  if (dir == 0 && pinned) {
    // the source is registered memory: the runtime reads it when the DMA runs, not before hipMemcpyAsync returns
    if (!x->h2d_done) CX_HIP(hipEventCreateWithFlags(&x->h2d_done, hipEventDisableTiming));
    CX_HIP(hipEventRecord(x->h2d_done, st));
    x->h2d_outstanding = true;
  }
  return CX_OK;
}

int cx_copy_h2d(cx_context* ctx, void* dst, const void* src, size_t bytes, hipStream_t st) { return Copy(ctx, dst, src, bytes, 0, st); }
int cx_copy_d2h(cx_context* ctx, void* dst, const void* src, size_t bytes, hipStream_t st) { return Copy(ctx, dst, src, bytes, 1, st); }

int cx_xfer_wait_h2d(cx_context* ctx) {
  cx_xfer_state* x = ctx->xfer;
  if (!x || !x->h2d_outstanding) return CX_OK;
  x->h2d_outstanding = false;
  CX_HIP(hipEventSynchronize(x->h2d_done));
  return CX_OK;
}

int cx_vector_in(cx_context* ctx, double* dst, const double* u, size_t count, int32_t memspace) {
  if (count == 0) return CX_OK;
  if (memspace == CX_HOST) return cx_copy_h2d(ctx, dst, u, count * sizeof(double));
  if (memspace == CX_HOST_SLICES) {
    const auto* s = reinterpret_cast<const cx_host_slices*>(u);
    CX_CHECK_ARG(size_t(s->nhead + s->ntail) == count);
    CX_TRY(cx_copy_h2d(ctx, dst, s->head, size_t(s->nhead) * sizeof(double)));
    return cx_copy_h2d(ctx, dst + s->nhead, s->tail, size_t(s->ntail) * sizeof(double));
  }
  CX_HIP(hipMemcpyAsync(dst, u, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return CX_OK;
}

int cx_vector_out(cx_context* ctx, double* u, const double* src, size_t count, int32_t memspace) {
  if (count == 0) return CX_OK;
  if (memspace == CX_HOST) return cx_copy_d2h(ctx, u, src, count * sizeof(double));
  if (memspace == CX_HOST_SLICES) {
    const auto* s = reinterpret_cast<const cx_host_slices*>(u);
    CX_CHECK_ARG(size_t(s->nhead + s->ntail) == count);
    CX_TRY(cx_copy_d2h(ctx, s->head, src, size_t(s->nhead) * sizeof(double)));
    if (s->tail && !s->skip_tail_out) CX_TRY(cx_copy_d2h(ctx, s->tail, src + s->nhead, size_t(s->ntail) * sizeof(double)));
    return CX_OK;
  }
  CX_HIP(hipMemcpyAsync(u, src, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return CX_OK;
}

DevBuf<double>* cx_stage_acquire(cx_context* ctx, size_t count) {
  cx_xfer_state* x = Xfer(ctx);
  std::lock_guard<std::mutex> lk(x->m);
  // best fit among the free buffers; none fits: the largest free one grows (its old storage goes back to the device
  // allocator once, not per call), or a new buffer joins the pool
  int best = -1, largest = -1;
  for (int i = 0; i < int(x->stage_free.size()); ++i) {
    const size_t n = x->stage_free[size_t(i)]->n;
    if (n >= count && (best < 0 || n < x->stage_free[size_t(best)]->n)) best = i;
    if (largest < 0 || n > x->stage_free[size_t(largest)]->n) largest = i;
  }
  const int pick = best >= 0 ? best : largest;
  DevBuf<double>* b = nullptr;
  if (pick >= 0) {
    b = x->stage_free[size_t(pick)];
    x->stage_free.erase(x->stage_free.begin() + pick);
  } else {
    b = new DevBuf<double>;
    x->stage_all.push_back(b);
  }
  if (b->alloc(count) != CX_OK) {
    x->stage_free.push_back(b);
    return nullptr;
  }
  return b;
}

void cx_stage_release(cx_context* ctx, DevBuf<double>* b) {
  if (!b || !ctx->xfer) return;
  std::lock_guard<std::mutex> lk(ctx->xfer->m);
  ctx->xfer->stage_free.push_back(b);
}

// ------------------------------------------------------------------ HostOrDevice
HostOrDevice::~HostOrDevice() {
  if (tmp) {
    // the staging buffer goes back to the pool: later users are ordered behind this call's work on the same stream;
    // the caller's array must have been read before the call returns
    (void)cx_xfer_wait_h2d(ctx);
    cx_stage_release(ctx, tmp);
  }
}

int HostOrDevice::bind(double* u, size_t count, int memspace) {
  n = count;
  is_host = cx_is_host_space(memspace);
  sliced = memspace == CX_HOST_SLICES;
  user = u;
  if (!u) { dptr = nullptr; return CX_OK; }
  if (!is_host) { dptr = u; return CX_OK; }
  if (sliced) {
    slices = *reinterpret_cast<const cx_host_slices*>(u);
    if (size_t(slices.nhead + slices.ntail) != count) {
      cx_set_error("internal: host slices of %lld + %lld entries for a vector of %zu", (long long)slices.nhead, (long long)slices.ntail, count);
      return CX_ERR_INVALID_ARGUMENT;
    }
  }
  tmp = cx_stage_acquire(ctx, count);
  if (!tmp) return CX_ERR_OUT_OF_MEMORY;
  dptr = tmp->p;
  return CX_OK;
}

int HostOrDevice::copy_in() {
  if (!sliced) return cx_copy_h2d(ctx, dptr, user, n * sizeof(double));
  CX_TRY(cx_copy_h2d(ctx, dptr, slices.head, size_t(slices.nhead) * sizeof(double)));
  return cx_copy_h2d(ctx, dptr + slices.nhead, slices.tail, size_t(slices.ntail) * sizeof(double));
}

int HostOrDevice::in(const double* u, size_t count, int memspace) {
  CX_TRY(bind(const_cast<double*>(u), count, memspace));
  if (!u || !is_host) return CX_OK;
  return copy_in();
}

int HostOrDevice::inout(double* u, size_t count, int memspace, bool with_copy_in) {
  CX_TRY(bind(u, count, memspace));
  if (!u || !is_host || !with_copy_in) return CX_OK;
  return copy_in();
}

int HostOrDevice::out_async(hipStream_t st) {
  if (!user || !is_host) return CX_OK;
  if (!sliced) return cx_copy_d2h(ctx, user, dptr, n * sizeof(double), st);
  CX_TRY(cx_copy_d2h(ctx, slices.head, dptr, size_t(slices.nhead) * sizeof(double), st));
  if (slices.tail && !slices.skip_tail_out) CX_TRY(cx_copy_d2h(ctx, slices.tail, dptr + slices.nhead, size_t(slices.ntail) * sizeof(double), st));
  return CX_OK;
}

int HostOrDevice::out() {
  if (!user || !is_host) return CX_OK;
  // (a copy into pageable memory blocks inside the runtime until the stream has drained: on a sharded context the wait
  // that cannot hang on a lost rank comes first)
  if (ctx->comm && ctx->nranks > 1) CX_TRY(cx_stream_sync(ctx, ctx->stream));
  CX_TRY(out_async(ctx->stream));
  return cx_stream_sync(ctx, ctx->stream);
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int cx_host_registration_policy(int32_t sightings, int64_t min_bytes, int64_t max_total_bytes) {
  CX_CHECK_ARG(sightings >= 0 && min_bytes >= 0 && max_total_bytes >= 0);
  PinRegistry& r = Registry();
  std::lock_guard<std::mutex> lk(r.m);
  r.read_env();  // an explicit call wins over the environment
  r.sightings_needed = sightings;
  r.min_bytes = size_t(min_bytes);
  r.max_total_bytes = size_t(max_total_bytes);
  if (sightings == 0) r.release_all();
  return CX_OK;
}

int cx_host_register(const void* p, size_t bytes) {
  CX_CHECK_ARG(p != nullptr);
  PinRegistry& r = Registry();
  std::lock_guard<std::mutex> lk(r.m);
  r.read_env();
  const int keep = r.sightings_needed;
  const size_t keep_min = r.min_bytes;
  r.sightings_needed = 1;
  r.min_bytes = 1;
  const bool ok = r.pin(p, bytes);
  r.sightings_needed = keep;
  r.min_bytes = keep_min;
  if (!ok) {
    cx_set_error("hipHostRegister refused %zu bytes at %p", bytes, p);
    return CX_ERR_HIP;
  }
  return CX_OK;
}

int cx_host_registrations_release(void) {
  PinRegistry& r = Registry();
  std::lock_guard<std::mutex> lk(r.m);
  r.release_all();
  return CX_OK;
}

int cx_transfer_stats_get(cx_context* ctx, cx_transfer_stats* out, int32_t reset) {
  CX_CHECK_ARG(ctx != nullptr && out != nullptr);
  std::memset(out, 0, sizeof(*out));
  // a front reports its shards: bytes and copies summed, durations of the slowest shard (they copy side by side)
  std::vector<cx_context*> list;
  if (ctx->shards.empty()) list.push_back(ctx);
  else list.assign(ctx->shards.begin(), ctx->shards.end());
  if (!ctx->shards.empty()) list.push_back(ctx);
  for (cx_context* c : list) {
    cx_xfer_state* x = c->xfer;
    if (!x) continue;
    std::lock_guard<std::mutex> lk(x->m);
    (void)hipSetDevice(c->device);
    x->collect();
    out->h2d_bytes += x->bytes[0];
    out->d2h_bytes += x->bytes[1];
    out->h2d_copies += x->copies[0];
    out->d2h_copies += x->copies[1];
    out->h2d_registered_bytes += x->pinned_bytes[0];
    out->d2h_registered_bytes += x->pinned_bytes[1];
    out->h2d_ms = std::max(out->h2d_ms, x->ms[0]);
    out->d2h_ms = std::max(out->d2h_ms, x->ms[1]);
    if (reset) {
      x->bytes[0] = x->bytes[1] = x->copies[0] = x->copies[1] = x->pinned_bytes[0] = x->pinned_bytes[1] = 0;
      x->ms[0] = x->ms[1] = 0.0;
    }
  }
  (void)hipSetDevice(ctx->device);
  PinRegistry& r = Registry();
  std::lock_guard<std::mutex> lk(r.m);
  out->num_registered = int32_t(r.entries.size());
  out->registered_bytes = int64_t(r.total_bytes);
  out->register_ms = r.register_ms;
  out->num_register_calls = r.num_register_calls;
  if (reset) {
    r.register_ms = 0.0;
    r.num_register_calls = 0;
  }
  return CX_OK;
}

}  // extern "C"
