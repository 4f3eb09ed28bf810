// The threading of a multi-shard front (cx_multi.hip): one worker thread per shard, job dispatch, and the rendezvous of
// the in-process exchange step.  Plain C++ -- no HIP, no RCCL -- so that tools/sanitize can drive it with dummy jobs
// under ThreadSanitizer and AddressSanitizer on the CPU (tests/test_sanitizers.py); everything device-specific comes in
// through three callbacks.
//
// Contract (what the sanitizer driver checks):
//   * run(fn) executes fn(shard) once on every worker thread and returns when all of them have returned; calls are
//     serialised; it returns CX_OK or the error of the shard that FAILED FIRST in program logic (a shard that was merely
//     released from a rendezvous by somebody else's failure reports CX_ERR_COMM and is not the one named);
//   * a shard whose job fails never leaves the others waiting: its worker calls abort_exchange() (in-process transport:
//     every current and future rendezvous of this run returns -1) and on_failure (RCCL transport: the owner aborts the
//     communicators, which ends the collectives the other shards' streams are stuck in);
//   * the rendezvous is reusable: after run() has returned the abort state is cleared;
//   * the destructor may run while the workers are idle, at any time after construction.
#ifndef CX_SHARD_GROUP_H_
#define CX_SHARD_GROUP_H_

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct cx_shard_group {
  static constexpr int kMaxShards = 16;
  // the two return codes this file knows (cx_multi.hip pins them to cxschur.h's CX_OK / CX_ERR_COMM)
  static constexpr int kOk = 0, kCommError = -4;
  int n = 0;
  // ---- supplied by the owner before start()
  std::function<void(int)> on_thread_start;       // e.g. hipSetDevice of the shard
  std::function<const char*()> last_error;        // the failing thread's error text (thread-local in the library)
  std::function<void(int)> on_failure;            // called on the failing worker's thread, after abort_exchange()
  // ---- job dispatch
  std::vector<std::thread> threads;
  std::mutex run_mutex;  // one front call at a time
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  uint64_t gen = 0;
  int pending = 0;
  bool stop = false;
  const std::function<int(int)>* job = nullptr;
  std::vector<int> rc;
  std::vector<std::string> err;
  // ---- rendezvous of the in-process exchange step: every shard deposits (pointer, length), all meet, shard 0 acts
  // (sums the buffers), all meet again
  std::mutex bm;
  std::condition_variable bcv;
  int arrived = 0;
  uint64_t bgen = 0;
  bool aborted = false;
  double* ptrs[kMaxShards] = {};
  double* ptrs2[kMaxShards] = {};  // a second buffer per shard (reduce-scatter: where the shard's own range of the sum goes)
  int64_t lens[kMaxShards] = {};

  void start(int num_shards) {
    n = num_shards;
    rc.assign(size_t(n), kOk);
    err.assign(size_t(n), std::string());
    for (int i = 0; i < n; ++i) threads.emplace_back([this, i] { worker(i); });
  }

  // returns 0 when all n shards have arrived, -1 when the exchange was aborted (before or while waiting)
  int barrier() {
    std::unique_lock<std::mutex> lk(bm);
    if (aborted) return -1;
    const uint64_t g0 = bgen;
    if (++arrived == n) {
      arrived = 0;
      ++bgen;
      bcv.notify_all();
      return 0;
    }
    bcv.wait(lk, [&] { return bgen != g0 || aborted; });
    return (bgen != g0) ? 0 : -1;
  }
  void abort_exchange() {
    std::lock_guard<std::mutex> lk(bm);
    aborted = true;
    bcv.notify_all();
  }

  // The in-process all-reduce as a protocol: deposit, meet, shard 0 runs `combine` over the deposited buffers, meet.
  // combine returns 0 on success; on failure the exchange is aborted for everybody.
  int exchange(int rank, double* p, int64_t len, const std::function<int(double* const*, int, int64_t)>& combine) {
    return exchange2(rank, p, nullptr, len, [&](double* const* a, double* const*, int n_, int64_t l) { return combine(a, n_, l); });
  }
  // the same with two buffers per shard (combine sees both arrays)
  int exchange2(int rank, double* p, double* q, int64_t len,
                const std::function<int(double* const*, double* const*, int, int64_t)>& combine) {
    {
      std::lock_guard<std::mutex> lk(bm);
      ptrs[rank] = p;
      ptrs2[rank] = q;
      lens[rank] = len;
    }
    if (barrier() != 0) return -1;
    int ok = 0;
    if (rank == 0) {
      {
        std::lock_guard<std::mutex> lk(bm);
        for (int r = 1; r < n; ++r)
          if (lens[r] != len) ok = -1;  // the shards disagree about the collective: a bug, not a transient
      }
      if (ok == 0) ok = combine(ptrs, ptrs2, n, len);
      if (ok != 0) abort_exchange();
    }
    if (barrier() != 0) return -1;
    return ok;
  }

  void worker(int i) {
    if (on_thread_start) on_thread_start(i);
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_start.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
        fn = job;
      }
      const int r = (*fn)(i);
      std::string text;
      if (r != kOk) {
        if (last_error) text = last_error();
        // a shard that fails will not reach the next exchange step: release the others instead of letting them wait
        abort_exchange();
        if (on_failure) on_failure(i);
      }
      {
        std::lock_guard<std::mutex> lk(m);
        rc[size_t(i)] = r;
        if (r != kOk) err[size_t(i)] = text;
        if (--pending == 0) cv_done.notify_one();
      }
    }
  }

  // *first_failed (may be NULL): the shard whose error is reported, -1 when none
  int run(const std::function<int(int)>& fn, int* first_failed = nullptr, std::string* message = nullptr) {
    std::lock_guard<std::mutex> serial(run_mutex);
    {
      std::unique_lock<std::mutex> lk(m);
      job = &fn;
      pending = n;
      ++gen;
      cv_start.notify_all();
      cv_done.wait(lk, [&] { return pending == 0; });
      job = nullptr;
    }
    {
      std::lock_guard<std::mutex> lk(bm);
      aborted = false;
      arrived = 0;
    }
    // report the failure that started it, not the CX_ERR_COMM of the shards that were released from the rendezvous
    int first = kOk, first_i = -1;
    {
      std::lock_guard<std::mutex> lk(m);
      for (int i = 0; i < n; ++i) {
        if (rc[size_t(i)] == kOk) continue;
        if (first == kOk || (first == kCommError && rc[size_t(i)] != kCommError)) { first = rc[size_t(i)]; first_i = i; }
      }
      if (first_i >= 0 && message) *message = err[size_t(first_i)];
    }
    if (first_failed) *first_failed = first_i;
    return first;
  }

  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      cv_start.notify_all();
    }
    for (auto& t : threads)
      if (t.joinable()) t.join();
    threads.clear();
  }

  ~cx_shard_group() { shutdown(); }
};

#endif
