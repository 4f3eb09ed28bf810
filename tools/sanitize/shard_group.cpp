// Sanitizer driver of the multi-shard front's threading (ceres-solver-ceres-solver_amd/csrc/cx_shard_group.h): the very
// header cx_multi.hip builds on, driven with dummy jobs -- no HIP, no GPU -- under ThreadSanitizer and AddressSanitizer +
// UBSan (tools/sanitize/Makefile, tests/test_sanitizers.py).
//
// Scenarios (each for 2, 3, 4 and 8 shards, repeated): a normal run whose jobs go through several exchange steps and
// must see the sum of all shards' buffers; one shard failing BEFORE the first rendezvous, BETWEEN two rendezvous and
// AFTER the last one while the others are still exchanging; the shard-0 combine step failing; shards disagreeing about
// the length of a collective; a run after a failed one (the rendezvous is reusable); construction + destruction while
// the workers are idle; many short runs back to back (job hand-over).  Every scenario must return (no thread may be
// left waiting) with the error of the shard that failed first in program logic.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../ceres-solver-ceres-solver_amd/csrc/cx_shard_group.h"

static int failures = 0;
#define EXPECT(cond, ...)                              \
  do {                                                 \
    if (!(cond)) {                                     \
      ++failures;                                      \
      std::printf("FAILED %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                        \
      std::printf("\n");                               \
    }                                                  \
  } while (0)

static thread_local std::string t_error;

// what InprocAllReduce does with the device buffers, on host arrays: every buffer <- sum of all, added in rank order
static int SumCombine(double* const* bufs, int n, int64_t len) {
  for (int64_t k = 0; k < len; ++k) {
    double s = bufs[0][k];
    for (int r = 1; r < n; ++r) s += bufs[r][k];
    for (int r = 0; r < n; ++r) bufs[r][k] = s;
  }
  return 0;
}

enum FailPoint { kNever, kBeforeFirst, kBetween, kAfterLast, kCombine, kLength };

struct Scenario {
  int n;
  int steps;
  FailPoint fail;
  int failing_shard;
};

// one job: `steps` exchange steps on a private buffer; returns 0 or an error code, error text in t_error
static int Job(cx_shard_group* g, const Scenario& sc, int i, std::vector<std::vector<double>>* bufs, std::atomic<int>* on_failure_calls) {
  (void)on_failure_calls;
  std::vector<double>& v = (*bufs)[size_t(i)];
  const int64_t len = int64_t(v.size());
  if (sc.fail == kBeforeFirst && i == sc.failing_shard) {
    t_error = "failed before the first exchange";
    return -2;
  }
  for (int s = 0; s < sc.steps; ++s) {
    for (int64_t k = 0; k < len; ++k) v[size_t(k)] = double(i + 1) * double(s + 1) + double(k);
    if (sc.fail == kBetween && i == sc.failing_shard && s == sc.steps / 2) {
      t_error = "failed between two exchanges";
      return -2;
    }
    const int64_t my_len = (sc.fail == kLength && i == sc.failing_shard && s == 1) ? len - 1 : len;
    auto combine = [&](double* const* b, int n, int64_t l) -> int {
      if (sc.fail == kCombine && s == 1) return -1;
      return SumCombine(b, n, l);
    };
    if (g->exchange(i, v.data(), my_len, combine) != 0) {
      t_error = "exchange aborted";
      return cx_shard_group::kCommError;
    }
    const double want0 = double(s + 1) * double(sc.n) * double(sc.n + 1) / 2.0;  // sum over ranks of (r + 1)(s + 1), at k = 0
    if (v[0] != want0) {
      t_error = "wrong sum";
      return -3;
    }
  }
  if (sc.fail == kAfterLast && i == sc.failing_shard) {
    t_error = "failed after the last exchange";
    return -2;
  }
  return 0;
}

static void RunScenario(cx_shard_group* g, const Scenario& sc, std::atomic<int>* on_failure_calls) {
  std::vector<std::vector<double>> bufs(size_t(sc.n), std::vector<double>(37));
  int failed = -2;
  std::string message;
  const int rc = g->run([&](int i) { return Job(g, sc, i, &bufs, on_failure_calls); }, &failed, &message);
  switch (sc.fail) {
    case kNever:
      EXPECT(rc == 0 && failed == -1, "clean run: rc %d, shard %d (%s)", rc, failed, message.c_str());
      break;
    case kBeforeFirst: case kBetween: case kAfterLast:
      EXPECT(rc == -2 && failed == sc.failing_shard, "n %d fail point %d: rc %d from shard %d (%s), wanted -2 from shard %d", sc.n, int(sc.fail), rc,
             failed, message.c_str(), sc.failing_shard);
      EXPECT(message.find("failed") == 0, "the failing shard's own message must be reported, got '%s'", message.c_str());
      break;
    case kCombine: case kLength:
      EXPECT(rc == cx_shard_group::kCommError, "n %d: a failed combine / a length mismatch must come back as a communication error, rc %d", sc.n, rc);
      break;
  }
}

int main() {
  using Clock = std::chrono::steady_clock;
  const auto t0 = Clock::now();
  int scenarios = 0;
  for (int n : {1, 2, 3, 4, 8}) {
    cx_shard_group g;
    std::atomic<int> started{0}, on_failure_calls{0};
    g.on_thread_start = [&](int) { started.fetch_add(1); };
    g.last_error = [] { return t_error.c_str(); };
    g.on_failure = [&](int) { on_failure_calls.fetch_add(1); };
    g.start(n);
    for (int repeat = 0; repeat < 6; ++repeat) {
      for (FailPoint fp : {kNever, kBeforeFirst, kNever, kBetween, kAfterLast, kNever, kCombine, kNever, kLength, kNever}) {
        if (n == 1 && (fp == kCombine || fp == kLength)) continue;  // one shard: nothing to disagree with
        const Scenario sc{n, 5, fp, (repeat * 3 + int(fp)) % n};
        RunScenario(&g, sc, &on_failure_calls);
        ++scenarios;
      }
    }
    // job hand-over: many empty runs back to back
    std::atomic<long> calls{0};
    for (int k = 0; k < 2000; ++k) {
      const int rc = g.run([&](int) { calls.fetch_add(1); return 0; });
      EXPECT(rc == 0, "empty run");
    }
    EXPECT(calls.load() == 2000L * n, "every worker runs every job exactly once: %ld calls for %d shards", calls.load(), n);
    EXPECT(started.load() == n, "on_thread_start once per worker");
    EXPECT(on_failure_calls.load() > 0 || n == 0, "on_failure must have been called for the failing shards");
  }  // destructor: workers idle
  // construction + immediate destruction, with and without a start
  for (int k = 0; k < 50; ++k) {
    cx_shard_group idle;
    if (k % 2) idle.start(1 + k % 5);
  }
  // a job that blocks in the rendezvous while another shard is slow to fail: the slow failure must still release it
  {
    cx_shard_group g;
    g.last_error = [] { return t_error.c_str(); };
    g.start(4);
    std::vector<std::vector<double>> bufs(4, std::vector<double>(8, 1.0));
    const auto t1 = Clock::now();
    int failed = -1;
    const int rc = g.run([&](int i) -> int {
      if (i == 2) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        t_error = "failed late";
        return -2;
      }
      return g.exchange(i, bufs[size_t(i)].data(), 8, SumCombine) == 0 ? 0 : cx_shard_group::kCommError;
    }, &failed);
    const double waited = std::chrono::duration<double>(Clock::now() - t1).count();
    EXPECT(rc == -2 && failed == 2, "late failure: rc %d shard %d", rc, failed);
    EXPECT(waited < 5.0, "the blocked shards must be released when the late shard fails (%.3f s)", waited);
  }
  std::printf("%d scenarios, %.2f s\n", scenarios, std::chrono::duration<double>(Clock::now() - t0).count());
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
