// Drives every host-only entry point of libcxschur (the threaded structure planners) on two synthetic scenes -- a camera
// ring and the same ring with long-range observations and loop closures -- under AddressSanitizer + UBSan and under
// ThreadSanitizer (tools/sanitize/Makefile).  No device is touched.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "../../include/cxschur.h"

static int failures = 0;
#define EXPECT(cond, ...)                         \
  do {                                            \
    if (!(cond)) {                                \
      ++failures;                                 \
      std::printf("FAILED %s: ", #cond);          \
      std::printf(__VA_ARGS__);                   \
      std::printf("\n");                          \
    }                                             \
  } while (0)

struct Scene {
  int32_t C, P;
  std::vector<int32_t> cam, pt;  // observations, point-major
  std::vector<cx_block> rows, cols;
  std::vector<int32_t> rcb;
  std::vector<cx_cell> cells;
  cx_block_structure bs{};
};

// points seen from a window of neighbouring cameras of a ring; a fraction of them from a second, far window as well
static Scene MakeScene(int32_t C, int32_t P, double mean_track, double long_range_fraction, unsigned seed) {
  Scene s;
  s.C = C;
  s.P = P;
  std::mt19937 prng(seed);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::geometric_distribution<int> extra(1.0 / (mean_track - 1.0));
  for (int32_t j = 0; j < P; ++j) {
    const int k = std::min(C, 2 + extra(prng));
    const int home = int((int64_t(j) * C) / P + int(uni(prng) * (C / 8 + 1))) % C;
    std::vector<int32_t> seen;
    for (int i = 0; i < k; ++i) seen.push_back((home + i) % C);
    if (uni(prng) < long_range_fraction) {
      const int far = (home + C / 3 + int(uni(prng) * (C / 3))) % C;
      for (int i = 0; i < 2; ++i) seen.push_back((far + i) % C);
    }
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    for (int32_t c : seen) {
      s.cam.push_back(c);
      s.pt.push_back(j);
    }
  }
  const int64_t O = int64_t(s.cam.size());
  s.rows.resize(size_t(O));
  s.rcb.resize(size_t(O) + 1);
  s.cells.resize(size_t(2 * O));
  s.cols.resize(size_t(P) + size_t(C));
  for (int32_t j = 0; j < P; ++j) s.cols[size_t(j)] = cx_block{3, 3 * j};
  for (int32_t i = 0; i < C; ++i) s.cols[size_t(P + i)] = cx_block{9, 3 * P + 9 * i};
  for (int64_t r = 0; r < O; ++r) {
    s.rows[size_t(r)] = cx_block{2, int32_t(2 * r)};
    s.rcb[size_t(r)] = int32_t(2 * r);
    s.cells[size_t(2 * r)] = cx_cell{s.pt[size_t(r)], int32_t(6 * r)};
    s.cells[size_t(2 * r + 1)] = cx_cell{P + s.cam[size_t(r)], int32_t(6 * O + 18 * r)};
  }
  s.rcb[size_t(O)] = int32_t(2 * O);
  s.bs = cx_block_structure{int32_t(O), P + C, s.rows.data(), s.cols.data(), s.rcb.data(), s.cells.data()};
  return s;
}

static void Exercise(const char* name, const Scene& s) {
  const int32_t C = s.C, P = s.P;
  const int64_t O = int64_t(s.cam.size());
  int32_t r = 0, e = 0, f = 0;
  EXPECT(cx_detect_structure(&s.bs, P, &r, &e, &f) == CX_OK && r == 2 && e == 3 && f == 9, "detect %d %d %d", r, e, f);
  for (int n : {1, 2, 3, 8}) {
    std::vector<int32_t> bounds(size_t(n) + 1, -1);
    EXPECT(cx_partition_points(&s.bs, P, n, bounds.data()) == CX_OK && bounds[0] == 0 && bounds[size_t(n)] == P &&
               std::is_sorted(bounds.begin(), bounds.end()), "partition into %d", n);
  }
  {
    std::vector<int32_t> ordering(size_t(C) + size_t(P));
    int32_t independent = 0;
    EXPECT(cx_stable_schur_ordering(C, P, O, s.cam.data(), s.pt.data(), ordering.data(), &independent) == CX_OK && independent >= P / 2,
           "stable Schur ordering: %d independent", independent);
    std::vector<int32_t> sorted = ordering;
    std::sort(sorted.begin(), sorted.end());
    bool perm = true;
    for (size_t i = 0; i < sorted.size(); ++i) perm = perm && sorted[i] == int32_t(i);
    EXPECT(perm, "ordering is a permutation");
  }
  int64_t num_cells = 0, num_pairs = 0, num_items = 0;
  EXPECT(cx_schur_pair_lists_host(&s.bs, P, &num_cells, &num_pairs, &num_items, nullptr, nullptr, 0, nullptr, 0) == CX_OK, "%s", cx_last_error());
  std::vector<int32_t> cell_row(size_t(std::max<int64_t>(num_cells, 1))), cell_col(cell_row.size()), pair_rows(size_t(std::max<int64_t>(2 * num_pairs, 1)));
  EXPECT(cx_schur_pair_lists_host(&s.bs, P, &num_cells, &num_pairs, &num_items, cell_row.data(), cell_col.data(), num_cells, pair_rows.data(),
                                  2 * num_pairs) == CX_OK && num_cells >= C, "%s", cx_last_error());
  for (int64_t q = 0; q < 2 * num_pairs; ++q) EXPECT(pair_rows[size_t(q)] >= 0 && pair_rows[size_t(q)] < O, "pair row out of range");
  int32_t T = 0, levels = 0;
  int64_t tiles = 0, updates = 0;
  std::vector<int32_t> first(static_cast<size_t>(C));
  EXPECT(cx_sparse_cholesky_plan_host(C, cell_row.data(), cell_col.data(), num_cells, first.data(), &T, &levels, &tiles, &updates, nullptr, nullptr, 0,
                                      nullptr, 0) == CX_OK && T > 0 && levels > 0, "%s", cx_last_error());
  std::vector<int32_t> level(static_cast<size_t>(T)), start(size_t(T) + 1), tcols(static_cast<size_t>(tiles));
  EXPECT(cx_sparse_cholesky_plan_host(C, cell_row.data(), cell_col.data(), num_cells, first.data(), &T, &levels, &tiles, &updates, level.data(),
                                      start.data(), T, tcols.data(), tiles) == CX_OK && start[size_t(T)] == tiles, "%s", cx_last_error());
  // the update schedule under a window (single plan and rank 1 of 3), with the "wide levels only" rule switched off so that
  // the windowing code runs on these small scenes
  setenv("CX_SPARSE_WINDOW_MIN_PRODUCTS", "0", 1);
  for (int nranks : {1, 3}) {
    int64_t products = 0, chains = 0, violations = -1;
    int32_t longest = 0;
    EXPECT(cx_sparse_cholesky_schedule_host(C, cell_row.data(), cell_col.data(), num_cells, nranks, nranks - 1, 4, &products, &chains, &longest,
                                            &violations) == T && violations == 0 && (nranks > 1 || products > 0) && chains <= products,
           "update schedule (%d ranks): %lld violations, %s", nranks, (long long)violations, cx_last_error());
  }
  unsetenv("CX_SPARSE_WINDOW_MIN_PRODUCTS");
  std::printf("%-22s %d cameras %d points %lld observations: S %lld cells, %lld pairs, %lld items; tile-sparse plan %d tile rows, %d levels, %lld tiles, %lld updates\n",
              name, C, P, (long long)O, (long long)num_cells, (long long)num_pairs, (long long)num_items, T, levels, (long long)tiles, (long long)updates);
  for (int pre : {CX_CLUSTER_JACOBI, CX_CLUSTER_TRIDIAGONAL})
    for (int clustering : {CX_CANONICAL_VIEWS, CX_SINGLE_LINKAGE}) {
      std::vector<int32_t> membership(static_cast<size_t>(C)), p1(size_t(4 * C)), p2(size_t(4 * C));
      int32_t K = 0, np = 0;
      EXPECT(cx_visibility_clusters_host(&s.bs, P, pre, clustering, membership.data(), &K, &np, p1.data(), p2.data(), int32_t(p1.size())) == CX_OK &&
                 K >= 1 && K <= C && np >= K, "visibility clusters (%d, %d): %s", pre, clustering, cx_last_error());
      for (int32_t m : membership) EXPECT(m >= 0 && m < K, "membership out of range");
    }
}

int main() {
  Exercise("ring", MakeScene(600, 20000, 6.0, 0.0, 1u));
  Exercise("ring + long range", MakeScene(600, 20000, 6.0, 0.05, 2u));
  Exercise("small", MakeScene(9, 60, 3.0, 0.2, 3u));
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
