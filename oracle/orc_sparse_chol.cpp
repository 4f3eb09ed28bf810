// ORACLE -- TEST INFRASTRUCTURE ONLY.  See orc_sparse_chol.h.
#include "orc_sparse_chol.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace orc {

// ------------------------------------------------------------------------------------------ ordering
namespace {

struct Dissector {
  int n;
  const std::vector<int>& xadj;
  const std::vector<int>& adj;
  std::vector<int> region;  // region id of every vertex still to be ordered; -1 once ordered
  std::vector<int> level, queue, out;
  int next_region = 1;

  Dissector(int n_, const std::vector<int>& xa, const std::vector<int>& a)
      : n(n_), xadj(xa), adj(a), region(size_t(n_), 0), level(size_t(n_), -1) { out.reserve(size_t(n_)); }

  // BFS inside region `reg` from `start`; fills queue (visit order) and level; returns the number of levels
  int Bfs(int start, int reg) {
    queue.clear();
    queue.push_back(start);
    level[size_t(start)] = 0;
    int depth = 0;
    for (size_t head = 0; head < queue.size(); ++head) {
      const int v = queue[head];
      depth = level[size_t(v)];
      for (int e = xadj[size_t(v)]; e < xadj[size_t(v) + 1]; ++e) {
        const int w = adj[size_t(e)];
        if (region[size_t(w)] == reg && level[size_t(w)] < 0) {
          level[size_t(w)] = depth + 1;
          queue.push_back(w);
        }
      }
    }
    return depth + 1;
  }
  void ClearLevels() { for (int v : queue) level[size_t(v)] = -1; }

  void Emit(const std::vector<int>& vs) {
    for (int v : vs) { out.push_back(v); region[size_t(v)] = -1; }
  }

  // vertices: all of region `reg`
  void Order(std::vector<int> vertices, int reg) {
    // components first
    size_t done = 0;
    std::vector<int> comp;
    while (done < vertices.size()) {
      // next vertex of the region not yet put into a component (components get fresh region ids)
      int start = -1;
      for (; done < vertices.size(); ++done)
        if (region[size_t(vertices[done])] == reg) { start = vertices[done]; break; }
      if (start < 0) break;
      Bfs(start, reg);
      comp = queue;
      ClearLevels();
      const int creg = next_region++;
      for (int v : comp) region[size_t(v)] = creg;
      OrderConnected(comp, creg);
    }
  }

  void OrderConnected(std::vector<int>& vertices, int reg) {
    if (vertices.size() <= 32) { Emit(vertices); return; }
    // pseudo-peripheral start: repeat the BFS from a minimum-degree vertex of the last level while it gets deeper
    int nlevels = Bfs(vertices[0], reg);
    for (int pass = 0; pass < 3; ++pass) {
      int best = queue.back();
      for (size_t i = queue.size(); i-- > 0 && level[size_t(queue[i])] == nlevels - 1;) {
        const int v = queue[i];
        if (xadj[size_t(v) + 1] - xadj[size_t(v)] < xadj[size_t(best) + 1] - xadj[size_t(best)]) best = v;
      }
      ClearLevels();
      const int got = Bfs(best, reg);
      const bool deeper = got > nlevels;
      nlevels = got;
      if (!deeper) break;
    }
    if (nlevels < 3) {  // no vertex separator from a level structure: close to a clique
      std::vector<int> all = queue;
      ClearLevels();
      Emit(all);
      return;
    }
    std::vector<int> count(size_t(nlevels), 0);
    for (int v : queue) count[size_t(level[size_t(v)])]++;
    // separator: the smallest level among those whose removal leaves both sides with >= 1/3 of the vertices,
    // else the level that balances best
    const int total = int(queue.size());
    int best = -1, best_size = total + 1, below = 0, balanced = 1, balanced_gap = total;
    for (int l = 0; l < nlevels; ++l) {
      const int above = total - below - count[size_t(l)];
      if (l > 0 && l < nlevels - 1) {
        if (3 * below >= total && 3 * above >= total && count[size_t(l)] < best_size) { best = l; best_size = count[size_t(l)]; }
        if (std::abs(below - above) < balanced_gap) { balanced_gap = std::abs(below - above); balanced = l; }
      }
      below += count[size_t(l)];
    }
    if (best < 0) best = balanced;
    std::vector<int> lo, hi, sep;
    for (int v : queue) {
      const int l = level[size_t(v)];
      (l < best ? lo : l > best ? hi : sep).push_back(v);
    }
    ClearLevels();
    const int rlo = next_region++, rhi = next_region++;
    for (int v : lo) region[size_t(v)] = rlo;
    for (int v : hi) region[size_t(v)] = rhi;
    for (int v : sep) region[size_t(v)] = -2;  // neither side sees it
    Order(std::move(lo), rlo);
    Order(std::move(hi), rhi);
    Emit(sep);
  }
};

}  // namespace

std::vector<int> NestedDissectionOrder(int n, const std::vector<int>& adj_begin, const std::vector<int>& adj) {
  Dissector d(n, adj_begin, adj);
  std::vector<int> all(static_cast<size_t>(n));
  std::iota(all.begin(), all.end(), 0);
  d.Order(std::move(all), 0);
  return d.out;
}

// ------------------------------------------------------------------------------------------ small dense kernels
namespace {

// C (m x n) -= A (m x k) * B (n x k)'   all row-major
template <int M, int N, int K>
inline void GemmNTSubFixed(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ B) {
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += A[i * K + k] * B[j * K + k];
      C[i * N + j] -= s;
    }
}
inline void GemmNTSub(double* C, const double* A, const double* B, int m, int n, int k) {
  if (m == 9 && n == 9 && k == 9) { GemmNTSubFixed<9, 9, 9>(C, A, B); return; }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int p = 0; p < k; ++p) s += A[i * k + p] * B[j * k + p];
      C[i * n + j] -= s;
    }
}
// in-place lower Cholesky of the e x e row-major block (upper part zeroed); false if not PD
inline bool CholLower(double* a, int e) {
  for (int j = 0; j < e; ++j) {
    double d = a[j * e + j];
    for (int k = 0; k < j; ++k) d -= a[j * e + k] * a[j * e + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    a[j * e + j] = d;
    for (int i = j + 1; i < e; ++i) {
      double s = a[i * e + j];
      for (int k = 0; k < j; ++k) s -= a[i * e + k] * a[j * e + k];
      a[i * e + j] = s / d;
    }
    for (int c = j + 1; c < e; ++c) a[j * e + c] = 0.0;
  }
  return true;
}
// X (m x e) <- X L^-T  (L lower e x e): row by row forward substitution
inline void TrsmRightLT(double* X, const double* L, int m, int e) {
  for (int i = 0; i < m; ++i) {
    double* x = X + i * e;
    for (int j = 0; j < e; ++j) {
      double s = x[j];
      for (int k = 0; k < j; ++k) s -= x[k] * L[j * e + k];
      x[j] = s / L[j * e + j];
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ analysis
void BlockSparseCholesky::Analyze(const std::vector<int>& block_sizes, const std::vector<std::pair<int, int>>& upper_cells) {
  n_ = int(block_sizes.size());
  size_ = block_sizes;
  pos_.assign(size_t(n_) + 1, 0);
  for (int i = 0; i < n_; ++i) pos_[size_t(i) + 1] = pos_[size_t(i)] + size_[size_t(i)];
  cells_ = upper_cells;
  // adjacency
  std::vector<int> xadj(size_t(n_) + 1, 0), adj;
  for (const auto& c : cells_)
    if (c.first != c.second) { xadj[size_t(c.first) + 1]++; xadj[size_t(c.second) + 1]++; }
  std::partial_sum(xadj.begin(), xadj.end(), xadj.begin());
  adj.resize(size_t(xadj.back()));
  {
    std::vector<int> cur(xadj.begin(), xadj.end() - 1);
    for (const auto& c : cells_)
      if (c.first != c.second) { adj[size_t(cur[size_t(c.first)]++)] = c.second; adj[size_t(cur[size_t(c.second)]++)] = c.first; }
  }
  perm_ = NestedDissectionOrder(n_, xadj, adj);
  iperm_.assign(size_t(n_), 0);
  for (int k = 0; k < n_; ++k) iperm_[size_t(perm_[size_t(k)])] = k;
  // lower pattern of P S P' by columns (rows > column)
  std::vector<std::vector<int>> below(static_cast<size_t>(n_));
  for (const auto& c : cells_) {
    if (c.first == c.second) continue;
    const int a = iperm_[size_t(c.first)], b = iperm_[size_t(c.second)];
    below[size_t(std::min(a, b))].push_back(std::max(a, b));
  }
  // symbolic factorisation: struct(L_k) = pattern(A_k) U (struct(L_c) \ {k}) over the children c of k
  std::vector<std::vector<int>> lstruct(static_cast<size_t>(n_));
  std::vector<std::vector<int>> children(static_cast<size_t>(n_));
  std::vector<int> mark(size_t(n_), -1), parent(size_t(n_), -1);
  for (int k = 0; k < n_; ++k) {
    std::vector<int>& s = lstruct[size_t(k)];
    mark[size_t(k)] = k;
    for (int r : below[size_t(k)]) if (mark[size_t(r)] != k) { mark[size_t(r)] = k; s.push_back(r); }
    for (int c : children[size_t(k)])
      for (int r : lstruct[size_t(c)]) if (r != k && mark[size_t(r)] != k) { mark[size_t(r)] = k; s.push_back(r); }
    std::sort(s.begin(), s.end());
    if (!s.empty()) { parent[size_t(k)] = s[0]; children[size_t(s[0])].push_back(k); }
    std::vector<int>().swap(below[size_t(k)]);
  }
  // storage
  col_begin_.assign(size_t(n_) + 1, 0);
  for (int k = 0; k < n_; ++k) col_begin_[size_t(k) + 1] = col_begin_[size_t(k)] + 1 + int64_t(lstruct[size_t(k)].size());
  const int64_t nb = col_begin_[size_t(n_)];
  row_of_.resize(size_t(nb));
  val_of_.resize(size_t(nb));
  int64_t nv = 0;
  flops_ = 0.0;
  for (int k = 0; k < n_; ++k) {
    const int ek = size_[size_t(perm_[size_t(k)])];
    int64_t b = col_begin_[size_t(k)];
    row_of_[size_t(b)] = k;
    val_of_[size_t(b)] = nv;
    nv += int64_t(ek) * ek;
    ++b;
    double rows_below = 0.0;
    for (int r : lstruct[size_t(k)]) {
      const int er = size_[size_t(perm_[size_t(r)])];
      row_of_[size_t(b)] = r;
      val_of_[size_t(b)] = nv;
      nv += int64_t(er) * ek;
      rows_below += er;
      ++b;
    }
    flops_ += double(ek) * ek * ek / 3.0 + rows_below * ek * ek + rows_below * (rows_below + 1.0) * ek;
  }
  values_.assign(size_t(nv), 0.0);
  // row lists
  rowlist_begin_.assign(size_t(n_) + 1, 0);
  for (int k = 0; k < n_; ++k)
    for (int64_t b = col_begin_[size_t(k)] + 1; b < col_begin_[size_t(k) + 1]; ++b) rowlist_begin_[size_t(row_of_[size_t(b)]) + 1]++;
  std::partial_sum(rowlist_begin_.begin(), rowlist_begin_.end(), rowlist_begin_.begin());
  rowlist_col_.resize(size_t(rowlist_begin_[size_t(n_)]));
  rowlist_blk_.resize(rowlist_col_.size());
  {
    std::vector<int64_t> cur(rowlist_begin_.begin(), rowlist_begin_.end() - 1);
    for (int k = 0; k < n_; ++k)
      for (int64_t b = col_begin_[size_t(k)] + 1; b < col_begin_[size_t(k) + 1]; ++b) {
        const int64_t at = cur[size_t(row_of_[size_t(b)])]++;
        rowlist_col_[size_t(at)] = k;
        rowlist_blk_[size_t(at)] = b;
      }
  }
  // heights of the elimination tree
  std::vector<int> height(size_t(n_), 0);
  int max_h = 0;
  for (int k = 0; k < n_; ++k) {
    for (int c : children[size_t(k)]) height[size_t(k)] = std::max(height[size_t(k)], height[size_t(c)] + 1);
    max_h = std::max(max_h, height[size_t(k)]);
  }
  height_begin_.assign(size_t(max_h) + 2, 0);
  for (int k = 0; k < n_; ++k) height_begin_[size_t(height[size_t(k)]) + 1]++;
  std::partial_sum(height_begin_.begin(), height_begin_.end(), height_begin_.begin());
  height_cols_.resize(size_t(n_));
  {
    std::vector<int> cur(height_begin_.begin(), height_begin_.end() - 1);
    for (int k = 0; k < n_; ++k) height_cols_[size_t(cur[size_t(height[size_t(k)])]++)] = k;
  }
  // where the input cells land
  cell_blk_.resize(cells_.size());
  for (size_t i = 0; i < cells_.size(); ++i) {
    const int a = iperm_[size_t(cells_[i].first)], b = iperm_[size_t(cells_[i].second)];
    const int col = std::min(a, b), row = std::max(a, b);
    const auto first = row_of_.begin() + col_begin_[size_t(col)], last = row_of_.begin() + col_begin_[size_t(col) + 1];
    cell_blk_[i] = std::lower_bound(first, last, row) - row_of_.begin();
  }
}

// ------------------------------------------------------------------------------------------ numeric
bool BlockSparseCholesky::FactorColumns(int threads) {
  bool ok = true;
  const int nh = int(height_begin_.size()) - 1;
  auto esize = [&](int k) { return size_[size_t(perm_[size_t(k)])]; };
  // update of the blocks [b_lo, b_hi) of column k (rows row_of_[b_lo] ..) by all earlier columns
  auto update = [&](int k, int64_t b_lo, int64_t b_hi, const std::vector<int64_t>& where) {
    if (b_lo >= b_hi) return;
    const int ek = esize(k);
    const int row_lo = row_of_[size_t(b_lo)], row_hi = row_of_[size_t(b_hi) - 1];
    for (int64_t q = rowlist_begin_[size_t(k)]; q < rowlist_begin_[size_t(k) + 1]; ++q) {
      const int c = rowlist_col_[size_t(q)];
      const int64_t bkc = rowlist_blk_[size_t(q)];  // block (k, c)
      const int ec = esize(c);
      const double* Lkc = &values_[size_t(val_of_[size_t(bkc)])];
      const int64_t cend = col_begin_[size_t(c) + 1];
      int64_t b2 = bkc;
      if (row_of_[size_t(b2)] < row_lo)
        b2 = std::lower_bound(row_of_.begin() + bkc, row_of_.begin() + cend, row_lo) - row_of_.begin();
      for (; b2 < cend; ++b2) {
        const int r = row_of_[size_t(b2)];
        if (r > row_hi) break;
        double* T = &values_[size_t(val_of_[size_t(where[size_t(r)])])];
        GemmNTSub(T, &values_[size_t(val_of_[size_t(b2)])], Lkc, esize(r), ek, ec);
      }
    }
  };
  std::vector<std::vector<int64_t>> where_tl(static_cast<size_t>(threads), std::vector<int64_t>(size_t(n_), -1));
  for (int h = 0; h < nh && ok; ++h) {
    const int hb = height_begin_[size_t(h)], he = height_begin_[size_t(h) + 1];
    if (he - hb >= 2 * threads || threads == 1) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
      for (int i = hb; i < he; ++i) {
        const int k = height_cols_[size_t(i)];
        std::vector<int64_t>& where = where_tl[size_t(omp_get_thread_num())];
        const int64_t cb = col_begin_[size_t(k)], ce = col_begin_[size_t(k) + 1];
        for (int64_t b = cb; b < ce; ++b) where[size_t(row_of_[size_t(b)])] = b;
        update(k, cb, ce, where);
        const int ek = esize(k);
        double* Lkk = &values_[size_t(val_of_[size_t(cb)])];
        if (!CholLower(Lkk, ek)) {
#pragma omp atomic write
          ok = false;
        } else {
          for (int64_t b = cb + 1; b < ce; ++b) TrsmRightLT(&values_[size_t(val_of_[size_t(b)])], Lkk, esize(row_of_[size_t(b)]), ek);
        }
      }
    } else {
      // few columns at this height (near the root): split every column's update over row slices
      std::vector<int64_t>& where = where_tl[0];
      for (int i = hb; i < he && ok; ++i) {
        const int k = height_cols_[size_t(i)];
        const int64_t cb = col_begin_[size_t(k)], ce = col_begin_[size_t(k) + 1];
        for (int64_t b = cb; b < ce; ++b) where[size_t(row_of_[size_t(b)])] = b;
        const int ek = esize(k);
        double* Lkk = &values_[size_t(val_of_[size_t(cb)])];
        bool pd = true;
#pragma omp parallel num_threads(threads)
        {
          const int t = omp_get_thread_num(), nt = omp_get_num_threads();
          const int64_t nblk = ce - cb;
          update(k, cb + nblk * t / nt, cb + nblk * (t + 1) / nt, where);
#pragma omp barrier
#pragma omp single
          pd = CholLower(Lkk, ek);
          if (pd) {
#pragma omp for schedule(static)
            for (int64_t b = cb + 1; b < ce; ++b) TrsmRightLT(&values_[size_t(val_of_[size_t(b)])], Lkk, esize(row_of_[size_t(b)]), ek);
          }
        }
        if (!pd) ok = false;
      }
    }
  }
  return ok;
}

bool BlockSparseCholesky::Factor(const std::function<const double*(int, int)>& cell_values, int threads) {
  std::fill(values_.begin(), values_.end(), 0.0);
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < int64_t(cells_.size()); ++i) {
    const int oi = cells_[size_t(i)].first, oj = cells_[size_t(i)].second;
    const int si = size_[size_t(oi)], sj = size_[size_t(oj)];
    const double* src = cell_values(oi, oj);  // si x sj
    double* dst = &values_[size_t(val_of_[size_t(cell_blk_[size_t(i)])])];
    const int a = iperm_[size_t(oi)], b = iperm_[size_t(oj)];
    if (a >= b) {  // block (a, b) = S_ij
      if (a == b) {  // diagonal cell: the reference stores the upper triangle; mirror it
        for (int r = 0; r < si; ++r)
          for (int c = 0; c < sj; ++c) dst[r * sj + c] = (c >= r) ? src[r * sj + c] : src[c * sj + r];
      } else {
        std::memcpy(dst, src, sizeof(double) * size_t(si) * size_t(sj));
      }
    } else {  // block (b, a) = S_ij'
      for (int r = 0; r < si; ++r)
        for (int c = 0; c < sj; ++c) dst[c * si + r] = src[r * sj + c];
    }
  }
  return FactorColumns(threads);
}

void BlockSparseCholesky::Solve(const double* rhs, double* x, int threads) const {
  auto esize = [&](int k) { return size_[size_t(perm_[size_t(k)])]; };
  std::vector<int> epos(size_t(n_) + 1, 0);
  for (int k = 0; k < n_; ++k) epos[size_t(k) + 1] = epos[size_t(k)] + esize(k);
  std::vector<double> y(static_cast<size_t>(epos[size_t(n_)]));
  for (int k = 0; k < n_; ++k) std::copy(rhs + pos_[size_t(perm_[size_t(k)])], rhs + pos_[size_t(perm_[size_t(k)]) + 1], y.begin() + epos[size_t(k)]);
  const int nh = int(height_begin_.size()) - 1;
  // L y = b, row oriented: every row only reads rows of smaller height
  for (int h = 0; h < nh; ++h) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
    for (int i = height_begin_[size_t(h)]; i < height_begin_[size_t(h) + 1]; ++i) {
      const int k = height_cols_[size_t(i)];
      const int ek = esize(k);
      double* yk = &y[size_t(epos[size_t(k)])];
      for (int64_t q = rowlist_begin_[size_t(k)]; q < rowlist_begin_[size_t(k) + 1]; ++q) {
        const int c = rowlist_col_[size_t(q)];
        const int ec = esize(c);
        const double* L = &values_[size_t(val_of_[size_t(rowlist_blk_[size_t(q)])])];  // ek x ec
        const double* yc = &y[size_t(epos[size_t(c)])];
        for (int r = 0; r < ek; ++r) {
          double s = 0.0;
          for (int p = 0; p < ec; ++p) s += L[r * ec + p] * yc[p];
          yk[r] -= s;
        }
      }
      const double* Lkk = &values_[size_t(val_of_[size_t(col_begin_[size_t(k)])])];
      for (int r = 0; r < ek; ++r) {
        double s = yk[r];
        for (int p = 0; p < r; ++p) s -= Lkk[r * ek + p] * yk[p];
        yk[r] = s / Lkk[r * ek + r];
      }
    }
  }
  // L' x = y, column oriented gather: every column only reads rows of larger height
  for (int h = nh - 1; h >= 0; --h) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
    for (int i = height_begin_[size_t(h)]; i < height_begin_[size_t(h) + 1]; ++i) {
      const int k = height_cols_[size_t(i)];
      const int ek = esize(k);
      double* yk = &y[size_t(epos[size_t(k)])];
      for (int64_t b = col_begin_[size_t(k)] + 1; b < col_begin_[size_t(k) + 1]; ++b) {
        const int r = row_of_[size_t(b)];
        const int er = esize(r);
        const double* L = &values_[size_t(val_of_[size_t(b)])];  // er x ek
        const double* xr = &y[size_t(epos[size_t(r)])];
        for (int p = 0; p < ek; ++p) {
          double s = 0.0;
          for (int q2 = 0; q2 < er; ++q2) s += L[q2 * ek + p] * xr[q2];
          yk[p] -= s;
        }
      }
      const double* Lkk = &values_[size_t(val_of_[size_t(col_begin_[size_t(k)])])];
      for (int r = ek - 1; r >= 0; --r) {
        double s = yk[r];
        for (int p = r + 1; p < ek; ++p) s -= Lkk[p * ek + r] * yk[p];
        yk[r] = s / Lkk[r * ek + r];
      }
    }
  }
  for (int k = 0; k < n_; ++k) std::copy(y.begin() + epos[size_t(k)], y.begin() + epos[size_t(k) + 1], x + pos_[size_t(perm_[size_t(k)])]);
}

}  // namespace orc
