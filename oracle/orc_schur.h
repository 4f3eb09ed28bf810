// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// SchurEliminator (schur_eliminator_impl.h), the BlockRandomAccess matrices it
// writes into, and the dense Cholesky of the reduced system
// (dense_cholesky.cc:153-207), restated with dynamic block sizes.
#ifndef ORC_SCHUR_H_
#define ORC_SCHUR_H_
#include <omp.h>

#include <algorithm>
#include <memory>

#include "orc_api.h"
#include "orc_common.h"

namespace orc {

// block_random_access_matrix.h:58-118.  GetCell returns the base pointer of the
// array the cell lives in plus (row, col, strides) of the cell inside it.
struct BRAM {
  virtual ~BRAM() = default;
  virtual double* GetCell(int i, int j, int* r, int* c, int* row_stride, int* col_stride) = 0;
  virtual void SetZero() = 0;
  virtual int num_rows() const = 0;
};

// block_random_access_dense_matrix.cc:41-72
struct DenseBRAM : BRAM {
  double* values;
  int n;
  std::vector<int> layout;
  DenseBRAM(double* v, const std::vector<int>& block_sizes) : values(v) {
    n = 0;
    for (int s : block_sizes) { layout.push_back(n); n += s; }
  }
  double* GetCell(int i, int j, int* r, int* c, int* rs, int* cs) override {
    *r = layout[i]; *c = layout[j]; *rs = n; *cs = n;
    return values;
  }
  void SetZero() override { std::fill(values, values + size_t(n) * n, 0.0); }
  int num_rows() const override { return n; }
};

// block_random_access_sparse_matrix.cc:48-122 as the eliminator sees it: only the cells of a given set of block
// pairs exist (GetCell is null for the others); kept in dense storage so the cell arithmetic is the same.
struct SubsetBRAM : DenseBRAM {
  std::vector<std::pair<int, int>> pairs;  // sorted
  SubsetBRAM(double* v, const std::vector<int>& block_sizes, const std::vector<std::pair<int, int>>& block_pairs)
      : DenseBRAM(v, block_sizes), pairs(block_pairs) {
    std::sort(pairs.begin(), pairs.end());
  }
  double* GetCell(int i, int j, int* r, int* c, int* rs, int* cs) override {
    if (!std::binary_search(pairs.begin(), pairs.end(), std::make_pair(i, j))) return nullptr;
    return DenseBRAM::GetCell(i, j, r, c, rs, cs);
  }
};

// block_random_access_diagonal_matrix.cc:50-85
struct DiagonalBRAM : BRAM {
  double* values;
  int n = 0;
  std::vector<int> sizes;
  std::vector<int64_t> offset;
  DiagonalBRAM(double* v, const std::vector<int>& block_sizes) : values(v), sizes(block_sizes) {
    int64_t o = 0;
    for (int s : block_sizes) { offset.push_back(o); o += int64_t(s) * s; n += s; }
    offset.push_back(o);
  }
  double* GetCell(int i, int j, int* r, int* c, int* rs, int* cs) override {
    if (i != j) return nullptr;
    *r = 0; *c = 0; *rs = sizes[i]; *cs = sizes[i];
    return values + offset[i];
  }
  void SetZero() override { std::fill(values, values + offset.back(), 0.0); }
  int num_rows() const override { return n; }
};

struct Chunk {
  int start = 0, size = 0;
  std::map<int, int> buffer_layout;  // f block id -> offset (BufferLayoutType, schur_eliminator.h:266)
};

struct Eliminator {
  BS bs;
  const double* values;
  int nelim;
  std::vector<Chunk> chunks;
  std::vector<int> lhs_row_layout;
  int buffer_size = 1;
  int uneliminated_row_begins = 0;
  static const int kLocks = 8192;
  std::vector<omp_lock_t> locks;

  Eliminator(const cx_block_structure* s, const double* v, int num_eliminate_blocks)
      : bs(s), values(v), nelim(num_eliminate_blocks), locks(kLocks) {
    for (auto& l : locks) omp_init_lock(&l);
    Init();
  }
  ~Eliminator() { for (auto& l : locks) omp_destroy_lock(&l); }

  omp_lock_t* LockFor(int i, int j) { return &locks[(uint32_t(i) * 2654435761u + uint32_t(j)) % kLocks]; }

  // schur_eliminator_impl.h:80-174
  void Init() {
    int n = 0;
    for (int i = nelim; i < bs.C; ++i) { lhs_row_layout.push_back(n); n += bs.cols[i].size; }
    int r = 0;
    while (r < bs.R) {
      if (bs.rcb[r + 1] == bs.rcb[r]) break;
      const int chunk_block_id = bs.cells[bs.rcb[r]].block_id;
      if (chunk_block_id >= nelim) break;
      chunks.emplace_back();
      Chunk& chunk = chunks.back();
      chunk.start = r;
      int buffer = 0;
      const int e_block_size = bs.cols[chunk_block_id].size;
      while (r + chunk.size < bs.R) {
        const int row = r + chunk.size;
        if (bs.cells[bs.rcb[row]].block_id != chunk_block_id) break;
        for (int c = bs.rcb[row] + 1; c < bs.rcb[row + 1]; ++c) {
          const int f = bs.cells[c].block_id;
          if (chunk.buffer_layout.emplace(f, buffer).second) buffer += e_block_size * bs.cols[f].size;
        }
        buffer_size = std::max(buffer, buffer_size);
        ++chunk.size;
      }
      r += chunk.size;
    }
    uneliminated_row_begins = chunks.empty() ? 0 : chunks.back().start + chunks.back().size;
  }

  std::vector<int> FBlockSizes() const {
    std::vector<int> s;
    for (int i = nelim; i < bs.C; ++i) s.push_back(bs.cols[i].size);
    return s;
  }

  // S += F_i' F_j for the f cells of one row (EBlockRowOuterProduct :665-714 for
  // first_f_cell = 1, NoEBlockRowOuterProduct :604-659 for first_f_cell = 0)
  void RowOuterProduct(int row, int first_f_cell, BRAM* lhs) {
    const int rs = bs.rows[row].size;
    const int b = bs.rcb[row], e = bs.rcb[row + 1];
    for (int i = b + first_f_cell; i < e; ++i) {
      const int block1 = bs.cells[i].block_id - nelim;
      const int s1 = bs.cols[bs.cells[i].block_id].size;
      int r, c, rst, cst;
      double* v = lhs->GetCell(block1, block1, &r, &c, &rst, &cst);
      if (v) {
        omp_lock_t* l = LockFor(block1, block1);
        omp_set_lock(l);
        MatTMat(values + bs.cells[i].position, rs, s1, values + bs.cells[i].position, rs, s1, v, r, c, rst, cst, 1);
        omp_unset_lock(l);
      }
      for (int j = i + 1; j < e; ++j) {
        const int block2 = bs.cells[j].block_id - nelim;
        const int s2 = bs.cols[bs.cells[j].block_id].size;
        double* v2 = lhs->GetCell(block1, block2, &r, &c, &rst, &cst);
        if (v2) {
          omp_lock_t* l = LockFor(block1, block2);
          omp_set_lock(l);
          MatTMat(values + bs.cells[i].position, rs, s1, values + bs.cells[j].position, rs, s2, v2, r, c, rst, cst, 1);
          omp_unset_lock(l);
        }
      }
    }
  }

  // schur_eliminator_impl.h:177-304
  void Eliminate(const double* b, const double* D, BRAM* lhs, double* rhs, int threads) {
    if (lhs->num_rows() > 0) {
      lhs->SetZero();
      if (rhs) std::fill(rhs, rhs + lhs->num_rows(), 0.0);
    }
    if (D) {
      for (int i = nelim; i < bs.C; ++i) {
        const int block_id = i - nelim;
        int r, c, rst, cst;
        double* v = lhs->GetCell(block_id, block_id, &r, &c, &rst, &cst);
        if (v) {
          const int sz = bs.cols[i].size;
          const double* d = D + bs.cols[i].position;
          for (int k = 0; k < sz; ++k) v[(r + k) * cst + c + k] += d[k] * d[k];
        }
      }
    }
    const int nchunks = int(chunks.size());
#pragma omp parallel num_threads(threads)
    {
      std::vector<double> buffer(buffer_size), outer(buffer_size);
      std::vector<double> ete, g, inv_g, sj;
#pragma omp for schedule(dynamic, 16)
      for (int ci = 0; ci < nchunks; ++ci) {
        const Chunk& chunk = chunks[ci];
        const int e_block_id = bs.cells[bs.rcb[chunk.start]].block_id;
        const int es = bs.cols[e_block_id].size;
        std::fill(buffer.begin(), buffer.end(), 0.0);
        ete.assign(size_t(es) * es, 0.0);
        if (D) {
          const double* d = D + bs.cols[e_block_id].position;
          for (int k = 0; k < es; ++k) ete[k * es + k] = d[k] * d[k];
        }
        g.assign(es, 0.0);

        // ChunkDiagonalBlockAndGradient :442-505
        for (int j = 0; j < chunk.size; ++j) {
          const int row = chunk.start + j;
          const int rs = bs.rows[row].size;
          const int cb = bs.rcb[row], ce = bs.rcb[row + 1];
          if (ce - cb > 1) RowOuterProduct(row, 1, lhs);
          const double* E = values + bs.cells[cb].position;
          MatTMat(E, rs, es, E, rs, es, ete.data(), 0, 0, es, es, 1);
          if (b) MatTVec(E, rs, es, b + bs.rows[row].position, g.data(), 1);
          for (int c = cb + 1; c < ce; ++c) {
            const int f = bs.cells[c].block_id;
            const int fs = bs.cols[f].size;
            double* buf = buffer.data() + chunk.buffer_layout.at(f);
            MatTMat(E, rs, es, values + bs.cells[c].position, rs, fs, buf, 0, 0, es, fs, 1);
          }
        }

        // InvertPSDMatrix<kEBlockSize>(assume_full_rank_ete = true, ete)
        // (invert_psd_matrix.h:48-72): closed form below 5, LLT otherwise.
        std::vector<double> inverse_ete(ete);
        if (es < 5) InvertSmall(inverse_ete.data(), es); else InvertPSD(inverse_ete.data(), es);

        if (rhs) {
          inv_g.assign(es, 0.0);
          MatVec(inverse_ete.data(), es, es, g.data(), inv_g.data(), 0);
          // UpdateRhs :379-420
          for (int j = 0; j < chunk.size; ++j) {
            const int row = chunk.start + j;
            const int rs = bs.rows[row].size;
            const int cb = bs.rcb[row], ce = bs.rcb[row + 1];
            sj.assign(b + bs.rows[row].position, b + bs.rows[row].position + rs);
            MatVec(values + bs.cells[cb].position, rs, es, inv_g.data(), sj.data(), -1);
            for (int c = cb + 1; c < ce; ++c) {
              const int f = bs.cells[c].block_id;
              const int block = f - nelim;
              omp_lock_t* l = LockFor(-1, block);
              omp_set_lock(l);
              MatTVec(values + bs.cells[c].position, rs, bs.cols[f].size, sj.data(), rhs + lhs_row_layout[block], 1);
              omp_unset_lock(l);
            }
          }
        }

        // ChunkOuterProduct :512-561
        for (auto it1 = chunk.buffer_layout.begin(); it1 != chunk.buffer_layout.end(); ++it1) {
          const int block1 = it1->first - nelim;
          const int s1 = bs.cols[it1->first].size;
          MatTMat(buffer.data() + it1->second, es, s1, inverse_ete.data(), es, es, outer.data(), 0, 0, s1, es, 0);
          for (auto it2 = it1; it2 != chunk.buffer_layout.end(); ++it2) {
            const int block2 = it2->first - nelim;
            int r, c, rst, cst;
            double* v = lhs->GetCell(block1, block2, &r, &c, &rst, &cst);
            if (v) {
              const int s2 = bs.cols[it2->first].size;
              omp_lock_t* l = LockFor(block1, block2);
              omp_set_lock(l);
              MatMat(outer.data(), s1, es, buffer.data() + it2->second, es, s2, v, r, c, rst, cst, -1);
              omp_unset_lock(l);
            }
          }
        }
      }
    }
    // NoEBlockRowsUpdate :567-593
    for (int row = uneliminated_row_begins; row < bs.R; ++row) {
      RowOuterProduct(row, 0, lhs);
      if (!rhs) continue;
      for (int c = bs.rcb[row]; c < bs.rcb[row + 1]; ++c) {
        const int f = bs.cells[c].block_id;
        MatTVec(values + bs.cells[c].position, bs.rows[row].size, bs.cols[f].size,
                b + bs.rows[row].position, rhs + lhs_row_layout[f - nelim], 1);
      }
    }
  }

  // schur_eliminator_impl.h:307-373
  void BackSubstitute(const double* b, const double* D, const double* z, double* y, int threads) {
    const int nchunks = int(chunks.size());
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int ci = 0; ci < nchunks; ++ci) {
      const Chunk& chunk = chunks[ci];
      const int e_block_id = bs.cells[bs.rcb[chunk.start]].block_id;
      const int es = bs.cols[e_block_id].size;
      double* y_ptr = y + bs.cols[e_block_id].position;
      std::vector<double> ete(size_t(es) * es, 0.0), sj;
      if (D) {
        const double* d = D + bs.cols[e_block_id].position;
        for (int k = 0; k < es; ++k) ete[k * es + k] = d[k] * d[k];
      }
      // NOTE: the reference accumulates into y_block without zeroing it first;
      // callers (schur_complement_solver.cc:137) zero x beforehand.
      for (int j = 0; j < chunk.size; ++j) {
        const int row = chunk.start + j;
        const int rs = bs.rows[row].size;
        const int cb = bs.rcb[row], ce = bs.rcb[row + 1];
        sj.assign(b + bs.rows[row].position, b + bs.rows[row].position + rs);
        for (int c = cb + 1; c < ce; ++c) {
          const int f = bs.cells[c].block_id;
          MatVec(values + bs.cells[c].position, rs, bs.cols[f].size, z + lhs_row_layout[f - nelim], sj.data(), -1);
        }
        const double* E = values + bs.cells[cb].position;
        MatTVec(E, rs, es, sj.data(), y_ptr, 1);
        MatTMat(E, rs, es, E, rs, es, ete.data(), 0, 0, es, es, 1);
      }
      if (es < 5) InvertSmall(ete.data(), es); else InvertPSD(ete.data(), es);
      std::vector<double> tmp(y_ptr, y_ptr + es);
      MatVec(ete.data(), es, es, tmp.data(), y_ptr, 0);
    }
  }
};

// Blocked right-looking Cholesky A = U'U on the upper triangle of row-major A
// (== Eigen::LLT<Lower> on the column-major view the reference factors,
// dense_cholesky.cc:153-178).  Returns false when not positive definite.
template <typename T>
inline bool CholeskyUpperT(T* A, int n, int threads) {
  const int NB = 64;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    // factor diagonal block
    for (int j = k0; j < k0 + kb; ++j) {
      for (int i = k0; i <= j; ++i) {
        T s = A[size_t(i) * n + j];
        for (int k = k0; k < i; ++k) s -= A[size_t(k) * n + i] * A[size_t(k) * n + j];
        if (i == j) {
          if (!(s > T(0))) return false;
          A[size_t(i) * n + i] = std::sqrt(s);
        } else {
          A[size_t(i) * n + j] = s / A[size_t(i) * n + i];
        }
      }
    }
    const int rest = k0 + kb;
    // panel: U(k0:k0+kb, rest:n) = U_kk^-T A(k0:k0+kb, rest:n)
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int j = rest; j < n; ++j) {
      for (int i = k0; i < k0 + kb; ++i) {
        T s = A[size_t(i) * n + j];
        for (int k = k0; k < i; ++k) s -= A[size_t(k) * n + i] * A[size_t(k) * n + j];
        A[size_t(i) * n + j] = s / A[size_t(i) * n + i];
      }
    }
    // trailing update: A(i,j) -= sum_k U(k,i) U(k,j), i<=j in rest..n
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int i = rest; i < n; ++i) {
      T* Ai = A + size_t(i) * n;
      for (int k = k0; k < k0 + kb; ++k) {
        const T uki = A[size_t(k) * n + i];
        const T* Uk = A + size_t(k) * n;
        for (int j = i; j < n; ++j) Ai[j] -= uki * Uk[j];
      }
    }
  }
  return true;
}

inline bool CholeskyUpper(double* A, int n, int threads) { return CholeskyUpperT<double>(A, n, threads); }

}  // namespace orc
#endif
