// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the Ceres algorithms on the LM inner-linear-solve path.
// Nothing in the product (libcxschur, the host mirror, bench.py's GPU leg) may
// include, link or call this code; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do, and only as the checker / baseline.
//
// Parity status: the reference cannot be compiled in this image (Eigen, abseil
// and SuiteSparse are absent), so this restatement is pinned by the reference's
// own fixtures (tests/golden/, re-typed from
// internal/ceres/linear_least_squares_problems.cc, block_sparse_matrix_test.cc,
// conjugate_gradients_solver_test.cc) and by dense-algebra procedures the
// reference's tests use (schur_eliminator_test.cc:82-177 etc.).
// The Snavely Jacobian arithmetic has no fixture in the reference (the BAL file
// is a stripped blob): that part is "parity unpinned" and is checked against
// finite differences and scipy's Rodrigues rotation instead.
#ifndef ORC_COMMON_H_
#define ORC_COMMON_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "../include/cxschur.h"

namespace orc {

// ---------------------------------------------------------------------------
// small_blas.h:164-555 / small_blas_generic.h restated as plain loops.
// All matrices row-major.  kOp: +1 accumulate, -1 subtract, 0 assign.
// ---------------------------------------------------------------------------

// C(r_c.., c_c..) op= A * B      (small_blas.h:180-262  MatrixMatrixMultiply)
inline void MatMat(const double* A, int ra, int ca, const double* B, int rb, int cb, double* C,
                   int start_row_c, int start_col_c, int /*row_stride_c*/, int col_stride_c,
                   int op) {
  (void)rb;
  for (int i = 0; i < ra; ++i) {
    for (int j = 0; j < cb; ++j) {
      double s = 0.0;
      for (int k = 0; k < ca; ++k) s += A[i * ca + k] * B[k * cb + j];
      double* c = &C[(i + start_row_c) * col_stride_c + start_col_c + j];
      if (op > 0) *c += s; else if (op < 0) *c -= s; else *c = s;
    }
  }
}

// C op= A' * B                   (small_blas.h:280-372  MatrixTransposeMatrixMultiply)
inline void MatTMat(const double* A, int ra, int ca, const double* B, int rb, int cb, double* C,
                    int start_row_c, int start_col_c, int /*row_stride_c*/, int col_stride_c,
                    int op) {
  (void)rb;
  for (int i = 0; i < ca; ++i) {
    for (int j = 0; j < cb; ++j) {
      double s = 0.0;
      for (int k = 0; k < ra; ++k) s += A[k * ca + i] * B[k * cb + j];
      double* c = &C[(i + start_row_c) * col_stride_c + start_col_c + j];
      if (op > 0) *c += s; else if (op < 0) *c -= s; else *c = s;
    }
  }
}

// c op= A * b                    (small_blas.h:385-463  MatrixVectorMultiply)
inline void MatVec(const double* A, int ra, int ca, const double* b, double* c, int op) {
  for (int i = 0; i < ra; ++i) {
    double s = 0.0;
    for (int k = 0; k < ca; ++k) s += A[i * ca + k] * b[k];
    if (op > 0) c[i] += s; else if (op < 0) c[i] -= s; else c[i] = s;
  }
}

// c op= A' * b                   (small_blas.h:470-555  MatrixTransposeVectorMultiply)
inline void MatTVec(const double* A, int ra, int ca, const double* b, double* c, int op) {
  for (int j = 0; j < ca; ++j) {
    double s = 0.0;
    for (int k = 0; k < ra; ++k) s += A[k * ca + j] * b[k];
    if (op > 0) c[j] += s; else if (op < 0) c[j] -= s; else c[j] = s;
  }
}

// In-place inverse of a symmetric positive definite n x n row-major matrix via
// LLT, reading the upper triangle only -- the arithmetic of
// m.selfadjointView<Eigen::Upper>().llt().solve(Identity)
// (invert_psd_matrix.h:64-66, block_random_access_diagonal_matrix.cc:90-100,
// implicit_schur_complement.cc:201-202).  Returns false if not PD.
bool InvertPSD(double* m, int n);
// Closed-form (adjugate) inverse used for fixed sizes < 5
// (invert_psd_matrix.h:60-63 -> Eigen's m.inverse()); general (not only PSD).
bool InvertSmall(double* m, int n);

// ---------------------------------------------------------------------------
// block_structure.h:52-182 view over the flat C structure.
// ---------------------------------------------------------------------------
struct BS {
  int R = 0, C = 0;
  const cx_block* rows = nullptr;
  const cx_block* cols = nullptr;
  const int32_t* rcb = nullptr;  // row_cell_begin
  const cx_cell* cells = nullptr;
  explicit BS(const cx_block_structure* s)
      : R(s->num_row_blocks), C(s->num_col_blocks), rows(s->row_blocks), cols(s->col_blocks),
        rcb(s->row_cell_begin), cells(s->cells) {}
  int num_rows() const { return R ? rows[R - 1].position + rows[R - 1].size : 0; }
  int num_cols() const { return C ? cols[C - 1].position + cols[C - 1].size : 0; }
  int64_t nnz() const {
    int64_t n = 0;
    for (int r = 0; r < R; ++r)
      for (int c = rcb[r]; c < rcb[r + 1]; ++c) n += int64_t(rows[r].size) * cols[cells[c].block_id].size;
    return n;
  }
};

// Transposed index (block_sparse_matrix.cc:784-808 CreateTranspose): for every
// column block the (row block, value position) pairs in ascending row order.
struct Transpose {
  std::vector<int32_t> col_cell_begin;  // C+1
  std::vector<int32_t> cell_row;        // row block id
  std::vector<int32_t> cell_pos;        // value position (cell is a column-major view)
  void Build(const BS& bs);
};

}  // namespace orc
#endif
