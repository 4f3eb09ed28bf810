// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// BlockSparseMatrix products and conversions, restating
// internal/ceres/block_sparse_matrix.cc.
#include <omp.h>

#include <algorithm>
#include <numeric>

#include "orc_api.h"
#include "orc_common.h"

namespace orc {

static int g_threads = 1;

// LLT-based inverse reading the upper triangle of row-major m.
bool InvertPSD(double* m, int n) {
  // With A row-major, its upper triangle (i<=j) is A(i,j).  Factor A = U' U.
  std::vector<double> U(size_t(n) * n, 0.0);
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i <= j; ++i) {
      double s = m[i * n + j];
      for (int k = 0; k < i; ++k) s -= U[k * n + i] * U[k * n + j];
      if (i == j) {
        if (!(s > 0.0)) return false;
        U[i * n + i] = std::sqrt(s);
      } else {
        U[i * n + j] = s / U[i * n + i];
      }
    }
  }
  // Solve U' U X = I column by column.
  std::vector<double> y(n);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) {  // U' y = e_c  (forward)
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= U[k * n + i] * y[k];
      y[i] = s / U[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {  // U x = y (backward)
      double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= U[i * n + k] * m[k * n + c];
      m[i * n + c] = s / U[i * n + i];
    }
  }
  return true;
}

// Cofactor inverse for n in {1,2,3,4} (Eigen's fixed-size inverse()), Gauss-Jordan otherwise.
bool InvertSmall(double* m, int n) {
  if (n == 1) {
    m[0] = 1.0 / m[0];
    return true;
  }
  if (n == 2) {
    const double a = m[0], b = m[1], c = m[2], d = m[3];
    const double invdet = 1.0 / (a * d - b * c);
    m[0] = d * invdet; m[1] = -b * invdet; m[2] = -c * invdet; m[3] = a * invdet;
    return true;
  }
  if (n == 3) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7],
                 i = m[8];
    const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const double det = a * c00 + b * c01 + c * c02;
    const double invdet = 1.0 / det;
    m[0] = c00 * invdet; m[1] = (c * h - b * i) * invdet; m[2] = (b * f - c * e) * invdet;
    m[3] = c01 * invdet; m[4] = (a * i - c * g) * invdet; m[5] = (c * d - a * f) * invdet;
    m[6] = c02 * invdet; m[7] = (b * g - a * h) * invdet; m[8] = (a * e - b * d) * invdet;
    return true;
  }
  // general: Gauss-Jordan with partial pivoting
  std::vector<double> a(m, m + n * n), inv(size_t(n) * n, 0.0);
  for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (std::fabs(a[r * n + col]) > std::fabs(a[piv * n + col])) piv = r;
    if (a[piv * n + col] == 0.0) return false;
    if (piv != col)
      for (int k = 0; k < n; ++k) {
        std::swap(a[piv * n + k], a[col * n + k]);
        std::swap(inv[piv * n + k], inv[col * n + k]);
      }
    const double p = 1.0 / a[col * n + col];
    for (int k = 0; k < n; ++k) { a[col * n + k] *= p; inv[col * n + k] *= p; }
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = a[r * n + col];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) { a[r * n + k] -= f * a[col * n + k]; inv[r * n + k] -= f * inv[col * n + k]; }
    }
  }
  std::copy(inv.begin(), inv.end(), m);
  return true;
}

void Transpose::Build(const BS& bs) {
  col_cell_begin.assign(bs.C + 1, 0);
  const int ncells = bs.rcb[bs.R];
  for (int c = 0; c < ncells; ++c) col_cell_begin[bs.cells[c].block_id + 1]++;
  std::partial_sum(col_cell_begin.begin(), col_cell_begin.end(), col_cell_begin.begin());
  cell_row.resize(ncells);
  cell_pos.resize(ncells);
  std::vector<int32_t> cursor(col_cell_begin.begin(), col_cell_begin.end() - 1);
  for (int r = 0; r < bs.R; ++r) {
    for (int c = bs.rcb[r]; c < bs.rcb[r + 1]; ++c) {
      const int k = cursor[bs.cells[c].block_id]++;
      cell_row[k] = r;
      cell_pos[k] = bs.cells[c].position;
    }
  }
}

}  // namespace orc

using namespace orc;

extern "C" {

void orc_set_num_threads(int n) { orc::g_threads = std::max(1, n); omp_set_num_threads(orc::g_threads); }
int orc_get_num_threads(void) { return orc::g_threads; }

// block_sparse_matrix.cc:239-274
void orc_right_multiply(const cx_block_structure* s, const double* values, const double* x, double* y) {
  BS bs(s);
#pragma omp parallel for schedule(static) num_threads(orc::g_threads)
  for (int r = 0; r < bs.R; ++r) {
    const int rs = bs.rows[r].size, rp = bs.rows[r].position;
    for (int c = bs.rcb[r]; c < bs.rcb[r + 1]; ++c) {
      const cx_cell& cell = bs.cells[c];
      MatVec(values + cell.position, rs, bs.cols[cell.block_id].size,
             x + bs.cols[cell.block_id].position, y + rp, 1);
    }
  }
}

// block_sparse_matrix.cc:278-349 (the transpose-structure form used when
// num_threads > 1; per output entry the summation runs over ascending rows,
// the order the single-threaded row loop produces as well)
void orc_left_multiply(const cx_block_structure* s, const double* values, const double* x, double* y) {
  BS bs(s);
  Transpose t;
  t.Build(bs);
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc::g_threads)
  for (int cb = 0; cb < bs.C; ++cb) {
    const int cs = bs.cols[cb].size, cp = bs.cols[cb].position;
    for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
      const int r = t.cell_row[k];
      MatTVec(values + t.cell_pos[k], bs.rows[r].size, cs, x + bs.rows[r].position, y + cp, 1);
    }
  }
}

// block_sparse_matrix.cc:351-401
void orc_squared_column_norm(const cx_block_structure* s, const double* values, double* x) {
  BS bs(s);
  std::fill(x, x + bs.num_cols(), 0.0);
  for (int r = 0; r < bs.R; ++r) {
    const int rs = bs.rows[r].size;
    for (int c = bs.rcb[r]; c < bs.rcb[r + 1]; ++c) {
      const cx_cell& cell = bs.cells[c];
      const int cs = bs.cols[cell.block_id].size, cp = bs.cols[cell.block_id].position;
      const double* m = values + cell.position;
      for (int j = 0; j < cs; ++j) {
        double sum = 0.0;
        for (int i = 0; i < rs; ++i) sum += m[i * cs + j] * m[i * cs + j];
        x[cp + j] += sum;
      }
    }
  }
}

// block_sparse_matrix.cc:403-450
void orc_scale_columns(const cx_block_structure* s, double* values, const double* scale) {
  BS bs(s);
#pragma omp parallel for schedule(static) num_threads(orc::g_threads)
  for (int r = 0; r < bs.R; ++r) {
    const int rs = bs.rows[r].size;
    for (int c = bs.rcb[r]; c < bs.rcb[r + 1]; ++c) {
      const cx_cell& cell = bs.cells[c];
      const int cs = bs.cols[cell.block_id].size, cp = bs.cols[cell.block_id].position;
      double* m = values + cell.position;
      for (int i = 0; i < rs; ++i)
        for (int j = 0; j < cs; ++j) m[i * cs + j] *= scale[cp + j];
    }
  }
}

// block_sparse_matrix.cc:69-118 + 451-492 (+ UpdateCompressedRowSparseMatrixImpl)
int64_t orc_to_crs(const cx_block_structure* s, const double* values, int transpose, int32_t* rows,
                   int32_t* cols, double* vals) {
  BS bs(s);
  int64_t off = 0;
  int32_t* rp = rows;
  *rp++ = 0;
  if (!transpose) {
    for (int r = 0; r < bs.R; ++r) {
      const int rs = bs.rows[r].size;
      for (int i = 0; i < rs; ++i) {
        for (int c = bs.rcb[r]; c < bs.rcb[r + 1]; ++c) {
          const cx_cell& cell = bs.cells[c];
          const int cs = bs.cols[cell.block_id].size, cp = bs.cols[cell.block_id].position;
          for (int j = 0; j < cs; ++j) {
            cols[off] = cp + j;
            vals[off] = values[cell.position + i * cs + j];
            ++off;
          }
        }
        *rp++ = int32_t(off);
      }
    }
  } else {
    Transpose t;
    t.Build(bs);
    for (int cb = 0; cb < bs.C; ++cb) {
      const int cs = bs.cols[cb].size;
      for (int j = 0; j < cs; ++j) {
        for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
          const int r = t.cell_row[k];
          const int rs = bs.rows[r].size, rpos = bs.rows[r].position;
          for (int i = 0; i < rs; ++i) {
            cols[off] = rpos + i;
            vals[off] = values[t.cell_pos[k] + i * cs + j];
            ++off;
          }
        }
        *rp++ = int32_t(off);
      }
    }
  }
  return off;
}

// detect_structure.cc:39-120; -1 == Eigen::Dynamic
void orc_detect_structure(const cx_block_structure* s, int num_eliminate_blocks, int* row_block_size,
                          int* e_block_size, int* f_block_size) {
  BS bs(s);
  *row_block_size = 0;
  *e_block_size = 0;
  *f_block_size = 0;
  const int kDynamic = -1;
  for (int r = 0; r < bs.R; ++r) {
    const int nc = bs.rcb[r + 1] - bs.rcb[r];
    if (nc == 0) continue;
    const cx_cell* cells = bs.cells + bs.rcb[r];
    if (cells[0].block_id >= num_eliminate_blocks) break;
    if (*row_block_size == 0) *row_block_size = bs.rows[r].size;
    else if (*row_block_size != kDynamic && *row_block_size != bs.rows[r].size) *row_block_size = kDynamic;
    const int e = cells[0].block_id;
    if (*e_block_size == 0) *e_block_size = bs.cols[e].size;
    else if (*e_block_size != kDynamic && *e_block_size != bs.cols[e].size) *e_block_size = kDynamic;
    if (nc > 1) {
      if (*f_block_size == 0) *f_block_size = bs.cols[cells[1].block_id].size;
      for (int c = 1; c < nc && *f_block_size != kDynamic; ++c)
        if (*f_block_size != bs.cols[cells[c].block_id].size) *f_block_size = kDynamic;
    }
    if (*row_block_size == kDynamic && *e_block_size == kDynamic && *f_block_size == kDynamic) break;
  }
}

}  // extern "C"
