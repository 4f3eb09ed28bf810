// The static <2,3,9> kernels for the reference's other small Schur structures, by embedding.
//
// The reference instantiates SchurEliminator / PartitionedMatrixView for 22 (row, e, f) size triples
// (schur_eliminator.cc:55-143, generated/) and lets rows without an e-block ride along (NoEBlockRowsUpdate,
// schur_eliminator_impl.h:567-659).  Here the tuned kernels exist for <2,3,9> with two cells per row; everything else
// used to drop to the dynamic-size path (cx_generic.hip: one thread per row block, atomics).  This file gives the
// structures with 2-row e-rows, e-blocks of ONE size e <= 3 and f-blocks of ONE size f <= 9 -- <2,3,6>, <2,3,3>, <2,3,4>,
// <2,2,2>, <2,2,3>, <2,2,4>, <2,3,9> in a layout other than BuildJacobianLayout's -- the static path as well, including
// trailing rows that hold a single f cell (priors on cameras):
//   * an inner matrix with the <2,3,9> layout is kept beside the caller's values: E cells widened to 2x3, F cells to
//     2x9 with zero columns.  A zero column of J is a zero row and column of J'J; the padded unknowns get diagonal 1
//     (D' = 1 there) and right-hand side 0, so they decouple exactly and stay 0 -- S, the reduced right-hand side and
//     the step of the real unknowns are the same sums as before, with extra zero terms.
//   * a trailing row with one f cell (s x f) becomes ceil(s / 2) two-row e-rows of a DUMMY point each, E = 0: such a
//     row's chunk contributes F'F to S and F'b to the right-hand side, which is exactly NoEBlockRowsUpdate.
//   * vectors are widened / narrowed around every call (num_cols-sized, small next to J).
// The wrapper stays a complete dynamic-size matrix (values in the caller's layout, all of cx_matrix_* works on it); only
// the products and the solvers go through the inner matrix.  Cost of the embedding: the padded cells are streamed too
// (<2,3,6>: 192 instead of 144 bytes per row block, 1.33 x); what does not embed -- e = 4, f > 9, rows of other sizes,
// several f cells in a row -- keeps the dynamic-size path.
#include <algorithm>
#include <numeric>

#include "cx_solver_internal.h"

namespace {

__global__ void k_embed_values(const double* __restrict__ src, const int32_t* __restrict__ epos, const int32_t* __restrict__ fpos,
                               const int32_t* __restrict__ valid, int64_t O_in, int e, int f, double* __restrict__ E,
                               double* __restrict__ F) {
  const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= 24 * O_in) return;
  const int64_t r = idx / 24;
  const int k = int(idx - 24 * r);
  if (k < 6) {
    const int i = k / 3, j = k - 3 * i;
    const int32_t p = epos[r];
    E[6 * r + k] = (p >= 0 && j < e) ? src[p + i * e + j] : 0.0;
  } else {
    const int kk = k - 6, i = kk / 9, j = kk - 9 * i;
    F[18 * r + kk] = (j < f && i < valid[r]) ? src[fpos[r] + i * f + j] : 0.0;
  }
}

// narrow -> wide column vector: [points (e each) | cameras (f each)] -> [points (3) | dummy points (3) | cameras (9)]
__global__ void k_widen_cols(const double* __restrict__ x, double* __restrict__ xw, int64_t P, int64_t P_in, int64_t C, int e, int f,
                             double fill) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t ne = 3 * P_in;
  if (i >= ne + 9 * C) return;
  double v = fill;
  if (i < ne) {
    const int64_t p = i / 3;
    const int k = int(i - 3 * p);
    if (p < P && k < e && x != nullptr) v = x[e * p + k];
    else if (p < P && k < e) v = 0.0;
  } else {
    const int64_t c = (i - ne) / 9;
    const int k = int(i - ne - 9 * c);
    if (k < f) v = x != nullptr ? x[e * P + f * c + k] : 0.0;
  }
  xw[i] = v;
}
// wide -> narrow; accumulate != 0: y += ...
__global__ void k_narrow_cols(const double* __restrict__ xw, double* __restrict__ x, int64_t P, int64_t P_in, int64_t C, int e, int f,
                              int accumulate) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= e * P + f * C) return;
  double v;
  if (i < e * P) {
    const int64_t p = i / e;
    v = xw[3 * p + (i - e * p)];
  } else {
    const int64_t c = (i - e * P) / f;
    v = xw[3 * P_in + 9 * c + (i - e * P - f * c)];
  }
  x[i] = accumulate ? x[i] + v : v;
}
// rows: the first n_main scalar rows map one to one, the rest through tail_src (original row, or -1 for padding)
__global__ void k_widen_rows(const double* __restrict__ b, double* __restrict__ bw, int64_t n_main, int64_t n_in,
                             const int32_t* __restrict__ tail_src) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  if (i < n_main) { bw[i] = b[i]; return; }
  const int32_t s = tail_src[i - n_main];
  bw[i] = s >= 0 ? b[s] : 0.0;
}
__global__ void k_narrow_rows_add(const double* __restrict__ yw, double* __restrict__ y, int64_t n_main, int64_t n_in,
                                  const int32_t* __restrict__ tail_src) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  if (i < n_main) { y[i] += yw[i]; return; }
  const int32_t s = tail_src[i - n_main];
  if (s >= 0) y[s] += yw[i];  // every original row has exactly one image
}

int Grid(int64_t n) { return int((n + 255) / 256); }

}  // namespace

// Builds A->embed when the structure qualifies; leaves it null (and returns CX_OK) when it does not.
int cxe_try_embed(cx_matrix* A) {
  static const bool disabled = std::getenv("CX_NO_EMBEDDING") != nullptr;  // A/B switch: keep the dynamic-size path
  if (disabled || A->is239 || A->nelim <= 0 || A->R == 0) return CX_OK;
  const int32_t P = A->nelim, C = A->Cb - P;
  if (C <= 0) return CX_OK;
  const int e = A->cols[0].size, f = A->cols[size_t(P)].size;
  if (e < 1 || e > 3 || f < 1 || f > 9) return CX_OK;
  for (int32_t j = 0; j < A->Cb; ++j) {
    const int want = j < P ? e : f;
    const int64_t pos = j < P ? int64_t(e) * j : int64_t(e) * P + int64_t(f) * (j - P);
    if (A->cols[size_t(j)].size != want || A->cols[size_t(j)].position != pos) return CX_OK;
  }
  // e-rows first: 2 rows, [e cell, f cell], sorted by e-block, no camera twice in a chunk; then rows with ONE f cell
  int64_t O_main = 0, pos = 0;
  {
    std::vector<int32_t> last_pt(size_t(C), -1);
    int32_t prev = 0;
    bool in_tail = false;
    for (int32_t r = 0; r < A->R; ++r) {
      const int32_t cb = A->rcb[size_t(r)], ce = A->rcb[size_t(r) + 1];
      if (A->rows[size_t(r)].position != pos) return CX_OK;
      pos += A->rows[size_t(r)].size;
      if (ce - cb == 2 && !in_tail) {
        const cx_cell &c0 = A->cells[size_t(cb)], &c1 = A->cells[size_t(cb) + 1];
        if (A->rows[size_t(r)].size != 2 || c0.block_id >= P || c1.block_id < P || c0.block_id < prev) return CX_OK;
        prev = c0.block_id;
        if (last_pt[size_t(c1.block_id - P)] == c0.block_id) return CX_OK;
        last_pt[size_t(c1.block_id - P)] = c0.block_id;
        ++O_main;
      } else if (ce - cb == 1 && A->cells[size_t(cb)].block_id >= P && A->rows[size_t(r)].size >= 1) {
        in_tail = true;
      } else {
        return CX_OK;
      }
    }
  }
  if (O_main == 0) return CX_OK;
  // the inner structure
  int64_t O_in = O_main;
  for (int32_t r = int32_t(O_main); r < A->R; ++r) O_in += (A->rows[size_t(r)].size + 1) / 2;
  if (24 * O_in >= (int64_t(1) << 31)) return CX_OK;
  const int64_t n_dummy = O_in - O_main;
  const int64_t P_in64 = int64_t(P) + n_dummy;
  if (P_in64 + C >= (int64_t(1) << 31)) return CX_OK;
  const int32_t P_in = int32_t(P_in64);
  std::vector<cx_block> rows(static_cast<size_t>(O_in)), cols(size_t(P_in) + size_t(C));
  std::vector<int32_t> rcb(size_t(O_in) + 1), epos(static_cast<size_t>(O_in)), fpos(static_cast<size_t>(O_in)), valid(static_cast<size_t>(O_in));
  std::vector<cx_cell> cells(size_t(2 * O_in));
  std::vector<int32_t> tail_src(size_t(2 * n_dummy), -1);
  for (int32_t j = 0; j < P_in; ++j) cols[size_t(j)] = cx_block{3, 3 * j};
  for (int32_t c = 0; c < C; ++c) cols[size_t(P_in) + size_t(c)] = cx_block{9, int32_t(3 * P_in64 + 9 * int64_t(c))};
  int64_t k = 0;
  for (int32_t r = 0; r < A->R; ++r) {
    const int32_t cb = A->rcb[size_t(r)];
    if (r < O_main) {
      epos[size_t(k)] = A->cells[size_t(cb)].position;
      fpos[size_t(k)] = A->cells[size_t(cb) + 1].position;
      valid[size_t(k)] = 2;
      cells[size_t(2 * k)] = cx_cell{A->cells[size_t(cb)].block_id, int32_t(6 * k)};
      cells[size_t(2 * k + 1)] = cx_cell{P_in + (A->cells[size_t(cb) + 1].block_id - P), int32_t(6 * O_in + 18 * k)};
      ++k;
    } else {
      const int s = A->rows[size_t(r)].size;
      for (int i0 = 0; i0 < s; i0 += 2, ++k) {
        epos[size_t(k)] = -1;
        fpos[size_t(k)] = A->cells[size_t(cb)].position + i0 * f;
        valid[size_t(k)] = std::min(2, s - i0);
        cells[size_t(2 * k)] = cx_cell{int32_t(P + (k - O_main)), int32_t(6 * k)};
        cells[size_t(2 * k + 1)] = cx_cell{P_in + (A->cells[size_t(cb)].block_id - P), int32_t(6 * O_in + 18 * k)};
        for (int i = 0; i < valid[size_t(k)]; ++i) tail_src[size_t(2 * (k - O_main) + i)] = A->rows[size_t(r)].position + i0 + i;
      }
    }
  }
  for (int64_t r = 0; r < O_in; ++r) {
    rows[size_t(r)] = cx_block{2, int32_t(2 * r)};
    rcb[size_t(r)] = int32_t(2 * r);
  }
  rcb[size_t(O_in)] = int32_t(2 * O_in);
  cx_block_structure bs{int32_t(O_in), P_in + C, rows.data(), cols.data(), rcb.data(), cells.data()};
  auto* E = new cx_embed;
  int rc = cx_matrix_create(A->ctx, &bs, P_in, &E->inner);
  if (rc == CX_OK && !E->inner->is239) {  // (cannot happen for a structure built here; keep the dynamic-size path if it does)
    cx_matrix_destroy(E->inner);
    delete E;
    return CX_OK;
  }
  hipStream_t st = A->ctx->stream;
  if (rc == CX_OK) rc = E->d_epos.upload(epos, st);
  if (rc == CX_OK) rc = E->d_fpos.upload(fpos, st);
  if (rc == CX_OK) rc = E->d_valid.upload(valid, st);
  if (rc == CX_OK && n_dummy > 0) rc = E->d_tail_src.upload(tail_src, st);
  if (rc != CX_OK) {
    if (E->inner) cx_matrix_destroy(E->inner);
    delete E;
    return rc;
  }
  E->e = e;
  E->f = f;
  E->P = P;
  E->C = C;
  E->P_in = P_in;
  E->O_main = O_main;
  E->O_in = O_in;
  // the visibility structure counts real points only (rows without an e-block see no point, visibility.cc:50-85)
  E->inner->P_vis = P;
  E->inner->O_vis = O_main;
  E->dirty = true;
  A->embed = E;
  return CX_OK;
}

void cxe_destroy(cx_matrix* A) {
  if (!A->embed) return;
  if (A->embed->inner) cx_matrix_destroy(A->embed->inner);
  delete A->embed;
  A->embed = nullptr;
}

// brings the inner values up to date with the caller's array
int cxe_sync(cx_matrix* A) {
  cx_embed* E = A->embed;
  if (!E->dirty) return CX_OK;
  cx_matrix* I = E->inner;
  hipLaunchKernelGGL(k_embed_values, dim3(Grid(24 * E->O_in)), dim3(256), 0, A->ctx->stream, (const double*)A->d_values.p,
                     (const int32_t*)E->d_epos.p, (const int32_t*)E->d_fpos.p, (const int32_t*)E->d_valid.p, E->O_in, E->e, E->f,
                     I->d_values.p, I->d_values.p + 6 * E->O_in);
  CX_HIP(hipGetLastError());
  CX_TRY(cx_matrix_values_changed(I));
  E->dirty = false;
  return CX_OK;
}

int cxe_widen_cols(cx_matrix* A, const double* x, double fill, double* xw) {
  cx_embed* E = A->embed;
  hipLaunchKernelGGL(k_widen_cols, dim3(Grid(3 * int64_t(E->P_in) + 9 * int64_t(E->C))), dim3(256), 0, A->ctx->stream, x, xw,
                     int64_t(E->P), int64_t(E->P_in), int64_t(E->C), E->e, E->f, fill);
  CX_HIP(hipGetLastError());
  return CX_OK;
}
int cxe_narrow_cols(cx_matrix* A, const double* xw, double* x, bool accumulate) {
  cx_embed* E = A->embed;
  hipLaunchKernelGGL(k_narrow_cols, dim3(Grid(A->num_cols)), dim3(256), 0, A->ctx->stream, xw, x, int64_t(E->P), int64_t(E->P_in),
                     int64_t(E->C), E->e, E->f, accumulate ? 1 : 0);
  CX_HIP(hipGetLastError());
  return CX_OK;
}
int cxe_widen_rows(cx_matrix* A, const double* b, double* bw) {
  cx_embed* E = A->embed;
  hipLaunchKernelGGL(k_widen_rows, dim3(Grid(2 * E->O_in)), dim3(256), 0, A->ctx->stream, b, bw, 2 * E->O_main, 2 * E->O_in,
                     (const int32_t*)E->d_tail_src.p);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// op 0: y += A x, 1: y += A'x, 2: y = diag(A'A); device pointers in the caller's spaces
int cxe_matrix_op(cx_matrix* A, int op, const double* x, double* y) {
  cx_embed* E = A->embed;
  cx_matrix* I = E->inner;
  hipStream_t st = A->ctx->stream;
  CX_TRY(cxe_sync(A));
  CX_TRY(E->d_cols.alloc(size_t(I->num_cols)));
  CX_TRY(E->d_rows.alloc(size_t(I->num_rows)));
  switch (op) {
    case 0:
      CX_TRY(cxe_widen_cols(A, x, 0.0, E->d_cols.p));
      if (E->O_in == E->O_main) return cxk_right_multiply(I, E->d_cols.p, y);  // no trailing rows: the row spaces coincide
      CX_HIP(hipMemsetAsync(E->d_rows.p, 0, size_t(I->num_rows) * sizeof(double), st));
      CX_TRY(cxk_right_multiply(I, E->d_cols.p, E->d_rows.p));
      hipLaunchKernelGGL(k_narrow_rows_add, dim3(Grid(I->num_rows)), dim3(256), 0, st, (const double*)E->d_rows.p, y, 2 * E->O_main,
                         2 * E->O_in, (const int32_t*)E->d_tail_src.p);
      break;
    case 1:
      if (E->O_in != E->O_main) CX_TRY(cxe_widen_rows(A, x, E->d_rows.p));
      CX_HIP(hipMemsetAsync(E->d_cols.p, 0, size_t(I->num_cols) * sizeof(double), st));
      CX_TRY(cx_matrix_ensure_ft(I));
      CX_TRY(cxk_left_multiply(I, E->O_in != E->O_main ? E->d_rows.p : x, E->d_cols.p));
      CX_TRY(cxe_narrow_cols(A, E->d_cols.p, y, true));
      break;
    default:
      CX_TRY(cx_matrix_ensure_ft(I));
      CX_TRY(cxk_squared_column_norm(I, E->d_cols.p));
      CX_TRY(cxe_narrow_cols(A, E->d_cols.p, y, false));
      break;
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// The solve: b, D, x are device pointers in the caller's spaces; `solve` runs a static solver on the inner matrix
int cxe_solve(cx_matrix* A, const double* b, const double* D, double* x,
              const std::function<int(cx_matrix*, const double*, const double*, double*)>& solve) {
  cx_embed* E = A->embed;
  cx_matrix* I = E->inner;
  CX_TRY(cxe_sync(A));
  CX_TRY(E->d_b.alloc(size_t(I->num_rows)));
  CX_TRY(E->d_D.alloc(size_t(I->num_cols)));
  CX_TRY(E->d_x.alloc(size_t(I->num_cols)));
  const double* bw = b;
  if (E->O_in != E->O_main) {
    CX_TRY(cxe_widen_rows(A, b, E->d_b.p));
    bw = E->d_b.p;
  }
  // the padded unknowns (zero columns of J, dummy points) get diagonal 1: they decouple and stay 0
  CX_TRY(cxe_widen_cols(A, D, 1.0, E->d_D.p));
  CX_HIP(hipMemsetAsync(E->d_x.p, 0, size_t(I->num_cols) * sizeof(double), A->ctx->stream));
  CX_TRY(solve(I, bw, E->d_D.p, E->d_x.p));
  return cxe_narrow_cols(A, E->d_x.p, x, false);
}
