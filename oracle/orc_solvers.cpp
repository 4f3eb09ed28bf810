// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// PartitionedMatrixView, ImplicitSchurComplement, ConjugateGradientsSolver, the
// preconditioners and the three LinearSolvers on the path, restating
//   partitioned_matrix_view_impl.h, implicit_schur_complement.cc,
//   conjugate_gradients_solver.h, block_jacobi_preconditioner.cc,
//   schur_jacobi_preconditioner.cc, power_series_expansion_preconditioner.cc,
//   cgnr_solver.cc, iterative_schur_complement_solver.cc, schur_complement_solver.cc.
// The optional all-reduce callback is NOT in the reference (single process);
// it restates the shard-by-point design of the product so that the gloo tests
// can check sharded == unsharded.
#include <omp.h>

#include <numeric>
#include <set>

#include <cstdio>
#include <functional>

#include "orc_schur.h"
#include "orc_sparse_chol.h"
#include "orc_visibility.h"

namespace orc {

using Vec = std::vector<double>;

// Wall time of the numeric part of the last solve: the clock starts once the
// structure-only objects (transpose index, chunks) exist, which the reference
// builds once per Solver::Solve and reuses across iterations (linear_solver.h:137-142).
static double g_solve_t0 = 0.0, g_last_solve_seconds = 0.0;
static void MarkSolveStart() { g_solve_t0 = omp_get_wtime(); }

struct Comm {
  orc_allreduce_fn fn = nullptr;
  void* user = nullptr;
  bool active() const { return fn != nullptr; }
  void Sum(double* buf, int64_t n) const { if (fn) fn(buf, n, user); }
};

// partitioned_matrix_view_impl.h:47-658
struct PMV {
  BS bs;
  const double* values;
  int nelim;
  int num_row_blocks_e = 0;
  int num_cols_e = 0, num_cols_f = 0;
  Transpose t;
  int threads;

  PMV(const cx_block_structure* s, const double* v, int num_eliminate_blocks, int nthreads)
      : bs(s), values(v), nelim(num_eliminate_blocks), threads(nthreads) {
    // :60-90 rows whose first cell is an e block
    for (int r = 0; r < bs.R; ++r) {
      if (bs.rcb[r + 1] == bs.rcb[r]) continue;  // the reference CHECKs non-empty rows
      if (bs.cells[bs.rcb[r]].block_id < nelim) num_row_blocks_e = r + 1;
    }
    for (int c = 0; c < bs.C; ++c) (c < nelim ? num_cols_e : num_cols_f) += bs.cols[c].size;
    t.Build(bs);
  }
  int first_f_cell(int r) const { return r < num_row_blocks_e ? 1 : 0; }

  // y += E x  (:92-118)
  void RightMultiplyE(const double* x, double* y) const {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int r = 0; r < num_row_blocks_e; ++r) {
      const cx_cell& cell = bs.cells[bs.rcb[r]];
      MatVec(values + cell.position, bs.rows[r].size, bs.cols[cell.block_id].size,
             x + bs.cols[cell.block_id].position, y + bs.rows[r].position, 1);
    }
  }
  // y += F x  (:120-170); x indexed from the first f column
  void RightMultiplyF(const double* x, double* y) const {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int r = 0; r < bs.R; ++r) {
      for (int c = bs.rcb[r] + first_f_cell(r); c < bs.rcb[r + 1]; ++c) {
        const cx_cell& cell = bs.cells[c];
        MatVec(values + cell.position, bs.rows[r].size, bs.cols[cell.block_id].size,
               x + bs.cols[cell.block_id].position - num_cols_e, y + bs.rows[r].position, 1);
      }
    }
  }
  // y += E' x (:172-230, transpose-structure form)
  void LeftMultiplyE(const double* x, double* y) const {
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int cb = 0; cb < nelim; ++cb) {
      for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
        const int r = t.cell_row[k];
        MatTVec(values + t.cell_pos[k], bs.rows[r].size, bs.cols[cb].size, x + bs.rows[r].position,
                y + bs.cols[cb].position, 1);
      }
    }
  }
  // y += F' x (:232-320)
  void LeftMultiplyF(const double* x, double* y) const {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int cb = nelim; cb < bs.C; ++cb) {
      for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
        const int r = t.cell_row[k];
        MatTVec(values + t.cell_pos[k], bs.rows[r].size, bs.cols[cb].size, x + bs.rows[r].position,
                y + bs.cols[cb].position - num_cols_e, 1);
      }
    }
  }
  // block diagonal of E'E, one size^2 row-major block per e block (:420-540)
  void BlockDiagonalEtE(std::vector<Vec>* blocks) const {
    blocks->resize(nelim);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int cb = 0; cb < nelim; ++cb) {
      const int cs = bs.cols[cb].size;
      Vec& m = (*blocks)[cb];
      m.assign(size_t(cs) * cs, 0.0);
      for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
        const int rs = bs.rows[t.cell_row[k]].size;
        const double* E = values + t.cell_pos[k];
        MatTMat(E, rs, cs, E, rs, cs, m.data(), 0, 0, cs, cs, 1);
      }
    }
  }
  // block diagonal of F'F (:542-658)
  void BlockDiagonalFtF(std::vector<Vec>* blocks) const {
    blocks->resize(bs.C - nelim);
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int cb = nelim; cb < bs.C; ++cb) {
      const int cs = bs.cols[cb].size;
      Vec& m = (*blocks)[cb - nelim];
      m.assign(size_t(cs) * cs, 0.0);
      for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
        const int rs = bs.rows[t.cell_row[k]].size;
        const double* F = values + t.cell_pos[k];
        MatTMat(F, rs, cs, F, rs, cs, m.data(), 0, 0, cs, cs, 1);
      }
    }
  }
};

// y += blockdiag(blocks) x   (BlockSparseMatrix product of the block-diagonal matrix)
static void BlockDiagMultiply(const std::vector<Vec>& blocks, const cx_block* cols, int first, int pos0,
                              const double* x, double* y) {
  for (size_t i = 0; i < blocks.size(); ++i) {
    const int s = cols[first + i].size, p = cols[first + i].position - pos0;
    MatVec(blocks[i].data(), s, s, x + p, y + p, 1);
  }
}

// Sum-all-reduce a list of equally ordered blocks in one message.
static void AllReduceBlocks(const Comm& comm, std::vector<Vec>* blocks) {
  if (!comm.active()) return;
  size_t total = 0;
  for (auto& b : *blocks) total += b.size();
  Vec flat(total);
  size_t o = 0;
  for (auto& b : *blocks) { std::copy(b.begin(), b.end(), flat.begin() + o); o += b.size(); }
  comm.Sum(flat.data(), int64_t(total));
  o = 0;
  for (auto& b : *blocks) { std::copy(flat.begin() + o, flat.begin() + o + b.size(), b.begin()); o += b.size(); }
}

// implicit_schur_complement.cc:49-276
struct ISC {
  PMV A;
  const double* D = nullptr;
  const double* b = nullptr;
  Comm comm;
  std::vector<Vec> ete_inv, ftf_inv;
  bool compute_ftf_inverse;
  Vec rhs, tmp_rows, tmp_e_cols, tmp_e_cols_2, tmp_f_cols;

  ISC(const cx_block_structure* s, const double* v, int nelim, int threads, bool need_ftf, Comm c)
      : A(s, v, nelim, threads), comm(c), compute_ftf_inverse(need_ftf) {}
  int num_rows() const { return A.num_cols_f; }

  // AddDiagonalAndInvert :179-204
  static bool AddDiagonalAndInvert(const double* D, const cx_block* cols, int first, std::vector<Vec>* blocks) {
    bool ok = true;
    for (size_t i = 0; i < blocks->size(); ++i) {
      const int s = cols[first + i].size;
      Vec& m = (*blocks)[i];
      if (D) {
        const double* d = D + cols[first + i].position;
        for (int k = 0; k < s; ++k) m[k * s + k] += d[k] * d[k];
      }
      ok = InvertPSD(m.data(), s) && ok;
    }
    return ok;
  }

  void Init(const double* D_, const double* b_) {
    D = D_;
    b = b_;
    A.BlockDiagonalEtE(&ete_inv);
    AddDiagonalAndInvert(D, A.bs.cols, 0, &ete_inv);
    if (compute_ftf_inverse) {
      A.BlockDiagonalFtF(&ftf_inv);
      AllReduceBlocks(comm, &ftf_inv);
      AddDiagonalAndInvert(D, A.bs.cols, A.nelim, &ftf_inv);
    }
    rhs.assign(A.num_cols_f, 0.0);
    tmp_rows.assign(A.bs.num_rows(), 0.0);
    tmp_e_cols.assign(A.num_cols_e, 0.0);
    tmp_e_cols_2.assign(A.num_cols_e, 0.0);
    tmp_f_cols.assign(A.num_cols_f, 0.0);
    if (b) UpdateRhs();
  }

  void EtEInvMultiply(const double* x, double* y) const {  // y += (E'E)^-1 x
    BlockDiagMultiply(ete_inv, A.bs.cols, 0, 0, x, y);
  }
  void FtFInvMultiply(const double* x, double* y) const {
    BlockDiagMultiply(ftf_inv, A.bs.cols, A.nelim, A.num_cols_e, x, y);
  }

  // :106-144  y = S x
  void RightMultiply(const double* x, double* y) {
    std::fill(tmp_rows.begin(), tmp_rows.end(), 0.0);
    A.RightMultiplyF(x, tmp_rows.data());
    std::fill(tmp_e_cols.begin(), tmp_e_cols.end(), 0.0);
    A.LeftMultiplyE(tmp_rows.data(), tmp_e_cols.data());
    std::fill(tmp_e_cols_2.begin(), tmp_e_cols_2.end(), 0.0);
    EtEInvMultiply(tmp_e_cols.data(), tmp_e_cols_2.data());
    for (auto& v : tmp_e_cols_2) v = -v;
    A.RightMultiplyE(tmp_e_cols_2.data(), tmp_rows.data());
    const int n = A.num_cols_f;
    if (!comm.active()) {
      if (D) { const double* Df = D + A.num_cols_e; for (int i = 0; i < n; ++i) y[i] = Df[i] * Df[i] * x[i]; }
      else std::fill(y, y + n, 0.0);
      A.LeftMultiplyF(tmp_rows.data(), y);
    } else {
      // shard: sum the F' part over ranks, then add the diagonal once
      std::fill(y, y + n, 0.0);
      A.LeftMultiplyF(tmp_rows.data(), y);
      comm.Sum(y, n);
      if (D) { const double* Df = D + A.num_cols_e; for (int i = 0; i < n; ++i) y[i] += Df[i] * Df[i] * x[i]; }
    }
  }

  // :146-177  y += Z x,  Z = (F'F)^-1 F'E (E'E)^-1 E'F
  void InversePowerSeriesOperatorRightMultiplyAccumulate(const double* x, double* y) {
    std::fill(tmp_rows.begin(), tmp_rows.end(), 0.0);
    A.RightMultiplyF(x, tmp_rows.data());
    std::fill(tmp_e_cols.begin(), tmp_e_cols.end(), 0.0);
    A.LeftMultiplyE(tmp_rows.data(), tmp_e_cols.data());
    std::fill(tmp_e_cols_2.begin(), tmp_e_cols_2.end(), 0.0);
    EtEInvMultiply(tmp_e_cols.data(), tmp_e_cols_2.data());
    std::fill(tmp_rows.begin(), tmp_rows.end(), 0.0);
    A.RightMultiplyE(tmp_e_cols_2.data(), tmp_rows.data());
    std::fill(tmp_f_cols.begin(), tmp_f_cols.end(), 0.0);
    A.LeftMultiplyF(tmp_rows.data(), tmp_f_cols.data());
    comm.Sum(tmp_f_cols.data(), int64_t(tmp_f_cols.size()));
    FtFInvMultiply(tmp_f_cols.data(), y);
  }

  // :208-243
  void BackSubstitute(const double* x, double* y) {
    const int num_cols = A.num_cols_e + A.num_cols_f;
    std::fill(tmp_rows.begin(), tmp_rows.end(), 0.0);
    A.RightMultiplyF(x, tmp_rows.data());
    for (size_t i = 0; i < tmp_rows.size(); ++i) tmp_rows[i] = b[i] - tmp_rows[i];
    std::fill(tmp_e_cols.begin(), tmp_e_cols.end(), 0.0);
    A.LeftMultiplyE(tmp_rows.data(), tmp_e_cols.data());
    std::fill(y, y + num_cols, 0.0);
    EtEInvMultiply(tmp_e_cols.data(), y);
    std::copy(x, x + A.num_cols_f, y + A.num_cols_e);
  }

  // :251-276
  void UpdateRhs() {
    std::fill(tmp_e_cols.begin(), tmp_e_cols.end(), 0.0);
    A.LeftMultiplyE(b, tmp_e_cols.data());
    std::fill(tmp_e_cols_2.begin(), tmp_e_cols_2.end(), 0.0);
    EtEInvMultiply(tmp_e_cols.data(), tmp_e_cols_2.data());
    std::fill(tmp_rows.begin(), tmp_rows.end(), 0.0);
    A.RightMultiplyE(tmp_e_cols_2.data(), tmp_rows.data());
    for (size_t i = 0; i < tmp_rows.size(); ++i) tmp_rows[i] = b[i] - tmp_rows[i];
    std::fill(rhs.begin(), rhs.end(), 0.0);
    A.LeftMultiplyF(tmp_rows.data(), rhs.data());
    comm.Sum(rhs.data(), int64_t(rhs.size()));
  }
};

struct CGOptions {
  int min_num_iterations = 0, max_num_iterations = 500, residual_reset_period = 10;
  double r_tolerance = -1.0, q_tolerance = 0.0;
};
using Op = std::function<void(const Vec&, Vec&)>;           // y += Op x
using DotFn = std::function<double(const Vec&, const Vec&)>;

static void SetMessage(cx_summary* s, const char* fmt, double a = 0, double b = 0, double c = 0, double d = 0) {
  std::snprintf(s->message, sizeof(s->message), fmt, a, b, c, d);
}

// conjugate_gradients_solver.h:107-305
static cx_summary ConjugateGradients(const CGOptions& options, const Op& lhs, const Vec& rhs,
                                     const Op& preconditioner, const DotFn& Dot, Vec& solution) {
  auto IsZeroOrInfinity = [](double x) { return x == 0.0 || std::isinf(x); };
  auto Norm = [&](const Vec& v) { return std::sqrt(Dot(v, v)); };
  const size_t n = rhs.size();
  Vec p(n), r(n), z(n), tmp(n);
  cx_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  summary.termination_type = CX_NO_CONVERGENCE;
  SetMessage(&summary, "Maximum number of iterations reached.");
  summary.num_iterations = 0;

  const double norm_rhs = Norm(rhs);
  if (norm_rhs == 0.0) {
    std::fill(solution.begin(), solution.end(), 0.0);
    summary.termination_type = CX_SUCCESS;
    SetMessage(&summary, "Convergence. |b| = 0.");
    return summary;
  }
  const double tol_r = options.r_tolerance * norm_rhs;

  std::fill(tmp.begin(), tmp.end(), 0.0);
  lhs(solution, tmp);
  for (size_t i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
  double norm_r = Norm(r);
  if (options.min_num_iterations == 0 && norm_r <= tol_r) {
    summary.termination_type = CX_SUCCESS;
    SetMessage(&summary, "Convergence. |r| = %e <= %e.", norm_r, tol_r);
    return summary;
  }
  double rho = 1.0;
  for (size_t i = 0; i < n; ++i) tmp[i] = rhs[i] + r[i];
  double Q0 = -Dot(solution, tmp);

  for (summary.num_iterations = 1;; ++summary.num_iterations) {
    std::fill(z.begin(), z.end(), 0.0);
    preconditioner(r, z);
    const double last_rho = rho;
    rho = Dot(r, z);
    if (IsZeroOrInfinity(rho)) {
      summary.termination_type = CX_FAILURE;
      SetMessage(&summary, "Numerical failure. rho = r'z = %e.", rho);
      break;
    }
    if (summary.num_iterations == 1) {
      p = z;
    } else {
      const double beta = rho / last_rho;
      if (IsZeroOrInfinity(beta)) {
        summary.termination_type = CX_FAILURE;
        SetMessage(&summary, "Numerical failure. beta = rho_n / rho_{n-1} = %e, rho_n = %e, rho_{n-1} = %e", beta, rho, last_rho);
        break;
      }
      for (size_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    Vec& q = z;
    std::fill(q.begin(), q.end(), 0.0);
    lhs(p, q);
    const double pq = Dot(p, q);
    if (pq <= 0 || std::isinf(pq)) {
      summary.termination_type = CX_NO_CONVERGENCE;
      SetMessage(&summary, "Matrix is indefinite, no more progress can be made. p'q = %e. |p| = %e, |q| = %e", pq, Norm(p), Norm(q));
      break;
    }
    const double alpha = rho / pq;
    if (std::isinf(alpha)) {
      summary.termination_type = CX_FAILURE;
      SetMessage(&summary, "Numerical failure. alpha = rho / pq = %e, rho = %e, pq = %e.", alpha, rho, pq);
      break;
    }
    for (size_t i = 0; i < n; ++i) solution[i] = solution[i] + alpha * p[i];
    if (summary.num_iterations % options.residual_reset_period == 0) {
      std::fill(tmp.begin(), tmp.end(), 0.0);
      lhs(solution, tmp);
      for (size_t i = 0; i < n; ++i) r[i] = rhs[i] - tmp[i];
    } else {
      for (size_t i = 0; i < n; ++i) r[i] = r[i] - alpha * q[i];
    }
    for (size_t i = 0; i < n; ++i) tmp[i] = rhs[i] + r[i];
    const double Q1 = -Dot(solution, tmp);
    const double zeta = summary.num_iterations * (Q1 - Q0) / Q1;
    if (zeta < options.q_tolerance && summary.num_iterations >= options.min_num_iterations) {
      summary.termination_type = CX_SUCCESS;
      SetMessage(&summary, "Iteration: %.0f Convergence: zeta = %e < %e. |r| = %e",
                 double(summary.num_iterations), zeta, options.q_tolerance, Norm(r));
      break;
    }
    Q0 = Q1;
    norm_r = Norm(r);
    if (norm_r <= tol_r && summary.num_iterations >= options.min_num_iterations) {
      summary.termination_type = CX_SUCCESS;
      SetMessage(&summary, "Iteration: %.0f Convergence. |r| = %e <= %e.", double(summary.num_iterations), norm_r, tol_r);
      break;
    }
    if (summary.num_iterations >= options.max_num_iterations) break;
  }
  return summary;
}

static double PlainDot(const Vec& a, const Vec& b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

// cgnr_solver.cc:146-207
static cx_summary SolveCgnr(const cx_block_structure* s, const double* values, const double* b,
                            const double* D, const cx_solver_options& o, double r_tol, double q_tol,
                            double* x, const Comm& comm, int threads) {
  BS bs(s);
  const int num_cols = bs.num_cols(), num_rows = bs.num_rows();
  // shard layout: column blocks [0, nelim) are local points, the rest replicated cameras
  int num_cols_e = 0;
  for (int c = 0; c < o.num_eliminate_blocks; ++c) num_cols_e += bs.cols[c].size;
  const int n_shared = comm.active() ? num_cols - num_cols_e : 0;  // entries summed over ranks
  const int shared0 = num_cols - n_shared;

  // BlockSparseJacobiPreconditioner::UpdateImpl (block_jacobi_preconditioner.cc:59-115)
  std::vector<Vec> jac;
  if (o.preconditioner_type == CX_JACOBI) {
    Transpose t;
    t.Build(bs);
    jac.resize(bs.C);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int cb = 0; cb < bs.C; ++cb) {
      const int cs = bs.cols[cb].size;
      jac[cb].assign(size_t(cs) * cs, 0.0);
      for (int k = t.col_cell_begin[cb]; k < t.col_cell_begin[cb + 1]; ++k) {
        const int rs = bs.rows[t.cell_row[k]].size;
        const double* m = values + t.cell_pos[k];
        MatTMat(m, rs, cs, m, rs, cs, jac[cb].data(), 0, 0, cs, cs, 1);
      }
    }
    if (comm.active()) {
      std::vector<Vec> shared(jac.begin() + o.num_eliminate_blocks, jac.end());
      AllReduceBlocks(comm, &shared);
      std::copy(shared.begin(), shared.end(), jac.begin() + o.num_eliminate_blocks);
    }
    for (int cb = 0; cb < bs.C; ++cb) {
      const int cs = bs.cols[cb].size;
      if (D) for (int k = 0; k < cs; ++k) jac[cb][k * cs + k] += D[bs.cols[cb].position + k] * D[bs.cols[cb].position + k];
      InvertPSD(jac[cb].data(), cs);  // BlockRandomAccessDiagonalMatrix::Invert
    }
  }
  Op precond = [&](const Vec& r, Vec& z) {
    if (o.preconditioner_type == CX_JACOBI) BlockDiagMultiply(jac, bs.cols, 0, 0, r.data(), z.data());
    else for (size_t i = 0; i < r.size(); ++i) z[i] += r[i];  // IdentityPreconditioner
  };

  Vec zrows(num_rows);
  // CgnrLinearOperator::RightMultiplyAndAccumulate (cgnr_solver.cc:98-114)
  Op lhs = [&](const Vec& xx, Vec& y) {
    std::fill(zrows.begin(), zrows.end(), 0.0);
    orc_right_multiply(s, values, xx.data(), zrows.data());
    if (!comm.active()) {
      orc_left_multiply(s, values, zrows.data(), y.data());
    } else {
      Vec t(num_cols, 0.0);
      orc_left_multiply(s, values, zrows.data(), t.data());
      comm.Sum(t.data() + shared0, n_shared);
      for (int i = 0; i < num_cols; ++i) y[i] += t[i];
    }
    if (D) for (int i = 0; i < num_cols; ++i) y[i] = y[i] + D[i] * D[i] * xx[i];
  };
  DotFn dot = [&](const Vec& a, const Vec& c) {
    if (!comm.active()) return PlainDot(a, c);
    // local point part summed over ranks + replicated camera part counted once
    double parts[2] = {0.0, 0.0};
    for (int i = 0; i < shared0; ++i) parts[0] += a[i] * c[i];
    for (int i = shared0; i < num_cols; ++i) parts[1] += a[i] * c[i];
    comm.Sum(&parts[0], 1);
    return parts[0] + parts[1];
  };

  Vec rhs(num_cols, 0.0);
  orc_left_multiply(s, values, b, rhs.data());
  comm.Sum(rhs.data() + shared0, n_shared);
  Vec sol(num_cols, 0.0);
  CGOptions cg;
  cg.min_num_iterations = o.min_num_iterations;
  cg.max_num_iterations = o.max_num_iterations;
  cg.residual_reset_period = o.residual_reset_period;
  cg.q_tolerance = q_tol;
  cg.r_tolerance = r_tol;
  cx_summary summary = ConjugateGradients(cg, lhs, rhs, precond, dot, sol);
  std::copy(sol.begin(), sol.end(), x);
  return summary;
}

// iterative_schur_complement_solver.cc:64-199
static cx_summary SolveIterativeSchur(const cx_block_structure* s, const double* values,
                                      const double* b, const double* D, const cx_solver_options& o,
                                      double r_tol, double q_tol, double* x, const Comm& comm,
                                      int threads) {
  const int nelim = o.num_eliminate_blocks;
  const bool need_ftf = o.use_spse_initialization || o.preconditioner_type == CX_JACOBI ||
                        o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION;
  ISC isc(s, values, nelim, threads, need_ftf, comm);
  MarkSolveStart();
  isc.Init(D, b);
  cx_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  const int n = isc.num_rows();
  if (isc.A.bs.C - nelim == 0) {
    summary.termination_type = CX_SUCCESS;
    isc.BackSubstitute(nullptr, x);
    return summary;
  }
  Vec sol(n, 0.0);

  // PowerSeriesExpansionPreconditioner::RightMultiplyAndAccumulate
  // (power_series_expansion_preconditioner.cc:57-84)
  auto spse = [&](const Vec& xin, Vec& y, int max_iter, double tol) {
    Vec series_term(n), previous(n);
    std::fill(y.begin(), y.end(), 0.0);
    isc.FtFInvMultiply(xin.data(), y.data());
    previous = y;
    const double norm_threshold = tol * std::sqrt(PlainDot(y, y));
    for (int i = 1;; ++i) {
      std::fill(series_term.begin(), series_term.end(), 0.0);
      isc.InversePowerSeriesOperatorRightMultiplyAccumulate(previous.data(), series_term.data());
      for (int k = 0; k < n; ++k) y[k] += series_term[k];
      if (i >= max_iter || std::sqrt(PlainDot(series_term, series_term)) < norm_threshold) break;
      std::swap(previous, series_term);
    }
  };
  if (o.use_spse_initialization) spse(isc.rhs, sol, o.max_num_spse_iterations, o.spse_tolerance);

  // CreatePreconditioner :159-199
  std::vector<Vec> sj;  // SCHUR_JACOBI block diagonal inverse
  if (o.preconditioner_type == CX_SCHUR_JACOBI) {
    Eliminator el(s, values, nelim);
    auto sizes = el.FBlockSizes();
    size_t total = 0;
    for (int f : sizes) total += size_t(f) * f;
    Vec flat(total);
    DiagonalBRAM m(flat.data(), sizes);
    if (!comm.active()) {
      el.Eliminate(nullptr, D, &m, nullptr, threads);
    } else {
      // shard: eliminate with the camera part of D left out, sum, add it once
      Vec Dmod;
      const double* Duse = nullptr;
      if (D) {
        Dmod.assign(D, D + isc.A.num_cols_e + n);
        std::fill(Dmod.begin() + isc.A.num_cols_e, Dmod.end(), 0.0);
        Duse = Dmod.data();
      }
      el.Eliminate(nullptr, Duse, &m, nullptr, threads);
      comm.Sum(flat.data(), int64_t(total));
      if (D) {
        for (size_t i = 0; i < sizes.size(); ++i) {
          const double* d = D + isc.A.bs.cols[nelim + i].position;
          for (int k = 0; k < sizes[i]; ++k) flat[m.offset[i] + k * sizes[i] + k] += d[k] * d[k];
        }
      }
    }
    sj.resize(sizes.size());
    for (size_t i = 0; i < sizes.size(); ++i) {
      sj[i].assign(flat.begin() + m.offset[i], flat.begin() + m.offset[i + 1]);
      InvertPSD(sj[i].data(), sizes[i]);
    }
  }
  // VisibilityBasedPreconditioner (visibility_based_preconditioner.cc:68-434): the eliminator runs on a storage
  // that only has the block pairs of the preconditioner (GetCell of any other cell is null), then a Cholesky
  // factorisation; CLUSTER_TRIDIAGONAL retries once with the cells between different clusters halved (:354-393).
  // The reference factors with a sparse Cholesky (SuiteSparse & co, not in the tree); dense here.
  Vec vis_factor;
  if (o.preconditioner_type == CX_CLUSTER_JACOBI || o.preconditioner_type == CX_CLUSTER_TRIDIAGONAL) {
    const SumFn shard_sum = [&](double* buf, int64_t count) { comm.Sum(buf, count); };
    const VisibilityStructure vs =
        ComputeVisibilityStructure(s, nelim, o.preconditioner_type, o.visibility_clustering_type, comm.active() ? &shard_sum : nullptr);
    Eliminator el(s, values, nelim);
    auto sizes = el.FBlockSizes();
    Vec m_values(size_t(n) * n, 0.0);
    SubsetBRAM m(m_values.data(), sizes, vs.block_pairs);
    if (!comm.active()) {
      el.Eliminate(nullptr, D, &m, nullptr, threads);
    } else {
      Vec Dmod;
      const double* Duse = nullptr;
      if (D) {
        Dmod.assign(D, D + isc.A.num_cols_e + n);
        std::fill(Dmod.begin() + isc.A.num_cols_e, Dmod.end(), 0.0);
        Duse = Dmod.data();
      }
      el.Eliminate(nullptr, Duse, &m, nullptr, threads);
      comm.Sum(m_values.data(), int64_t(m_values.size()));
      if (D) for (int i = 0; i < n; ++i) m_values[size_t(i) * n + i] += D[isc.A.num_cols_e + i] * D[isc.A.num_cols_e + i];
    }
    vis_factor = m_values;
    bool ok = CholeskyUpper(vis_factor.data(), n, threads);
    if (!ok && o.preconditioner_type == CX_CLUSTER_TRIDIAGONAL) {
      // ScaleOffDiagonalCells :371-393
      for (const auto& bp : vs.block_pairs) {
        if (vs.membership[size_t(bp.first)] == vs.membership[size_t(bp.second)]) continue;
        for (int a = 0; a < sizes[size_t(bp.first)]; ++a)
          for (int c = 0; c < sizes[size_t(bp.second)]; ++c) m_values[size_t(m.layout[size_t(bp.first)] + a) * n + m.layout[size_t(bp.second)] + c] *= 0.5;
      }
      vis_factor = m_values;
      ok = CholeskyUpper(vis_factor.data(), n, threads);
    }
    if (!ok) {
      // IterativeSchurComplementSolver::SolveImpl :115-121
      summary.termination_type = CX_FAILURE;
      SetMessage(&summary, "Preconditioner update failed.");
      return summary;
    }
  }
  Op precond = [&](const Vec& r, Vec& z) {
    switch (o.preconditioner_type) {
      case CX_CLUSTER_JACOBI:
      case CX_CLUSTER_TRIDIAGONAL: {
        // SparseCholesky::Solve: U' y = r, U x = y (the operator overwrites, visibility_based_preconditioner.cc:427-434)
        Vec y(n);
        for (int i = 0; i < n; ++i) {
          double sum = r[i];
          for (int k = 0; k < i; ++k) sum -= vis_factor[size_t(k) * n + i] * y[k];
          y[i] = sum / vis_factor[size_t(i) * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
          double sum = y[i];
          const double* Ui = vis_factor.data() + size_t(i) * n;
          for (int k = i + 1; k < n; ++k) sum -= Ui[k] * z[k];
          z[i] = sum / Ui[i];
        }
        break;
      }
      case CX_JACOBI: isc.FtFInvMultiply(r.data(), z.data()); break;
      case CX_SCHUR_JACOBI: BlockDiagMultiply(sj, isc.A.bs.cols, nelim, isc.A.num_cols_e, r.data(), z.data()); break;
      case CX_SCHUR_POWER_SERIES_EXPANSION: {
        Vec y(n);
        spse(r, y, o.max_num_spse_iterations, 0.0);
        for (int i = 0; i < n; ++i) z[i] += y[i];
        break;
      }
      default: for (int i = 0; i < n; ++i) z[i] += r[i];
    }
  };
  Op lhs = [&](const Vec& xx, Vec& y) {  // LinearOperatorAdapter: y += S x
    Vec t(n);
    isc.RightMultiply(xx.data(), t.data());
    for (int i = 0; i < n; ++i) y[i] += t[i];
  };
  CGOptions cg;
  cg.min_num_iterations = o.min_num_iterations;
  cg.max_num_iterations = o.max_num_iterations;
  cg.residual_reset_period = o.residual_reset_period;
  cg.q_tolerance = q_tol;
  cg.r_tolerance = r_tol;
  summary = ConjugateGradients(cg, lhs, isc.rhs, precond, PlainDot, sol);
  if (summary.termination_type != CX_FAILURE && summary.termination_type != CX_FATAL_ERROR)
    isc.BackSubstitute(sol.data(), x);
  return summary;
}

// schur_complement_solver.cc:101-203 (DenseSchurComplementSolver)
static cx_summary SolveDenseSchur(const cx_block_structure* s, const double* values, const double* b,
                                  const double* D, const cx_solver_options& o, double* x,
                                  const Comm& comm, int threads) {
  const int nelim = o.num_eliminate_blocks;
  Eliminator el(s, values, nelim);
  auto sizes = el.FBlockSizes();
  int n = 0;
  for (int f : sizes) n += f;
  const int num_cols = el.bs.num_cols();
  Vec lhs(size_t(n) * n), rhs(n);
  DenseBRAM m(lhs.data(), sizes);
  MarkSolveStart();
  std::fill(x, x + num_cols, 0.0);
  if (!comm.active()) {
    el.Eliminate(b, D, &m, rhs.data(), threads);
  } else {
    Vec Dmod;
    const double* Duse = nullptr;
    if (D) {
      Dmod.assign(D, D + num_cols);
      std::fill(Dmod.begin() + (num_cols - n), Dmod.end(), 0.0);
      Duse = Dmod.data();
    }
    el.Eliminate(b, Duse, &m, rhs.data(), threads);
    comm.Sum(lhs.data(), int64_t(lhs.size()));
    comm.Sum(rhs.data(), n);
    if (D) for (int i = 0; i < n; ++i) lhs[size_t(i) * n + i] += D[num_cols - n + i] * D[num_cols - n + i];
  }
  cx_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  double* reduced = x + num_cols - n;
  if (n == 0) {
    summary.termination_type = CX_SUCCESS;
  } else {
    if (o.use_mixed_precision_solves || o.max_num_refinement_iterations > 0)  // DenseCholesky::Create, dense_cholesky.cc:84-136
      summary.termination_type = orc_dense_cholesky_solve_refined(n, lhs.data(), rhs.data(), reduced, o.use_mixed_precision_solves != 0,
                                                                  std::max(0, o.max_num_refinement_iterations));
    else
      summary.termination_type = orc_dense_cholesky_solve(n, lhs.data(), rhs.data(), reduced);
    summary.num_iterations = 1;
    if (summary.termination_type != CX_SUCCESS) SetMessage(&summary, "Eigen failure. Unable to perform dense Cholesky factorization.");
    else SetMessage(&summary, "Success.");
  }
  if (summary.termination_type == CX_SUCCESS) el.BackSubstitute(b, D, reduced, x, threads);
  return summary;
}

// SparseSchurComplementSolver::InitStorage (schur_complement_solver.cc:224-290): the cell set of the
// block-sparse reduced matrix -- every (i, i), every (i, j), i < j, of f-blocks met in one chunk, and the
// f x f cells of rows without an e-block; std::set order = lexicographic.
static std::vector<std::pair<int, int>> SparseSchurCells(const cx_block_structure* s, int num_eliminate_blocks) {
  BS bs(s);
  const int num_f = bs.C - num_eliminate_blocks;
  // The reference inserts into a std::set<pair<int,int>>; the set's iteration order is lexicographic, which is
  // what sorting and de-duplicating every block row's list gives as well (and scales to 10^8 insertions).
  std::vector<std::vector<int>> row_cols(static_cast<size_t>(num_f));
  for (int i = 0; i < num_f; ++i) row_cols[size_t(i)].push_back(i);
  auto compact = [](std::vector<int>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  int r = 0;
  while (r < bs.R) {
    const int e_block_id = bs.cells[bs.rcb[r]].block_id;
    if (e_block_id >= num_eliminate_blocks) break;
    std::vector<int> f_blocks;
    for (; r < bs.R; ++r) {
      if (bs.cells[bs.rcb[r]].block_id != e_block_id) break;
      for (int c = bs.rcb[r] + 1; c < bs.rcb[r + 1]; ++c) f_blocks.push_back(bs.cells[c].block_id - num_eliminate_blocks);
    }
    compact(f_blocks);
    for (size_t i = 0; i < f_blocks.size(); ++i) {
      std::vector<int>& row = row_cols[size_t(f_blocks[i])];
      for (size_t j = i + 1; j < f_blocks.size(); ++j) row.push_back(f_blocks[j]);
      if (row.size() > 4096 && row.size() > 4 * size_t(num_f)) compact(row);
    }
  }
  for (; r < bs.R; ++r) {
    for (int i = bs.rcb[r]; i < bs.rcb[r + 1]; ++i) {
      const int b1 = bs.cells[i].block_id - num_eliminate_blocks;
      for (int j = bs.rcb[r]; j < bs.rcb[r + 1]; ++j) {
        const int b2 = bs.cells[j].block_id - num_eliminate_blocks;
        if (b1 <= b2) row_cols[size_t(b1)].push_back(b2);
      }
    }
  }
  std::vector<std::pair<int, int>> cells;
  for (int i = 0; i < num_f; ++i) {
    compact(row_cols[size_t(i)]);
    for (int j : row_cols[size_t(i)]) cells.emplace_back(i, j);
  }
  return cells;
}

// BlockRandomAccessSparseMatrix (block_random_access_sparse_matrix.cc:48-122): the cells of a block-pair set, each
// a contiguous row-major array of its own (row_stride = col_stride = column block size), looked up by block pair
// (a hash map in the reference; a binary search in the sorted row here).
struct SparseBRAM : BRAM {
  std::vector<int> sizes, pos;
  std::vector<int64_t> row_begin;   // cells of block row i: [row_begin[i], row_begin[i+1])
  std::vector<int> cell_col;
  std::vector<int64_t> cell_off;
  std::vector<double> values;
  int n = 0;
  SparseBRAM(const std::vector<int>& block_sizes, const std::vector<std::pair<int, int>>& cells) : sizes(block_sizes) {
    for (int f : sizes) { pos.push_back(n); n += f; }
    row_begin.assign(sizes.size() + 1, 0);
    for (const auto& c : cells) row_begin[size_t(c.first) + 1]++;
    std::partial_sum(row_begin.begin(), row_begin.end(), row_begin.begin());
    int64_t off = 0;
    for (const auto& c : cells) {  // lexicographic: already grouped by row, ascending column
      cell_col.push_back(c.second);
      cell_off.push_back(off);
      off += int64_t(sizes[size_t(c.first)]) * sizes[size_t(c.second)];
    }
    values.assign(size_t(off), 0.0);
  }
  double* Find(int i, int j) {
    const auto first = cell_col.begin() + row_begin[size_t(i)], last = cell_col.begin() + row_begin[size_t(i) + 1];
    const auto it = std::lower_bound(first, last, j);
    if (it == last || *it != j) return nullptr;
    return values.data() + cell_off[size_t(it - cell_col.begin())];
  }
  double* GetCell(int i, int j, int* r, int* c, int* rs, int* cs) override {
    double* v = Find(i, j);
    if (!v) return nullptr;
    *r = 0; *c = 0; *rs = sizes[size_t(i)]; *cs = sizes[size_t(j)];
    return v;
  }
  void SetZero() override { std::fill(values.begin(), values.end(), 0.0); }
  int num_rows() const override { return n; }
};

static double g_sparse_chol_stats[8] = {0};

// SparseSchurComplementSolver with a sparse direct reduced solve (schur_complement_solver.cc:101-159, 224-335): the
// eliminator writes into the block-sparse storage of InitStorage, the reduced system is factored by the block-sparse
// Cholesky of orc_sparse_chol.h (the stand-in for SuiteSparse, third party).
static cx_summary SolveSparseSchur(const cx_block_structure* s, const double* values, const double* b,
                                   const double* D, const cx_solver_options& o, double* x,
                                   const Comm& comm, int threads) {
  // use_mixed_precision_solves / max_num_refinement_iterations (SparseCholesky::Create, sparse_cholesky.cc:45-118: a float
  // factorisation and / or RefinedSparseCholesky): the float factor and the refinement loop are restated once, on the dense
  // reduced matrix (orc_dense_cholesky_solve_refined) -- the same mathematics as a float sparse factor of the same S
  if (o.use_mixed_precision_solves || o.max_num_refinement_iterations > 0) return SolveDenseSchur(s, values, b, D, o, x, comm, threads);
  const int nelim = o.num_eliminate_blocks;
  Eliminator el(s, values, nelim);
  auto sizes = el.FBlockSizes();
  const int num_cols = el.bs.num_cols();
  const auto cells = SparseSchurCells(s, nelim);
  SparseBRAM m(sizes, cells);
  const int n = m.n;
  Vec rhs(static_cast<size_t>(n));
  BlockSparseCholesky chol;
  double t0 = omp_get_wtime();
  if (n > 0) chol.Analyze(sizes, cells);
  g_sparse_chol_stats[0] = omp_get_wtime() - t0;
  MarkSolveStart();
  std::fill(x, x + num_cols, 0.0);
  t0 = omp_get_wtime();
  if (!comm.active()) {
    el.Eliminate(b, D, &m, rhs.data(), threads);
  } else {
    Vec Dmod;
    const double* Duse = nullptr;
    if (D) {
      Dmod.assign(D, D + num_cols);
      std::fill(Dmod.begin() + (num_cols - n), Dmod.end(), 0.0);
      Duse = Dmod.data();
    }
    el.Eliminate(b, Duse, &m, rhs.data(), threads);
    comm.Sum(m.values.data(), int64_t(m.values.size()));
    comm.Sum(rhs.data(), n);
    if (D)
      for (size_t i = 0; i < sizes.size(); ++i) {
        double* v = m.Find(int(i), int(i));
        for (int k = 0; k < sizes[i]; ++k) v[k * sizes[i] + k] += D[num_cols - n + m.pos[i] + k] * D[num_cols - n + m.pos[i] + k];
      }
  }
  g_sparse_chol_stats[1] = omp_get_wtime() - t0;
  cx_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  double* reduced = x + num_cols - n;
  if (n == 0) {
    summary.termination_type = CX_SUCCESS;
  } else {
    t0 = omp_get_wtime();
    const bool ok = chol.Factor([&](int i, int j) { return static_cast<const double*>(m.Find(i, j)); }, threads);
    g_sparse_chol_stats[2] = omp_get_wtime() - t0;
    summary.num_iterations = 1;
    if (!ok) {
      summary.termination_type = CX_FAILURE;
      SetMessage(&summary, "Sparse Cholesky factorization failed: the reduced camera matrix is not positive definite.");
    } else {
      t0 = omp_get_wtime();
      chol.Solve(rhs.data(), reduced, threads);
      g_sparse_chol_stats[3] = omp_get_wtime() - t0;
      summary.termination_type = CX_SUCCESS;
      SetMessage(&summary, "Success.");
    }
    g_sparse_chol_stats[4] = double(chol.num_factor_blocks());
    g_sparse_chol_stats[5] = chol.factor_flops();
    g_sparse_chol_stats[6] = double(chol.num_heights());
    g_sparse_chol_stats[7] = double(cells.size());
  }
  if (summary.termination_type == CX_SUCCESS) el.BackSubstitute(b, D, reduced, x, threads);
  return summary;
}

// ITERATIVE_SCHUR with use_explicit_schur_complement: SparseSchurComplementSolver (linear_solver.cc:111-116)
// with SolveReducedLinearSystemUsingConjugateGradients (schur_complement_solver.cc:337-420).  The eliminator
// writes the same cells into a dense matrix here (identical arithmetic per cell); the operator walks the
// InitStorage cell list as BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate does
// (block_random_access_sparse_matrix.cc:124-163).
static cx_summary SolveExplicitSchur(const cx_block_structure* s, const double* values, const double* b,
                                     const double* D, const cx_solver_options& o, double r_tol, double q_tol,
                                     double* x, const Comm& comm, int threads) {
  const int nelim = o.num_eliminate_blocks;
  Eliminator el(s, values, nelim);
  auto sizes = el.FBlockSizes();
  int n = 0;
  std::vector<int> pos;
  for (int f : sizes) { pos.push_back(n); n += f; }
  const int num_cols = el.bs.num_cols();
  Vec lhs(size_t(n) * n), rhs(n);
  DenseBRAM m(lhs.data(), sizes);
  const auto cells = SparseSchurCells(s, nelim);
  MarkSolveStart();
  std::fill(x, x + num_cols, 0.0);
  if (!comm.active()) {
    el.Eliminate(b, D, &m, rhs.data(), threads);
  } else {
    Vec Dmod;
    const double* Duse = nullptr;
    if (D) {
      Dmod.assign(D, D + num_cols);
      std::fill(Dmod.begin() + (num_cols - n), Dmod.end(), 0.0);
      Duse = Dmod.data();
    }
    el.Eliminate(b, Duse, &m, rhs.data(), threads);
    comm.Sum(lhs.data(), int64_t(lhs.size()));
    comm.Sum(rhs.data(), n);
    if (D) for (int i = 0; i < n; ++i) lhs[size_t(i) * n + i] += D[num_cols - n + i] * D[num_cols - n + i];
  }
  cx_summary summary;
  std::memset(&summary, 0, sizeof(summary));
  if (n == 0) {
    summary.termination_type = CX_SUCCESS;
    SetMessage(&summary, "Success.");
    el.BackSubstitute(b, D, x + num_cols, x, threads);
    return summary;
  }
  // block Jacobi preconditioner: diagonal cells, BlockRandomAccessDiagonalMatrix::Invert (LLT per block)
  std::vector<Vec> pre(sizes.size());
  for (size_t i = 0; i < sizes.size(); ++i) {
    const int f = sizes[i];
    pre[i].resize(size_t(f) * f);
    for (int a = 0; a < f; ++a)
      for (int c = 0; c < f; ++c) pre[i][a * f + c] = lhs[size_t(pos[i] + a) * n + pos[i] + c];
    if (!InvertPSD(pre[i].data(), f)) {
      summary.termination_type = CX_FAILURE;
      SetMessage(&summary, "Preconditioner update failed.");
      return summary;
    }
  }
  Op op = [&](const Vec& xx, Vec& y) {
    for (const auto& cell : cells) {
      const int r = cell.first, c = cell.second;
      const int fr = sizes[r], fc = sizes[c];
      for (int a = 0; a < fr; ++a) {
        double sum = 0.0;
        for (int k = 0; k < fc; ++k) sum += lhs[size_t(pos[r] + a) * n + pos[c] + k] * xx[pos[c] + k];
        y[pos[r] + a] += sum;
      }
      if (r == c) continue;
      for (int k = 0; k < fc; ++k) {
        double sum = 0.0;
        for (int a = 0; a < fr; ++a) sum += lhs[size_t(pos[r] + a) * n + pos[c] + k] * xx[pos[r] + a];
        y[pos[c] + k] += sum;
      }
    }
  };
  Op precond = [&](const Vec& r, Vec& z) {
    for (size_t i = 0; i < sizes.size(); ++i) {
      const int f = sizes[i];
      for (int a = 0; a < f; ++a) {
        double sum = 0.0;
        for (int k = 0; k < f; ++k) sum += pre[i][a * f + k] * r[pos[i] + k];
        z[pos[i] + a] += sum;
      }
    }
  };
  CGOptions cg;
  cg.min_num_iterations = o.min_num_iterations;
  cg.max_num_iterations = o.max_num_iterations;
  cg.residual_reset_period = o.residual_reset_period;
  cg.q_tolerance = q_tol;
  cg.r_tolerance = r_tol;
  Vec sol(n, 0.0);
  summary = ConjugateGradients(cg, op, rhs, precond, PlainDot, sol);
  double* reduced = x + num_cols - n;
  std::copy(sol.begin(), sol.end(), reduced);
  if (summary.termination_type != CX_FAILURE && summary.termination_type != CX_FATAL_ERROR)
    el.BackSubstitute(b, D, reduced, x, threads);
  return summary;
}

static int Solve(const cx_block_structure* bs, const double* values, const double* b, const double* D,
                 const cx_solver_options* o, double r_tol, double q_tol, double* x, cx_summary* out,
                 const Comm& comm) {
  const int threads = orc_get_num_threads();
  cx_summary s;
  MarkSolveStart();
  switch (o->type) {
    case CX_CGNR: s = SolveCgnr(bs, values, b, D, *o, r_tol, q_tol, x, comm, threads); break;
    case CX_ITERATIVE_SCHUR:
      s = o->use_explicit_schur_complement ? SolveExplicitSchur(bs, values, b, D, *o, r_tol, q_tol, x, comm, threads)
                                           : SolveIterativeSchur(bs, values, b, D, *o, r_tol, q_tol, x, comm, threads);
      break;
    case CX_DENSE_SCHUR: s = SolveDenseSchur(bs, values, b, D, *o, x, comm, threads); break;
    case CX_SPARSE_SCHUR: s = SolveSparseSchur(bs, values, b, D, *o, x, comm, threads); break;
    default: return -1;
  }
  g_last_solve_seconds = omp_get_wtime() - g_solve_t0;
  if (out) *out = s;
  return 0;
}

}  // namespace orc

using namespace orc;

extern "C" {

double orc_last_solve_seconds(void) { return g_last_solve_seconds; }

// last SPARSE_SCHUR solve: seconds of {analysis, eliminate, factor, triangular solves}, then blocks of L, factor
// flops, elimination-tree heights, cells of S
void orc_sparse_schur_stats(double* out8) { std::copy(g_sparse_chol_stats, g_sparse_chol_stats + 8, out8); }

int64_t orc_schur_sparse_structure(const cx_block_structure* bs, int num_eliminate_blocks, int32_t* cell_row,
                                   int32_t* cell_col, int64_t capacity) {
  const auto cells = SparseSchurCells(bs, num_eliminate_blocks);
  for (int64_t i = 0; i < std::min<int64_t>(capacity, int64_t(cells.size())); ++i) {
    if (cell_row) cell_row[i] = cells[size_t(i)].first;
    if (cell_col) cell_col[i] = cells[size_t(i)].second;
  }
  return int64_t(cells.size());
}

int orc_solve(const cx_block_structure* bs, const double* values, const double* b, const double* D,
              const cx_solver_options* o, double r_tol, double q_tol, double* x, cx_summary* out) {
  return Solve(bs, values, b, D, o, r_tol, q_tol, x, out, Comm());
}

int orc_solve_sharded(const cx_block_structure* bs, const double* values, const double* b,
                      const double* D, const cx_solver_options* o, double r_tol, double q_tol,
                      double* x, cx_summary* out, orc_allreduce_fn fn, void* user) {
  Comm c;
  c.fn = fn;
  c.user = user;
  return Solve(bs, values, b, D, o, r_tol, q_tol, x, out, c);
}

int orc_implicit_schur_multiply(const cx_block_structure* bs, const double* values, const double* D,
                                const double* b, int nelim, const double* x, double* y, double* rhs) {
  ISC isc(bs, values, nelim, orc_get_num_threads(), false, Comm());
  isc.Init(D, b);
  if (x && y) isc.RightMultiply(x, y);
  if (rhs && b) std::copy(isc.rhs.begin(), isc.rhs.end(), rhs);
  return 0;
}

int orc_block_diagonal_inverses(const cx_block_structure* bs, const double* values, const double* D,
                                int nelim, double* ete_inv, double* ftf_inv) {
  ISC isc(bs, values, nelim, orc_get_num_threads(), ftf_inv != nullptr, Comm());
  isc.Init(D, nullptr);
  if (ete_inv) for (auto& m : isc.ete_inv) { std::copy(m.begin(), m.end(), ete_inv); ete_inv += m.size(); }
  if (ftf_inv) for (auto& m : isc.ftf_inv) { std::copy(m.begin(), m.end(), ftf_inv); ftf_inv += m.size(); }
  return 0;
}

int orc_cg_dense(int n, const double* A, const double* b, double* x, int min_num_iterations,
                 int max_num_iterations, int residual_reset_period, double r_tolerance,
                 double q_tolerance, cx_summary* out) {
  Op lhs = [&](const Vec& xx, Vec& y) { MatVec(A, n, n, xx.data(), y.data(), 1); };
  Op id = [&](const Vec& r, Vec& z) { for (int i = 0; i < n; ++i) z[i] += r[i]; };
  Vec rhs(b, b + n), sol(x, x + n);
  CGOptions cg;
  cg.min_num_iterations = min_num_iterations;
  cg.max_num_iterations = max_num_iterations;
  cg.residual_reset_period = residual_reset_period;
  cg.r_tolerance = r_tolerance;
  cg.q_tolerance = q_tolerance;
  cx_summary s = ConjugateGradients(cg, lhs, rhs, id, PlainDot, sol);
  std::copy(sol.begin(), sol.end(), x);
  if (out) *out = s;
  return 0;
}

}  // extern "C"
