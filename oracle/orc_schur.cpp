// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// extern "C" entry points of the Schur eliminator restatement (orc_schur.h).
#include "orc_schur.h"

using namespace orc;

extern "C" {

int orc_schur_eliminate_dense(const cx_block_structure* s, const double* values, const double* b,
                              const double* D, int nelim, double* lhs, double* rhs) {
  Eliminator el(s, values, nelim);
  DenseBRAM m(lhs, el.FBlockSizes());
  el.Eliminate(b, D, &m, rhs, orc_get_num_threads());
  return 0;
}

int orc_schur_eliminate_diagonal(const cx_block_structure* s, const double* values, const double* D,
                                 int nelim, double* blocks) {
  Eliminator el(s, values, nelim);
  DiagonalBRAM m(blocks, el.FBlockSizes());
  el.Eliminate(nullptr, D, &m, nullptr, orc_get_num_threads());
  return 0;
}

int orc_schur_back_substitute(const cx_block_structure* s, const double* values, const double* b,
                              const double* D, int nelim, const double* z, double* x) {
  Eliminator el(s, values, nelim);
  el.BackSubstitute(b, D, z, x, orc_get_num_threads());
  return 0;
}

int orc_dense_cholesky_solve(int n, double* lhs, const double* rhs, double* x) {
  if (!CholeskyUpper(lhs, n, orc_get_num_threads())) return CX_FAILURE;
  // U' y = rhs ; U x = y
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double sum = rhs[i];
    for (int k = 0; k < i; ++k) sum -= lhs[size_t(k) * n + i] * y[k];
    y[i] = sum / lhs[size_t(i) * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double sum = y[i];
    const double* Ui = lhs + size_t(i) * n;
    for (int k = i + 1; k < n; ++k) sum -= Ui[k] * x[k];
    x[i] = sum / Ui[i];
  }
  return CX_SUCCESS;
}


// use_mixed_precision_solves / max_num_refinement_iterations on the dense reduced solve: DenseCholesky::Create
// (dense_cholesky.cc:84-136) picks FloatEigenDenseCholesky (dense_cholesky.cc:180-204: the matrix cast to float, LLT in
// float, right-hand side cast to float, solution cast back) when use_float != 0, and wraps whichever factorisation it picked
// into RefinedDenseCholesky (dense_cholesky.cc:322-347) when refinements > 0: Solve, then DenseIterativeRefiner::Refine
// (iterative_refiner.cc:83-99): `refinements` times residual = rhs - lhs * solution in double, solution += factor^-1 residual.
// lhs: upper triangle of the row-major matrix, kept intact (the factor is a copy).
int orc_dense_cholesky_solve_refined(int n, const double* lhs, const double* rhs, double* x, int use_float, int refinements) {
  const int threads = orc_get_num_threads();
  std::vector<double> Ud;
  std::vector<float> Uf;
  if (use_float) {
    Uf.resize(size_t(n) * n);
    for (size_t i = 0; i < Uf.size(); ++i) Uf[i] = float(lhs[i]);
    if (!CholeskyUpperT<float>(Uf.data(), n, threads)) return CX_FAILURE;
  } else {
    Ud.assign(lhs, lhs + size_t(n) * n);
    if (!CholeskyUpperT<double>(Ud.data(), n, threads)) return CX_FAILURE;
  }
  auto solve = [&](const double* r, double* out) {
    if (use_float) {
      std::vector<float> y(n), z(n);
      for (int i = 0; i < n; ++i) {
        float sum = float(r[i]);
        for (int k = 0; k < i; ++k) sum -= Uf[size_t(k) * n + i] * y[k];
        y[i] = sum / Uf[size_t(i) * n + i];
      }
      for (int i = n - 1; i >= 0; --i) {
        float sum = y[i];
        for (int k = i + 1; k < n; ++k) sum -= Uf[size_t(i) * n + k] * z[k];
        z[i] = sum / Uf[size_t(i) * n + i];
      }
      for (int i = 0; i < n; ++i) out[i] = double(z[i]);
    } else {
      std::vector<double> y(n);
      for (int i = 0; i < n; ++i) {
        double sum = r[i];
        for (int k = 0; k < i; ++k) sum -= Ud[size_t(k) * n + i] * y[k];
        y[i] = sum / Ud[size_t(i) * n + i];
      }
      for (int i = n - 1; i >= 0; --i) {
        double sum = y[i];
        for (int k = i + 1; k < n; ++k) sum -= Ud[size_t(i) * n + k] * out[k];
        out[i] = sum / Ud[size_t(i) * n + i];
      }
    }
  };
  solve(rhs, x);
  std::vector<double> residual(n), correction(n);
  for (int it = 0; it < refinements; ++it) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; ++i) {  // symmetric product from the upper triangle
      double sum = 0.0;
      for (int k = 0; k < i; ++k) sum += lhs[size_t(k) * n + i] * x[k];
      for (int k = i; k < n; ++k) sum += lhs[size_t(i) * n + k] * x[k];
      residual[i] = rhs[i] - sum;
    }
    solve(residual.data(), correction.data());
    for (int i = 0; i < n; ++i) x[i] += correction[i];
  }
  return CX_SUCCESS;
}

}  // extern "C"
