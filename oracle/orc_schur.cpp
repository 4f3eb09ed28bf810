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

}  // extern "C"
