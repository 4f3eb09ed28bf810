// Dynamic-block-size solvers (the Eigen::Dynamic instantiations of the reference).
#include "cx_internal.h"

struct cx_solver;

int cxg_solve(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol, double q_tol, double* x,
              cx_summary* summary) {
  (void)S; (void)A; (void)b; (void)D; (void)r_tol; (void)q_tol; (void)x; (void)summary;
  cx_set_error("solver for matrices outside the static <2,3,9> layout is not built yet");
  return CX_ERR_UNSUPPORTED;
}
