// Exercises CxLinearSolver the way the reference's own callers and tests do:
//   * the LM strategy call sequence   levenberg_marquardt_strategy.cc:97-133
//     (InvalidateArray, Solve(J, residuals, {D, q_tol = eta, r_tol = -1}), IsArrayValid, negate)
//   * schur_complement_solver_test.cc / iterative_schur_complement_solver_test.cc:
//     every solver type against the normal equations, |dx| / n < 1e-10
// on a synthetic bundle-adjustment Jacobian in the spirit of
// CreateFakeBundleAdjustmentJacobian (fake_bundle_adjustment_jacobian.cc:44-97).
// Needs a gfx950 device.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "cx_linear_solver.h"

using namespace ceres;
using namespace ceres::internal;

static std::unique_ptr<BlockSparseMatrix> FakeBundleAdjustmentJacobian(int num_cameras, int num_points, double visibility,
                                                                       std::mt19937& prng) {
  auto* bs = new CompressedRowBlockStructure;
  int pos = 0;
  for (int j = 0; j < num_points; ++j) { bs->cols.emplace_back(3, pos); pos += 3; }
  for (int i = 0; i < num_cameras; ++i) { bs->cols.emplace_back(9, pos); pos += 9; }
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::vector<std::pair<int, int>> obs;  // (point, camera), point-major
  for (int j = 0; j < num_points; ++j) {
    int seen = 0;
    for (int i = 0; i < num_cameras; ++i) {
      if (uni(prng) < visibility || (seen < 2 && i >= num_cameras - 2 + seen)) { obs.emplace_back(j, i); ++seen; }
    }
  }
  const int O = int(obs.size());
  int e_pos = 0, f_pos = 6 * O, row_pos = 0;
  for (auto& o : obs) {
    bs->rows.emplace_back(2);
    CompressedRow& row = bs->rows.back();
    row.block = Block(2, row_pos);
    row.cells[0] = Cell(o.first, e_pos);
    row.cells[1] = Cell(num_points + o.second, f_pos);
    row_pos += 2; e_pos += 6; f_pos += 18;
  }
  auto A = std::make_unique<BlockSparseMatrix>(bs);
  std::normal_distribution<double> normal(0.0, 1.0);
  for (int64_t i = 0; i < A->num_nonzeros(); ++i) A->mutable_values()[i] = normal(prng);
  return A;
}

// y = (J'J + D'D) x - J'b computed with plain host loops (independent of the library)
static double NormalEquationResidual(const BlockSparseMatrix& A, const double* b, const double* D, const double* x) {
  const auto* bs = A.block_structure();
  std::vector<double> Jx(A.num_rows(), 0.0), g(A.num_cols(), 0.0);
  for (auto& row : bs->rows)
    for (auto& cell : row.cells) {
      const Block& col = bs->cols[cell.block_id];
      for (int i = 0; i < row.block.size; ++i)
        for (int j = 0; j < col.size; ++j) Jx[row.block.position + i] += A.values()[cell.position + i * col.size + j] * x[col.position + j];
    }
  for (int i = 0; i < A.num_rows(); ++i) Jx[i] -= b[i];
  for (auto& row : bs->rows)
    for (auto& cell : row.cells) {
      const Block& col = bs->cols[cell.block_id];
      for (int i = 0; i < row.block.size; ++i)
        for (int j = 0; j < col.size; ++j) g[col.position + j] += A.values()[cell.position + i * col.size + j] * Jx[row.block.position + i];
    }
  double n2 = 0.0;
  for (int j = 0; j < A.num_cols(); ++j) { const double v = g[j] + D[j] * D[j] * x[j]; n2 += v * v; }
  return std::sqrt(n2);
}

int main() {
  std::mt19937 prng(5489u);
  const int kCameras = 12, kPoints = 300;
  auto A = FakeBundleAdjustmentJacobian(kCameras, kPoints, 0.4, prng);
  std::normal_distribution<double> normal(0.0, 1.0);
  std::vector<double> b(A->num_rows()), D(A->num_cols());
  for (auto& v : b) v = normal(prng);
  for (auto& v : D) v = 0.5 + std::abs(normal(prng));
  std::printf("J: %d x %d, %lld non-zeros\n", A->num_rows(), A->num_cols(), (long long)A->num_nonzeros());
  int failures = 0;
  std::vector<double> reference;
  const std::vector<double> zero(A->num_cols(), 0.0);
  const double norm_rhs = NormalEquationResidual(*A, b.data(), D.data(), zero.data());  // |J'b|
  struct Case { LinearSolverType type; PreconditionerType pre; const char* name; };
  const Case cases[] = {{DENSE_SCHUR, IDENTITY, "DENSE_SCHUR"}, {SPARSE_SCHUR, IDENTITY, "SPARSE_SCHUR"},
                        {ITERATIVE_SCHUR, JACOBI, "ITERATIVE_SCHUR+JACOBI"}, {ITERATIVE_SCHUR, SCHUR_JACOBI, "ITERATIVE_SCHUR+SCHUR_JACOBI"},
                        {CGNR, JACOBI, "CGNR+JACOBI"}};
  for (const Case& c : cases) {
    LinearSolver::Options options;
    options.type = c.type;
    options.preconditioner_type = c.pre;
    options.elimination_groups = {kPoints, kCameras};
    options.min_num_iterations = 0;
    options.max_num_iterations = A->num_cols();
    CxLinearSolver solver(options);
    LinearSolver::PerSolveOptions ps;
    ps.D = D.data();
    ps.r_tolerance = 1e-13;   // run to convergence, as the reference's solver tests do
    ps.q_tolerance = 0.0;
    std::vector<double> x(A->num_cols());
    InvalidateArray(A->num_cols(), x.data());
    LinearSolver::Summary s = solver.Solve(A.get(), b.data(), ps, x.data());
    const bool valid = IsArrayValid(A->num_cols(), x.data());
    const double res = NormalEquationResidual(*A, b.data(), D.data(), x.data());
    double diff = 0.0;
    if (reference.empty()) reference = x;
    for (size_t i = 0; i < x.size(); ++i) diff += (x[i] - reference[i]) * (x[i] - reference[i]);
    diff = std::sqrt(diff) / x.size();
    const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && valid && res < 1e-9 * norm_rhs && diff < 1e-10;
    std::printf("%-30s %s  iterations %3d  |normal eq residual| %.2e  |x - x_dense_schur|/n %.2e  (%s)\n", c.name,
                ok ? "ok  " : "FAIL", s.num_iterations, res, diff, s.message.c_str());
    if (!ok) ++failures;
    if (solver.Statistics().count("LinearSolver::Solve") != 1) ++failures;
  }
  // the LM call: truncated solve with q_tolerance = eta, step = -x
  {
    LinearSolver::Options options;
    options.type = ITERATIVE_SCHUR;
    options.preconditioner_type = JACOBI;
    options.elimination_groups = {kPoints, kCameras};
    options.min_num_iterations = 0;
    options.max_num_iterations = 500;
    CxLinearSolver solver(options);
    LinearSolver::PerSolveOptions ps;
    ps.D = D.data();
    ps.q_tolerance = 0.1;
    ps.r_tolerance = -1.0;
    std::vector<double> step(A->num_cols());
    InvalidateArray(A->num_cols(), step.data());
    LinearSolver::Summary s = solver.Solve(A.get(), b.data(), ps, step.data());
    const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && IsArrayValid(A->num_cols(), step.data());
    for (auto& v : step) v = -v;
    std::printf("LM-style truncated solve        %s  iterations %3d  (%s)\n", ok ? "ok  " : "FAIL", s.num_iterations, s.message.c_str());
    if (!ok) ++failures;
  }
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
