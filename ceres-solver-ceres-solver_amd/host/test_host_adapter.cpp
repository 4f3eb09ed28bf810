// Exercises the drop-in adapters the way the reference's own callers and tests do.
//
//  0. A signature table: every virtual the adapters override, every Options field they read and every enumerator
//     they switch on is pinned by static_assert to the type / value the reference declares (file:line beside each
//     line), so the adapters cannot drift away from the interface they claim to implement.
//  1. CxLinearSolver on a host BlockSparseMatrix: every solver type against the normal equations
//     (schur_complement_solver_test.cc:186-227 / iterative_schur_complement_solver_test.cc: |dx| / n < 1e-10) and the
//     LM call sequence (levenberg_marquardt_strategy.cc:97-133: InvalidateArray, Solve(J, residuals, {D, q_tol = eta,
//     r_tol = -1}), IsArrayValid, negate), on a Jacobian in the spirit of CreateFakeBundleAdjustmentJacobian
//     (fake_bundle_adjustment_jacobian.cc:44-97).
//  2. Evaluator -> LM strategy -> LinearSolver through the mirrored virtuals only: a bundle-adjustment Program goes
//     through CxBalEvaluator::TryCreate (the factory hook), the Jacobian is the CxDeviceJacobian it creates, and a
//     trust-region loop written against Evaluator / SparseMatrix / LinearSolver (the calls of
//     trust_region_minimizer.cc:246-313, 381-463 and levenberg_marquardt_strategy.cc:69-156) minimises it with J never
//     leaving the device.  Checked against (a) the same loop run on a host BlockSparseMatrix that holds a copy of
//     every Jacobian (host products, values uploaded per solve) and (b) cx_minimize, the device-resident minimizer.
// Needs a gfx950 device.  Exit code 0 = all checks passed.
#include <functional>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <type_traits>

#include "cx_bal_evaluator.h"
#include "cx_linear_solver.h"

using namespace ceres;
using namespace ceres::internal;

// ------------------------------------------------------------------------------------------------ 0. signatures
namespace signature_table {
template <typename T, typename U> constexpr bool same = std::is_same<T, U>::value;
using StatsMap = std::map<std::string, CallStatistics>;

// linear_solver.h:338-341  virtual Summary Solve(LinearOperator* A, const double* b, const PerSolveOptions&, double* x) = 0;
static_assert(same<decltype(&LinearSolver::Solve),
                   LinearSolver::Summary (LinearSolver::*)(LinearOperator*, const double*, const LinearSolver::PerSolveOptions&, double*)>);
// linear_solver.h:348-350  virtual std::map<std::string, CallStatistics> Statistics() const
static_assert(same<decltype(&LinearSolver::Statistics), StatsMap (LinearSolver::*)() const>);
// execution_summary.h:45-49
static_assert(same<decltype(CallStatistics::time), absl::Duration> && same<decltype(CallStatistics::calls), int>);
// CxLinearSolver overrides exactly those two
static_assert(std::is_base_of<LinearSolver, CxLinearSolver>::value && !std::is_abstract<CxLinearSolver>::value);
static_assert(same<decltype(&CxLinearSolver::Solve),
                   LinearSolver::Summary (CxLinearSolver::*)(LinearOperator*, const double*, const LinearSolver::PerSolveOptions&, double*)>);
static_assert(same<decltype(&CxLinearSolver::Statistics), StatsMap (CxLinearSolver::*)() const>);
// linear_solver.h:57-74: numeric order the C ABI's cx_termination repeats
static_assert(int(LinearSolverTerminationType::SUCCESS) == CX_SUCCESS && int(LinearSolverTerminationType::NO_CONVERGENCE) == CX_NO_CONVERGENCE &&
              int(LinearSolverTerminationType::FAILURE) == CX_FAILURE && int(LinearSolverTerminationType::FATAL_ERROR) == CX_FATAL_ERROR);
// linear_solver.h:150-230: the Options fields the adapter and the factory patch of INTEGRATION.md read
using O = LinearSolver::Options;
static_assert(same<decltype(O::type), LinearSolverType> && same<decltype(O::preconditioner_type), PreconditionerType> &&
              same<decltype(O::visibility_clustering_type), VisibilityClusteringType> &&
              same<decltype(O::dense_linear_algebra_library_type), DenseLinearAlgebraLibraryType> &&
              same<decltype(O::sparse_linear_algebra_library_type), SparseLinearAlgebraLibraryType> &&
              same<decltype(O::ordering_type), OrderingType> && same<decltype(O::dynamic_sparsity), bool> &&
              same<decltype(O::use_explicit_schur_complement), bool> && same<decltype(O::min_num_iterations), int> &&
              same<decltype(O::max_num_iterations), int> && same<decltype(O::max_num_spse_iterations), int> &&
              same<decltype(O::use_spse_initialization), bool> && same<decltype(O::spse_tolerance), double> &&
              same<decltype(O::num_threads), int> && same<decltype(O::elimination_groups), std::vector<int>> &&
              same<decltype(O::residual_reset_period), int> && same<decltype(O::row_block_size), int> &&
              same<decltype(O::e_block_size), int> && same<decltype(O::f_block_size), int> &&
              same<decltype(O::use_mixed_precision_solves), bool> && same<decltype(O::max_num_refinement_iterations), int> &&
              same<decltype(O::subset_preconditioner_start_row_block), int> && same<decltype(O::context), ContextImpl*>);
// linear_solver.h:232-318, :320-326
static_assert(same<decltype(LinearSolver::PerSolveOptions::D), double*> &&
              same<decltype(LinearSolver::PerSolveOptions::preconditioner), LinearOperator*> &&
              same<decltype(LinearSolver::PerSolveOptions::r_tolerance), double> &&
              same<decltype(LinearSolver::PerSolveOptions::q_tolerance), double>);
static_assert(same<decltype(LinearSolver::Summary::residual_norm), double> && same<decltype(LinearSolver::Summary::num_iterations), int> &&
              same<decltype(LinearSolver::Summary::termination_type), LinearSolverTerminationType> &&
              same<decltype(LinearSolver::Summary::message), std::string>);
// include/ceres/types.h:57-141: enumerator values the adapter switches on
static_assert(DENSE_SCHUR == 3 && SPARSE_SCHUR == 4 && ITERATIVE_SCHUR == 5 && CGNR == 6);
static_assert(IDENTITY == 0 && JACOBI == 1 && SCHUR_JACOBI == 2 && SCHUR_POWER_SERIES_EXPANSION == 3 && CLUSTER_JACOBI == 4 &&
              CLUSTER_TRIDIAGONAL == 5 && SUBSET == 6);
static_assert(CANONICAL_VIEWS == 0 && SINGLE_LINKAGE == 1);

// evaluator.h:95  virtual std::unique_ptr<SparseMatrix> CreateJacobian() const = 0;
static_assert(same<decltype(&Evaluator::CreateJacobian), std::unique_ptr<SparseMatrix> (Evaluator::*)() const>);
// evaluator.h:116-121 (the virtual; the 5-argument overload :127-134 forwards to it)
using EvaluateVirtual = bool (Evaluator::*)(const Evaluator::EvaluateOptions&, const double*, double*, double*, double*, SparseMatrix*);
static_assert(same<decltype(static_cast<EvaluateVirtual>(&Evaluator::Evaluate)), EvaluateVirtual>);
// evaluator.h:146-148, :151, :155, :158, :164-166
static_assert(same<decltype(&Evaluator::Plus), bool (Evaluator::*)(const double*, const double*, double*) const>);
static_assert(same<decltype(&Evaluator::NumParameters), int (Evaluator::*)() const> &&
              same<decltype(&Evaluator::NumEffectiveParameters), int (Evaluator::*)() const> &&
              same<decltype(&Evaluator::NumResiduals), int (Evaluator::*)() const>);
static_assert(same<decltype(&Evaluator::Statistics), StatsMap (Evaluator::*)() const>);
// evaluator.h:64-73, :99-112
static_assert(same<decltype(Evaluator::Options::num_threads), int> && same<decltype(Evaluator::Options::num_eliminate_blocks), int> &&
              same<decltype(Evaluator::Options::linear_solver_type), LinearSolverType> &&
              same<decltype(Evaluator::Options::sparse_linear_algebra_library_type), SparseLinearAlgebraLibraryType> &&
              same<decltype(Evaluator::Options::dynamic_sparsity), bool> && same<decltype(Evaluator::Options::context), ContextImpl*> &&
              same<decltype(Evaluator::Options::evaluation_callback), EvaluationCallback*>);
static_assert(same<decltype(Evaluator::EvaluateOptions::apply_loss_function), bool> &&
              same<decltype(Evaluator::EvaluateOptions::new_evaluation_point), bool>);
static_assert(std::is_base_of<Evaluator, CxBalEvaluator>::value && !std::is_abstract<CxBalEvaluator>::value);

// sparse_matrix.h:66-113 / linear_operator.h:46-86: what CxDeviceJacobian overrides
using RightLeft = void (SparseMatrix::*)(const double*, double*) const;
static_assert(same<decltype(static_cast<RightLeft>(&SparseMatrix::RightMultiplyAndAccumulate)), RightLeft>);   // :72-73
static_assert(same<decltype(static_cast<RightLeft>(&SparseMatrix::LeftMultiplyAndAccumulate)), RightLeft>);    // :76
using Norm1 = void (SparseMatrix::*)(double*) const;
using Norm3 = void (SparseMatrix::*)(double*, ContextImpl*, int) const;
static_assert(same<decltype(static_cast<Norm1>(&SparseMatrix::SquaredColumnNorm)), Norm1> &&                    // :79
              same<decltype(static_cast<Norm3>(&SparseMatrix::SquaredColumnNorm)), Norm3>);                     // :80-82
using Scale1 = void (SparseMatrix::*)(const double*);
using Scale3 = void (SparseMatrix::*)(const double*, ContextImpl*, int);
static_assert(same<decltype(static_cast<Scale1>(&SparseMatrix::ScaleColumns)), Scale1> &&                       // :84
              same<decltype(static_cast<Scale3>(&SparseMatrix::ScaleColumns)), Scale3>);                        // :85-87
static_assert(same<decltype(&SparseMatrix::ToDenseMatrix), void (SparseMatrix::*)(Matrix*) const> &&            // :98
              same<decltype(&SparseMatrix::ToTextFile), void (SparseMatrix::*)(FILE*) const> &&                 // :101
              same<decltype(&SparseMatrix::mutable_values), double* (SparseMatrix::*)()> &&                     // :107
              same<decltype(&SparseMatrix::values), const double* (SparseMatrix::*)() const> &&                 // :108
              same<decltype(&SparseMatrix::num_nonzeros), int (SparseMatrix::*)() const>);                      // :112
static_assert(same<decltype(&LinearOperator::num_rows), int (LinearOperator::*)() const> &&                     // linear_operator.h:85-86
              same<decltype(&LinearOperator::num_cols), int (LinearOperator::*)() const>);
static_assert(std::is_base_of<SparseMatrix, CxDeviceJacobian>::value && !std::is_abstract<CxDeviceJacobian>::value);
static_assert(std::is_final<BlockSparseMatrix>::value);  // block_sparse_matrix.h:60 -- why CxDeviceJacobian is a sibling
// block_structure.h:54-75: Block / Cell are laid out like the C ABI's cx_block / cx_cell
static_assert(sizeof(Block) == sizeof(cx_block) && sizeof(Cell) == sizeof(cx_cell));
}  // namespace signature_table

static int failures = 0;
#define EXPECT(cond, ...)                                   \
  do {                                                      \
    if (!(cond)) {                                          \
      ++failures;                                           \
      std::printf("FAIL %s:%d  %s  ", __FILE__, __LINE__, #cond); \
      std::printf(__VA_ARGS__);                             \
      std::printf("\n");                                    \
    }                                                       \
  } while (0)

// ------------------------------------------------------------------------------------------------ 1. host Jacobian
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

// |(J'J + D'D) x - J'b| with the SparseMatrix virtuals of the host container (independent of the library)
static double NormalEquationResidual(const SparseMatrix& A, const double* b, const double* D, const double* x) {
  std::vector<double> Jx(size_t(A.num_rows()), 0.0), g(size_t(A.num_cols()), 0.0);
  A.RightMultiplyAndAccumulate(x, Jx.data());
  for (int i = 0; i < A.num_rows(); ++i) Jx[size_t(i)] -= b[i];
  A.LeftMultiplyAndAccumulate(Jx.data(), g.data());
  double n2 = 0.0;
  for (int j = 0; j < A.num_cols(); ++j) { const double v = g[size_t(j)] + D[j] * D[j] * x[j]; n2 += v * v; }
  return std::sqrt(n2);
}

static void TestSolversOnHostJacobian() {
  std::mt19937 prng(5489u);
  const int kCameras = 12, kPoints = 300;
  auto A = FakeBundleAdjustmentJacobian(kCameras, kPoints, 0.4, prng);
  std::normal_distribution<double> normal(0.0, 1.0);
  std::vector<double> b(static_cast<size_t>(A->num_rows())), D(static_cast<size_t>(A->num_cols()));
  for (auto& v : b) v = normal(prng);
  for (auto& v : D) v = 0.5 + std::abs(normal(prng));
  std::printf("J: %d x %d, %d non-zeros\n", A->num_rows(), A->num_cols(), A->num_nonzeros());
  std::vector<double> reference;
  const std::vector<double> zero(size_t(A->num_cols()), 0.0);
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
    EXPECT(CxLinearSolver::Supports(options), "%s", c.name);
    std::unique_ptr<LinearSolver> solver = std::make_unique<CxLinearSolver>(options);  // used through the base class only
    LinearSolver::PerSolveOptions ps;
    ps.D = D.data();
    ps.r_tolerance = 1e-13;   // run to convergence, as the reference's solver tests do
    ps.q_tolerance = 0.0;
    std::vector<double> x(static_cast<size_t>(A->num_cols()));
    InvalidateArray(A->num_cols(), x.data());
    LinearSolver::Summary s = solver->Solve(A.get(), b.data(), ps, x.data());
    const bool valid = IsArrayValid(A->num_cols(), x.data());
    const double res = NormalEquationResidual(*A, b.data(), D.data(), x.data());
    double diff = 0.0;
    if (reference.empty()) reference = x;
    for (size_t i = 0; i < x.size(); ++i) diff += (x[i] - reference[i]) * (x[i] - reference[i]);
    diff = std::sqrt(diff) / double(x.size());
    const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && valid && res < 1e-9 * norm_rhs && diff < 1e-10;
    std::printf("%-30s %s  iterations %3d  |normal eq residual| %.2e  |x - x_dense_schur|/n %.2e  (%s)\n", c.name,
                ok ? "ok  " : "FAIL", s.num_iterations, res, diff, s.message.c_str());
    EXPECT(ok, "%s", c.name);
    // TypedLinearSolver's bookkeeping (linear_solver.h:366-380): one timed call
    const auto stats = solver->Statistics();
    EXPECT(stats.count("LinearSolver::Solve") == 1 && stats.at("LinearSolver::Solve").calls == 1 &&
               absl::ToDoubleSeconds(stats.at("LinearSolver::Solve").time) > 0.0, "%s statistics", c.name);
  }
  {  // use_mixed_precision_solves / max_num_refinement_iterations (solver.h:572-590) through LinearSolver::Options, as
     // TrustRegionPreprocessor copies them (trust_region_preprocessor.cc:220-223): the float factor alone is visibly single
     // precision, with four refinement steps (dense_cholesky_test.cc:86-89) it is the double precision answer again
    for (LinearSolverType type : {DENSE_SCHUR, SPARSE_SCHUR})
      for (int refinements : {0, 4}) {
        LinearSolver::Options options;
        options.type = type;
        options.elimination_groups = {kPoints, kCameras};
        options.use_mixed_precision_solves = true;
        options.max_num_refinement_iterations = refinements;
        EXPECT(CxLinearSolver::Supports(options), "mixed precision %d", int(type));
        CxLinearSolver solver(options);
        LinearSolver::PerSolveOptions ps;
        ps.D = D.data();
        std::vector<double> x(static_cast<size_t>(A->num_cols()));
        InvalidateArray(A->num_cols(), x.data());
        LinearSolver::Summary s = solver.Solve(A.get(), b.data(), ps, x.data());
        double diff = 0.0, scale = 0.0;
        for (size_t i = 0; i < x.size(); ++i) {
          diff = std::max(diff, std::abs(x[i] - reference[i]));
          scale = std::max(scale, std::abs(reference[i]));
        }
        diff /= scale;
        const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && IsArrayValid(A->num_cols(), x.data()) &&
                        (refinements == 0 ? (diff > 1e-11 && diff < 1e-3) : diff < 1e-9) &&
                        s.message.find("single precision") != std::string::npos;
        std::printf("%-12s mixed precision, %d refinement steps  %s  max |x - x_dense_schur| / max |x| %.2e  (%s)\n",
                    type == DENSE_SCHUR ? "DENSE_SCHUR" : "SPARSE_SCHUR", refinements, ok ? "ok  " : "FAIL", diff, s.message.c_str());
        EXPECT(ok, "mixed precision solve");
      }
  }
  {  // the LM call: truncated solve with q_tolerance = eta, step = -x
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
    std::vector<double> step(static_cast<size_t>(A->num_cols()));
    InvalidateArray(A->num_cols(), step.data());
    LinearSolver::Summary s = solver.Solve(A.get(), b.data(), ps, step.data());
    const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && IsArrayValid(A->num_cols(), step.data());
    std::printf("LM-style truncated solve        %s  iterations %3d  (%s)\n", ok ? "ok  " : "FAIL", s.num_iterations, s.message.c_str());
    EXPECT(ok, "LM-style solve");
  }
  {  // no exception, no abort on misuse: failures come back as FATAL_ERROR summaries (linear_solver.h:66-73)
    LinearSolver::Options options;
    options.type = DENSE_QR;
    EXPECT(!CxLinearSolver::Supports(options), "DENSE_QR must not be claimed");
    CxLinearSolver solver(options);
    std::vector<double> x(static_cast<size_t>(A->num_cols()));
    LinearSolver::PerSolveOptions ps;
    ps.D = D.data();
    LinearSolver::Summary s = solver.Solve(A.get(), b.data(), ps, x.data());
    EXPECT(s.termination_type == LinearSolverTerminationType::FATAL_ERROR, "unsupported type -> FATAL_ERROR (%s)", s.message.c_str());
    options.type = ITERATIVE_SCHUR;
    options.elimination_groups = {kPoints, kCameras};
    CxLinearSolver solver2(options);
    ps.preconditioner = A.get();  // a user-supplied preconditioner operator cannot run on the device
    s = solver2.Solve(A.get(), b.data(), ps, x.data());
    EXPECT(s.termination_type == LinearSolverTerminationType::FATAL_ERROR, "user preconditioner -> FATAL_ERROR");
  }
}

// ------------------------------------------------------------------------------------------------ 2. the program
// Owns a synthetic bundle-adjustment problem modelled with the (mirrored) modelling-layer classes, in the state
// Solver::Solve leaves it in for a Schur-type solver: parameter blocks = points then cameras
// (ApplyOrdering, reorder_program.cc:216-254), residual blocks ordered by LexicographicallyOrderResidualBlocks.
struct BalProgram {
  using Snavely = AutoDiffCostFunction<examples::SnavelyReprojectionError, 2, 9, 3>;
  using SnavelyQuaternion = AutoDiffCostFunction<examples::SnavelyReprojectionErrorWithQuaternions, 2, 10, 3>;
  int C = 0, P = 0;
  int camera_size = 9;             // 10: quaternion cameras on ProductManifold<QuaternionManifold, EuclideanManifold<6>>
  std::unique_ptr<Manifold> camera_manifold;
  std::vector<double> user_state;  // [cameras (9 or 10 each) | points (3 each)] like BALProblem (bal_problem.cc:93-108)
  std::vector<std::unique_ptr<ParameterBlock>> parameter_blocks;
  std::vector<std::unique_ptr<CostFunction>> cost_functions;
  std::vector<std::unique_ptr<ResidualBlock>> residual_blocks;
  std::vector<std::unique_ptr<const LossFunction>> owned_losses;
  std::vector<int32_t> camera_index, point_index;  // input order
  std::vector<double> observations;
  Program program;

  // SnavelyReprojectionError::operator() on doubles (snavely_reprojection_error.h:60-93, rotation.h:792-857)
  static void Project(const double* cam, const double* pt, double* xy) {
    const double theta = std::sqrt(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2]);
    double p[3];
    if (theta > 0.0) {
      const double ct = std::cos(theta), st = std::sin(theta), ti = 1.0 / theta;
      const double w[3] = {cam[0] * ti, cam[1] * ti, cam[2] * ti};
      const double wx[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
      const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - ct);
      for (int i = 0; i < 3; ++i) p[i] = pt[i] * ct + wx[i] * st + w[i] * tmp;
    } else {
      p[0] = pt[0] + cam[1] * pt[2] - cam[2] * pt[1];
      p[1] = pt[1] + cam[2] * pt[0] - cam[0] * pt[2];
      p[2] = pt[2] + cam[0] * pt[1] - cam[1] * pt[0];
    }
    p[0] += cam[3]; p[1] += cam[4]; p[2] += cam[5];
    const double xp = -p[0] / p[2], yp = -p[1] / p[2];
    const double r2 = xp * xp + yp * yp;
    const double distortion = 1.0 + r2 * (cam[7] + cam[8] * r2);
    xy[0] = cam[6] * distortion * xp;
    xy[1] = cam[6] * distortion * yp;
  }

  // SnavelyReprojectionErrorWithQuaternions::operator() on doubles (snavely_reprojection_error.h:120-158,
  // QuaternionRotatePoint rotation.h:722-760): camera = quaternion (w x y z), translation, focal, k1, k2
  static void ProjectQuaternion(const double* cam, const double* pt, double* xy) {
    const double scale = 1.0 / std::sqrt(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2] + cam[3] * cam[3]);
    const double q[4] = {scale * cam[0], scale * cam[1], scale * cam[2], scale * cam[3]};
    double uv0 = q[2] * pt[2] - q[3] * pt[1], uv1 = q[3] * pt[0] - q[1] * pt[2], uv2 = q[1] * pt[1] - q[2] * pt[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    double p[3] = {pt[0] + q[0] * uv0 + (q[2] * uv2 - q[3] * uv1), pt[1] + q[0] * uv1 + (q[3] * uv0 - q[1] * uv2),
                   pt[2] + q[0] * uv2 + (q[1] * uv1 - q[2] * uv0)};
    p[0] += cam[4]; p[1] += cam[5]; p[2] += cam[6];
    const double xp = -p[0] / p[2], yp = -p[1] / p[2];
    const double r2 = xp * xp + yp * yp;
    const double distortion = 1.0 + r2 * (cam[8] + cam[9] * r2);
    xy[0] = cam[7] * distortion * xp;
    xy[1] = cam[7] * distortion * yp;
  }
  void ProjectModel(const double* cam, const double* pt, double* xy) const {
    if (camera_size == 10) ProjectQuaternion(cam, pt, xy);
    else Project(cam, pt, xy);
  }
  // AngleAxisToQuaternion (rotation.h:246-270)
  static void AngleAxisToQuaternion(const double* aa, double* q) {
    const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > 0.0) {
      const double theta = std::sqrt(theta2), k = std::sin(theta * 0.5) / theta;
      q[0] = std::cos(theta * 0.5); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
    } else {
      q[0] = 1.0; q[1] = aa[0] * 0.5; q[2] = aa[1] * 0.5; q[3] = aa[2] * 0.5;
    }
  }

  // losses: one object per residual block when `loss_factory` is set (as bundle_adjuster does), else `loss` for all
  BalProgram(int num_cameras, int num_points, double visibility, unsigned seed, const LossFunction* loss = nullptr, bool quaternion = false,
             std::function<const LossFunction*(size_t)> loss_factory = nullptr)
      : C(num_cameras), P(num_points), camera_size(quaternion ? 10 : 9) {
    std::mt19937 prng(seed);
    std::normal_distribution<double> normal(0.0, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    user_state.assign(size_t(9 * C + 3 * P), 0.0);
    std::vector<double> truth(user_state.size());
    double* cams = truth.data();
    double* pts = truth.data() + 9 * C;
    for (int i = 0; i < C; ++i) {  // cameras around the origin looking roughly down -z at a cloud near z = -10
      double* c = cams + 9 * i;
      for (int k = 0; k < 3; ++k) c[k] = 0.1 * normal(prng);
      c[3] = 0.5 * normal(prng); c[4] = 0.5 * normal(prng); c[5] = -10.0 + 0.5 * normal(prng);
      c[6] = 500.0 + 20.0 * normal(prng); c[7] = 1e-3 * normal(prng); c[8] = 1e-5 * normal(prng);
    }
    for (int j = 0; j < P; ++j) for (int k = 0; k < 3; ++k) pts[3 * j + k] = 2.0 * normal(prng);
    for (int j = 0; j < P; ++j) {
      int seen = 0;
      for (int i = 0; i < C; ++i) {
        if (uni(prng) < visibility || (seen < 2 && i >= C - 2 + seen)) {
          double xy[2];
          Project(cams + 9 * i, pts + 3 * j, xy);
          camera_index.push_back(i);
          point_index.push_back(j);
          observations.push_back(xy[0] + 0.5 * normal(prng));
          observations.push_back(xy[1] + 0.5 * normal(prng));
          ++seen;
        }
      }
    }
    // shuffle the observations so that the input order is not already the Schur order
    std::vector<int> perm(camera_index.size());
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), prng);
    {
      std::vector<int32_t> ci(perm.size()), pi(perm.size());
      std::vector<double> ob(2 * perm.size());
      for (size_t k = 0; k < perm.size(); ++k) {
        ci[k] = camera_index[size_t(perm[k])]; pi[k] = point_index[size_t(perm[k])];
        ob[2 * k] = observations[size_t(2 * perm[k])]; ob[2 * k + 1] = observations[size_t(2 * perm[k] + 1)];
      }
      camera_index.swap(ci); point_index.swap(pi); observations.swap(ob);
    }
    // the start point: the truth, perturbed (BALProblem::Perturb, bal_problem.cc:294-333)
    for (size_t k = 0; k < truth.size(); ++k) user_state[k] = truth[k];
    for (int i = 0; i < C; ++i) { for (int k = 0; k < 3; ++k) user_state[size_t(9 * i + k)] += 0.01 * normal(prng); for (int k = 3; k < 6; ++k) user_state[size_t(9 * i + k)] += 0.05 * normal(prng); }
    for (int j = 0; j < 3 * P; ++j) user_state[size_t(9 * C + j)] += 0.05 * normal(prng);
    if (quaternion) {  // BALProblem(filename, use_quaternions): the angle-axis of every camera becomes a quaternion (bal_problem.cc:110-128)
      std::vector<double> q_state(size_t(10 * C + 3 * P));
      for (int i = 0; i < C; ++i) {
        AngleAxisToQuaternion(&user_state[size_t(9 * i)], &q_state[size_t(10 * i)]);
        for (int k = 3; k < 9; ++k) q_state[size_t(10 * i + k + 1)] = user_state[size_t(9 * i + k)];
      }
      std::copy(user_state.begin() + 9 * C, user_state.end(), q_state.begin() + 10 * C);
      user_state.swap(q_state);
      camera_manifold.reset(new ProductManifold<QuaternionManifold, EuclideanManifold<6>>());
    }

    // Problem -> Program: parameter blocks in the order ApplyOrdering gives for ordering {points: 0, cameras: 1}
    for (int j = 0; j < P; ++j) parameter_blocks.emplace_back(new ParameterBlock(&user_state[size_t(camera_size * C + 3 * j)], 3, j));
    for (int i = 0; i < C; ++i) {
      parameter_blocks.emplace_back(new ParameterBlock(&user_state[size_t(camera_size * i)], camera_size, P + i));
      if (quaternion) parameter_blocks.back()->SetManifold(camera_manifold.get());
    }
    for (auto& pb : parameter_blocks) program.mutable_parameter_blocks()->push_back(pb.get());
    program.SetParameterOffsetsAndIndex();
    // residual blocks in input order, then LexicographicallyOrderResidualBlocks (reorder_program.cc:256-338)
    const size_t O = camera_index.size();
    std::vector<ResidualBlock*> input(O);
    for (size_t k = 0; k < O; ++k) {
      if (quaternion) cost_functions.emplace_back(new SnavelyQuaternion(new examples::SnavelyReprojectionErrorWithQuaternions(observations[2 * k], observations[2 * k + 1])));
      else cost_functions.emplace_back(new Snavely(new examples::SnavelyReprojectionError(observations[2 * k], observations[2 * k + 1])));
      std::vector<ParameterBlock*> blocks = {parameter_blocks[size_t(P + camera_index[k])].get(), parameter_blocks[size_t(point_index[k])].get()};
      if (loss_factory) {
        owned_losses.emplace_back(loss_factory(k));
        residual_blocks.emplace_back(new ResidualBlock(cost_functions.back().get(), owned_losses.back().get(), blocks, int(k)));
      } else {
        residual_blocks.emplace_back(new ResidualBlock(cost_functions.back().get(), loss, blocks, int(k)));
      }
      input[k] = residual_blocks.back().get();
    }
    std::vector<int> offsets(size_t(P) + 1, 0);
    for (size_t k = 0; k < O; ++k) offsets[size_t(point_index[k])]++;
    std::partial_sum(offsets.begin(), offsets.end(), offsets.begin());
    std::vector<ResidualBlock*> reordered(O, nullptr);
    for (size_t k = 0; k < O; ++k) reordered[size_t(--offsets[size_t(point_index[k])])] = input[k];
    *program.mutable_residual_blocks() = reordered;
  }

  // Program::ParameterBlocksToStateVector (program.cc:186-193)
  std::vector<double> StateVector() const {
    std::vector<double> state(static_cast<size_t>(program.NumParameters()));
    for (const ParameterBlock* pb : program.parameter_blocks())
      std::copy(pb->user_state(), pb->user_state() + pb->Size(), state.begin() + pb->state_offset());
    return state;
  }
};

// The trust-region loop of the reference reduced to the calls that cross the boundary: everything it does with the
// evaluator, the Jacobian and the linear solver goes through the abstract interfaces.
struct LmTrace {
  std::vector<double> costs;        // cost after every successful step (cost[0] = initial)
  std::vector<int> linear_iterations;
  std::vector<double> final_state;
  int num_successful = 0, num_unsuccessful = 0;
  bool ok = true;
};

// Wall time of the loop, split by the call that crosses the boundary (--time mode): iterations after the first
// `skip_iterations` are accumulated.  "caller" is everything between the calls -- the reference's own Eigen expressions.
struct LoopClock {
  int skip_iterations = 0;
  int timed_iterations = 0;
  double evaluate_jacobian_ms = 0, evaluate_cost_ms = 0, squared_column_norm_ms = 0, scale_columns_ms = 0, solve_ms = 0,
         model_cost_product_ms = 0, total_ms = 0;
  std::function<void()> on_timing_starts;  // called once, when the first timed iteration begins
};

static LmTrace RunTrustRegionLoop(Evaluator* evaluator, SparseMatrix* jacobian, LinearSolver* linear_solver, std::vector<double> x,
                                  int max_iterations, double eta, LoopClock* clock = nullptr) {
  LmTrace trace;
  using SteadyClock = std::chrono::steady_clock;
  bool timing = false;
  auto timed = [&](double LoopClock::*slot, auto&& call) {
    if (!timing) return call();
    const auto t0 = SteadyClock::now();
    auto r = call();
    clock->*slot += std::chrono::duration<double, std::milli>(SteadyClock::now() - t0).count();
    return r;
  };
  SteadyClock::time_point timing_started;
  const int num_parameters = evaluator->NumParameters(), num_effective = evaluator->NumEffectiveParameters();
  const int num_residuals = evaluator->NumResiduals();
  std::vector<double> residuals(static_cast<size_t>(num_residuals)), gradient(static_cast<size_t>(num_effective)), scaling(static_cast<size_t>(num_effective)),
      diagonal(static_cast<size_t>(num_effective)), lm_diagonal(static_cast<size_t>(num_effective)), step(static_cast<size_t>(num_effective)), delta(static_cast<size_t>(num_effective)),
      model_residuals(static_cast<size_t>(num_residuals)), candidate(static_cast<size_t>(num_parameters));
  double cost = 0.0;
  double radius = 1e4, decrease_factor = 2.0;  // Solver::Options defaults (solver.h:270-290)
  const double max_radius = 1e16, min_relative_decrease = 1e-3, min_diagonal = 1e-6, max_diagonal = 1e32;
  bool reuse_diagonal = false;
  // TrustRegionMinimizer::EvaluateGradientAndJacobian (trust_region_minimizer.cc:246-313)
  auto evaluate_gradient_and_jacobian = [&](bool first) {
    Evaluator::EvaluateOptions evaluate_options;
    evaluate_options.new_evaluation_point = true;
    if (!timed(&LoopClock::evaluate_jacobian_ms, [&] { return evaluator->Evaluate(evaluate_options, x.data(), &cost, residuals.data(), gradient.data(), jacobian); })) return false;
    if (first) {
      jacobian->SquaredColumnNorm(scaling.data());
      for (auto& s : scaling) s = 1.0 / (1.0 + std::sqrt(s));
    }
    timed(&LoopClock::scale_columns_ms, [&] { jacobian->ScaleColumns(scaling.data(), nullptr, 1); return true; });
    return true;
  };
  if (!evaluate_gradient_and_jacobian(true)) { trace.ok = false; return trace; }
  trace.costs.push_back(cost);
  for (int iteration = 1; iteration <= max_iterations; ++iteration) {
    if (clock != nullptr && !timing && iteration > clock->skip_iterations) {
      if (clock->on_timing_starts) clock->on_timing_starts();
      timing = true;
      timing_started = SteadyClock::now();
    }
    if (timing) {
      clock->timed_iterations = iteration - 1 - clock->skip_iterations;
      clock->total_ms = std::chrono::duration<double, std::milli>(SteadyClock::now() - timing_started).count();
    }
    // LevenbergMarquardtStrategy::ComputeStep (levenberg_marquardt_strategy.cc:69-156)
    if (!reuse_diagonal) {
      timed(&LoopClock::squared_column_norm_ms, [&] { jacobian->SquaredColumnNorm(diagonal.data(), nullptr, 1); return true; });
      for (auto& d : diagonal) d = std::min(std::max(d, min_diagonal), max_diagonal);
    }
    for (int i = 0; i < num_effective; ++i) lm_diagonal[size_t(i)] = std::sqrt(diagonal[size_t(i)] / radius);
    LinearSolver::PerSolveOptions solve_options;
    solve_options.D = lm_diagonal.data();
    solve_options.q_tolerance = eta;
    solve_options.r_tolerance = -1.0;
    InvalidateArray(num_effective, step.data());
    LinearSolver::Summary summary = timed(&LoopClock::solve_ms, [&] { return linear_solver->Solve(jacobian, residuals.data(), solve_options, step.data()); });
    if (summary.termination_type == LinearSolverTerminationType::FATAL_ERROR) { trace.ok = false; std::printf("  %s\n", summary.message.c_str()); return trace; }
    bool step_is_valid = false;
    if (summary.termination_type != LinearSolverTerminationType::FAILURE && IsArrayValid(num_effective, step.data())) {
      for (auto& v : step) v = -v;
      reuse_diagonal = true;
      trace.linear_iterations.push_back(summary.num_iterations);
      // TrustRegionMinimizer::ComputeTrustRegionStep (trust_region_minimizer.cc:381-463)
      std::fill(model_residuals.begin(), model_residuals.end(), 0.0);
      timed(&LoopClock::model_cost_product_ms, [&] { jacobian->RightMultiplyAndAccumulate(step.data(), model_residuals.data(), nullptr, 1); return true; });
      double model_cost_change = 0.0;
      for (int i = 0; i < num_residuals; ++i) model_cost_change -= model_residuals[size_t(i)] * (residuals[size_t(i)] + model_residuals[size_t(i)] / 2.0);
      step_is_valid = model_cost_change > 0.0;
      if (step_is_valid) {
        for (int i = 0; i < num_effective; ++i) delta[size_t(i)] = step[size_t(i)] * scaling[size_t(i)];
        // ComputeCandidatePointAndEvaluateCost (:720-748): cost only
        double candidate_cost = 0.0;
        if (!evaluator->Plus(x.data(), delta.data(), candidate.data()) ||
            !timed(&LoopClock::evaluate_cost_ms, [&] { return evaluator->Evaluate(candidate.data(), &candidate_cost, nullptr, nullptr, nullptr); })) { trace.ok = false; return trace; }
        const double relative_decrease = (cost - candidate_cost) / model_cost_change;  // TrustRegionStepEvaluator, monotonic
        if (relative_decrease > min_relative_decrease) {
          // HandleSuccessfulStep (:790-812) + LevenbergMarquardtStrategy::StepAccepted (:158-166)
          x = candidate;
          radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3)));
          decrease_factor = 2.0;
          reuse_diagonal = false;
          ++trace.num_successful;
          if (!evaluate_gradient_and_jacobian(false)) { trace.ok = false; return trace; }
          trace.costs.push_back(cost);
          continue;
        }
      }
    }
    // HandleUnsuccessfulStep / StepRejected (:168-173)
    radius /= decrease_factor;
    decrease_factor *= 2.0;
    reuse_diagonal = true;
    ++trace.num_unsuccessful;
    (void)step_is_valid;
  }
  if (timing) {
    clock->timed_iterations = max_iterations - clock->skip_iterations;
    clock->total_ms = std::chrono::duration<double, std::milli>(SteadyClock::now() - timing_started).count();
  }
  trace.final_state = x;
  return trace;
}

// A reference-style host evaluator for the comparison run: Evaluate() forwards to the device evaluator (the Jacobian
// arithmetic under test elsewhere) but hands J out as a host BlockSparseMatrix -- the path a Ceres build without
// CxBalEvaluator takes: host container, host products, values uploaded by CxLinearSolver on every Solve.
class HostJacobianEvaluator final : public Evaluator {
 public:
  explicit HostJacobianEvaluator(Evaluator* device_evaluator) : device_(device_evaluator), device_jacobian_(device_evaluator->CreateJacobian()) {}
  std::unique_ptr<SparseMatrix> CreateJacobian() const final {
    const auto* dj = static_cast<const CxDeviceJacobian*>(device_jacobian_.get());
    return std::make_unique<BlockSparseMatrix>(new CompressedRowBlockStructure(*dj->block_structure()));
  }
  bool Evaluate(const EvaluateOptions& o, const double* state, double* cost, double* residuals, double* gradient, SparseMatrix* jacobian) final {
    if (!device_->Evaluate(o, state, cost, residuals, gradient, jacobian ? device_jacobian_.get() : nullptr)) return false;
    if (jacobian) std::copy(device_jacobian_->values(), device_jacobian_->values() + jacobian->num_nonzeros(), jacobian->mutable_values());
    return true;
  }
  bool Plus(const double* state, const double* delta, double* out) const final { return device_->Plus(state, delta, out); }
  int NumParameters() const final { return device_->NumParameters(); }
  int NumEffectiveParameters() const final { return device_->NumEffectiveParameters(); }
  int NumResiduals() const final { return device_->NumResiduals(); }
 private:
  Evaluator* device_;
  std::unique_ptr<SparseMatrix> device_jacobian_;
};

class SoftLOneStandIn final : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const final { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};

static void TestEvaluatorToSolverThroughTheInterfaces() {
  const int kCameras = 24, kPoints = 1500, kIterations = 6;
  const double kEta = 1e-2;
  BalProgram bal(kCameras, kPoints, 0.35, 11u);
  Evaluator::Options evaluator_options;
  evaluator_options.linear_solver_type = ITERATIVE_SCHUR;
  evaluator_options.num_eliminate_blocks = kPoints;
  std::string why;
  std::unique_ptr<Evaluator> evaluator = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
  EXPECT(evaluator != nullptr, "TryCreate declined: %s", why.c_str());
  if (!evaluator) return;
  EXPECT(evaluator->NumParameters() == 3 * kPoints + 9 * kCameras && evaluator->NumEffectiveParameters() == evaluator->NumParameters() &&
             evaluator->NumResiduals() == 2 * bal.program.NumResidualBlocks(), "sizes");
  std::printf("BA program: %d cameras, %d points, %d residual blocks\n", kCameras, kPoints, bal.program.NumResidualBlocks());

  // the hook declines what the device does not implement, with a reason, and the factory falls through
  {
    Evaluator::Options o = evaluator_options;
    o.linear_solver_type = CGNR;
    EXPECT(CxBalEvaluator::TryCreate(o, &bal.program, &why) == nullptr && !why.empty(), "CGNR program must be declined");
    o = evaluator_options;
    o.num_eliminate_blocks = kPoints - 1;
    EXPECT(CxBalEvaluator::TryCreate(o, &bal.program, &why) == nullptr, "wrong elimination group must be declined (%s)", why.c_str());
    // a loss the device does not evaluate (ScaledLoss around a built-in one) is declined; so are blocks whose losses differ
    ScaledLoss scaled(new HuberLoss(1.0), 2.0);
    BalProgram robust(4, 40, 0.6, 3u, &scaled);
    o = evaluator_options;
    o.num_eliminate_blocks = 40;
    EXPECT(CxBalEvaluator::TryCreate(o, &robust.program, &why) == nullptr && why.find("built-in") != std::string::npos, "ScaledLoss: %s", why.c_str());
    BalProgram mixed(4, 40, 0.6, 3u, nullptr, false, [](size_t k) -> const LossFunction* { return new HuberLoss(k % 2 ? 1.0 : 2.0); });
    EXPECT(CxBalEvaluator::TryCreate(o, &mixed.program, &why) == nullptr && why.find("different loss") != std::string::npos, "mixed losses: %s", why.c_str());
    BalProgram partly(4, 40, 0.6, 3u, nullptr, false, [](size_t k) -> const LossFunction* { return k % 2 ? new HuberLoss(1.0) : nullptr; });
    EXPECT(CxBalEvaluator::TryCreate(o, &partly.program, &why) == nullptr, "loss on some blocks only: %s", why.c_str());
    // a loss that is the identity (TrivialLoss, or a user's class behaving like it) is no loss
    SoftLOneStandIn identity;
    BalProgram trivial(4, 40, 0.6, 3u, &identity);
    std::unique_ptr<Evaluator> te = CxBalEvaluator::TryCreate(o, &trivial.program, &why);
    EXPECT(te != nullptr && static_cast<CxBalEvaluator*>(te.get())->loss_type() == CX_LOSS_NONE, "identity loss: %s", why.c_str());
  }

  // residual r of the program's block k must be the reprojection error of ITS observation: checks the row order
  std::unique_ptr<SparseMatrix> jacobian = evaluator->CreateJacobian();
  auto* device_jacobian = dynamic_cast<CxDeviceJacobian*>(jacobian.get());
  EXPECT(device_jacobian != nullptr, "CreateJacobian must return the device-resident matrix");
  const std::vector<double> x0 = bal.StateVector();
  {
    std::vector<double> residuals(static_cast<size_t>(evaluator->NumResiduals()));
    double cost = 0.0;
    EXPECT(evaluator->Evaluate(x0.data(), &cost, residuals.data(), nullptr, nullptr), "residual-only evaluation");
    double worst = 0.0, half_sum = 0.0;
    const auto& blocks = bal.program.residual_blocks();
    for (size_t k = 0; k < blocks.size(); ++k) {
      const auto* cost_function = static_cast<const BalProgram::Snavely*>(blocks[k]->cost_function());
      const double* cam = x0.data() + blocks[k]->parameter_blocks()[0]->state_offset();
      const double* pt = x0.data() + blocks[k]->parameter_blocks()[1]->state_offset();
      double xy[2];
      BalProgram::Project(cam, pt, xy);
      const double r0 = xy[0] - cost_function->functor().observed_x, r1 = xy[1] - cost_function->functor().observed_y;
      worst = std::max(worst, std::max(std::abs(r0 - residuals[2 * k]), std::abs(r1 - residuals[2 * k + 1])));
      half_sum += 0.5 * (r0 * r0 + r1 * r1);
    }
    EXPECT(worst < 1e-9, "residuals are in program order: max |difference| %.3e", worst);
    EXPECT(std::abs(cost - half_sum) < 1e-10 * half_sum, "cost %.12e vs %.12e", cost, half_sum);
  }

  LinearSolver::Options solver_options;
  solver_options.type = ITERATIVE_SCHUR;
  solver_options.preconditioner_type = JACOBI;
  solver_options.elimination_groups = {kPoints, kCameras};
  solver_options.min_num_iterations = 0;
  solver_options.max_num_iterations = 500;

  // (i) device-resident: evaluator -> strategy -> solver, J never crosses PCIe
  CxLinearSolver device_solver(solver_options);
  LmTrace device_trace = RunTrustRegionLoop(evaluator.get(), jacobian.get(), &device_solver, x0, kIterations, kEta);
  EXPECT(device_trace.ok && device_trace.num_successful >= 3, "device-resident loop: %d successful steps", device_trace.num_successful);
  EXPECT(device_jacobian->num_downloads() == 0 && device_jacobian->num_uploads() == 0,
         "J must stay in HBM: %d downloads, %d uploads", device_jacobian->num_downloads(), device_jacobian->num_uploads());
  const auto evaluator_stats = evaluator->Statistics();
  EXPECT(evaluator_stats.count("Evaluator::Total") == 1 && evaluator_stats.count("Evaluator::Jacobian") == 1 &&
             evaluator_stats.count("Evaluator::Residual") == 1, "Evaluator::Statistics keys (program_evaluator.h:143-147)");
  EXPECT(device_solver.Statistics().at("LinearSolver::Solve").calls == device_trace.num_successful + device_trace.num_unsuccessful,
         "LinearSolver::Solve call count");

  // (ii) the residual vector is aliased to the evaluator's device copy by default (round 4: the pointer match is backed by
  // a sampled bit-for-bit comparison): identical arithmetic with it switched off ...
  EXPECT(device_solver.last_solve_aliased_residuals(), "residual aliasing is the default");
  {
    std::unique_ptr<Evaluator> evaluator2 = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
    std::unique_ptr<SparseMatrix> jacobian2 = evaluator2->CreateJacobian();
    CxLinearSolver uploading_solver(solver_options);
    uploading_solver.set_alias_evaluator_residuals(false);
    LmTrace t = RunTrustRegionLoop(evaluator2.get(), jacobian2.get(), &uploading_solver, x0, kIterations, kEta);
    EXPECT(t.ok && !uploading_solver.last_solve_aliased_residuals(), "aliasing switched off must upload b");
    EXPECT(t.costs == device_trace.costs && t.linear_iterations == device_trace.linear_iterations, "aliased residuals change nothing");
    // ... and a caller that HAS touched its residual array between Evaluate and Solve gets its own b, not the stale copy
    std::vector<double> residuals(static_cast<size_t>(evaluator2->NumResiduals())), step_a(x0.size()), step_b(x0.size()), D(x0.size(), 1.0);
    double cost = 0.0;
    Evaluator::EvaluateOptions eo;
    EXPECT(evaluator2->Evaluate(eo, x0.data(), &cost, residuals.data(), nullptr, jacobian2.get()), "evaluate");
    CxLinearSolver solver_a(solver_options), solver_b(solver_options);
    LinearSolver::PerSolveOptions ps;
    ps.D = D.data();
    ps.q_tolerance = 1e-6;
    ps.r_tolerance = -1.0;
    solver_a.Solve(jacobian2.get(), residuals.data(), ps, step_a.data());
    EXPECT(solver_a.last_solve_aliased_residuals(), "untouched residuals are aliased");
    for (size_t i = 0; i < residuals.size(); ++i) residuals[i] *= 2.0;
    solver_b.Solve(jacobian2.get(), residuals.data(), ps, step_b.data());
    EXPECT(!solver_b.last_solve_aliased_residuals(), "modified residuals must not be aliased");
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < step_a.size(); ++i) { worst = std::max(worst, std::abs(step_b[i] - 2.0 * step_a[i])); scale = std::max(scale, std::abs(step_a[i])); }
    EXPECT(worst <= 1e-4 * scale, "the solve of the doubled residuals is twice the step: %.3e vs scale %.3e", worst, scale);
  }

  // (ii b) Jacobi scaling folded into the evaluation (set_fuse_jacobi_scaling): J carries the same bits as after
  // Evaluate + ScaleColumns, so the loop walks the same costs to the last bit -- with one pass over J less per iteration
  {
    std::unique_ptr<Evaluator> evaluator3 = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
    static_cast<CxBalEvaluator*>(evaluator3.get())->set_fuse_jacobi_scaling(true);
    std::unique_ptr<SparseMatrix> jacobian3 = evaluator3->CreateJacobian();
    CxLinearSolver solver3(solver_options);
    LmTrace t = RunTrustRegionLoop(evaluator3.get(), jacobian3.get(), &solver3, x0, kIterations, kEta);
    EXPECT(t.ok && t.costs == device_trace.costs && t.linear_iterations == device_trace.linear_iterations,
           "fused Jacobi scaling must not change a bit of the loop (%zu vs %zu accepted)", t.costs.size(), device_trace.costs.size());
    EXPECT(static_cast<CxDeviceJacobian*>(jacobian3.get())->num_downloads() == 0, "fused scaling keeps J in HBM");
  }

  // (ii c) the boundary's transfer machinery changes no bit: caller arrays registered with the HIP runtime (first
  // sight, what CxSharedContext sets) or not at all, and the model-cost product into a target the caller has just zeroed
  // taken as y = J x (set_assume_zeroed_product_target) -- all three opt-ins on, as a TrustRegionMinimizer build would
  for (int registration = 0; registration <= 1; ++registration) {
    cx_host_registration_policy(registration, 1024, int64_t(1) << 30);  // small arrays here: register from 1 KiB
    std::unique_ptr<Evaluator> evaluator4 = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
    static_cast<CxBalEvaluator*>(evaluator4.get())->set_fuse_jacobi_scaling(true);
    std::unique_ptr<SparseMatrix> jacobian4 = evaluator4->CreateJacobian();
    static_cast<CxDeviceJacobian*>(jacobian4.get())->set_assume_zeroed_product_target(true);
    CxLinearSolver solver4(solver_options);
    solver4.set_alias_evaluator_residuals(true);
    cx_transfer_stats before{}, after{};
    cx_transfer_stats_get(CxSharedContext(), &before, 1);
    LmTrace t = RunTrustRegionLoop(evaluator4.get(), jacobian4.get(), &solver4, x0, kIterations, kEta);
    cx_transfer_stats_get(CxSharedContext(), &after, 0);
    EXPECT(t.ok && t.costs == device_trace.costs && t.linear_iterations == device_trace.linear_iterations,
           "registration policy %d + zeroed product target + aliasing + fused scaling must not change a bit of the loop", registration);
    const double registered = double(after.h2d_registered_bytes + after.d2h_registered_bytes) / double(after.h2d_bytes + after.d2h_bytes);
    std::printf("  registration policy %d: %.2f MB H2D, %.2f MB D2H over the loop, %.0f %% through registered arrays (%d arrays, %.1f ms registering)\n",
                registration, after.h2d_bytes / 1e6, after.d2h_bytes / 1e6, 100.0 * registered, after.num_registered, after.register_ms);
    EXPECT(registration == 0 ? registered == 0.0 : registered > 0.9, "registered share of the traffic: %.3f", registered);
    // the zeroed target saves the upload of num_rows doubles per product: H2D per iteration stays below 4 column vectors
    EXPECT(double(after.h2d_bytes) < double(kIterations + 1) * 8.0 * (4.5 * evaluator4->NumEffectiveParameters()), "H2D bytes %lld", (long long)after.h2d_bytes);
  }
  cx_host_registration_policy(0, int64_t(256) << 10, int64_t(16) << 30);

  // (iii) the reference-style path: host BlockSparseMatrix, host products, values uploaded on every Solve
  HostJacobianEvaluator host_evaluator(evaluator.get());
  std::unique_ptr<SparseMatrix> host_jacobian = host_evaluator.CreateJacobian();
  CxLinearSolver host_solver(solver_options);
  LmTrace host_trace = RunTrustRegionLoop(&host_evaluator, host_jacobian.get(), &host_solver, x0, kIterations, kEta);
  EXPECT(host_trace.ok && host_trace.costs.size() == device_trace.costs.size() && host_trace.linear_iterations == device_trace.linear_iterations,
         "host-Jacobian loop takes the same steps (%zu vs %zu successful)", host_trace.costs.size(), device_trace.costs.size());
  for (size_t k = 0; k < std::min(host_trace.costs.size(), device_trace.costs.size()); ++k)
    EXPECT(std::abs(host_trace.costs[k] - device_trace.costs[k]) <= 1e-9 * device_trace.costs[k], "cost %zu: %.12e vs %.12e", k,
           host_trace.costs[k], device_trace.costs[k]);

  // (iv) cx_minimize, the device-resident minimizer of the C ABI, on the same problem in INPUT observation order
  {
    cx_context* ctx = CxSharedContext();
    cx_evaluator* e = nullptr;
    cx_solver* s = nullptr;
    EXPECT(cx_evaluator_create_bal(ctx, kCameras, kPoints, int64_t(bal.camera_index.size()), bal.camera_index.data(), bal.point_index.data(),
                                   bal.observations.data(), &e) == CX_OK, "%s", cx_last_error());
    cx_solver_options so;
    cx_solver_default_options(&so);
    so.type = CX_ITERATIVE_SCHUR;
    so.preconditioner_type = CX_JACOBI;
    so.num_eliminate_blocks = kPoints;
    EXPECT(cx_solver_create(ctx, &so, &s) == CX_OK, "%s", cx_last_error());
    cx_minimizer_options mo;
    cx_minimizer_default_options(&mo);
    mo.max_num_iterations = kIterations;
    mo.eta = kEta;
    mo.function_tolerance = 0.0;
    mo.gradient_tolerance = 0.0;
    mo.parameter_tolerance = 0.0;
    std::vector<double> state = x0;
    cx_minimizer_summary ms;
    std::vector<cx_iteration_summary> its(static_cast<size_t>(kIterations + 2));
    EXPECT(cx_minimize(e, s, &mo, state.data(), CX_HOST, &ms, its.data(), int32_t(its.size())) == CX_OK, "%s", cx_last_error());
    std::vector<double> costs;
    for (int k = 0; k < ms.num_iterations && k < int(its.size()); ++k)
      if (k == 0 || its[size_t(k)].step_is_successful) costs.push_back(its[size_t(k)].cost);
    EXPECT(costs.size() == device_trace.costs.size(), "cx_minimize: %zu accepted costs vs %zu", costs.size(), device_trace.costs.size());
    for (size_t k = 0; k < std::min(costs.size(), device_trace.costs.size()); ++k)
      EXPECT(std::abs(costs[k] - device_trace.costs[k]) <= 1e-7 * device_trace.costs[k], "cx_minimize cost %zu: %.12e vs %.12e", k, costs[k],
             device_trace.costs[k]);
    cx_solver_destroy(s);
    cx_evaluator_destroy(e);
  }
  std::printf("LM through Evaluator/LinearSolver: cost %.6e -> %.6e in %d successful + %d unsuccessful steps, CG iterations",
              device_trace.costs.front(), device_trace.costs.back(), device_trace.num_successful, device_trace.num_unsuccessful);
  for (int n : device_trace.linear_iterations) std::printf(" %d", n);
  std::printf("\n");

  // host access on demand: values() materialises a copy, mutable_values() is written back before the next product
  {
    std::vector<double> ones(size_t(jacobian->num_cols()), 1.0), y0(size_t(jacobian->num_rows()), 0.0), y1 = y0;
    jacobian->RightMultiplyAndAccumulate(ones.data(), y0.data());
    double* v = jacobian->mutable_values();
    EXPECT(device_jacobian->num_downloads() == 1, "values are downloaded on first request only");
    for (int i = 0; i < jacobian->num_nonzeros(); ++i) v[i] *= 2.0;
    jacobian->RightMultiplyAndAccumulate(ones.data(), y1.data());
    EXPECT(device_jacobian->num_uploads() == 1, "modified host values are written back once");
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < y0.size(); ++i) { worst = std::max(worst, std::abs(y1[i] - 2.0 * y0[i])); scale = std::max(scale, std::abs(y0[i])); }
    EXPECT(worst <= 1e-12 * scale, "product after mutable_values(): %.3e", worst);
    Matrix dense;
    jacobian->ToDenseMatrix(&dense);
    EXPECT(dense.rows() == jacobian->num_rows() && dense.cols() == jacobian->num_cols(), "ToDenseMatrix shape");
    std::vector<double> yd(size_t(jacobian->num_rows()), 0.0);
    for (int r = 0; r < jacobian->num_rows(); ++r) for (int c = 0; c < jacobian->num_cols(); ++c) yd[size_t(r)] += dense(r, c);
    worst = 0.0;
    for (size_t i = 0; i < yd.size(); ++i) worst = std::max(worst, std::abs(yd[i] - y1[i]));
    EXPECT(worst <= 1e-11 * 2.0 * scale, "ToDenseMatrix row sums: %.3e", worst);
  }
  // the Jacobian may outlive the evaluator (shared ownership of the device objects)
  evaluator.reset();
  std::vector<double> norms(static_cast<size_t>(jacobian->num_cols()));
  jacobian->SquaredColumnNorm(norms.data());
  EXPECT(IsArrayValid(jacobian->num_cols(), norms.data()), "Jacobian usable after the evaluator is gone");
}

// the N-GPU path behind the boundary.  The same unmodified trust-region loop, written against the
// abstract Evaluator / SparseMatrix / LinearSolver interfaces only, on 1, 2 and 4 shards (CxSetDevices: logical shards
// of device 0 on a one-GPU box, one shard per GPU with RCCL where there are several): the caller still is ONE process
// passing whole vectors (context_impl.h:74-83, linear_solver.h:363-390); costs equal the 1-shard run to 1e-9, CG
// iteration counts are equal, J is never copied to or from the host.
static void TestShardsBehindTheInterfaces() {
  const int kCameras = 24, kPoints = 1500, kIterations = 6;
  const double kEta = 1e-2;
  BalProgram bal(kCameras, kPoints, 0.35, 11u);
  Evaluator::Options evaluator_options;
  evaluator_options.linear_solver_type = ITERATIVE_SCHUR;
  evaluator_options.num_eliminate_blocks = kPoints;
  LinearSolver::Options solver_options;
  solver_options.type = ITERATIVE_SCHUR;
  solver_options.preconditioner_type = JACOBI;
  solver_options.elimination_groups = {kPoints, kCameras};
  solver_options.min_num_iterations = 0;
  solver_options.max_num_iterations = 500;
  const std::vector<double> x0 = bal.StateVector();
  LmTrace one;
  for (int shards : {1, 2, 4}) {
    CxSetDevices(std::vector<int>(size_t(shards), 0));
    cx_context* ctx = CxSharedContext();
    EXPECT(ctx != nullptr && cx_context_num_shards(ctx) == shards, "context of %d shard(s): %s", shards, cx_last_error());
    if (ctx == nullptr) break;
    std::string why;
    std::unique_ptr<Evaluator> evaluator = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
    EXPECT(evaluator != nullptr, "TryCreate on %d shards declined: %s", shards, why.c_str());
    if (!evaluator) break;
    std::unique_ptr<SparseMatrix> jacobian = evaluator->CreateJacobian();
    auto* device_jacobian = dynamic_cast<CxDeviceJacobian*>(jacobian.get());
    CxLinearSolver solver(solver_options);
    solver.set_alias_evaluator_residuals(shards == 4);  // the residual token of a front: every shard reads its own rows in HBM
    LmTrace t = RunTrustRegionLoop(evaluator.get(), jacobian.get(), &solver, x0, kIterations, kEta);
    EXPECT(t.ok && t.num_successful >= 3, "%d shard(s): %d successful steps", shards, t.num_successful);
    EXPECT(device_jacobian != nullptr && device_jacobian->num_downloads() == 0 && device_jacobian->num_uploads() == 0,
           "%d shard(s): J must stay in HBM", shards);
    if (shards == 4) EXPECT(solver.last_solve_aliased_residuals(), "residual aliasing on a front was not taken");
    if (shards == 1) {
      one = t;
    } else {
      EXPECT(t.costs.size() == one.costs.size() && t.linear_iterations == one.linear_iterations,
             "%d shards take the steps of one (%zu vs %zu accepted)", shards, t.costs.size(), one.costs.size());
      for (size_t k = 0; k < std::min(t.costs.size(), one.costs.size()); ++k)
        EXPECT(std::abs(t.costs[k] - one.costs[k]) <= 1e-9 * one.costs[k], "%d shards, cost %zu: %.12e vs %.12e", shards, k, t.costs[k],
               one.costs[k]);
      double worst = 0.0, scale = 0.0;
      for (size_t i = 0; i < one.final_state.size(); ++i) {
        worst = std::max(worst, std::abs(t.final_state[i] - one.final_state[i]));
        scale = std::max(scale, std::abs(one.final_state[i]));
      }
      EXPECT(worst <= 1e-7 * scale, "%d shards, final state differs by %.3e", shards, worst);
      // the reference-style path on shards: a host BlockSparseMatrix is cut, uploaded and solved by the same front
      HostJacobianEvaluator host_evaluator(evaluator.get());
      std::unique_ptr<SparseMatrix> host_jacobian = host_evaluator.CreateJacobian();
      CxLinearSolver host_solver(solver_options);
      LmTrace h = RunTrustRegionLoop(&host_evaluator, host_jacobian.get(), &host_solver, x0, kIterations, kEta);
      EXPECT(h.ok && h.linear_iterations == one.linear_iterations, "%d shards, host Jacobian: same CG iterations", shards);
      for (size_t k = 0; k < std::min(h.costs.size(), one.costs.size()); ++k)
        EXPECT(std::abs(h.costs[k] - one.costs[k]) <= 1e-9 * one.costs[k], "%d shards, host Jacobian, cost %zu", shards, k);
    }
    std::printf("LM through Evaluator/LinearSolver on %d shard(s): cost %.9e -> %.9e, CG iterations", shards, t.costs.front(), t.costs.back());
    for (int n : t.linear_iterations) std::printf(" %d", n);
    std::printf("\n");
  }
  CxSetDevices({0});
}

// the programs bundle_adjuster builds with --robustify and with --use_quaternions --use_manifolds
// (examples/bundle_adjuster.cc:316-346) go through TryCreate: the loss is recovered from the LossFunction objects
// (one HuberLoss(1.0) per residual block, as the example news them), the camera manifold is recognised by type.
// The trust-region loop through the interfaces must take the steps of cx_minimize with the same settings, and the
// cost at the start must equal the host sum 1/2 sum rho(|r|^2) over the program's residual blocks.
static void TestRobustAndQuaternionPrograms() {
  const int kCameras = 16, kPoints = 900, kIterations = 5;
  const double kEta = 1e-2;
  for (int variant = 0; variant < 3; ++variant) {
    const bool robust = variant != 1, quaternion = variant != 0;
    const char* name = variant == 0 ? "--robustify" : (variant == 1 ? "--use_quaternions --use_manifolds" : "--robustify --use_quaternions --use_manifolds");
    std::function<const LossFunction*(size_t)> factory;
    if (robust) factory = [](size_t) -> const LossFunction* { return new HuberLoss(1.0); };
    BalProgram bal(kCameras, kPoints, 0.4, 23u, nullptr, quaternion, factory);
    Evaluator::Options evaluator_options;
    evaluator_options.linear_solver_type = ITERATIVE_SCHUR;
    evaluator_options.num_eliminate_blocks = kPoints;
    std::string why;
    std::unique_ptr<Evaluator> evaluator = CxBalEvaluator::TryCreate(evaluator_options, &bal.program, &why);
    EXPECT(evaluator != nullptr, "%s: TryCreate declined: %s", name, why.c_str());
    if (!evaluator) continue;
    auto* cxe = static_cast<CxBalEvaluator*>(evaluator.get());
    EXPECT(cxe->loss_type() == (robust ? CX_LOSS_HUBER : CX_LOSS_NONE) && (!robust || cxe->loss_a() == 1.0), "%s: loss %d (%.17g)", name,
           cxe->loss_type(), cxe->loss_a());
    EXPECT(cxe->camera_model() == (quaternion ? CX_CAMERA_QUATERNION_MANIFOLD : CX_CAMERA_ANGLE_AXIS), "%s: camera model", name);
    EXPECT(evaluator->NumParameters() == 3 * kPoints + bal.camera_size * kCameras && evaluator->NumEffectiveParameters() == 3 * kPoints + 9 * kCameras,
           "%s: ambient %d / tangent %d", name, evaluator->NumParameters(), evaluator->NumEffectiveParameters());
    const std::vector<double> x0 = bal.StateVector();
    // cost at the start against the host sum over the program's residual blocks (with and without the loss)
    for (bool apply_loss : {true, false}) {
      Evaluator::EvaluateOptions eo;
      eo.apply_loss_function = apply_loss;
      double cost = 0.0, want = 0.0;
      EXPECT(evaluator->Evaluate(eo, x0.data(), &cost, nullptr, nullptr, nullptr), "%s: cost evaluation", name);
      for (const ResidualBlock* rb : bal.program.residual_blocks()) {
        double xy[2], ox, oy;
        bal.ProjectModel(x0.data() + rb->parameter_blocks()[0]->state_offset(), x0.data() + rb->parameter_blocks()[1]->state_offset(), xy);
        if (quaternion) { const auto& f = static_cast<const BalProgram::SnavelyQuaternion*>(rb->cost_function())->functor(); ox = f.observed_x; oy = f.observed_y; }
        else { const auto& f = static_cast<const BalProgram::Snavely*>(rb->cost_function())->functor(); ox = f.observed_x; oy = f.observed_y; }
        const double sq = (xy[0] - ox) * (xy[0] - ox) + (xy[1] - oy) * (xy[1] - oy);
        double rho[3] = {sq, 1.0, 0.0};
        if (apply_loss && rb->loss_function() != nullptr) rb->loss_function()->Evaluate(sq, rho);
        want += 0.5 * rho[0];
      }
      EXPECT(std::abs(cost - want) <= 1e-10 * want, "%s (apply_loss_function %d): cost %.12e vs %.12e", name, int(apply_loss), cost, want);
    }
    LinearSolver::Options solver_options;
    solver_options.type = ITERATIVE_SCHUR;
    solver_options.preconditioner_type = JACOBI;
    solver_options.elimination_groups = {kPoints, kCameras};
    solver_options.max_num_iterations = 500;
    std::unique_ptr<SparseMatrix> jacobian = evaluator->CreateJacobian();
    CxLinearSolver solver(solver_options);
    LmTrace t = RunTrustRegionLoop(evaluator.get(), jacobian.get(), &solver, x0, kIterations, kEta);
    EXPECT(t.ok && t.num_successful >= 3 && t.costs.back() < 0.2 * t.costs.front(), "%s: %d successful steps, cost %.4e -> %.4e", name,
           t.num_successful, t.costs.front(), t.costs.back());
    // cx_minimize with the same settings stated directly
    cx_context* ctx = CxSharedContext();
    cx_evaluator* e = nullptr;
    cx_solver* s = nullptr;
    EXPECT(cx_evaluator_create_bal(ctx, kCameras, kPoints, int64_t(bal.camera_index.size()), bal.camera_index.data(), bal.point_index.data(),
                                   bal.observations.data(), &e) == CX_OK, "%s", cx_last_error());
    if (quaternion) EXPECT(cx_evaluator_set_camera_model(e, CX_CAMERA_QUATERNION_MANIFOLD) == CX_OK, "%s", cx_last_error());
    if (robust) EXPECT(cx_evaluator_set_loss(e, CX_LOSS_HUBER, 1.0, 0.0) == CX_OK, "%s", cx_last_error());
    cx_solver_options so;
    cx_solver_default_options(&so);
    so.type = CX_ITERATIVE_SCHUR;
    so.preconditioner_type = CX_JACOBI;
    so.num_eliminate_blocks = kPoints;
    EXPECT(cx_solver_create(ctx, &so, &s) == CX_OK, "%s", cx_last_error());
    cx_minimizer_options mo;
    cx_minimizer_default_options(&mo);
    mo.max_num_iterations = kIterations;
    mo.eta = kEta;
    mo.function_tolerance = mo.gradient_tolerance = mo.parameter_tolerance = 0.0;
    std::vector<double> state = x0;
    cx_minimizer_summary ms;
    std::vector<cx_iteration_summary> its(static_cast<size_t>(kIterations + 2));
    EXPECT(cx_minimize(e, s, &mo, state.data(), CX_HOST, &ms, its.data(), int32_t(its.size())) == CX_OK, "%s", cx_last_error());
    std::vector<double> costs;
    for (int k = 0; k < ms.num_iterations && k < int(its.size()); ++k)
      if (k == 0 || its[size_t(k)].step_is_successful) costs.push_back(its[size_t(k)].cost);
    EXPECT(costs.size() == t.costs.size(), "%s: cx_minimize %zu accepted costs vs %zu", name, costs.size(), t.costs.size());
    for (size_t k = 0; k < std::min(costs.size(), t.costs.size()); ++k)
      EXPECT(std::abs(costs[k] - t.costs[k]) <= 1e-7 * t.costs[k], "%s: cost %zu: %.12e vs %.12e", name, k, costs[k], t.costs[k]);
    cx_solver_destroy(s);
    cx_evaluator_destroy(e);
    std::printf("%-50s cost %.6e -> %.6e in %d steps\n", name, t.costs.front(), t.costs.back(), t.num_successful);
  }
}

// ------------------------------------------------------------------------------------------------ --time
// What one LM iteration costs THROUGH the interfaces at a given size: the unmodified call sequence of RunTrustRegionLoop
// on CxBalEvaluator / CxDeviceJacobian / CxLinearSolver with host vectors, wall ms per call kind, bytes across PCIe.
//   test_host_adapter --time --problem FILE [--iterations N] [--warmup W] [--eta X] [--shards S] [--plain]
// FILE (written by tools/boundary_timing.py from a bal.py preset): int64 C, P, O; int32 camera[O]; int32 point[O]
// (non-decreasing: the Schur order); double xy[2 O]; double state[3 P + 9 C] (points, then cameras).
// --plain switches the three opt-ins off (fused scaling, residual aliasing, zeroed product target).
static int TimeBoundary(int argc, char** argv) {
  const char* path = nullptr;
  int iterations = 4, warmup = 1, shards = 1;
  double eta = 0.1;
  bool plain = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--problem" && i + 1 < argc) path = argv[++i];
    else if (a == "--iterations" && i + 1 < argc) iterations = std::atoi(argv[++i]);
    else if (a == "--warmup" && i + 1 < argc) warmup = std::atoi(argv[++i]);
    else if (a == "--shards" && i + 1 < argc) shards = std::atoi(argv[++i]);
    else if (a == "--eta" && i + 1 < argc) eta = std::atof(argv[++i]);
    else if (a == "--plain") plain = true;
  }
  if (path == nullptr) { std::fprintf(stderr, "--time needs --problem FILE\n"); return 2; }
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); return 2; }
  int64_t header[3];
  if (std::fread(header, sizeof(int64_t), 3, f) != 3) { std::fprintf(stderr, "short file\n"); return 2; }
  const int64_t C = header[0], P = header[1], O = header[2];
  std::vector<int32_t> cam(static_cast<size_t>(O)), pt(static_cast<size_t>(O));
  std::vector<double> xy(static_cast<size_t>(2 * O)), state(static_cast<size_t>(3 * P + 9 * C));
  if (std::fread(cam.data(), 4, size_t(O), f) != size_t(O) || std::fread(pt.data(), 4, size_t(O), f) != size_t(O) ||
      std::fread(xy.data(), 8, size_t(2 * O), f) != size_t(2 * O) || std::fread(state.data(), 8, state.size(), f) != state.size()) {
    std::fprintf(stderr, "short file\n");
    return 2;
  }
  std::fclose(f);
  if (shards > 1) CxSetDevices(std::vector<int>(size_t(shards), 0));  // logical shards on device 0
  if (std::getenv("CX_PIN") == nullptr) CxRegisterCallerArrays(true);  // the loop's vectors live until it returns; CX_PIN=0 for the A/B
  CxBalProblemView view;
  view.num_cameras = int32_t(C);
  view.num_points = int32_t(P);
  view.num_observations = O;
  view.camera_index = cam.data();
  view.point_index = pt.data();
  view.observations_xy = xy.data();
  Evaluator::Options evaluator_options;
  evaluator_options.linear_solver_type = ITERATIVE_SCHUR;
  evaluator_options.num_eliminate_blocks = int(P);
  std::string error;
  std::unique_ptr<CxBalEvaluator> evaluator = CxBalEvaluator::Create(evaluator_options, view, &error);
  if (!evaluator) { std::fprintf(stderr, "CxBalEvaluator::Create: %s\n", error.c_str()); return 1; }
  if (!plain) evaluator->set_fuse_jacobi_scaling(true);
  std::unique_ptr<SparseMatrix> jacobian = evaluator->CreateJacobian();
  if (!plain) static_cast<CxDeviceJacobian*>(jacobian.get())->set_assume_zeroed_product_target(true);
  LinearSolver::Options solver_options;
  solver_options.type = ITERATIVE_SCHUR;
  solver_options.preconditioner_type = JACOBI;
  solver_options.elimination_groups = {int(P), int(C)};
  solver_options.min_num_iterations = 0;
  solver_options.max_num_iterations = 500;
  CxLinearSolver solver(solver_options);
  solver.set_alias_evaluator_residuals(!plain);
  cx_context* ctx = CxSharedContext();
  LoopClock clock;
  clock.skip_iterations = warmup;
  clock.on_timing_starts = [&] { cx_transfer_stats t; cx_transfer_stats_get(ctx, &t, 1); };
  LmTrace trace = RunTrustRegionLoop(evaluator.get(), jacobian.get(), &solver, state, warmup + iterations, eta, &clock);
  if (!trace.ok) { std::fprintf(stderr, "the loop failed\n"); return 1; }
  cx_transfer_stats t;
  cx_transfer_stats_get(ctx, &t, 0);
  const double k = double(std::max(1, clock.timed_iterations));
  const double inside = clock.evaluate_jacobian_ms + clock.evaluate_cost_ms + clock.squared_column_norm_ms + clock.scale_columns_ms +
                        clock.solve_ms + clock.model_cost_product_ms;
  std::printf("{\"through\": \"CxBalEvaluator / CxDeviceJacobian / CxLinearSolver (host vectors)\", \"cameras\": %lld, \"points\": %lld, "
              "\"residual_blocks\": %lld, \"shards\": %d, \"opt_ins\": %s, \"iterations\": %d, \"accepted\": %d, "
              "\"lm_iteration_through_interfaces_ms\": %.3f, \"calls_ms\": {\"evaluate_jacobian_ms\": %.3f, \"evaluate_cost_ms\": %.3f, "
              "\"squared_column_norm_ms\": %.3f, \"scale_columns_ms\": %.3f, \"solve_ms\": %.3f, \"model_cost_product_ms\": %.3f}, "
              "\"caller_ms\": %.3f, \"wall_ms_per_iteration\": %.3f, \"h2d_bytes\": %.0f, \"d2h_bytes\": %.0f, \"h2d_ms\": %.3f, "
              "\"d2h_ms\": %.3f, \"transfer_ms\": %.3f, \"registered_fraction\": %.4f, \"registered_arrays\": %d, "
              "\"register_ms_total\": %.1f, \"final_cost\": %.17g, \"cg_iterations_last\": %d}\n",
              (long long)C, (long long)P, (long long)O, shards, plain ? "false" : "true", clock.timed_iterations, trace.num_successful,
              inside / k, clock.evaluate_jacobian_ms / k, clock.evaluate_cost_ms / k, clock.squared_column_norm_ms / k, clock.scale_columns_ms / k,
              clock.solve_ms / k, clock.model_cost_product_ms / k, (clock.total_ms - inside) / k, clock.total_ms / k, t.h2d_bytes / k, t.d2h_bytes / k,
              t.h2d_ms / k, t.d2h_ms / k, (t.h2d_ms + t.d2h_ms) / k,
              double(t.h2d_registered_bytes + t.d2h_registered_bytes) / std::max(1.0, double(t.h2d_bytes + t.d2h_bytes)), t.num_registered, t.register_ms,
              trace.costs.empty() ? 0.0 : trace.costs.back(), trace.linear_iterations.empty() ? 0 : trace.linear_iterations.back());
  return 0;
}

int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i)
    if (std::string(argv[i]) == "--time") return TimeBoundary(argc, argv);
  TestSolversOnHostJacobian();
  TestEvaluatorToSolverThroughTheInterfaces();
  TestShardsBehindTheInterfaces();
  TestRobustAndQuaternionPrograms();
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
