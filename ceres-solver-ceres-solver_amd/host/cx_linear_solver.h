// CxLinearSolver: the adapter that puts libcxschur behind Ceres' LinearSolver interface.
//
// It derives from TypedLinearSolver<BlockSparseMatrix> exactly like
// SchurComplementSolver / IterativeSchurComplementSolver / CgnrSolver do
// (schur_complement_solver.h:107-140, iterative_schur_complement_solver.h:72-98,
// cgnr_solver.h:52-80), caches the device structure on the first Solve as they cache
// theirs (schur_complement_solver.cc:109-135) and returns the reference's Summary.
// Inside a Ceres checkout replace "ceres_mirror.h" by the real headers.
#ifndef CX_LINEAR_SOLVER_H_
#define CX_LINEAR_SOLVER_H_

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cxschur.h"
#include "ceres_mirror.h"

namespace ceres::internal {

// One process-wide device context, the analogue of ContextImpl::InitCuda (context_impl.cc:125-203).
inline cx_context* CxSharedContext(int device = 0) {
  static cx_context* ctx = nullptr;
  if (ctx == nullptr && cx_context_create(device, &ctx) != CX_OK) {
    throw std::runtime_error(std::string("cxschur: ") + cx_last_error());
  }
  return ctx;
}

// Flatten vector<CompressedRow> into the C ABI arrays (what CudaBlockSparseStructure does
// for the reference's own device code, cuda_block_structure.cc:50-236).
struct CxFlatStructure {
  std::vector<cx_block> rows, cols;
  std::vector<int32_t> row_cell_begin;
  std::vector<cx_cell> cells;
  cx_block_structure view{};
  explicit CxFlatStructure(const CompressedRowBlockStructure& bs) {
    cols.reserve(bs.cols.size());
    for (const Block& b : bs.cols) cols.push_back(cx_block{b.size, b.position});
    rows.reserve(bs.rows.size());
    row_cell_begin.reserve(bs.rows.size() + 1);
    row_cell_begin.push_back(0);
    for (const CompressedRow& r : bs.rows) {
      rows.push_back(cx_block{r.block.size, r.block.position});
      for (const Cell& c : r.cells) cells.push_back(cx_cell{c.block_id, c.position});
      row_cell_begin.push_back(int32_t(cells.size()));
    }
    view.num_row_blocks = int32_t(rows.size());
    view.num_col_blocks = int32_t(cols.size());
    view.row_blocks = rows.data();
    view.col_blocks = cols.data();
    view.row_cell_begin = row_cell_begin.data();
    view.cells = cells.data();
  }
};

class CxLinearSolver final : public BlockSparseMatrixSolver {
 public:
  explicit CxLinearSolver(LinearSolver::Options options) : options_(std::move(options)) {}
  ~CxLinearSolver() override {
    if (solver_) cx_solver_destroy(solver_);
    if (matrix_) cx_matrix_destroy(matrix_);
  }
  // execution_summary.h:45-83 -> Solver::Summary::linear_solver_time_in_seconds (solver.cc:636-643)
  std::map<std::string, double> Statistics() const override { return {{"LinearSolver::Solve", total_seconds_}}; }
  const cx_solve_timing& last_timing() const { return timing_; }

 private:
  LinearSolver::Summary SolveImpl(BlockSparseMatrix* A, const double* b,
                                  const LinearSolver::PerSolveOptions& per_solve_options, double* x) final {
    LinearSolver::Summary summary;
    cx_context* ctx = CxSharedContext();
    const int num_eliminate_blocks =
        (options_.type == CGNR || options_.elimination_groups.empty()) ? 0 : options_.elimination_groups[0];
    if (matrix_ == nullptr) {  // structure is fixed for the life of the solver (linear_solver.h:137-142)
      CxFlatStructure flat(*A->block_structure());
      if (cx_matrix_create(ctx, &flat.view, num_eliminate_blocks, &matrix_) != CX_OK) return Fatal(&summary);
      cx_solver_options o;
      cx_solver_default_options(&o);
      switch (options_.type) {
        case DENSE_SCHUR: o.type = CX_DENSE_SCHUR; break;
        case SPARSE_SCHUR: o.type = CX_SPARSE_SCHUR; break;
        case ITERATIVE_SCHUR: o.type = CX_ITERATIVE_SCHUR; break;
        case CGNR: o.type = CX_CGNR; break;
        default:
          summary.termination_type = LinearSolverTerminationType::FATAL_ERROR;
          summary.message = "cxschur implements DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR and CGNR only.";
          return summary;
      }
      switch (options_.preconditioner_type) {
        case IDENTITY: o.preconditioner_type = CX_IDENTITY; break;
        case JACOBI: o.preconditioner_type = CX_JACOBI; break;
        case SCHUR_JACOBI: o.preconditioner_type = CX_SCHUR_JACOBI; break;
        case SCHUR_POWER_SERIES_EXPANSION: o.preconditioner_type = CX_SCHUR_POWER_SERIES_EXPANSION; break;
        case CLUSTER_JACOBI: o.preconditioner_type = CX_CLUSTER_JACOBI; break;
        case CLUSTER_TRIDIAGONAL: o.preconditioner_type = CX_CLUSTER_TRIDIAGONAL; break;
        default:
          summary.termination_type = LinearSolverTerminationType::FATAL_ERROR;
          summary.message = "Preconditioner not available in cxschur.";
          return summary;
      }
      o.min_num_iterations = options_.min_num_iterations;
      o.max_num_iterations = options_.max_num_iterations;
      o.residual_reset_period = options_.residual_reset_period;
      o.num_eliminate_blocks = num_eliminate_blocks;
      o.use_mixed_precision_solves = options_.use_mixed_precision_solves;
      o.max_num_refinement_iterations = options_.max_num_refinement_iterations;
      o.max_num_spse_iterations = options_.max_num_spse_iterations;
      o.use_spse_initialization = options_.use_spse_initialization;
      o.spse_tolerance = options_.spse_tolerance;
      o.use_explicit_schur_complement = options_.use_explicit_schur_complement;
      o.visibility_clustering_type = options_.visibility_clustering_type == SINGLE_LINKAGE ? CX_SINGLE_LINKAGE : CX_CANONICAL_VIEWS;
      if (cx_solver_create(ctx, &o, &solver_) != CX_OK) return Fatal(&summary);
    }
    // Values change every LM iteration: upload them verbatim (same cell layout).
    if (cx_matrix_set_values(matrix_, A->values(), CX_HOST) != CX_OK) return Fatal(&summary);
    cx_per_solve_options ps{};
    ps.D = per_solve_options.D;
    ps.r_tolerance = per_solve_options.r_tolerance;
    ps.q_tolerance = per_solve_options.q_tolerance;
    ps.memspace = CX_HOST;
    cx_summary s;
    if (cx_solver_solve(solver_, matrix_, b, &ps, x, &s) != CX_OK) return Fatal(&summary);
    summary.residual_norm = s.residual_norm;
    summary.num_iterations = s.num_iterations;
    summary.termination_type = static_cast<LinearSolverTerminationType>(s.termination_type);
    summary.message = s.message;
    cx_solver_last_timing(solver_, &timing_);
    total_seconds_ += timing_.total_ms * 1e-3;
    return summary;
  }

  // HIP / RCCL errors map to FATAL_ERROR, which makes TrustRegionMinimizer abort the solve
  // (trust_region_minimizer.cc:404-411)
  static LinearSolver::Summary Fatal(LinearSolver::Summary* s) {
    s->termination_type = LinearSolverTerminationType::FATAL_ERROR;
    s->num_iterations = 0;
    s->message = std::string("cxschur: ") + cx_last_error();
    return *s;
  }

  LinearSolver::Options options_;
  cx_matrix* matrix_ = nullptr;
  cx_solver* solver_ = nullptr;
  cx_solve_timing timing_{};
  double total_seconds_ = 0.0;
};

}  // namespace ceres::internal
#endif
