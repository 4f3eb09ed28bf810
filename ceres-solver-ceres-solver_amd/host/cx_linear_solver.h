// CxLinearSolver: the adapter that puts libcxschur behind Ceres' LinearSolver interface (linear_solver.h:148-354).
//
// It overrides LinearSolver::Solve(LinearOperator*, ...) itself instead of deriving from
// TypedLinearSolver<BlockSparseMatrix> (linear_solver.h:363-390) because it accepts two kinds of Jacobian:
//   * CxDeviceJacobian (from CxBalEvaluator::CreateJacobian): values already in HBM, nothing is uploaded;
//   * BlockSparseMatrix (from the reference's ProgramEvaluator): values() uploaded verbatim every Solve, as
//     SchurComplementSolver / IterativeSchurComplementSolver / CgnrSolver would read them
//     (schur_complement_solver.h:107-140, iterative_schur_complement_solver.h:72-98, cgnr_solver.h:52-80).
// Like those it caches its per-structure state on the first Solve (schur_complement_solver.cc:109-135; "a single
// instance ... solves multiple systems with the same sparsity structure", linear_solver.h:137-142), times
// "LinearSolver::Solve" into an ExecutionSummary exactly as TypedLinearSolver does (:366-380), and never throws:
// every failure comes back as a Summary (FATAL_ERROR for device / library errors, which makes TrustRegionMinimizer
// stop, trust_region_minimizer.cc:404-411).
// Inside a Ceres checkout compile with -DCX_USE_CERES_HEADERS; nothing else changes.
#ifndef CX_LINEAR_SOLVER_H_
#define CX_LINEAR_SOLVER_H_

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cxschur.h"
#include "cx_device_jacobian.h"
#ifdef CX_USE_CERES_HEADERS
#include "ceres/block_sparse_matrix.h"
#include "ceres/execution_summary.h"
#include "ceres/linear_solver.h"
#endif

namespace ceres::internal {

// The process-wide device context(s), the analogue of ContextImpl::InitCuda (context_impl.cc:125-203): one context per
// device list, created on first use, shared by every adapter object, never destroyed (as ContextImpl's CUDA half lives
// as long as the ceres::Context).  The list names the GPUs a Solver::Solve may use:
//   {d}              one device (default {0}),
//   {d0, d1, ...}    one shard per device, RCCL all-reduce of the camera-space sums between them,
//   {d, d, ...}      logical shards on one device (how one-GPU machines exercise the sharded path),
// set by CxSetDevices() -- the hook for a Solver::Options / ContextImpl field in a Ceres tree -- or by the environment
// variable CX_DEVICES ("0,1,2,3") read on first use.  Thread-safe.  nullptr (and cx_last_error()) when the devices are
// not usable; the adapters turn that into FATAL_ERROR.
struct CxContextRegistry {
  std::mutex mutex;
  std::vector<int> devices;  // empty: not configured yet
  std::map<std::vector<int>, cx_context*> contexts;
  static CxContextRegistry& Get() {
    static CxContextRegistry registry;
    return registry;
  }
};

// Register the arrays the minimizer hands to Evaluate / Solve / the Jacobian's products with the HIP runtime the first
// time they are seen (cx_host_registration_policy).  A Ceres tree may switch this on: TrustRegionMinimizer and
// LevenbergMarquardtStrategy allocate those vectors once and keep them until Minimize returns
// (trust_region_minimizer.cc:181-203), after which no call reaches this library before the evaluator and the solver are
// destroyed -- and their destructors release the registrations.  Off by default: it buys host time only (pageable copies
// of such vectors already run at the PCIe rate), and a registered array that is freed early faults the GPU.
inline void CxRegisterCallerArrays(bool on) { cx_host_registration_policy(on ? 1 : 0, int64_t(256) << 10, int64_t(16) << 30); }

inline void CxSetDevices(const std::vector<int>& devices) {
  CxContextRegistry& r = CxContextRegistry::Get();
  std::lock_guard<std::mutex> lock(r.mutex);
  r.devices = devices;
}

inline cx_context* CxSharedContext() {
  CxContextRegistry& r = CxContextRegistry::Get();
  std::lock_guard<std::mutex> lock(r.mutex);
  if (r.devices.empty()) {
    if (const char* env = std::getenv("CX_DEVICES")) {
      for (const char* p = env; *p != '\0';) {
        char* end = nullptr;
        const long d = std::strtol(p, &end, 10);
        if (end == p) break;
        r.devices.push_back(int(d));
        p = (*end == ',') ? end + 1 : end;
      }
    }
    if (r.devices.empty()) r.devices.push_back(0);
  }
  auto it = r.contexts.find(r.devices);
  if (it != r.contexts.end()) return it->second;
  cx_context* ctx = nullptr;
  const int rc = r.devices.size() == 1 ? cx_context_create(r.devices[0], &ctx)
                                       : cx_context_create_multi(int(r.devices.size()), r.devices.data(), &ctx);
  if (rc != CX_OK) return nullptr;  // not cached: a later call may succeed (or report again)
  r.contexts.emplace(r.devices, ctx);
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

class CxLinearSolver final : public LinearSolver {
 public:
  explicit CxLinearSolver(LinearSolver::Options options) : options_(std::move(options)) {}
  ~CxLinearSolver() override {
    if (solver_) cx_solver_destroy(solver_);
    if (owned_matrix_) cx_matrix_destroy(owned_matrix_);
    // b / D / x / values() arrays registered on first sight belong to a caller that is about to free them
    cx_host_registrations_release();
  }
  CxLinearSolver(const CxLinearSolver&) = delete;
  void operator=(const CxLinearSolver&) = delete;

  // True for the option sets this library implements; the factory patch asks before constructing.
  static bool Supports(const LinearSolver::Options& options) {
    switch (options.type) {
      case DENSE_SCHUR: case SPARSE_SCHUR: case ITERATIVE_SCHUR: case CGNR: break;
      default: return false;
    }
    if (options.preconditioner_type == SUBSET) return false;
    if (options.dynamic_sparsity) return false;
    return true;
  }

  // When the right-hand side handed to Solve is the very host array the CxBalEvaluator behind the Jacobian last
  // wrote its residuals to -- which is what LevenbergMarquardtStrategy::ComputeStep passes
  // (levenberg_marquardt_strategy.cc:113, trust_region_minimizer.cc:404-408) -- use the copy the evaluator kept in
  // HBM instead of uploading 2 * num_residual_blocks doubles again.  LinearSolver::Solve promises nothing about who
  // owns b (TrustRegionMinimizer does not modify the residual array between Evaluate and Solve, another caller might).
  // Round 4: ON by default, because the pointer alone is not trusted: before the device copy is used, 64 entries of it spread over
  // the vector are compared bit for bit with the host array (cx_evaluator_device_residuals_match: a 512-byte copy); a
  // caller that has touched its residuals between Evaluate and Solve gets them uploaded as before.
  void set_alias_evaluator_residuals(bool on) { alias_evaluator_residuals_ = on; }

  LinearSolver::Summary Solve(LinearOperator* A, const double* b, const LinearSolver::PerSolveOptions& per_solve_options,
                              double* x) final {
    ScopedExecutionTimer total_time("LinearSolver::Solve", &execution_summary_);
    LinearSolver::Summary summary;
    if (A == nullptr || b == nullptr || x == nullptr) return Fatal(&summary, "null argument");  // CHECKs in the reference (:369-371)
    if (per_solve_options.preconditioner != nullptr)
      return Fatal(&summary, "a user-supplied preconditioner operator cannot be applied on the device");
    // a device-resident Jacobian brings its context along (the evaluator that owns it chose the devices); a host
    // Jacobian goes to the context of the configured device list, fixed at the first Solve like the rest of the
    // per-structure state
    auto* device_jacobian = dynamic_cast<CxDeviceJacobian*>(A);
    if (ctx_ == nullptr) ctx_ = device_jacobian != nullptr ? device_jacobian->handle()->ctx : CxSharedContext();
    cx_context* ctx = ctx_;
    if (ctx == nullptr) return Fatal(&summary);
    if (device_jacobian != nullptr && device_jacobian->handle()->ctx != ctx)
      return Fatal(&summary, "the Jacobian lives on another device context than the one this solver was first used with");
    const int num_eliminate_blocks =
        (options_.type == CGNR || options_.elimination_groups.empty()) ? 0 : options_.elimination_groups[0];

    cx_matrix* matrix = nullptr;
    const double* device_b = nullptr;
    int64_t uploaded_values_bytes = 0;
    if (device_jacobian != nullptr) {
      matrix = device_jacobian->device_matrix();  // values are in HBM already
      if (alias_evaluator_residuals_ && device_jacobian->handle()->last_residuals_host == b) {
        int32_t same = 0;
        if (cx_evaluator_device_residuals_match(device_jacobian->handle()->evaluator, b, &same) == CX_OK && same)
          device_b = cx_evaluator_device_residuals(device_jacobian->handle()->evaluator);
      }
    } else if (auto* host_jacobian = dynamic_cast<BlockSparseMatrix*>(A)) {
      if (owned_matrix_ == nullptr) {  // structure is fixed for the life of the solver (linear_solver.h:137-142)
        CxFlatStructure flat(*host_jacobian->block_structure());
        if (cx_matrix_create(ctx, &flat.view, num_eliminate_blocks, &owned_matrix_) != CX_OK) return Fatal(&summary);
      }
      // values change every LM iteration: upload them verbatim (same cell layout)
      if (cx_matrix_set_values(owned_matrix_, host_jacobian->values(), CX_HOST) != CX_OK) return Fatal(&summary);
      matrix = owned_matrix_;
      uploaded_values_bytes = int64_t(host_jacobian->num_nonzeros()) * int64_t(sizeof(double));
    } else {
      return Fatal(&summary, "the Jacobian is neither a BlockSparseMatrix nor a CxDeviceJacobian");
    }

    if (solver_ == nullptr) {
      cx_solver_options o;
      if (!TranslateOptions(num_eliminate_blocks, &o, &summary)) return summary;
      if (cx_solver_create(ctx, &o, &solver_) != CX_OK) return Fatal(&summary);
    }
    cx_per_solve_options ps{};
    ps.D = per_solve_options.D;
    ps.r_tolerance = per_solve_options.r_tolerance;
    ps.q_tolerance = per_solve_options.q_tolerance;
    ps.memspace = CX_HOST;
    ps.b_on_device = device_b != nullptr ? 1 : 0;
    cx_summary s;
    if (cx_solver_solve(solver_, matrix, device_b != nullptr ? device_b : b, &ps, x, &s) != CX_OK) return Fatal(&summary);
    summary.residual_norm = s.residual_norm;
    summary.num_iterations = s.num_iterations;
    summary.termination_type = static_cast<LinearSolverTerminationType>(s.termination_type);
    summary.message = s.message;
    last_notes_ = s.notes;  // CX_NOTE_*: options answered differently from how they were asked (the message says so in words)
    if (uploaded_values_bytes > 0) {
      // said once per Solve, where Solver::Summary / the iteration log will show it: this Jacobian is a host matrix (the
      // evaluator is not CxBalEvaluator) and its values cross PCIe every Solve
      char note[160];
      std::snprintf(note, sizeof(note), " [cxschur: host BlockSparseMatrix, %.1f MB of values uploaded for this Solve]", double(uploaded_values_bytes) / 1e6);
      summary.message += note;
    }
    host_values_uploaded_bytes_ += uploaded_values_bytes;
    cx_solver_last_timing(solver_, &timing_);
    aliased_last_b_ = device_b != nullptr;
    return summary;
  }

  // execution_summary.h:45-83 -> Solver::Summary::linear_solver_time_in_seconds (solver.cc:636-643)
  std::map<std::string, CallStatistics> Statistics() const final { return execution_summary_.statistics(); }

  const cx_solve_timing& last_timing() const { return timing_; }  // device-side phase times of the last Solve
  bool last_solve_aliased_residuals() const { return aliased_last_b_; }
  int last_notes() const { return last_notes_; }
  int64_t host_values_uploaded_bytes() const { return host_values_uploaded_bytes_; }  // over the life of this solver

 private:
  bool TranslateOptions(int num_eliminate_blocks, cx_solver_options* o, LinearSolver::Summary* summary) const {
    cx_solver_default_options(o);
    switch (options_.type) {
      case DENSE_SCHUR: o->type = CX_DENSE_SCHUR; break;
      case SPARSE_SCHUR: o->type = CX_SPARSE_SCHUR; break;
      case ITERATIVE_SCHUR: o->type = CX_ITERATIVE_SCHUR; break;
      case CGNR: o->type = CX_CGNR; break;
      default:
        Fatal(summary, "cxschur implements DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR and CGNR only.");
        return false;
    }
    switch (options_.preconditioner_type) {
      case IDENTITY: o->preconditioner_type = CX_IDENTITY; break;
      case JACOBI: o->preconditioner_type = CX_JACOBI; break;
      case SCHUR_JACOBI: o->preconditioner_type = CX_SCHUR_JACOBI; break;
      case SCHUR_POWER_SERIES_EXPANSION: o->preconditioner_type = CX_SCHUR_POWER_SERIES_EXPANSION; break;
      case CLUSTER_JACOBI: o->preconditioner_type = CX_CLUSTER_JACOBI; break;
      case CLUSTER_TRIDIAGONAL: o->preconditioner_type = CX_CLUSTER_TRIDIAGONAL; break;
      default:
        Fatal(summary, "Preconditioner not available in cxschur.");
        return false;
    }
    o->min_num_iterations = options_.min_num_iterations;
    o->max_num_iterations = options_.max_num_iterations;
    o->residual_reset_period = options_.residual_reset_period;
    o->num_eliminate_blocks = num_eliminate_blocks;
    o->use_mixed_precision_solves = options_.use_mixed_precision_solves;
    o->max_num_refinement_iterations = options_.max_num_refinement_iterations;
    o->max_num_spse_iterations = options_.max_num_spse_iterations;
    o->use_spse_initialization = options_.use_spse_initialization;
    o->spse_tolerance = options_.spse_tolerance;
    o->use_explicit_schur_complement = options_.use_explicit_schur_complement;
    o->visibility_clustering_type = options_.visibility_clustering_type == SINGLE_LINKAGE ? CX_SINGLE_LINKAGE : CX_CANONICAL_VIEWS;
    // ordering_type (linear_solver.h:157): the tile-sparse Cholesky chooses its own fill-reducing ordering and the
    // eliminator keeps the natural camera order; dense / sparse library types do not apply.
    return true;
  }

  // HIP / RCCL errors map to FATAL_ERROR, which makes TrustRegionMinimizer abort the solve
  // (trust_region_minimizer.cc:404-411)
  static LinearSolver::Summary Fatal(LinearSolver::Summary* s, const char* what = nullptr) {
    s->termination_type = LinearSolverTerminationType::FATAL_ERROR;
    s->num_iterations = 0;
    s->message = std::string("cxschur: ") + (what != nullptr ? what : cx_last_error());
    return *s;
  }

  LinearSolver::Options options_;
  cx_context* ctx_ = nullptr;          // shared, not owned (CxSharedContext)
  cx_matrix* owned_matrix_ = nullptr;  // only for host BlockSparseMatrix Jacobians
  cx_solver* solver_ = nullptr;
  cx_solve_timing timing_{};
  bool alias_evaluator_residuals_ = true;
  bool aliased_last_b_ = false;
  int last_notes_ = 0;
  int64_t host_values_uploaded_bytes_ = 0;
  ExecutionSummary execution_summary_;
};

}  // namespace ceres::internal
#endif
