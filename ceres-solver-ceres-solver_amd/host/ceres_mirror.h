// Header-compatible mirror of the reference declarations at the drop-in boundary.
//
// The Ceres sources cannot be compiled in this image (Eigen / abseil absent), so the adapters in
// cx_linear_solver.h / cx_device_jacobian.h / cx_bal_evaluator.h are written against these declarations.  They
// repeat the reference's names, members, virtual signatures and defaults one to one; the reference line of every
// declaration is given beside it and test_host_adapter.cpp holds a static_assert table that pins each mirrored
// signature, so a drift between this file and the adapters cannot compile.  What is NOT mirrored is listed at the
// end of this comment.  Inside a Ceres checkout the adapters include the real headers instead of this file
// (-DCX_USE_CERES_HEADERS) and nothing else changes.
//
//   include/ceres/types.h:57-200              LinearSolverType ... DenseLinearAlgebraLibraryType, kImpossibleValue
//   internal/ceres/block_structure.h:52-182   Block, Cell, CompressedList, CompressedRowBlockStructure
//   internal/ceres/linear_operator.h:46-83    LinearOperator
//   internal/ceres/sparse_matrix.h:66-113     SparseMatrix
//   internal/ceres/block_sparse_matrix.h:60-176  BlockSparseMatrix (final)
//   internal/ceres/execution_summary.h:45-92  CallStatistics, ExecutionSummary, ScopedExecutionTimer
//   internal/ceres/linear_solver.h:57-394     LinearSolverTerminationType, OrderingType, LinearSolver, TypedLinearSolver
//   internal/ceres/evaluator.h:60-167         Evaluator
//   internal/ceres/array_utils.cc:42-75       InvalidateArray, IsArrayValid
//   internal/ceres/casts.h                    down_cast
//   internal/ceres/program.h, parameter_block.h, residual_block.h, include/ceres/cost_function.h,
//   autodiff_cost_function.h, examples/snavely_reprojection_error.h   the accessors CxBalEvaluator::TryCreate reads
//
// Not mirrored (the adapters do not touch them): the Eigen `Vector` overloads of LinearOperator (non-pure, inherited),
// BlockSparseMatrix' CRS conversions / AppendRows / CreateRandomMatrix, LinearSolver::Create and
// Evaluator::Create (the factories INTEGRATION.md patches), EvaluationCallback's members.
#ifndef CX_CERES_MIRROR_H_
#define CX_CERES_MIRROR_H_

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

// ---- the two abseil names execution_summary.h uses (absl/time/time.h); seconds in a double
namespace absl {
class Duration {
 public:
  constexpr Duration() = default;
  constexpr explicit Duration(double s) : s_(s) {}
  Duration& operator+=(Duration d) { s_ += d.s_; return *this; }
  constexpr double seconds() const { return s_; }
 private:
  double s_ = 0.0;
};
constexpr Duration ZeroDuration() { return Duration(); }
inline double ToDoubleSeconds(Duration d) { return d.seconds(); }
class Time {
 public:
  std::chrono::steady_clock::time_point t;
};
inline Time Now() { return Time{std::chrono::steady_clock::now()}; }
inline Duration operator-(Time a, Time b) { return Duration(std::chrono::duration<double>(a.t - b.t).count()); }
}  // namespace absl

namespace ceres {

// include/ceres/types.h:57-91
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
// :93-141
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, SCHUR_POWER_SERIES_EXPANSION, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL, SUBSET };
// :143-173
enum VisibilityClusteringType { CANONICAL_VIEWS, SINGLE_LINKAGE };
// :175-200
enum SparseLinearAlgebraLibraryType { SUITE_SPARSE, EIGEN_SPARSE, ACCELERATE_SPARSE, CUDA_SPARSE, NO_SPARSE };
// :214-218
enum DenseLinearAlgebraLibraryType { EIGEN, LAPACK, CUDA };
// :480
const double kImpossibleValue = 1e302;

class EvaluationCallback;  // include/ceres/evaluation_callback.h (only named)

// ---- the modelling-layer accessors CxBalEvaluator::TryCreate reads (read-only)
class CostFunction {  // include/ceres/cost_function.h:66-140
 public:
  virtual ~CostFunction() = default;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }
 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }
 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_ = 0;
};
class LossFunction {  // include/ceres/loss_function.h:84-88
 public:
  virtual ~LossFunction() = default;
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
// the built-in losses, include/ceres/loss_function.h:131-292 with loss_function.cc:46-144: parameters are PRIVATE, as in
// the reference -- which is why the adapter recovers them through Evaluate (cx_loss_probe.h)
class TrivialLoss final : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
class HuberLoss final : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
 private:
  const double a_, b_;
};
class SoftLOneLoss final : public LossFunction {
 public:
  explicit SoftLOneLoss(double a) : b_(a * a), c_(1 / b_) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, tmp = std::sqrt(sum);
    rho[0] = 2.0 * b_ * (tmp - 1.0);
    rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / tmp);
    rho[2] = -(c_ * rho[1]) / (2.0 * sum);
  }
 private:
  const double b_, c_;
};
class CauchyLoss final : public LossFunction {
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
 private:
  const double b_, c_;
};
class ArctanLoss final : public LossFunction {
 public:
  explicit ArctanLoss(double a) : a_(a), b_(1 / (a * a)) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1 + s * s * b_, inv = 1 / sum;
    rho[0] = a_ * std::atan2(s, a_);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -2.0 * s * b_ * (inv * inv);
  }
 private:
  const double a_, b_;
};
class TolerantLoss final : public LossFunction {
 public:
  TolerantLoss(double a, double b) : a_(a), b_(b), c_(b * std::log(1.0 + std::exp(-a / b))) {}
  void Evaluate(double s, double rho[3]) const override {
    const double x = (s - a_) / b_;
    if (x > 36.7) {
      rho[0] = s - a_ - c_; rho[1] = 1.0; rho[2] = 0.0;
    } else {
      const double e_x = std::exp(x);
      rho[0] = b_ * std::log(1.0 + e_x) - c_;
      rho[1] = std::max(std::numeric_limits<double>::min(), e_x / (1.0 + e_x));
      rho[2] = 0.5 / (b_ * (1.0 + std::cosh(x)));
    }
  }
 private:
  const double a_, b_, c_;
};
class TukeyLoss final : public LossFunction {
 public:
  explicit TukeyLoss(double a) : a_squared_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s <= a_squared_) {
      const double value = 1.0 - s / a_squared_, value_sq = value * value;
      rho[0] = a_squared_ / 3.0 * (1.0 - value_sq * value);
      rho[1] = value_sq;
      rho[2] = -2.0 / a_squared_ * value;
    } else {
      rho[0] = a_squared_ / 3.0; rho[1] = 0.0; rho[2] = 0.0;
    }
  }
 private:
  const double a_squared_;
};
class ScaledLoss final : public LossFunction {  // :329-350 (a loss the device does not offer: must be declined)
 public:
  ScaledLoss(const LossFunction* rho, double a) : rho_(rho), a_(a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (rho_ == nullptr) { rho[0] = a_ * s; rho[1] = a_; rho[2] = 0.0; return; }
    rho_->Evaluate(s, rho);
    rho[0] *= a_; rho[1] *= a_; rho[2] *= a_;
  }
 private:
  std::unique_ptr<const LossFunction> rho_;
  const double a_;
};
class Manifold {  // include/ceres/manifold.h:126-222 (only AmbientSize / TangentSize are read)
 public:
  virtual ~Manifold() = default;
  virtual int AmbientSize() const = 0;
  virtual int TangentSize() const = 0;
};
// the manifold types bundle_adjuster --use_quaternions --use_manifolds composes (bundle_adjuster.cc:337-345;
// include/ceres/manifold.h:224-300, 310-345, product_manifold.h:67-120): identified by type, arithmetic on the device
template <int Size>
class EuclideanManifold final : public Manifold {
 public:
  int AmbientSize() const override { return Size; }
  int TangentSize() const override { return Size; }
};
class QuaternionManifold final : public Manifold {
 public:
  int AmbientSize() const override { return 4; }
  int TangentSize() const override { return 3; }
};
template <typename Manifold0, typename Manifold1, typename... ManifoldN>
class ProductManifold final : public Manifold {
 public:
  int AmbientSize() const override { return (Manifold0().AmbientSize() + Manifold1().AmbientSize()) + (0 + ... + ManifoldN().AmbientSize()); }
  int TangentSize() const override { return (Manifold0().TangentSize() + Manifold1().TangentSize()) + (0 + ... + ManifoldN().TangentSize()); }
};
// include/ceres/autodiff_cost_function.h:151-240: owns the functor, exposes it through functor()
template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction final : public CostFunction {
 public:
  explicit AutoDiffCostFunction(CostFunctor* functor) : functor_(functor) {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
  }
  const CostFunctor& functor() const { return *functor_; }
 private:
  std::unique_ptr<CostFunctor> functor_;
};
namespace examples {
// examples/snavely_reprojection_error.h:53-104 (the data members; operator() lives on the device, cx_eval.hip)
struct SnavelyReprojectionError {
  SnavelyReprojectionError(double observed_x_, double observed_y_) : observed_x(observed_x_), observed_y(observed_y_) {}
  double observed_x;
  double observed_y;
};
// :111-170, the 10-parameter camera (quaternion w x y z, translation, focal, k1, k2)
struct SnavelyReprojectionErrorWithQuaternions {
  SnavelyReprojectionErrorWithQuaternions(double observed_x_, double observed_y_) : observed_x(observed_x_), observed_y(observed_y_) {}
  double observed_x;
  double observed_y;
};
}  // namespace examples

namespace internal {

class ContextImpl;  // context_impl.h (only named by the interfaces)

// Eigen row-major dynamic matrix (internal/eigen.h:44): the three members SparseMatrix::ToDenseMatrix callers use
class Matrix {
 public:
  void resize(int64_t rows, int64_t cols) { rows_ = rows; cols_ = cols; v_.assign(size_t(rows * cols), 0.0); }
  void setZero() { v_.assign(v_.size(), 0.0); }
  double& operator()(int64_t r, int64_t c) { return v_[size_t(r * cols_ + c)]; }
  double operator()(int64_t r, int64_t c) const { return v_[size_t(r * cols_ + c)]; }
  int64_t rows() const { return rows_; }
  int64_t cols() const { return cols_; }
 private:
  int64_t rows_ = 0, cols_ = 0;
  std::vector<double> v_;
};

// casts.h: static_cast in optimised builds, checked in debug builds
template <typename To, typename From>
inline To down_cast(From* f) { return static_cast<To>(f); }

// ---------------------------------------------------------------- block_structure.h:52-182
using BlockSize = int32_t;
struct Block {
  Block() = default;
  Block(int size_, int position_) noexcept : size(size_), position(position_) {}
  BlockSize size{-1};
  int position{-1};
};
struct Cell {
  Cell() = default;
  Cell(int block_id_, int position_) noexcept : block_id(block_id_), position(position_) {}
  int block_id{-1};
  int position{-1};
};
struct CompressedList {
  CompressedList() = default;
  explicit CompressedList(int num_cells) noexcept : cells(num_cells) {}
  Block block;
  std::vector<Cell> cells;
  int nnz{-1};
  int cumulative_nnz{-1};
};
using CompressedRow = CompressedList;
using CompressedColumn = CompressedList;
struct CompressedRowBlockStructure {
  std::vector<Block> cols;
  std::vector<CompressedRow> rows;
};

// ---------------------------------------------------------------- linear_operator.h:46-83
class LinearOperator {
 public:
  virtual ~LinearOperator() = default;
  virtual void RightMultiplyAndAccumulate(const double* x, double* y) const = 0;                       // :51
  virtual void RightMultiplyAndAccumulate(const double* x, double* y, ContextImpl* context, int num_threads) const {  // :52-55
    (void)context; (void)num_threads;
    RightMultiplyAndAccumulate(x, y);
  }
  virtual void LeftMultiplyAndAccumulate(const double* x, double* y) const = 0;                        // :57
  virtual void LeftMultiplyAndAccumulate(const double* x, double* y, ContextImpl* context, int num_threads) const {   // :58-61
    (void)context; (void)num_threads;
    LeftMultiplyAndAccumulate(x, y);
  }
  virtual int num_rows() const = 0;  // :85
  virtual int num_cols() const = 0;  // :86
};

// ---------------------------------------------------------------- sparse_matrix.h:66-113
class SparseMatrix : public LinearOperator {
 public:
  ~SparseMatrix() override = default;
  using LinearOperator::RightMultiplyAndAccumulate;
  void RightMultiplyAndAccumulate(const double* x, double* y) const override = 0;  // :72-73
  void LeftMultiplyAndAccumulate(const double* x, double* y) const override = 0;   // :76
  virtual void SquaredColumnNorm(double* x) const = 0;                              // :79
  virtual void SquaredColumnNorm(double* x, ContextImpl* context, int num_threads) const {  // :80-82
    (void)context; (void)num_threads;
    SquaredColumnNorm(x);
  }
  virtual void ScaleColumns(const double* scale) = 0;                               // :84
  virtual void ScaleColumns(const double* scale, ContextImpl* context, int num_threads) {  // :85-87
    (void)context; (void)num_threads;
    ScaleColumns(scale);
  }
  virtual void SetZero() = 0;                                                       // :90
  virtual void SetZero(ContextImpl* /*context*/, int /*num_threads*/) { SetZero(); }  // :91-93
  virtual void ToDenseMatrix(Matrix* dense_matrix) const = 0;                       // :98
  virtual void ToTextFile(FILE* file) const = 0;                                    // :101
  virtual double* mutable_values() = 0;                                             // :107
  virtual const double* values() const = 0;                                         // :108
  int num_rows() const override = 0;                                                // :110
  int num_cols() const override = 0;                                                // :111
  virtual int num_nonzeros() const = 0;                                             // :112
};

// ---------------------------------------------------------------- block_sparse_matrix.h:60-176
// The host container with plain-loop products (block_sparse_matrix.cc:220-450); `final`, as in the reference --
// which is why the device-resident Jacobian is a sibling SparseMatrix (cx_device_jacobian.h), not a subclass.
class BlockSparseMatrix final : public SparseMatrix {
 public:
  explicit BlockSparseMatrix(CompressedRowBlockStructure* block_structure, bool use_page_locked_memory = false)  // :73-74
      : block_structure_(block_structure) {
    (void)use_page_locked_memory;
    num_rows_ = num_cols_ = num_nonzeros_ = 0;
    for (auto& c : block_structure_->cols) num_cols_ += c.size;
    for (auto& r : block_structure_->rows) {
      num_rows_ += r.block.size;
      for (auto& cell : r.cells) num_nonzeros_ += r.block.size * block_structure_->cols[cell.block_id].size;
    }
    values_.assign(size_t(num_nonzeros_), 0.0);
  }
  BlockSparseMatrix(const BlockSparseMatrix&) = delete;
  void operator=(const BlockSparseMatrix&) = delete;

  void SetZero() final { values_.assign(values_.size(), 0.0); }
  void RightMultiplyAndAccumulate(const double* x, double* y) const final {
    ForEachEntry([&](int r, int c, double v) { y[r] += v * x[c]; });
  }
  void LeftMultiplyAndAccumulate(const double* x, double* y) const final {
    ForEachEntry([&](int r, int c, double v) { y[c] += v * x[r]; });
  }
  void SquaredColumnNorm(double* x) const final {
    for (int j = 0; j < num_cols_; ++j) x[j] = 0.0;
    ForEachEntry([&](int, int c, double v) { x[c] += v * v; });
  }
  void ScaleColumns(const double* scale) final {
    const CompressedRowBlockStructure* bs = block_structure_.get();
    for (auto& row : bs->rows)
      for (auto& cell : row.cells) {
        const Block& col = bs->cols[cell.block_id];
        for (int i = 0; i < row.block.size; ++i)
          for (int j = 0; j < col.size; ++j) values_[size_t(cell.position + i * col.size + j)] *= scale[col.position + j];
      }
  }
  void ToDenseMatrix(Matrix* dense_matrix) const final {
    dense_matrix->resize(num_rows_, num_cols_);
    dense_matrix->setZero();
    ForEachEntry([&](int r, int c, double v) { (*dense_matrix)(r, c) = v; });
  }
  void ToTextFile(FILE* file) const final {
    ForEachEntry([&](int r, int c, double v) { std::fprintf(file, "% 10d % 10d %17f\n", r, c, v); });
  }
  int num_rows() const final { return num_rows_; }
  int num_cols() const final { return num_cols_; }
  int num_nonzeros() const final { return num_nonzeros_; }
  const double* values() const final { return values_.data(); }
  double* mutable_values() final { return values_.data(); }
  const CompressedRowBlockStructure* block_structure() const { return block_structure_.get(); }  // :136

 private:
  template <typename Fn>
  void ForEachEntry(Fn fn) const {
    const CompressedRowBlockStructure* bs = block_structure_.get();
    for (auto& row : bs->rows)
      for (auto& cell : row.cells) {
        const Block& col = bs->cols[cell.block_id];
        for (int i = 0; i < row.block.size; ++i)
          for (int j = 0; j < col.size; ++j)
            fn(row.block.position + i, col.position + j, values_[size_t(cell.position + i * col.size + j)]);
      }
  }
  std::unique_ptr<CompressedRowBlockStructure> block_structure_;
  std::vector<double> values_;
  int num_rows_, num_cols_, num_nonzeros_;
};

// ---------------------------------------------------------------- execution_summary.h:45-92
struct CallStatistics {
  CallStatistics() = default;
  absl::Duration time = absl::ZeroDuration();
  int calls{0};
};
class ExecutionSummary {
 public:
  void IncrementTimeBy(const std::string& name, absl::Duration delta) {
    std::lock_guard<std::mutex> l(mutex_);
    CallStatistics& call_stats = statistics_[name];
    call_stats.time += delta;
    ++call_stats.calls;
  }
  const std::map<std::string, CallStatistics>& statistics() const { return statistics_; }
 private:
  std::mutex mutex_;
  std::map<std::string, CallStatistics> statistics_;
};
class ScopedExecutionTimer {
 public:
  ScopedExecutionTimer(std::string name, ExecutionSummary* summary)
      : start_time_(absl::Now()), name_(std::move(name)), summary_(summary) {}
  ~ScopedExecutionTimer() { summary_->IncrementTimeBy(name_, absl::Now() - start_time_); }
 private:
  absl::Time start_time_;
  const std::string name_;
  ExecutionSummary* summary_;
};

// ---------------------------------------------------------------- linear_solver.h:57-394
enum class LinearSolverTerminationType { SUCCESS, NO_CONVERGENCE, FAILURE, FATAL_ERROR };  // :57-74
enum class OrderingType { NATURAL, AMD, NESDIS };                                           // :100-110

class LinearSolver {
 public:
  struct Options {  // :150-230, same members, same defaults (-1 stands for Eigen::Dynamic)
    LinearSolverType type = SPARSE_NORMAL_CHOLESKY;
    PreconditionerType preconditioner_type = JACOBI;
    VisibilityClusteringType visibility_clustering_type = CANONICAL_VIEWS;
    DenseLinearAlgebraLibraryType dense_linear_algebra_library_type = EIGEN;
    SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type = SUITE_SPARSE;
    OrderingType ordering_type = OrderingType::NATURAL;
    bool dynamic_sparsity = false;
    bool use_explicit_schur_complement = false;
    int min_num_iterations = 1;
    int max_num_iterations = 1;
    int max_num_spse_iterations = 5;
    bool use_spse_initialization = false;
    double spse_tolerance = 0.1;
    int num_threads = 1;
    std::vector<int> elimination_groups;
    int residual_reset_period = 10;
    int row_block_size = -1;
    int e_block_size = -1;
    int f_block_size = -1;
    bool use_mixed_precision_solves = false;
    int max_num_refinement_iterations = 0;
    int subset_preconditioner_start_row_block = -1;
    ContextImpl* context = nullptr;
  };
  struct PerSolveOptions {  // :232-318
    double* D = nullptr;
    LinearOperator* preconditioner = nullptr;
    double r_tolerance = 0.0;
    double q_tolerance = 0.0;
  };
  struct Summary {  // :320-326
    double residual_norm = -1.0;
    int num_iterations = -1;
    LinearSolverTerminationType termination_type = LinearSolverTerminationType::FAILURE;
    std::string message;
  };
  virtual ~LinearSolver() = default;                                                           // :335
  virtual Summary Solve(LinearOperator* A, const double* b, const PerSolveOptions& per_solve_options, double* x) = 0;  // :338-341
  virtual std::map<std::string, CallStatistics> Statistics() const { return {}; }              // :348-350
};

template <typename MatrixType>
class TypedLinearSolver : public LinearSolver {  // :363-390
 public:
  LinearSolver::Summary Solve(LinearOperator* A, const double* b, const LinearSolver::PerSolveOptions& per_solve_options,
                              double* x) override {
    ScopedExecutionTimer total_time("LinearSolver::Solve", &execution_summary_);
    return SolveImpl(down_cast<MatrixType*>(A), b, per_solve_options, x);
  }
  std::map<std::string, CallStatistics> Statistics() const override { return execution_summary_.statistics(); }
 private:
  virtual LinearSolver::Summary SolveImpl(MatrixType* A, const double* b,
                                          const LinearSolver::PerSolveOptions& per_solve_options, double* x) = 0;
  ExecutionSummary execution_summary_;
};
using BlockSparseMatrixSolver = TypedLinearSolver<BlockSparseMatrix>;  // :392

// ---------------------------------------------------------------- parameter_block.h / residual_block.h / program.h
class ParameterBlock {  // parameter_block.h:63-380, the read-only accessors
 public:
  ParameterBlock(double* user_state, int size, int index) : user_state_(user_state), size_(size), index_(index) {}
  int Size() const { return size_; }                                                         // :153
  int TangentSize() const { return manifold_ ? manifold_->TangentSize() : size_; }           // :156-158
  bool IsConstant() const { return is_constant_; }                                           // :118
  int index() const { return index_; }                                                       // :162
  int state_offset() const { return state_offset_; }                                         // :166
  int delta_offset() const { return delta_offset_; }                                         // :170
  const double* user_state() const { return user_state_; }                                   // :147
  const Manifold* manifold() const { return manifold_; }                                     // :173
  const double* lower_bounds() const { return nullptr; }                                     // bounds: none in the mirror
  const double* upper_bounds() const { return nullptr; }
  void set_state_offset(int o) { state_offset_ = o; }
  void set_delta_offset(int o) { delta_offset_ = o; }
  void SetManifold(Manifold* m) { manifold_ = m; }
 private:
  double* user_state_;
  int size_, index_;
  int state_offset_ = -1, delta_offset_ = -1;
  bool is_constant_ = false;
  Manifold* manifold_ = nullptr;
};
class ResidualBlock {  // residual_block.h:66-143
 public:
  ResidualBlock(const CostFunction* cost_function, const LossFunction* loss_function,
                const std::vector<ParameterBlock*>& parameter_blocks, int index)
      : cost_function_(cost_function), loss_function_(loss_function), parameter_blocks_(parameter_blocks), index_(index) {}
  const CostFunction* cost_function() const { return cost_function_; }                       // :107
  const LossFunction* loss_function() const { return loss_function_; }                       // :108
  ParameterBlock* const* parameter_blocks() const { return parameter_blocks_.data(); }       // :111-113
  int NumParameterBlocks() const { return int(cost_function_->parameter_block_sizes().size()); }  // :116-118
  int NumResiduals() const { return cost_function_->num_residuals(); }                       // :121
  int index() const { return index_; }                                                       // :128
 private:
  const CostFunction* cost_function_;
  const LossFunction* loss_function_;
  std::vector<ParameterBlock*> parameter_blocks_;
  int index_;
};
class Program {  // program.h:58-197
 public:
  const std::vector<ParameterBlock*>& parameter_blocks() const { return parameter_blocks_; }  // :64
  const std::vector<ResidualBlock*>& residual_blocks() const { return residual_blocks_; }     // :65
  std::vector<ParameterBlock*>* mutable_parameter_blocks() { return &parameter_blocks_; }     // :66
  std::vector<ResidualBlock*>* mutable_residual_blocks() { return &residual_blocks_; }        // :67
  int NumParameterBlocks() const { return int(parameter_blocks_.size()); }                    // :147
  int NumResidualBlocks() const { return int(residual_blocks_.size()); }                      // :146
  int NumResiduals() const { int n = 0; for (auto* r : residual_blocks_) n += r->NumResiduals(); return n; }          // :148
  int NumParameters() const { int n = 0; for (auto* p : parameter_blocks_) n += p->Size(); return n; }                // :149
  int NumEffectiveParameters() const { int n = 0; for (auto* p : parameter_blocks_) n += p->TangentSize(); return n; }  // :150
  // Program::SetParameterOffsetsAndIndex (program.cc:248-278), the offsets half
  void SetParameterOffsetsAndIndex() {
    int state = 0, delta = 0;
    for (auto* p : parameter_blocks_) {
      p->set_state_offset(state);
      p->set_delta_offset(delta);
      state += p->Size();
      delta += p->TangentSize();
    }
  }
 private:
  std::vector<ParameterBlock*> parameter_blocks_;
  std::vector<ResidualBlock*> residual_blocks_;
};

// ---------------------------------------------------------------- evaluator.h:60-167
class Evaluator {
 public:
  virtual ~Evaluator() = default;
  struct Options {  // :64-73
    int num_threads = 1;
    int num_eliminate_blocks = -1;
    LinearSolverType linear_solver_type = DENSE_QR;
    SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type = NO_SPARSE;
    bool dynamic_sparsity = false;
    ContextImpl* context = nullptr;
    EvaluationCallback* evaluation_callback = nullptr;
  };
  virtual std::unique_ptr<SparseMatrix> CreateJacobian() const = 0;  // :95
  struct EvaluateOptions {  // :99-112
    bool apply_loss_function = true;
    bool new_evaluation_point = true;
  };
  virtual bool Evaluate(const EvaluateOptions& evaluate_options, const double* state, double* cost, double* residuals,
                        double* gradient, SparseMatrix* jacobian) = 0;  // :116-121
  bool Evaluate(const double* state, double* cost, double* residuals, double* gradient, SparseMatrix* jacobian) {  // :127-134
    return Evaluate(EvaluateOptions(), state, cost, residuals, gradient, jacobian);
  }
  virtual bool Plus(const double* state, const double* delta, double* state_plus_delta) const = 0;  // :146-148
  virtual int NumParameters() const = 0;           // :151
  virtual int NumEffectiveParameters() const = 0;  // :155
  virtual int NumResiduals() const = 0;            // :158
  virtual std::map<std::string, CallStatistics> Statistics() const { return {}; }  // :164-166
};

// ---------------------------------------------------------------- array_utils.cc:42-75
inline void InvalidateArray(const int64_t size, double* x) {
  if (x != nullptr) for (int64_t i = 0; i < size; ++i) x[i] = kImpossibleValue;
}
inline bool IsArrayValid(const int64_t size, const double* x) {
  if (x != nullptr) for (int64_t i = 0; i < size; ++i) if (!std::isfinite(x[i]) || (x[i] == kImpossibleValue)) return false;
  return true;
}

}  // namespace internal
}  // namespace ceres
#endif
