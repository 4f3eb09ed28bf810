// Header-compatible mirror of the reference classes at the drop-in boundary.
//
// The Ceres sources cannot be compiled in this image (Eigen / abseil absent), so the
// adapter in cx_linear_solver.h is written against these declarations, which repeat
// the reference's names, members and conventions one to one:
//   Block, Cell, CompressedRow, CompressedRowBlockStructure   block_structure.h:52-182
//   BlockSparseMatrix (container half)                        block_sparse_matrix.h:60-176
//   LinearSolverTerminationType, LinearSolver::{Options,PerSolveOptions,Summary},
//   TypedLinearSolver<MatrixType>                             linear_solver.h:57-390
//   Evaluator::{EvaluateOptions}                              evaluator.h:60-167
//   InvalidateArray / IsArrayValid                            array_utils.h / array_utils.cc:42-75
// Inside a Ceres checkout the adapter includes the real headers instead of this file
// (INTEGRATION.md).
#ifndef CX_CERES_MIRROR_H_
#define CX_CERES_MIRROR_H_

#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace ceres {

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, SCHUR_POWER_SERIES_EXPANSION, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL, SUBSET };
enum VisibilityClusteringType { CANONICAL_VIEWS, SINGLE_LINKAGE };  // types.h:143-173

namespace internal {

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

struct CompressedRowBlockStructure {
  std::vector<Block> cols;
  std::vector<CompressedRow> rows;
};

// Container half of BlockSparseMatrix: structure + values_ (cells row-major at Cell::position).
class BlockSparseMatrix {
 public:
  explicit BlockSparseMatrix(CompressedRowBlockStructure* block_structure) : block_structure_(block_structure) {
    num_rows_ = num_cols_ = 0;
    num_nonzeros_ = 0;
    for (auto& c : block_structure_->cols) num_cols_ += c.size;
    for (auto& r : block_structure_->rows) {
      num_rows_ += r.block.size;
      for (auto& cell : r.cells) num_nonzeros_ += int64_t(r.block.size) * block_structure_->cols[cell.block_id].size;
    }
    values_.assign(size_t(num_nonzeros_), 0.0);
  }
  int num_rows() const { return num_rows_; }
  int num_cols() const { return num_cols_; }
  int64_t num_nonzeros() const { return num_nonzeros_; }
  const double* values() const { return values_.data(); }
  double* mutable_values() { return values_.data(); }
  const CompressedRowBlockStructure* block_structure() const { return block_structure_.get(); }

 private:
  std::unique_ptr<CompressedRowBlockStructure> block_structure_;
  std::vector<double> values_;
  int num_rows_, num_cols_;
  int64_t num_nonzeros_;
};

enum class LinearSolverTerminationType { SUCCESS, NO_CONVERGENCE, FAILURE, FATAL_ERROR };

class LinearOperator;  // linear_operator.h (only named by the interface)

class LinearSolver {
 public:
  struct Options {
    LinearSolverType type = ITERATIVE_SCHUR;
    PreconditionerType preconditioner_type = JACOBI;
    int min_num_iterations = 1;
    int max_num_iterations = 1;
    int num_threads = 1;
    int residual_reset_period = 10;
    std::vector<int> elimination_groups;
    int row_block_size = -1, e_block_size = -1, f_block_size = -1;
    VisibilityClusteringType visibility_clustering_type = CANONICAL_VIEWS;  // linear_solver.h:153
    bool use_explicit_schur_complement = false;  // linear_solver.h:161
    bool use_mixed_precision_solves = false;
    int max_num_refinement_iterations = 0;
    int max_num_spse_iterations = 5;
    bool use_spse_initialization = false;
    double spse_tolerance = 0.1;
    void* context = nullptr;  // ContextImpl* in the reference
  };
  struct PerSolveOptions {
    double* D = nullptr;
    LinearOperator* preconditioner = nullptr;
    double r_tolerance = 0.0;
    double q_tolerance = 0.0;
  };
  struct Summary {
    double residual_norm = -1.0;
    int num_iterations = -1;
    LinearSolverTerminationType termination_type = LinearSolverTerminationType::FAILURE;
    std::string message;
  };
  virtual ~LinearSolver() = default;
  virtual std::map<std::string, double> Statistics() const { return {}; }
};

template <typename MatrixType>
class TypedLinearSolver : public LinearSolver {
 public:
  // linear_solver.h:366-376 (the reference down-casts a LinearOperator*; the mirror takes the typed matrix)
  LinearSolver::Summary Solve(MatrixType* A, const double* b, const LinearSolver::PerSolveOptions& per_solve_options,
                              double* x) {
    return SolveImpl(A, b, per_solve_options, x);
  }

 private:
  virtual LinearSolver::Summary SolveImpl(MatrixType* A, const double* b,
                                          const LinearSolver::PerSolveOptions& per_solve_options, double* x) = 0;
};
using BlockSparseMatrixSolver = TypedLinearSolver<BlockSparseMatrix>;

// array_utils.cc:42-75
constexpr double kImpossibleValue = 1e302;
inline void InvalidateArray(int64_t size, double* x) {
  if (x) for (int64_t i = 0; i < size; ++i) x[i] = kImpossibleValue;
}
inline bool IsArrayValid(int64_t size, const double* x) {
  if (x) for (int64_t i = 0; i < size; ++i) if (!std::isfinite(x[i]) || x[i] == kImpossibleValue) return false;
  return true;
}

}  // namespace internal
}  // namespace ceres
#endif
