// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// Block-sparse Cholesky of the reduced camera matrix: the CPU stand-in for what the reference gets from
// SuiteSparse at SparseSchurComplementSolver::SolveReducedLinearSystem (schur_complement_solver.cc:292-335 ->
// SuiteSparseCholesky::Factorize / Solve, suitesparse.cc:397-469, with the block ordering of
// suitesparse.cc:218-279).  CHOLMOD is third-party and not under /root/reference: nothing here follows its
// source.  It is the textbook algorithm (George & Liu: elimination tree, symbolic factorisation by merging
// children, left-looking numeric factorisation) on the BLOCKS of S (one row / column per f-block), with a
// nested-dissection ordering by recursive level-structure bisection.  The ordering is "parity unpinned"; parity
// is stated on the solution (sparse_cholesky_test.cc:160-169: against a dense LLT).
//
// Parallelism (OpenMP): columns of one elimination-tree height are independent in a left-looking
// factorisation, so they are computed concurrently; near the root, where a height holds fewer columns than
// threads, the update of a single column is split over row slices instead.
#ifndef ORC_SPARSE_CHOL_H_
#define ORC_SPARSE_CHOL_H_
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

namespace orc {

// Fill-reducing permutation of an undirected graph given as CSR adjacency (no self loops): order[k] = vertex
// eliminated k-th.  Nested dissection: a BFS level structure from a pseudo-peripheral vertex, the middle level
// as separator, both sides first, the separator last; components and small pieces (<= 32 vertices) in BFS order.
std::vector<int> NestedDissectionOrder(int n, const std::vector<int>& adj_begin, const std::vector<int>& adj);

class BlockSparseCholesky {
 public:
  // block_sizes[i]: rows of block i; cells: (i, j) with i <= j present in the upper triangle of S (all (i, i)
  // must be there) -- the cell set SparseSchurComplementSolver::InitStorage creates.
  void Analyze(const std::vector<int>& block_sizes, const std::vector<std::pair<int, int>>& upper_cells);
  // cell_values(i, j) -> pointer to the row-major s_i x s_j block of S for a cell of the analysed set
  // Returns false when S is not positive definite.
  bool Factor(const std::function<const double*(int, int)>& cell_values, int threads);
  void Solve(const double* rhs, double* x, int threads) const;

  int64_t num_factor_blocks() const { return int64_t(row_of_.size()); }
  int64_t num_factor_nonzeros() const { return int64_t(values_.size()); }
  double factor_flops() const { return flops_; }
  int num_heights() const { return int(height_begin_.size()) - 1; }

 private:
  bool FactorColumns(int threads);
  int n_ = 0;
  std::vector<int> size_, pos_;       // block sizes, scalar offsets (original numbering)
  std::vector<int> perm_, iperm_;     // perm_[k] = original block eliminated k-th
  // L by block columns in elimination order: column k owns blocks [col_begin_[k], col_begin_[k+1]), first = diagonal
  std::vector<int64_t> col_begin_;
  std::vector<int> row_of_;           // row (elimination numbering) of each block
  std::vector<int64_t> val_of_;       // offset of each block in values_ (rows(row) x rows(col), row-major)
  std::vector<double> values_;
  // row lists: for row j the (column k < j, block index in column k) pairs, ascending k
  std::vector<int64_t> rowlist_begin_;
  std::vector<int> rowlist_col_;
  std::vector<int64_t> rowlist_blk_;
  // columns grouped by elimination-tree height
  std::vector<int> height_begin_, height_cols_;
  // where the input cells go: (column k, block index) per upper cell, transposed flag
  std::vector<std::pair<int, int>> cells_;
  std::vector<int64_t> cell_blk_;
  double flops_ = 0.0;
};

}  // namespace orc
#endif
