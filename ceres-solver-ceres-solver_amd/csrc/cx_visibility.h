// Visibility based preconditioners (CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL) for ITERATIVE_SCHUR:
// host-side structure (cx_visibility.cpp) and the banded Cholesky that factors and applies them
// (cx_band_chol.hip).  Reference: visibility_based_preconditioner.cc, visibility.cc,
// canonical_views_clustering.cc, single_linkage_clustering.cc, graph_algorithms.h:259-339.
#ifndef CX_VISIBILITY_H_
#define CX_VISIBILITY_H_

#include <utility>
#include <vector>

#include "cx_internal.h"

// One plan per matrix structure and (preconditioner type, clustering type).
struct cx_vis_plan {
  int preconditioner_type = -1, clustering_type = -1;
  // ---- structure, as VisibilityBasedPreconditioner holds it
  int num_clusters = 0;
  std::vector<int32_t> membership;                            // cluster_membership_[camera]
  std::vector<std::pair<int32_t, int32_t>> cluster_pairs;     // cluster_pairs_ (c1 <= c2), lexicographic
  std::vector<int32_t> sel_cells;                             // block_pairs_: ids into the matrix' S cell list, ascending
  // sharded matrix only: the block pairs of ALL ranks (c1 <= c2, lexicographic; from the summed co-visibility counts), the
  // same list on every rank -- what the tile-sparse factorisation has to be planned from, because a rank's own cells
  // (sel_cells) differ from rank to rank while the tile pool is summed over the ranks entry by entry
  std::vector<int32_t> global_pair_c1, global_pair_c2;
  // ---- band layout of the preconditioner matrix M
  // Cameras are ordered path by path of the cluster forest, cluster by cluster along a path, so M is block
  // tridiagonal in cluster blocks: every row r has its non-zeros in columns [r, col_end(r)).  Stored
  // diagonal-aligned: entry (r, c), r <= c <= r + ld, at r * ld + c -- a dense row-major matrix with
  // leading dimension ld as long as nothing left of the diagonal or right of the band is touched.  Paths start
  // at multiples of 32 rows (padding rows carry a unit diagonal), so step s of every path is one launch.
  int32_t N = 0;    // rows incl. padding, multiple of 32
  int32_t ld = 0;
  int32_t num_paths = 0;
  std::vector<int32_t> cam_row;          // [C] first of the camera's 9 rows
  std::vector<int32_t> row_src;          // [N] index into a camera-ordered vector (9 c + a), -1 for padding
  std::vector<int32_t> path_first_blk;   // [num_paths] first 32-row block; paths sorted by decreasing length
  std::vector<int32_t> path_num_blk;     // [num_paths]
  std::vector<int32_t> blk_cend;         // [N / 32] exclusive column end of the block's rows
  std::vector<int32_t> step_paths;       // [max blocks] paths that still have a block at step s
  std::vector<int32_t> step_tiles;       // [max blocks] largest number of 64x64 trailing tiles among them
  int64_t num_sel_items = 0;
  // ---- device
  DevBuf<int32_t> d_sel_cells, d_sel_items, d_sel_offdiag, d_cam_row, d_row_src, d_path_first_blk, d_path_num_blk, d_blk_cend;
  DevBuf<double> d_W, d_F, d_uinv, d_y;
  // ---- alternative to the band: the level-scheduled tile-sparse Cholesky (cx_sparse_chol.hip) on the same cells.
  // Chosen (cxv_use_sparse) when a path of the cluster forest is long: the band walk is then a chain of thousands of
  // dependent block steps per application, while nested dissection of a path gives an elimination tree of logarithmic
  // depth -- block cyclic reduction in effect.  -1 undecided, 0 band, 1 tile-sparse.
  int use_sparse = -1;
  cx_sp_plan sp;
};

// builds (or returns the cached) plan; CX_ERR_UNSUPPORTED with a message when it cannot be built
int cxv_get_plan(cx_matrix* A, int preconditioner_type, int clustering_type, cx_vis_plan** plan);
// M = selected cells of S (+ D_f^2), factor M = U'U; *d_flag raised when a pivot is not positive.
// halve_offdiag: the cells between different clusters are scaled by 1/2 (ScaleOffDiagonalCells).
int cxv_factor(cx_matrix* A, cx_vis_plan* plan, const double* D, bool halve_offdiag, int* d_flag);
// z = M^-1 r (camera order, 9C)
int cxv_solve(cx_matrix* A, cx_vis_plan* plan, const double* r, double* z);

#endif
