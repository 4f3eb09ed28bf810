// Host drivers of the Schur kernels (cx_schur.hip); device pointers throughout.
#ifndef CX_SCHUR_H_
#define CX_SCHUR_H_
#include "cx_internal.h"

// ete_inv[9P] = (E'E + D_e^2)^-1 per point; g[3P] = E'b (optional).  llt selects
// the LLT inverse (ImplicitSchurComplement) or the cofactor inverse (SchurEliminator).
int cxs_compute_ete_inverse(cx_matrix* A, const double* D, const double* b, double* ete_inv, double* g,
                            bool llt, int* d_flag);
// mode 0: out[2O] = (I - E (E'E)^-1 E') F xf ; 1: out[2O] = (I - E (E'E)^-1 E') b ;
// 2: out[3P] = (E'E)^-1 E' (b - F xf) ; 3: out[2O] = E (E'E)^-1 E' F xf
int cxs_chunk_pass(cx_matrix* A, int mode, const double* ete_inv, const double* xf, const double* b, double* out);
// blocks[81C] = block diagonal of F'F (with_schur = false) or of S without D_f^2 (true)
int cxs_camera_block_diagonal(cx_matrix* A, double* blocks);
// fused set-up of the implicit Schur complement (k_chunk_init + k_cam_init), see cx_schur.hip
int cxs_implicit_init(cx_matrix* A, const double* D, const double* b, bool want_blocks, bool with_schur,
                      double* ete_inv, double* rows_scratch, double* blocks, double* rhs_out, int* d_flag,
                      bool defer_reduce = false);
// blocks[81C] = block diagonal of F'F, ftb[9C] = F't (t: one value per scalar row), one pass over the camera-major copy
int cxs_camera_blocks_and_ft(cx_matrix* A, const double* t, double* blocks, double* ftb);
int cxs_block9_add_diag_invert(cx_context* ctx, double* blocks, const double* Df, int C, int* d_flag);
// dense lhs (9C x 9C row-major, upper block triangle) and rhs of the reduced system
int cxs_eliminate_dense(cx_matrix* A, const double* b, const double* D, bool add_df, double* lhs, double* rhs);
// the same elimination into the block-sparse upper-stored S of the matrix (A->d_S, cell list A->d_cell_*;
// D_f^2 NOT added, see k_pair_cells); CX_ERR_UNSUPPORTED when the pair list is too large to build
int cxs_build_pair_lists(cx_matrix* A);
int cxs_eliminate_sparse(cx_matrix* A, const double* b, const double* D, double* rhs, bool f32_operands = false);
// the two halves of the gather elimination on their own (tile-sparse SPARSE_SCHUR): per-item pair sums, F'F
// diagonal blocks and (E'E + D^2)^-1 into the matrix' scratch; rhs = F'(b - E (E'E)^-1 E'b)
// item_ids (device, optional): only these work items are summed (the cells a visibility preconditioner keeps)
// f32_operands (use_mixed_precision_solves: the cells feed a single precision factor): the per-row operand H = K'B is kept
// in float, one 128-byte line per row, products and sums in double
// rhs (optional): the reduced right-hand side F'(b - E (E'E)^-1 E'b) out of the same passes (b may be NULL: zeros)
int cxs_assemble_pair_items(cx_matrix* A, const double* D, const int32_t* item_ids = nullptr, int64_t num_selected = 0,
                            bool f32_operands = false, const double* b = nullptr, double* rhs = nullptr);
int cxs_eliminate_rhs(cx_matrix* A, const double* b, double* rhs);
// tile-sparse Cholesky (cx_sparse_chol.hip).  For the explicit S of a matrix: plan into A->sp, then assemble + factor
// + solve in one call.
int cxsp_build_plan(cx_matrix* A);
int cxsp_factor_and_solve(cx_matrix* A, const double* Df, const double* rhs, double* z, int* d_flag);
// Sharded matrix (A->ctx->nranks > 1): the same plan on every rank, from the union of the ranks' cells (one dense
// presence exchange, at most 16 384 cameras); per solve A->d_S (this rank's cell values, cxs_eliminate_sparse) is summed
// over the ranks in the union's cell order and the factorisation and the triangular solves are replicated.
int cxsp_build_plan_sharded(cx_matrix* A);
int cxsp_factor_and_solve_sharded(cx_matrix* A, const double* Df, const double* rhs, double* z, int* d_flag);
// The pieces, for any symmetric matrix of 9x9 camera blocks given by a subset of A's S cells (the visibility based
// preconditioners): plan from the cells (c1 <= c2, every diagonal cell present), assembly of the selected cells
// (sel_cells = ids into A's cell list or NULL for all; sel_offdiag / offdiag_scale as k_band_assemble), numeric
// factorisation in place, and M^-1 r by a forward and a backward sweep over the levels (r, z in camera order).
// distribute: on a sharded context, give every rank the factorisation of its own subtrees of the elimination tree (only
// the solve-at-once path, cxsp_factor_and_solve_sharded, knows how to use such a plan)
int cxsp_plan_from_cells(cx_context* ctx, int C, const int32_t* cell_c1, const int32_t* cell_c2, int64_t num_cells, cx_sp_plan* plan,
                         bool distribute = false);
int cxsp_assemble(cx_matrix* A, cx_sp_plan* plan, const double* Df, const int32_t* sel_cells, const int32_t* sel_offdiag,
                  int64_t num_sel, double offdiag_scale);
int cxsp_factor(cx_context* ctx, cx_sp_plan* plan, int* d_flag);
int cxsp_solve(cx_context* ctx, cx_sp_plan* plan, const double* r, double* z);
// y = S x with that storage (BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate); blocks[c] = S(c,c)
int cxs_sparse_multiply(cx_matrix* A, const double* x, double* y);
int cxs_sparse_diagonal(cx_matrix* A, double* blocks);

// dense Cholesky (cx_cholesky.hip): factor the upper triangle of row-major a (n x n) in
// place (a = U'U) and solve a x = rhs.  *d_flag set to 1 when not positive definite.
int cxd_cholesky_solve(cx_context* ctx, int n, double* a, const double* rhs, double* x, int* d_flag);
#endif
