// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// extern "C" surface of liborc.so, consumed through ctypes by tests/ and by
// bench.py's cpu_baseline leg.
#ifndef ORC_API_H_
#define ORC_API_H_

#include "../include/cxschur.h"

#ifdef __cplusplus
extern "C" {
#endif

void orc_set_num_threads(int n);
int orc_get_num_threads(void);

/* BlockSparseMatrix (block_sparse_matrix.cc) */
void orc_right_multiply(const cx_block_structure* bs, const double* values, const double* x, double* y);
void orc_left_multiply(const cx_block_structure* bs, const double* values, const double* x, double* y);
void orc_squared_column_norm(const cx_block_structure* bs, const double* values, double* x);
void orc_scale_columns(const cx_block_structure* bs, double* values, const double* scale);
/* ToCompressedRowSparseMatrix / ...Transpose (block_sparse_matrix.cc:451-492);
 * rows has num_rows+1 (or num_cols+1) entries; returns nnz */
int64_t orc_to_crs(const cx_block_structure* bs, const double* values, int transpose,
                   int32_t* rows, int32_t* cols, double* vals);

void orc_detect_structure(const cx_block_structure* bs, int num_eliminate_blocks,
                          int* row_block_size, int* e_block_size, int* f_block_size);

/* SchurEliminator (schur_eliminator_impl.h).  lhs: dense row-major n x n with
 * n = sum of f block sizes, upper block triangle written (BlockRandomAccessDenseMatrix);
 * b / rhs / D may be NULL as in the reference. */
int orc_schur_eliminate_dense(const cx_block_structure* bs, const double* values, const double* b,
                              const double* D, int num_eliminate_blocks, double* lhs, double* rhs);
/* same with a BlockRandomAccessDiagonalMatrix lhs (SchurJacobiPreconditioner):
 * blocks = concatenated f_i x f_i row-major diagonal blocks of S (not inverted) */
int orc_schur_eliminate_diagonal(const cx_block_structure* bs, const double* values,
                                 const double* D, int num_eliminate_blocks, double* blocks);
int orc_schur_back_substitute(const cx_block_structure* bs, const double* values, const double* b,
                              const double* D, int num_eliminate_blocks, const double* z, double* x);
/* DenseCholesky::FactorAndSolve on the upper triangle of row-major lhs (lhs is
 * overwritten by the factor).  Returns cx_termination. */
int orc_dense_cholesky_solve(int n, double* lhs, const double* rhs, double* x);
/* ... with a single precision factor (use_float) and / or `refinements` steps of iterative refinement in double
   (FloatEigenDenseCholesky, RefinedDenseCholesky, DenseIterativeRefiner: dense_cholesky.cc:180-204, 322-347, iterative_refiner.cc:83-99);
   lhs (upper triangle, row-major) is left intact */
int orc_dense_cholesky_solve_refined(int n, const double* lhs, const double* rhs, double* x, int use_float, int refinements);

/* ImplicitSchurComplement (implicit_schur_complement.cc): y = S x; rhs optional */
int orc_implicit_schur_multiply(const cx_block_structure* bs, const double* values, const double* D,
                                const double* b, int num_eliminate_blocks, const double* x,
                                double* y, double* rhs);
/* block diagonal of E'E (+D^2) inverse and F'F (+D^2) inverse
 * (partitioned_matrix_view_impl.h:420-658 + AddDiagonalAndInvert) */
int orc_block_diagonal_inverses(const cx_block_structure* bs, const double* values, const double* D,
                                int num_eliminate_blocks, double* ete_inv, double* ftf_inv);

/* LinearSolver::Solve for DENSE_SCHUR / SPARSE_SCHUR(dense S) / ITERATIVE_SCHUR / CGNR */
int orc_solve(const cx_block_structure* bs, const double* values, const double* b, const double* D,
              const cx_solver_options* options, double r_tolerance, double q_tolerance, double* x,
              cx_summary* summary);

/* Visibility based preconditioning (orc_visibility.cpp).  Graphs are edge lists (u, v, w) over vertices 0..n-1.
 * CreateSchurComplementGraph (visibility.cc:77-146): edges u <= v incl. the self edges; returns their number. */
int orc_schur_complement_graph(const cx_block_structure* bs, int num_eliminate_blocks, int32_t* u, int32_t* v, double* w,
                               int capacity);
/* ComputeCanonicalViewsClustering (canonical_views_clustering.cc:81-222): centers[<= n], membership[n] = index of
 * the vertex' centre or -1; returns the number of centres */
int orc_canonical_views(int n, const double* vertex_weights, int num_edges, const int32_t* u, const int32_t* v,
                        const double* w, int min_views, double size_penalty_weight, double similarity_penalty_weight,
                        double view_score_weight, int32_t* centers, int32_t* membership);
/* ComputeSingleLinkageClustering (single_linkage_clustering.cc:42-92): membership[n] = cluster representative;
 * returns the number of clusters */
int orc_single_linkage(int n, int num_edges, const int32_t* u, const int32_t* v, const double* w, double min_similarity,
                       int32_t* membership);
/* Degree2MaximumSpanningForest (graph_algorithms.h:259-339): forest edges (fu < fv), at most n - 1; returns their number */
int orc_degree2_forest(int n, int num_edges, const int32_t* u, const int32_t* v, const double* w, int32_t* fu, int32_t* fv);
/* VisibilityBasedPreconditioner's structure (visibility_based_preconditioner.cc:121-302): membership[num f-blocks],
 * cluster pairs and the f-block pairs of the preconditioner matrix (lexicographic); returns the number of block pairs */
int64_t orc_visibility_structure(const cx_block_structure* bs, int num_eliminate_blocks, int preconditioner_type,
                                 int clustering_type, int32_t* membership, int32_t* num_clusters, int32_t* num_cluster_pairs,
                                 int32_t* cluster_pair_1, int32_t* cluster_pair_2, int32_t cluster_pair_capacity,
                                 int32_t* block_pair_1, int32_t* block_pair_2, int64_t block_pair_capacity);

/* SparseSchurComplementSolver::InitStorage (schur_complement_solver.cc:224-290): the (row, column) f-block
 * ids of the cells of the block-sparse reduced matrix in creation order; returns their number */
int64_t orc_schur_sparse_structure(const cx_block_structure* bs, int num_eliminate_blocks, int32_t* cell_row,
                                   int32_t* cell_col, int64_t capacity);

/* wall seconds of the numeric part of the last orc_solve (structure set-up excluded) */
double orc_last_solve_seconds(void);
void orc_sparse_schur_stats(double* out8);

/* The same solve on one shard of a point-partitioned J; camera-space sums go
 * through the callback (sum-all-reduce in place).  Used by the gloo tests. */
typedef void (*orc_allreduce_fn)(double* buf, int64_t n, void* user);
int orc_solve_sharded(const cx_block_structure* bs, const double* values, const double* b,
                      const double* D, const cx_solver_options* options, double r_tolerance,
                      double q_tolerance, double* x, cx_summary* summary, orc_allreduce_fn fn,
                      void* user);

/* ConjugateGradientsSolver on a small dense SPD system with identity
 * preconditioner (conjugate_gradients_solver_test.cc:46-153) */
int orc_cg_dense(int n, const double* A, const double* b, double* x, int min_num_iterations,
                 int max_num_iterations, int residual_reset_period, double r_tolerance,
                 double q_tolerance, cx_summary* summary);

/* ---- bundle adjustment ---- */
/* SnavelyReprojectionError via forward-mode duals (Jet<double,12>):
 * residual[2], jac_cam[2x9 row-major], jac_pt[2x3]; jacobians may be NULL */
void orc_snavely(const double* camera9, const double* point3, const double* obs2,
                 double* residual, double* jac_cam, double* jac_pt);
void orc_angle_axis_rotate_point(const double* aa, const double* pt, double* out);
void orc_angle_axis_to_rotation_matrix(const double* angle_axis, double* R_column_major);

/* LexicographicallyOrderResidualBlocks (reorder_program.cc:256-338) for BAL:
 * order[k] = input observation index placed at row block k. */
void orc_bal_residual_order(int num_points, int64_t num_obs, const int32_t* point_index,
                            int64_t* order);
/* ComputeStableSchurOrdering (parameter_block_ordering.cc:50-83) on the BAL
 * Hessian graph with parameter blocks in program order cameras 0..C-1, points
 * 0..P-1 (bal_problem / bundle_adjuster.cc:253-267): vertex ids camera i -> i,
 * point j -> C + j.  Returns the independent set size. */
int orc_stable_schur_ordering(int num_cameras, int num_points, int64_t num_obs,
                              const int32_t* camera_index, const int32_t* point_index,
                              int32_t* ordering);
/* Fill a block structure for BAL in the layout BuildJacobianLayout gives
 * (block_jacobian_writer.cc:68-167): caller allocates row_blocks[O],
 * col_blocks[P+C], row_cell_begin[O+1], cells[2O]. */
void orc_bal_structure(int num_cameras, int num_points, int64_t num_obs,
                       const int32_t* camera_index, const int32_t* point_index,
                       const int64_t* order, cx_block* row_blocks, cx_block* col_blocks,
                       int32_t* row_cell_begin, cx_cell* cells);
/* ProgramEvaluator::Evaluate for BAL (program_evaluator.h:137-304): state =
 * [points | cameras]; any of cost/residuals/gradient/values may be NULL */
void orc_bal_evaluate(const cx_block_structure* bs, int num_cameras, int num_points,
                      int64_t num_obs, const int32_t* camera_index, const int32_t* point_index,
                      const double* observations, const int64_t* order, const double* state,
                      double* cost, double* residuals, double* gradient, double* values);
/* LossFunction::Evaluate (loss_function.cc:46-144) for cx_loss_type; rho[3] */
void orc_loss_evaluate(int type, double a, double b, double s, double* rho);
/* Corrector (corrector.cc:41-155): corrects jacobian (row-major num_rows x num_cols, may be
 * NULL) and then residuals in place */
void orc_corrector_apply(double sq_norm, const double* rho, int num_rows, int num_cols,
                         double* residuals, double* jacobian);
/* orc_bal_evaluate with a robust loss on every residual block (residual_block.cc:160-196) */
void orc_bal_evaluate_robust(const cx_block_structure* bs, int num_cameras, int num_points,
                             int64_t num_obs, const int32_t* camera_index, const int32_t* point_index,
                             const double* observations, const int64_t* order, const double* state,
                             int loss_type, double loss_a, double loss_b, double* cost,
                             double* residuals, double* gradient, double* values);

/* ---- quaternion cameras on a manifold (bundle_adjuster --use_quaternions --use_manifolds) ---- */
/* AngleAxisToQuaternion / QuaternionToAngleAxis (rotation.h:315-388), w first */
void orc_angle_axis_to_quaternion(const double* angle_axis, double* quaternion);
void orc_quaternion_to_angle_axis(const double* quaternion, double* angle_axis);
/* QuaternionManifold::Plus / PlusJacobian (manifold.cc:27-78, 4x3 row-major) */
void orc_quaternion_plus(const double* x, const double* delta, double* x_plus_delta);
void orc_quaternion_plus_jacobian(const double* x, double* jacobian);
/* SnavelyReprojectionErrorWithQuaternions + PlusJacobian projection: residual[2], tangent jac_cam[2x9], jac_pt[2x3] */
void orc_snavely_quaternion(const double* camera10, const double* point3, const double* obs2,
                            double* residual, double* jac_cam, double* jac_pt);
/* Evaluator::Plus of the BAL program for a cx_camera_model (state [points | cameras]) */
void orc_bal_plus(int num_cameras, int num_points, int camera_model, const double* x, const double* delta, double* out);
/* orc_bal_evaluate_robust for a cx_camera_model: state has 3P + (9 or 10)C entries, gradient 3P + 9C */
void orc_bal_evaluate_model(const cx_block_structure* bs, int num_cameras, int num_points, int64_t num_obs,
                            const int32_t* camera_index, const int32_t* point_index, const double* observations,
                            const int64_t* order, const double* state, int camera_model, int loss_type,
                            double loss_a, double loss_b, double* cost, double* residuals, double* gradient,
                            double* values);

/* ---- trust region minimizer ---- */
/* The Evaluator / SparseMatrix / LinearSolver operations TrustRegionMinimizer and
 * LevenbergMarquardtStrategy call, as callbacks so that tests can drive the loop with the
 * reference's own test problem (PowellEvaluator2, trust_region_minimizer_test.cc:63-213). */
typedef struct {
  int32_t num_parameters;
  int32_t num_residuals;
  void* user;
  /* Evaluator::Evaluate: returns nonzero on success; residuals/gradient may be NULL;
   * want_jacobian != 0 refreshes the Jacobian the other callbacks act on */
  int (*evaluate)(void* user, const double* x, double* cost, double* residuals, double* gradient,
                  int want_jacobian);
  void (*squared_column_norm)(void* user, double* out);
  void (*scale_columns)(void* user, const double* scale);
  void (*right_multiply)(void* user, const double* x, double* y); /* y += J x */
  /* LinearSolver::Solve: min |J x - b|^2 + |D x|^2; returns cx_termination */
  int (*solve)(void* user, const double* b, const double* D, double q_tolerance, double* x,
               int* num_iterations);
  /* Evaluator::Plus and NumEffectiveParameters for programs with manifolds; plus == NULL: Euclidean,
   * num_effective_parameters is then ignored (= num_parameters) */
  int32_t num_effective_parameters;
  void (*plus)(void* user, const double* x, const double* delta, double* x_plus_delta);
} orc_min_problem;
int orc_minimize(const orc_min_problem* problem, const cx_minimizer_options* options,
                 double* parameters, cx_minimizer_summary* summary,
                 cx_iteration_summary* iterations, int capacity);
/* the same loop on the BAL program with orc_bal_evaluate_robust and orc_solve */
int orc_minimize_bal(int num_cameras, int num_points, int64_t num_obs, const int32_t* camera_index,
                     const int32_t* point_index, const double* observations, int camera_model, int loss_type,
                     double loss_a, double loss_b, const cx_solver_options* solver_options,
                     const cx_minimizer_options* options, double* state,
                     cx_minimizer_summary* summary, cx_iteration_summary* iterations, int capacity);

#ifdef __cplusplus
}
#endif
#endif
