/*
 * cxschur.h -- C ABI of libcxschur: the MI355X (gfx950) implementation of the
 * Levenberg-Marquardt inner linear solve of Ceres Solver (block-sparse
 * Jacobian evaluation, J products, Schur elimination, reduced camera solve,
 * back substitution, CGNR).
 *
 * Every entry point names the reference interface it stands in for
 * (file:line relative to the Ceres source tree).  Signatures use only plain
 * pointers and sizes; nothing here depends on torch, Eigen or abseil.  All
 * functions return 0 on success and a negative cx_status on error; the text of
 * the last error of the calling thread is available from cx_last_error().
 *
 * Conventions shared with the reference:
 *   - all small dense blocks are ROW-major (block_random_access_matrix.h:66-68)
 *   - the solver minimises |A x - b|^2 + |D x|^2; D may be NULL
 *     (linear_solver.h:148-354)
 *   - termination types carry the numeric order of linear_solver.h:57-74
 */
#ifndef CXSCHUR_H_
#define CXSCHUR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* == ceres::internal::Block (block_structure.h:54-60) */
typedef struct { int32_t size; int32_t position; } cx_block;
/* == ceres::internal::Cell (block_structure.h:66-75) */
typedef struct { int32_t block_id; int32_t position; } cx_cell;

/* Flattened ceres::internal::CompressedRowBlockStructure
 * (block_structure.h:84-182).  The reference keeps one std::vector<Cell> per
 * row; here row i owns cells[row_cell_begin[i] .. row_cell_begin[i+1]).  This is
 * the flattening the reference itself performs for its device code
 * (cuda_block_structure.cc:50-236). */
typedef struct {
  int32_t num_row_blocks;
  int32_t num_col_blocks;
  const cx_block* row_blocks;      /* [num_row_blocks] size/position of each row block */
  const cx_block* col_blocks;      /* [num_col_blocks] */
  const int32_t* row_cell_begin;   /* [num_row_blocks + 1] */
  const cx_cell* cells;            /* [row_cell_begin[num_row_blocks]] */
} cx_block_structure;

typedef enum {
  CX_OK = 0,
  CX_ERR_INVALID_ARGUMENT = -1,
  CX_ERR_HIP = -2,            /* a HIP runtime call failed            */
  CX_ERR_NO_DEVICE = -3,      /* no gfx950 device / code object       */
  CX_ERR_COMM = -4,           /* RCCL failure                          */
  CX_ERR_UNSUPPORTED = -5,
  CX_ERR_OUT_OF_MEMORY = -6
} cx_status;

/* linear_solver.h:57-74, same numeric order */
typedef enum {
  CX_SUCCESS = 0,
  CX_NO_CONVERGENCE = 1,
  CX_FAILURE = 2,
  CX_FATAL_ERROR = 3
} cx_termination;

/* include/ceres/types.h:57-91 (LinearSolverType); only the Schur / CGNR
 * members exist here. */
typedef enum {
  CX_DENSE_SCHUR = 0,      /* explicit S, dense storage, dense Cholesky          */
  CX_SPARSE_SCHUR = 1,     /* explicit S; dense below 2048 cameras, tile-sparse Cholesky above */
  CX_ITERATIVE_SCHUR = 2,  /* implicit S + PCG                                   */
  CX_CGNR = 3              /* PCG on J'J + D'D                                   */
} cx_linear_solver_type;

/* include/ceres/types.h:93-141 (PreconditionerType) */
typedef enum {
  CX_IDENTITY = 0,
  CX_JACOBI = 1,
  CX_SCHUR_JACOBI = 2,
  CX_SCHUR_POWER_SERIES_EXPANSION = 3,
  /* VisibilityBasedPreconditioner (visibility_based_preconditioner.cc:68-577): the cells of S whose cameras share
   * a visibility cluster (CLUSTER_JACOBI), plus the cells between clusters joined by an edge of the degree-2
   * maximum spanning forest of the cluster graph (CLUSTER_TRIDIAGONAL), factored by Cholesky.  ITERATIVE_SCHUR
   * on the implicit S only, as in the reference. */
  CX_CLUSTER_JACOBI = 4,
  CX_CLUSTER_TRIDIAGONAL = 5
} cx_preconditioner_type;

/* include/ceres/types.h:143-173 (VisibilityClusteringType) */
typedef enum { CX_CANONICAL_VIEWS = 0, CX_SINGLE_LINKAGE = 1 } cx_visibility_clustering_type;

/* Built-in LossFunction of a residual block (include/ceres/loss_function.h:171-292); a, b are
 * the constructor arguments (b only for TOLERANT). */
typedef enum {
  CX_LOSS_NONE = 0, /* loss_function == nullptr / TrivialLoss */
  CX_LOSS_HUBER = 1,
  CX_LOSS_SOFT_L_ONE = 2,
  CX_LOSS_CAUCHY = 3,
  CX_LOSS_ARCTAN = 4,
  CX_LOSS_TOLERANT = 5,
  CX_LOSS_TUKEY = 6
} cx_loss_type;

/* Camera parameterisation of the bundle-adjustment evaluator (examples/bundle_adjuster.cc:316-346):
 * CX_CAMERA_ANGLE_AXIS: 9 parameters (angle-axis 3, translation 3, focal, k1, k2), Euclidean --
 *   SnavelyReprojectionError, snavely_reprojection_error.h:53-104;
 * CX_CAMERA_QUATERNION_MANIFOLD: 10 parameters (quaternion w x y z, translation 3, focal, k1, k2) on
 *   ProductManifold<QuaternionManifold, EuclideanManifold<6>> (--use_quaternions --use_manifolds):
 *   SnavelyReprojectionErrorWithQuaternions :111-170; tangent size 9, so J keeps the <2,3,9> structure. */
typedef enum { CX_CAMERA_ANGLE_AXIS = 0, CX_CAMERA_QUATERNION_MANIFOLD = 1 } cx_camera_model;

/* Where b, D, x, state ... pointers of a call live. */
typedef enum { CX_HOST = 0, CX_DEVICE = 1 } cx_memspace;

/* LinearSolver::Options (linear_solver.h:150-230), the fields this path reads. */
typedef struct {
  int32_t type;                       /* cx_linear_solver_type */
  int32_t preconditioner_type;        /* cx_preconditioner_type */
  int32_t min_num_iterations;         /* default 0   */
  int32_t max_num_iterations;         /* default 500 (solver.h max_linear_solver_iterations) */
  int32_t residual_reset_period;      /* default 10  */
  int32_t num_eliminate_blocks;       /* == elimination_groups[0]; 0 for CGNR */
  int32_t use_mixed_precision_solves; /* DENSE_SCHUR / SPARSE_SCHUR (solver.h:572-585; dense_cholesky.cc:84-136, sparse_cholesky.cc:
                                       * 45-118): S, computed in double, is factored in SINGLE precision (float tile pool, fp32
                                       * matrix instructions).  CGNR / ITERATIVE_SCHUR (not in the reference): the CG operator
                                       * streams fp32 copies of the J values (fp64 accumulation and vectors) */
  int32_t max_num_refinement_iterations; /* solver.h:587-590; DENSE_SCHUR / SPARSE_SCHUR: that many steps of residual = rhs - S z in
                                       * double, z += factor^-1 residual, after the solve -- with either factor, as
                                       * RefinedDenseCholesky / RefinedSparseCholesky wrap either (iterative_refiner.cc) */
  int32_t max_num_spse_iterations;    /* default 5 */
  int32_t use_spse_initialization;    /* default 0 */
  double spse_tolerance;              /* default 0.1 */
  int32_t deterministic;              /* 1: camera-space sums in fixed order (bitwise reproducible) */
  int32_t use_explicit_schur_complement; /* ITERATIVE_SCHUR on an explicitly computed block-sparse S (solver.h:518-540) */
  int32_t visibility_clustering_type; /* cx_visibility_clustering_type, CLUSTER_* preconditioners (solver.h:337-338) */
  int32_t reserved;
} cx_solver_options;

/* LinearSolver::PerSolveOptions (linear_solver.h:232-318) */
typedef struct {
  const double* D;       /* [num_cols] or NULL */
  double r_tolerance;    /* default -1 */
  double q_tolerance;    /* default 0  */
  int32_t memspace;      /* cx_memspace of b, D and x */
  int32_t b_on_device;   /* non-zero: b is a device pointer whatever memspace says (the residuals an evaluator
                          * left in HBM, cx_evaluator_device_residuals) */
} cx_per_solve_options;

/* LinearSolver::Summary (linear_solver.h:320-326) + notes: options that were answered differently from how they were
 * asked, machine-readable (the message says the same in words; the host adapter logs them):
 *   CX_NOTE_DOUBLE_PRECISION_FACTOR      use_mixed_precision_solves on a dynamic-size structure (or a dense S no tile plan
 *                                        fits): the factorisation ran in double precision -- more accurate than asked, not faster;
 *   CX_NOTE_SPSE_INITIALIZATION_SKIPPED  use_spse_initialization on a dynamic-size structure: CG started from zero. */
enum { CX_NOTE_DOUBLE_PRECISION_FACTOR = 1, CX_NOTE_SPSE_INITIALIZATION_SKIPPED = 2 };
typedef struct {
  double residual_norm;
  int32_t num_iterations;
  int32_t termination_type;  /* cx_termination */
  char message[256];
  int32_t notes;             /* CX_NOTE_* bits */
  int32_t reserved;
} cx_summary;

/* Phase timings of the last solve in milliseconds (device time, HIP events);
 * the phases are the reference's EventLogger events
 * (schur_complement_solver.cc:106-155, iterative_schur_complement_solver.cc:64-157). */
typedef struct {
  double setup_ms;
  double eliminate_ms;        /* Eliminate, or implicit-Schur Init */
  double reduced_solve_ms;    /* Cholesky, or the CG loop          */
  double back_substitute_ms;
  double total_ms;
  double allreduce_ms;        /* DEVICE time inside the exchange step of a sharded solve: HIP event pairs around every
                               * ncclAllReduce on the context's stream, summed (the first 256 collectives of a solve are
                               * timed, more are extrapolated from them); 0 on one rank */
  double allreduce_host_ms;   /* host time spent enqueueing them (or, with the callback transport, inside the callback) */
  double allreduce_calls;     /* collectives of the solve */
  double allreduce_bytes;     /* payload summed over them (8 bytes per double) */
  double sampled;             /* 1: the phase times above (and cx_solver_kernel_stats) were taken during THIS solve; 0: they
                                 are those of the last solve that took them (launch-bound solvers, see cx_solver_last_timing) */
} cx_solve_timing;

/* Device time of one kernel (or short kernel sequence) of the hot loop during the
 * last solve: HIP event pairs on the context's stream around the first
 * sampled_launches launches (at most 64); launches counts all of them. */
typedef struct {
  char name[64];
  double sampled_ms;
  int32_t sampled_launches;
  int32_t launches;
} cx_kernel_stat;

typedef struct cx_context cx_context;
typedef struct cx_matrix cx_matrix;
typedef struct cx_solver cx_solver;
typedef struct cx_evaluator cx_evaluator;

/* ---------------------------------------------------------------- context */

/* One context per process and GPU: a HIP stream plus, after cx_context_set_comm,
 * an RCCL communicator.  Stands in for ContextImpl (context_impl.h:60-150),
 * whose CUDA half owns the stream and library handles. */
int cx_context_create(int device_id, cx_context** out);
void cx_context_destroy(cx_context* ctx);

/* Multi-GPU (new; the reference is single-device, context_impl.h:74-83).
 * cx_comm_unique_id fills a 128-byte id on one rank; the caller distributes it
 * (any out-of-band channel) and every rank calls cx_context_set_comm.  From then
 * on solvers and evaluators on this context treat their matrix as one shard of
 * a point-partitioned Jacobian and all-reduce camera-space sums. */
int cx_comm_unique_id(void* out_128_bytes);
int cx_context_set_comm(cx_context* ctx, int rank, int nranks, const void* unique_id_128_bytes);
/* Alternative transport for the same exchange step: fn must sum the n doubles at device_ptr over
 * all ranks in place and return 0 (it is called with the context's stream drained).  Meant for
 * rehearsing the sharded path where RCCL peers are not available (two ranks sharing one GPU with
 * a host-staged gloo all-reduce in tests/); RCCL remains the production data path. */
typedef int (*cx_allreduce_fn)(double* device_ptr, int64_t n, void* user);
int cx_context_set_comm_callback(cx_context* ctx, int rank, int nranks, cx_allreduce_fn fn, void* user);
int cx_context_rank(const cx_context* ctx);
int cx_context_num_ranks(const cx_context* ctx);
/* sum-all-reduce of n doubles in place on the context's stream (exposed for tests) */
int cx_allreduce_sum(cx_context* ctx, double* device_ptr, int64_t n);
/* A sharded call cannot hang on a lost rank.  (1) Where a sharded phase starts (cx_solver_solve, cx_evaluator_evaluate
 * after their inputs are staged; the structure exchange of a sharded SPARSE_SCHUR) every rank first tells the others
 * whether it is still healthy -- a rank that failed locally still takes part, flag set, and ALL ranks return an error
 * together.  (2) On an RCCL communicator every wait polls with a deadline; when no progress is seen for `seconds`
 * (default: CX_COMM_TIMEOUT_S in the environment, else 120) the communicator is aborted (ncclCommAbort ends the
 * collective the stream is stuck in), the context is marked broken and the call returns CX_ERR_COMM -- FATAL_ERROR in
 * the summary, which makes TrustRegionMinimizer stop (trust_region_minimizer.cc:404-411).  Nothing is retried.  (3) In a
 * multi-shard front the worker of a failing shard releases the others at once (in-process transport) or aborts every
 * shard's communicator (RCCL); an RCCL front is unusable afterwards and says so. */
int cx_context_set_comm_timeout(cx_context* ctx, double seconds);
/* Test hooks.  cx_debug_inject_failure: the context (on a multi-shard front: its shard `shard`) fails locally right
 * before the nth collective it enters from now on (0 = the next one; -1 clears) -- how the tests make one shard drop
 * out in the middle of a solve.  cx_debug_stall_stream: a host callback that sleeps that long is enqueued on the
 * context stream (a stream that makes no progress, for the deadline test). */
int cx_debug_inject_failure(cx_context* ctx, int32_t shard, int64_t nth_collective);
int cx_debug_stall_stream(cx_context* ctx, int32_t milliseconds);
/* ... and the context reports (and waits as if it had) that many ranks: on a one-GPU box RCCL forms one-rank
 * communicators only, this makes the deadline path of a several-rank communicator reachable there. */
int cx_debug_force_rank_count(cx_context* ctx, int32_t nranks);

/* Several GPUs behind ONE set of handles in ONE process -- how a Solver::Solve caller (one process, one ContextImpl,
 * context_impl.h:74-83; LinearSolver::Solve and Evaluator::Evaluate called with whole vectors, linear_solver.h:363-390,
 * evaluator.h:120-150) reaches the sharded path.  device_ids[num_shards] names either distinct devices (one shard per
 * GPU; the shards' exchange step is ncclAllReduce on one RCCL communicator per device, each driven by its own worker
 * thread) or the same device throughout (logical shards on one GPU with an in-process sum: the configuration the
 * one-GPU test boxes run).  On such a context
 *   cx_matrix_create        cuts the e-blocks (points) into num_shards contiguous ranges of about equal non-zeros
 *                           (cx_partition_points) and gives every shard the rows of its range;
 *   cx_evaluator_create_bal does the same with the observations;
 *   cx_solver_solve / cx_evaluator_evaluate / cx_minimize / the matrix products take and return WHOLE vectors in host
 *                           memory (memspace must be CX_HOST; per_solve.b_on_device accepts the token of
 *                           cx_evaluator_device_residuals), scatter them, run the per-rank solvers of the shards side by
 *                           side and gather the result;
 * everything else of the ABI (cx_malloc, the parity-test entry points, ...) addresses the first device only.
 * At most 16 shards. */
int cx_context_create_multi(int num_shards, const int* device_ids, cx_context** out);
int cx_context_num_shards(const cx_context* ctx);   /* 1 for a plain context */
/* How a matrix on a multi-shard context is cut: returns the number of shards n and, when capacity >= n + 1, the first
 * e-block and the first row block of every shard (n + 1 entries each; either array may be NULL). */
int cx_matrix_shard_layout(const cx_matrix* A, int32_t* e_block_bounds, int32_t* row_block_bounds, int32_t capacity);

/* ---- host vectors at the boundary.
 * LinearSolver::Solve and Evaluator::Evaluate hand over HOST arrays (linear_solver.h:363-390, evaluator.h:120-150), and
 * TrustRegionMinimizer / LevenbergMarquardtStrategy pass the same long-lived arrays every LM iteration
 * (trust_region_minimizer.cc:181-203, levenberg_marquardt_strategy.cc:77-99).  A caller that can vouch for the lifetime
 * of those arrays may have the library register them with the HIP runtime (hipHostRegister): their copies then are plain
 * DMA that do not block the calling thread.  OFF by default, for two measured reasons: pageable
 * copies of vectors this size already run at the PCIe rate on MI355X hosts (the runtime pins a large copy's pages in
 * place), and a registration cannot see its array die -- a registered range that is freed and whose addresses come back
 * with a later allocation faults the GPU when it is copied.  THE CONTRACT of switching it on: every array handed to the
 * library stays allocated until cx_host_registrations_release() has been called (the host adapters call it in their
 * destructors; TrustRegionMinimizer's vectors live until Minimize returns).
 *   sightings        0 (default): never register; 1: the first time an array is handed in; 2: the second time;
 *   min_bytes        smaller arrays stay pageable (default 256 KiB);
 *   max_total_bytes  default 16 GiB; at most 64 arrays, least recently used go first.
 * Environment (read once, an explicit call wins): CX_PIN=0|1|2, CX_PIN_MIN_KB, CX_PIN_MAX_MB.
 * cx_host_register pins one array now (whatever the policy); cx_host_registrations_release unregisters everything. */
int cx_host_registration_policy(int32_t sightings, int64_t min_bytes, int64_t max_total_bytes);
int cx_host_register(const void* host_ptr, size_t bytes);
int cx_host_registrations_release(void);
/* What crossed PCIe through this context's calls since the last reset (a multi-shard front: bytes and copies summed
 * over the shards, durations of the slowest shard).  *_ms are device-side durations of the copies (HIP events on the
 * stream each copy ran on), *_registered_bytes the part that went to / from registered memory. */
typedef struct {
  int64_t h2d_bytes, d2h_bytes;
  int64_t h2d_copies, d2h_copies;
  int64_t h2d_registered_bytes, d2h_registered_bytes;
  double h2d_ms, d2h_ms;
  int64_t registered_bytes;      /* registry: bytes registered now */
  int32_t num_registered;        /* registry: arrays registered now */
  int32_t reserved;
  double register_ms;            /* host time spent in hipHostRegister since the last reset */
  int64_t num_register_calls;
} cx_transfer_stats;
int cx_transfer_stats_get(cx_context* ctx, cx_transfer_stats* out, int32_t reset);

int cx_malloc(cx_context* ctx, size_t bytes, void** device_ptr);
int cx_free(cx_context* ctx, void* device_ptr);
int cx_memcpy_h2d(cx_context* ctx, void* dst_device, const void* src_host, size_t bytes);
int cx_memcpy_d2h(cx_context* ctx, void* dst_host, const void* src_device, size_t bytes);
int cx_memset_zero(cx_context* ctx, void* device_ptr, size_t bytes);
int cx_synchronize(cx_context* ctx);
/* the hipStream_t all work of this context is enqueued on */
void* cx_context_stream(cx_context* ctx);
const char* cx_last_error(void);
/* "gfx950 ..." description of the device the context is bound to */
int cx_device_name(cx_context* ctx, char* out, size_t n);

/* ----------------------------------------------------------------- matrix */

/* Device-resident BlockSparseMatrix (block_sparse_matrix.h:60-176).  The value
 * array has exactly the reference layout (cells row-major at Cell::position), so
 * a host BlockSparseMatrix::values() array can be uploaded verbatim.
 * num_eliminate_blocks > 0 declares the [E F] partition of
 * PartitionedMatrixView (partitioned_matrix_view.h:60-140). */
int cx_matrix_create(cx_context* ctx, const cx_block_structure* bs,
                     int32_t num_eliminate_blocks, cx_matrix** out);
void cx_matrix_destroy(cx_matrix* A);
int64_t cx_matrix_num_rows(const cx_matrix* A);
int64_t cx_matrix_num_cols(const cx_matrix* A);
int64_t cx_matrix_num_nonzeros(const cx_matrix* A);
/* 1 when the <2,3,9> bundle-adjustment kernels are in use (detect_structure.cc:39-120) */
int cx_matrix_is_static_239(const cx_matrix* A);
/* Which kernels serve this matrix: 0 the dynamic-size path (Eigen::Dynamic instantiations of the reference,
 * schur_eliminator.cc:140-142), 1 the static <2,3,9> path on the caller's own layout, 2 the static path through an
 * embedded <2,3,9> image -- structures with 2-row e-rows, e-blocks of one size e <= 3 and f-blocks of one size f <= 9
 * (<2,3,6>, <2,3,3>, <2,3,4>, <2,2,2>, <2,2,3>, <2,2,4>, or <2,3,9> in another cell layout), optionally followed by rows
 * that hold a single f cell (NoEBlockRowsUpdate, schur_eliminator_impl.h:567-659). */
int cx_matrix_static_path(const cx_matrix* A);
/* BlockSparseMatrix::mutable_values(): device pointer to num_nonzeros doubles */
double* cx_matrix_device_values(cx_matrix* A);
/* copy values in (memspace says where src lives) */
int cx_matrix_set_values(cx_matrix* A, const double* src, int32_t memspace);
int cx_matrix_get_values(const cx_matrix* A, double* dst_host);
/* call after writing through cx_matrix_device_values(): drops cached copies */
int cx_matrix_values_changed(cx_matrix* A);
/* BlockSparseMatrix::SetZero (block_sparse_matrix.cc:220-225) */
int cx_matrix_set_zero(cx_matrix* A);
/* y += A x   BlockSparseMatrix::RightMultiplyAndAccumulate (block_sparse_matrix.cc:239-274) */
int cx_matrix_right_multiply(cx_matrix* A, const double* x, double* y, int32_t memspace);
/* y = A x: the product for a caller that knows y to be zero beforehand -- TrustRegionMinimizer's model cost
 * (trust_region_minimizer.cc:430-433: model_residuals_.setZero() immediately before RightMultiplyAndAccumulate).  With
 * host vectors it saves the upload of num_rows zeros; the bits of y equal those of zero-filling y and calling
 * cx_matrix_right_multiply. */
int cx_matrix_right_multiply_overwrite(cx_matrix* A, const double* x, double* y, int32_t memspace);
/* y += A' x  BlockSparseMatrix::LeftMultiplyAndAccumulate (block_sparse_matrix.cc:278-349) */
int cx_matrix_left_multiply(cx_matrix* A, const double* x, double* y, int32_t memspace);
/* x = diag(A'A)  BlockSparseMatrix::SquaredColumnNorm (block_sparse_matrix.cc:351-401) */
int cx_matrix_squared_column_norm(cx_matrix* A, double* x, int32_t memspace);
/* A <- A diag(scale)  BlockSparseMatrix::ScaleColumns (block_sparse_matrix.cc:403-450) */
int cx_matrix_scale_columns(cx_matrix* A, const double* scale, int32_t memspace);
/* PartitionedMatrixView<2,3,9> (partitioned_matrix_view_impl.h:92-320): products with the E
 * half (first num_eliminate_blocks column blocks) or the F half of A.  x_e / y_e have
 * num_cols_e entries, x_f / y_f num_cols_f; all accumulate (y += ...). */
int cx_matrix_right_multiply_e(cx_matrix* A, const double* x_e, double* y, int32_t memspace);
int cx_matrix_right_multiply_f(cx_matrix* A, const double* x_f, double* y, int32_t memspace);
int cx_matrix_left_multiply_e(cx_matrix* A, const double* x, double* y_e, int32_t memspace);
int cx_matrix_left_multiply_f(cx_matrix* A, const double* x, double* y_f, int32_t memspace);
/* device time of the last cx_matrix_* product in ms (HIP events on the context stream) */
double cx_matrix_last_kernel_ms(const cx_matrix* A);

/* ----------------------------------------------------------------- solver */

/* LinearSolver::Create (linear_solver.cc:75-128) for the Schur / CGNR members. */
int cx_solver_create(cx_context* ctx, const cx_solver_options* options, cx_solver** out);
void cx_solver_destroy(cx_solver* s);
void cx_solver_default_options(cx_solver_options* options);
/* TypedLinearSolver<BlockSparseMatrix>::Solve (linear_solver.h:363-390):
 * b has num_rows entries, x num_cols; x is fully overwritten. */
int cx_solver_solve(cx_solver* s, cx_matrix* A, const double* b,
                    const cx_per_solve_options* per_solve, double* x, cx_summary* summary);
/* Phase times of the last solve.  A launch-bound solver (one rank, at most 4096 reduced unknowns: some twenty enqueues of
 * 2-17 us kernels, where every timing event costs about as much as a kernel) takes phase times and kernel samples on its
 * first solve and on every 16th after it (CX_DIAG_PERIOD=n in the environment: every n-th, 1 = always); in between the
 * phases and kernel samples of the last sampled solve stay (cx_solve_timing.sampled = 0 says so), and total_ms is host
 * wall time around the solve. */
int cx_solver_last_timing(const cx_solver* s, cx_solve_timing* out);
/* The next solve of s takes phase times and kernel samples whatever the period (a caller that wants them for a particular
 * solve of a launch-bound solver). */
int cx_solver_sample_next(cx_solver* s);
/* per-kernel device times of the last (sampled, see above) solve (ExecutionSummary of the reference is host
 * wall time per phase, execution_summary.h:45-83; this is its device-side counterpart) */
int cx_solver_kernel_stats(const cx_solver* s, cx_kernel_stat* out, int32_t capacity, int32_t* count);

/* The Schur pieces on their own, for parity tests against
 * schur_eliminator_test.cc / implicit_schur_complement_test.cc.
 * SchurEliminator::Eliminate (schur_eliminator_impl.h:177-304) into a dense
 * row-major lhs of order num_cols_f (upper block triangle, as the reference) */
int cx_schur_eliminate_dense(cx_context* ctx, cx_matrix* A, const double* b, const double* D,
                             double* lhs, double* rhs, int32_t memspace);
/* SchurEliminator::BackSubstitute (schur_eliminator_impl.h:307-373): z has
 * num_cols_f entries, x num_cols; only the e-part of x is written. */
/* The cell set of the block-sparse reduced camera matrix, in the order
 * SparseSchurComplementSolver::InitStorage creates it (schur_complement_solver.cc:224-290): every diagonal
 * cell (i, i) and every (i, j), i < j, of two f-blocks observed together in some chunk, lexicographic.
 * Static <2,3,9> matrices only.  *num_cells is always set; the arrays (may be NULL) receive up to
 * capacity entries of the (row block, column block) ids. */
int cx_schur_sparse_structure(cx_matrix* A, int64_t* num_cells, int32_t* cell_row, int32_t* cell_col, int64_t capacity);
/* The structure of a CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL preconditioner as VisibilityBasedPreconditioner holds it
 * (visibility_based_preconditioner.h:130-199): cluster_membership_[num cameras], num_clusters_, cluster_pairs_
 * (c1 <= c2, lexicographic) and block_pairs_ (f-block pairs b1 <= b2 of the preconditioner matrix, lexicographic).
 * Counts are always set; arrays (may be NULL) receive up to their capacity.  Static <2,3,9> matrices only.
 * Ties the reference leaves to hash-table iteration order are broken by ascending id.  On a sharded
 * matrix membership and cluster pairs are global (the visibility counts are summed over the ranks), block pairs are
 * the rank's own. */
int cx_visibility_structure(cx_matrix* A, int32_t preconditioner_type, int32_t clustering_type, int32_t* membership,
                            int32_t* num_clusters, int32_t* num_cluster_pairs, int32_t* cluster_pair_1,
                            int32_t* cluster_pair_2, int32_t cluster_pair_capacity, int64_t* num_block_pairs,
                            int32_t* block_pair_1, int32_t* block_pair_2, int64_t block_pair_capacity);
/* The clusters and cluster pairs of the same preconditioner computed from the flat block structure alone: no device,
 * no matrix object: the host part of the structure analysis.  Static <2,3,9> layout as the device path requires,
 * two cells per row block, e-block first, rows sorted by e-block. */
int cx_visibility_clusters_host(const cx_block_structure* bs, int32_t num_eliminate_blocks, int32_t preconditioner_type,
                                int32_t clustering_type, int32_t* membership, int32_t* num_clusters,
                                int32_t* num_cluster_pairs, int32_t* cluster_pair_1, int32_t* cluster_pair_2,
                                int32_t cluster_pair_capacity);
int cx_schur_back_substitute(cx_context* ctx, cx_matrix* A, const double* b, const double* D,
                             const double* z, double* x, int32_t memspace);
/* ImplicitSchurComplement: Init + RightMultiplyAndAccumulate
 * (implicit_schur_complement.cc:49-144); y = S x (overwrites y), and rhs of the
 * reduced system (UpdateRhs, :251-276) when rhs != NULL. */
int cx_implicit_schur_multiply(cx_context* ctx, cx_matrix* A, const double* D, const double* b,
                               const double* x, double* y, double* rhs, int32_t memspace);
/* DenseCholesky::FactorAndSolve (dense_cholesky.cc:139-151): lhs is n x n
 * row-major, only its upper triangle is read; solves lhs x = rhs. */
int cx_dense_cholesky_solve(cx_context* ctx, int32_t n, double* lhs, const double* rhs,
                            double* x, int32_t memspace, cx_summary* summary);

/* -------------------------------------------------------------- evaluator */

/* Bundle-adjustment Evaluator: ProgramEvaluator<BlockEvaluatePreparer,
 * BlockJacobianWriter> (program_evaluator.h:137-304) specialised to
 * SnavelyReprojectionError (examples/snavely_reprojection_error.h:53-104).
 * Parameter blocks: column block j < num_points is point j, column block
 * num_points + i is camera i (the order ApplyOrdering produces for
 * bundle_adjuster's user ordering, reorder_program.cc:216-254).  Observations
 * are given in INPUT order; the evaluator orders residual blocks as
 * LexicographicallyOrderResidualBlocks does (reorder_program.cc:256-338) and
 * lays out J as BlockJacobianWriter::BuildJacobianLayout
 * (block_jacobian_writer.cc:68-167).  row_of_observation (may be NULL) receives
 * for each input observation its row block in J. */
int cx_evaluator_create_bal(cx_context* ctx, int32_t num_cameras, int32_t num_points,
                            int64_t num_observations, const int32_t* camera_index,
                            const int32_t* point_index, const double* observations_xy,
                            cx_evaluator** out);
void cx_evaluator_destroy(cx_evaluator* e);
/* Evaluator::CreateJacobian (evaluator.h:105): matrix owned by the evaluator */
cx_matrix* cx_evaluator_jacobian(cx_evaluator* e);
int cx_evaluator_row_of_observation(const cx_evaluator* e, int64_t* out_host);
/* Evaluator::Evaluate (evaluator.h:120-150).  state = [points (3 each) |
 * cameras (9 each)] in column order; cost/residuals/gradient may be NULL;
 * evaluate_jacobian != 0 writes the evaluator's matrix. */
int cx_evaluator_evaluate(cx_evaluator* e, const double* state, double* cost,
                          double* residuals, double* gradient, int32_t evaluate_jacobian,
                          int32_t memspace);
/* Robust loss applied to every residual block: cost = 1/2 rho(|r|^2), residuals and Jacobian
 * corrected as Corrector does (residual_block.cc:160-196, corrector.cc:41-155,
 * loss_function.cc:46-144; bundle_adjuster --robustify uses HuberLoss(1.0), bundle_adjuster.cc:327).
 * CX_LOSS_NONE restores the plain squared loss (also what Evaluate does with
 * apply_loss_function = false, evaluator.h:84-92). */
int cx_evaluator_set_loss(cx_evaluator* e, int32_t loss_type, double a, double b);
/* Camera parameterisation (cx_camera_model; default CX_CAMERA_ANGLE_AXIS).  With
 * CX_CAMERA_QUATERNION_MANIFOLD the state of evaluate / minimize has 3P + 10C entries ([points | cameras as
 * w x y z, translation, focal, k1, k2]) while gradient, step and J stay in the 3P + 9C tangent space: the
 * kernel multiplies the ambient 2x10 camera Jacobian by the manifold's PlusJacobian (residual_block.cc:136-159). */
int cx_evaluator_set_camera_model(cx_evaluator* e, int32_t camera_model);
int64_t cx_evaluator_num_parameters(const cx_evaluator* e);            /* Evaluator::NumParameters (ambient) */
int64_t cx_evaluator_num_effective_parameters(const cx_evaluator* e);  /* Evaluator::NumEffectiveParameters (tangent) */
/* Evaluator::Plus (evaluator.h:152-158): x_plus_delta = x [+] delta; x, x_plus_delta ambient, delta tangent */
int cx_evaluator_plus(cx_evaluator* e, const double* x, const double* delta, double* x_plus_delta, int32_t memspace);
double cx_evaluator_last_kernel_ms(const cx_evaluator* e);
/* The library keeps a second, camera-major copy of the F cells (what the reference's transpose block structure
 * is for, block_sparse_matrix.cc:784-808).  By default a Jacobian evaluation writes it in the same kernel
 * (+148 B per residual block); a caller that always follows Evaluate with ScaleColumns -- TrustRegionMinimizer
 * with jacobi_scaling, trust_region_minimizer.cc:263-279 -- turns that off (on = 0) because cx_matrix_scale_columns
 * rewrites the copy anyway.  Either way the copy is rebuilt lazily if it is ever found stale. */
int cx_evaluator_set_emit_camera_major(cx_evaluator* e, int32_t on);
/* Column scaling inside the evaluation (round 3).  TrustRegionMinimizer computes its Jacobi scaling once, at iteration 0,
 * and calls jacobian->ScaleColumns(jacobian_scaling_) after EVERY evaluation (trust_region_minimizer.cc:263-279): a second
 * pass over all of J per iteration.  With a scale vector registered here ([num_effective_parameters], copied; NULL clears
 * it) every Jacobian evaluation writes J diag(scale) directly -- the same bits as Evaluate followed by ScaleColumns; the
 * gradient it returns is still J'r of the UNscaled problem, as the reference evaluates it before scaling.  A gradient-only
 * evaluation (no Jacobian) ignores the scale.  (Whether the kernel also writes the camera-major copy is
 * cx_evaluator_set_emit_camera_major's business; without it the copy is rebuilt by a gather pass at its first use.) */
int cx_evaluator_set_column_scale(cx_evaluator* e, const double* scale, int32_t memspace);
/* Device copy of the residuals the last cx_evaluator_evaluate wrote to HOST memory (NULL when it wrote none, or
 * wrote them to a caller's device buffer): the same vector can go into cx_solver_solve as b with
 * cx_per_solve_options.b_on_device = 1, without a second trip across PCIe.  Valid until the next evaluate that
 * produces residuals. */
const double* cx_evaluator_device_residuals(const cx_evaluator* e);
/* *match = 1 when the host array still holds what the device copy holds: 64 entries spread over the vector (first and last
 * included) are brought back and compared bit for bit -- a 512-byte copy instead of the 16 bytes per residual block that
 * re-uploading b costs.  What lets the host adapter alias b to the device copy BY DEFAULT: LinearSolver::Solve promises
 * nothing about who owns b, so the adapter does not take the pointer's word for it. */
int cx_evaluator_device_residuals_match(cx_evaluator* e, const double* host_residuals, int32_t* match);

/* -------------------------------------------------------------- minimizer */

/* Minimizer::Options for TRUST_REGION + LEVENBERG_MARQUARDT (include/ceres/solver.h:250-330,
 * 620-640; minimizer.h:60-190); defaults are Solver::Options' defaults. */
typedef struct {
  int32_t max_num_iterations;                 /* 50   */
  int32_t max_num_consecutive_invalid_steps;  /* 5    */
  int32_t jacobi_scaling;                     /* 1    */
  int32_t use_nonmonotonic_steps;             /* 0    */
  int32_t max_consecutive_nonmonotonic_steps; /* 5    */
  int32_t reserved;
  double initial_trust_region_radius;         /* 1e4  */
  double max_trust_region_radius;             /* 1e16 */
  double min_trust_region_radius;             /* 1e-32 */
  double min_relative_decrease;               /* 1e-3 */
  double min_lm_diagonal;                     /* 1e-6 */
  double max_lm_diagonal;                     /* 1e32 */
  double function_tolerance;                  /* 1e-6 */
  double gradient_tolerance;                  /* 1e-10 */
  double parameter_tolerance;                 /* 1e-8 */
  double eta;                                 /* 1e-1 */
} cx_minimizer_options;

/* IterationSummary (include/ceres/iteration_callback.h:45-160); times in ms */
typedef struct {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_nonmonotonic;
  int32_t step_is_successful;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double eta;
  int32_t linear_solver_iterations;
  int32_t reserved;
  double iteration_ms;      /* wall time of the iteration                          */
  double linear_solver_ms;  /* wall time inside LinearSolver::Solve                 */
  double jacobian_ms;       /* device time of the residual + Jacobian evaluation     */
  double residual_ms;       /* device time of the candidate-point cost evaluation    */
} cx_iteration_summary;

/* TerminationType (include/ceres/types.h:401-432) restricted to what the minimizer sets */
typedef enum { CX_CONVERGENCE = 0, CX_MIN_NO_CONVERGENCE = 1, CX_MIN_FAILURE = 2 } cx_minimizer_termination;

typedef struct {
  int32_t termination_type;  /* cx_minimizer_termination */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_iterations;    /* iteration summaries produced (iteration 0 included) */
  double initial_cost;
  double final_cost;         /* cost at the returned state (the minimum-cost iterate) */
  double total_ms;
  char message[256];
} cx_minimizer_summary;

void cx_minimizer_default_options(cx_minimizer_options* options);
/* TrustRegionMinimizer::Minimize (trust_region_minimizer.cc:68-840) with
 * LevenbergMarquardtStrategy (levenberg_marquardt_strategy.cc:50-175) and
 * TrustRegionStepEvaluator (trust_region_step_evaluator.cc:40-117), for the bundle-adjustment
 * evaluator: state, residuals, gradient, step, LM diagonal, Jacobi scaling and J stay in HBM for
 * the whole minimisation; per iteration only the handful of scalars the control flow needs
 * cross to the host.  state ([points | cameras], like cx_evaluator_evaluate) is updated in place
 * to the minimum-cost iterate.  iterations (may be NULL) receives up to capacity summaries.
 * On a sharded context state holds this rank's points and all cameras. */
int cx_minimize(cx_evaluator* e, cx_solver* s, const cx_minimizer_options* options, double* state,
                int32_t memspace, cx_minimizer_summary* summary, cx_iteration_summary* iterations,
                int32_t capacity);

/* ------------------------------------------------------ host-side helpers */

/* DetectStructure (detect_structure.cc:39-120); -1 stands for Eigen::Dynamic */
int cx_detect_structure(const cx_block_structure* bs, int32_t num_eliminate_blocks,
                        int32_t* row_block_size, int32_t* e_block_size, int32_t* f_block_size);
/* ComputeStableSchurOrdering (parameter_block_ordering.cc:50-83, graph_algorithms.h:165-227) for a
 * bundle-adjustment program whose parameter blocks are listed cameras 0..C-1 then points 0..P-1
 * (bundle_adjuster.cc:253-267): vertex ids camera i -> i, point j -> C + j.  ordering has C + P
 * entries: the independent set (eliminated first) followed by the rest. */
int cx_stable_schur_ordering(int32_t num_cameras, int32_t num_points, int64_t num_observations,
                             const int32_t* camera_index, const int32_t* point_index,
                             int32_t* ordering, int32_t* independent_set_size);
/* Host half of the plan of the tile-sparse Cholesky that stands in for SuiteSparse in SPARSE_SCHUR
 * (schur_complement_solver.cc:292-335, suitesparse.cc:218-279, 397-469) -- no device needed.  Input: the upper cells
 * (cell_row <= cell_col, every diagonal cell present) of a symmetric matrix of num_cameras 9x9 blocks, e.g. the
 * InitStorage cell list (cx_schur_sparse_structure).  Output: the padded nested-dissection layout (first scalar row of
 * every camera; pieces start at multiples of 64 rows), the number of 64-row tile rows, of levels of the tile elimination
 * tree, of tiles after symbolic fill and of tile-pair updates; per tile row its level and its list of column tiles
 * (tile_row_start[num_tile_rows + 1] into tile_cols; every list is closed by the pseudo tile `num_tile_rows` that
 * carries the right-hand side).  Arrays may be NULL; they are filled when their capacity suffices. */
int cx_sparse_cholesky_plan_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                 int32_t* camera_first_row, int32_t* num_tile_rows, int32_t* num_levels, int64_t* num_tiles,
                                 int64_t* num_tile_pair_updates, int32_t* tile_row_level, int32_t* tile_row_start,
                                 int32_t capacity_rows, int32_t* tile_cols, int64_t capacity_tiles);
/* Host half of the gather assembly of the explicit S (SchurEliminator::Eliminate into the InitStorage cell set,
 * schur_eliminator_impl.h:512-561, schur_complement_solver.cc:224-290) -- no device needed: the cells of the upper block
 * triangle of S in InitStorage order and, per cell, the row pairs (row of the cell's first camera, row of its second) that
 * update it, in chunk order.  Static <2,3,9> layout (two cells per row block, e-block first, rows sorted by e-block).
 * Counts are always set (-1 cells / items when the pair list would exceed 2^28 pairs and the scatter path is used
 * instead); the arrays (may be NULL) are filled when their capacity suffices: cell_row / cell_col [num_cells],
 * pair_rows [2 * num_pairs]. */
int cx_schur_pair_lists_host(const cx_block_structure* bs, int32_t num_eliminate_blocks, int64_t* num_cells,
                             int64_t* num_pairs, int64_t* num_items, int32_t* cell_row, int32_t* cell_col,
                             int64_t cell_capacity, int32_t* pair_rows, int64_t pair_capacity);
/* The schedule of the tile-pair updates of that factorisation under an update window (a target tile's
 * contributions of `window` consecutive levels are subtracted as one chain), checked from its definition -- no device: every
 * product of the symbolic factor is scheduled exactly once, after its source row is factored and before its target's row is
 * (for rank `rank` of a plan distributed over `nranks`: own rows' products before the exchange of the replicated tiles), source
 * rows ascending inside a chain.  Same input as cx_sparse_cholesky_plan_host.  Returns the number of tile rows (< 0: error);
 * *violations must come back 0. */
int cx_sparse_cholesky_schedule_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                     int32_t nranks, int32_t rank, int32_t window, int64_t* num_products, int64_t* num_chains,
                                     int32_t* longest_chain, int64_t* violations);
/* How the DISTRIBUTED factorisation of a sharded SPARSE_SCHUR (several ranks) divides this structure:
 * every rank factors the subtrees of the tile elimination tree it owns, the rows above them (the top separators of the
 * dissection) are factored by every rank after ONE exchange of their tiles.  Same input as cx_sparse_cholesky_plan_host.
 * Returns the number of tile rows (negative: error); updates_per_rank[nranks] = tile-pair updates of each rank's own rows,
 * *updates_replicated / *tiles_replicated = the replicated top, tile_row_owner[rows] (may be NULL) = owning rank or -1. */
int cx_sparse_cholesky_distribution_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                         int32_t nranks, int64_t* updates_per_rank, int64_t* updates_replicated,
                                         int64_t* tiles_replicated, int32_t* tile_row_owner, int32_t capacity_rows);
/* Partition the first num_eliminate_blocks column blocks (points) into nranks
 * contiguous ranges holding about equal numbers of non-zeros -- the balancing
 * PartitionRangeForParallelFor does on cumulative_nnz
 * (partition_range_for_parallel_for.h:60-150).  bounds has nranks+1 entries. */
int cx_partition_points(const cx_block_structure* bs, int32_t num_eliminate_blocks,
                        int32_t nranks, int32_t* bounds);

#ifdef __cplusplus
}
#endif
#endif /* CXSCHUR_H_ */
