// Internal declarations of libcxschur (not part of the C ABI).
#ifndef CX_INTERNAL_H_
#define CX_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <memory>
#include <vector>

#include "../../include/cxschur.h"

// ------------------------------------------------------------------ errors
void cx_set_error(const char* fmt, ...);

#define CX_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      cx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (_e == hipErrorOutOfMemory) ? CX_ERR_OUT_OF_MEMORY : CX_ERR_HIP;                \
    }                                                                                        \
  } while (0)

#define CX_CHECK_ARG(cond)                                                        \
  do {                                                                            \
    if (!(cond)) {                                                                \
      cx_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);    \
      return CX_ERR_INVALID_ARGUMENT;                                             \
    }                                                                             \
  } while (0)

#define CX_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != CX_OK) return _rc; \
  } while (0)

// RAII device array
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  int alloc(size_t count) {
    if (count <= n && p) return CX_OK;
    release();
    if (count == 0) count = 1;
    CX_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    n = count;
    return CX_OK;
  }
  int upload(const T* h, size_t count, hipStream_t s) {
    CX_TRY(alloc(count));
    if (count > 0) CX_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    CX_HIP(hipStreamSynchronize(s));
    return CX_OK;
  }
  int upload(const std::vector<T>& h, hipStream_t s) { return upload(h.data(), h.size(), s); }
};

// ---------------------------------------------------------------- context
struct cx_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[16] = {};  // 0-7 general purpose pairs, 8-15 the solver's phase stopwatch
  // RCCL (resolved at run time from librccl.so.1; see cx_context.cpp)
  void* comm = nullptr;
  int (*allreduce_cb)(double*, int64_t, void*) = nullptr;  // rehearsal transport (cx_context_set_comm_callback)
  void* allreduce_cb_user = nullptr;
  // in-process transport of a multi-shard front: recv[0, count) = sum over the ranks of send[rank * count ...) (NULL: a
  // callback transport without it sums the whole send buffer and keeps its own range)
  int (*reduce_scatter_cb)(double*, double*, int64_t, void*) = nullptr;
  int rank = 0;
  int nranks = 1;
  bool comm_broken = false;        // the communicator was aborted after a lost rank (cx_comm_abort): no sharded work any more
  double comm_timeout_s = 0.0;     // 0: CX_COMM_TIMEOUT_S or 120 s
  int64_t fail_countdown = -1;     // cx_debug_inject_failure: fail before the n-th collective from now (-1: off)
  DevBuf<double> agree;            // one word for cx_comm_agree
  double allreduce_host_ms = 0.0;
  // device-side timing of the exchange step: event pairs around the collectives since cx_allreduce_reset()
  static constexpr int kTimedCollectives = 256;
  std::vector<hipEvent_t> ar_events;   // 2 * kTimedCollectives, created on first use
  int ar_timed = 0;                    // pairs recorded
  int64_t ar_calls = 0, ar_bytes = 0;
  int num_cus = 256;
  char name[128] = {};
  DevBuf<double> chol_scratch;  // dense Cholesky work vectors (cx_cholesky.hip)
  // ---- front of several shards driven from ONE process (cx_context_create_multi, cx_multi.hip): shards[i] is an ordinary
  // context with rank i of shards.size(); the front itself stays a plain context on the first device
  std::vector<cx_context*> shards;
  struct cx_shard_group* group = nullptr;  // worker threads (one per shard) + the in-process exchange step
  // ---- host <-> device traffic of the calls that take host vectors (cx_transfer.cpp)
  struct cx_xfer_state* xfer = nullptr;    // staging pool, copy stream, counters; created on first use
};

// ------------------------------------------------- tile-sparse Cholesky (cx_sparse_chol.hip)
// Plan and storage of the level-scheduled tile-sparse Cholesky of a symmetric positive definite matrix made of 9x9
// camera blocks: the explicit Schur complement S (SPARSE_SCHUR) or a visibility based preconditioner.
struct cx_sp_plan {
  int state = 0;       // 0 not built, 1 ready, 2 not available (too much fill)
  int C = 0;           // 9x9 block rows
  int T = 0;           // tile rows (64 scalar rows each, padding included)
  int num_levels = 0;  // heights of the tile elimination tree
  int64_t num_tiles = 0;
  std::vector<int32_t> h_level_row_begin, h_level_panel_begin, h_level_tgt_begin;  // [levels + 1]
  DevBuf<int32_t> d_cam_pos, d_valid, d_row_start, d_row_tiles;  // first row of each camera; valid rows per tile row; tile lists
  DevBuf<int32_t> d_level_rows, d_panel_row, d_panel_pool;
  DevBuf<int32_t> d_tgt_pool, d_tgt_flags, d_src_begin, d_src_a, d_src_b;
  DevBuf<int32_t> d_col_start, d_col_pool;  // transposed index (forward solves): [T + 1] list ranges of the tile columns (tiles (K < I, I), ascending K); [tiles] every tile's slot in its column's list
  DevBuf<double> d_W, d_x;                  // tile pool (factored in place); vectors, block inverses, partial products
  // use_mixed_precision_solves: the pool is kept (and factored) in single precision, d_W32 instead of d_W; the caller sets f32
  // before cxsp_assemble / cxsp_factor_and_solve*, vectors and the 32 x 32 block inverses stay double
  bool f32 = false;
  bool replicate = false;  // sharded: keep the whole factor on every rank (refinement solves need cxsp_solve), see cxsp_build_plan_sharded
  bool built_replicate = false;                 // how the current plan was built (the option may differ from solve to solve)
  std::vector<int32_t> h_union_c1, h_union_c2;  // the union cell list, kept so that re-planning needs no second exchange
  DevBuf<float> d_W32;
  // sharded matrix (points over ranks): the plan is built from the UNION of the ranks' S cells; a rank scatters its own
  // cell values into the common cell-major array, which is summed over the ranks and assembled into the pool
  int64_t num_union_cells = 0;
  DevBuf<int32_t> d_union_c1, d_union_c2, d_local_to_union;
  DevBuf<double> d_union_values;
  // distributed factorisation (sharded SPARSE_SCHUR, cx_sparse_chol.hip): levels [0, level_split) hold this rank's own tile
  // rows, levels [level_split, num_levels) the rows every rank factors; -1: not distributed
  int level_split = -1;
  int64_t num_shared_tiles = 0;
  DevBuf<int32_t> d_shared_tiles;    // pool slots of the replicated rows' tiles (summed over the ranks at the split)
  DevBuf<int32_t> d_row_keep;        // [T] 1: this rank contributes the row's part of the solution to the final sum
  DevBuf<double> d_exchange;         // packed shared tiles + one slot for the not-positive-definite flag
  // cell values by owner (round 4): [nranks][rs_chunk_cells] union cell ids (-1 padding), rank r's range = the cells its
  // tile rows (rank 0: and the replicated rows) are assembled from -- the ranges of the reduce-scatter of a solve
  int64_t rs_chunk_cells = 0;
  DevBuf<int32_t> d_rs_cells;
  DevBuf<int32_t> d_cell_mine;       // [union cells] 1: this rank assembles the cell
  // stored-factor sweeps on the distributed factor: per replicated row with incoming products from owned rows, their slots
  int32_t xs_count = 0;
  DevBuf<int32_t> d_xs_begin, d_xs_slots;
  DevBuf<double> d_xs_t;
  DevBuf<double> d_rs_send, d_rs_recv;
  std::vector<int64_t> h_work_per_rank;
  int64_t h_work_shared = 0;
};

// -------------------------------------------------- embedding into the static layout (cx_embed.hip)
// A matrix whose structure is <2, e <= 3, f <= 9> (+ trailing rows with one f cell) keeps, beside its own values, an
// inner matrix in the <2,3,9> layout that the static kernels run on.
struct cx_matrix;
struct cx_embed {
  cx_matrix* inner = nullptr;
  int e = 3, f = 9;
  int32_t P = 0, C = 0;   // e-blocks and f-blocks of the caller's matrix
  int32_t P_in = 0;       // points of the inner matrix: P + one dummy point per two-row slice of a trailing row
  int64_t O_main = 0;     // e-rows
  int64_t O_in = 0;       // row blocks of the inner matrix
  bool dirty = true;      // the caller's values changed since the inner values were written
  DevBuf<int32_t> d_epos, d_fpos, d_valid;  // [O_in] where the inner row's E and F cell come from (E: -1 = zero), rows it has
  DevBuf<int32_t> d_tail_src;               // [2 (O_in - O_main)] caller's scalar row of every trailing inner row, -1 = padding
  DevBuf<double> d_cols, d_rows, d_b, d_D, d_x;  // vectors in the inner spaces
};

// ----------------------------------------------------------------- matrix
// Tile of the point-major ("chunk") kernels: whole chunks, at most kTileRows rows,
// unless a single chunk is larger than that (then the tile is that one chunk).
constexpr int kTileRows = 256;
// Segment of the camera-major kernels: a run of rows of ONE camera.
constexpr int kSegRows = 512;

struct cx_matrix {
  cx_context* ctx = nullptr;
  // host copy of the structure (flat)
  int32_t R = 0, Cb = 0;  // row blocks / column blocks
  int32_t nelim = 0;
  std::vector<cx_block> rows, cols;
  std::vector<int32_t> rcb;
  std::vector<cx_cell> cells;
  int64_t num_rows = 0, num_cols = 0, nnz = 0;
  int64_t num_cols_e = 0, num_cols_f = 0;
  int32_t num_row_blocks_e = 0;  // rows whose first cell is an e block (partitioned_matrix_view_impl.h:60-90)
  int32_t row_size = 0, e_size = 0, f_size = 0;  // DetectStructure result (-1 dynamic)

  // generic device structure
  DevBuf<cx_block> d_rows, d_cols;
  DevBuf<int32_t> d_rcb;
  DevBuf<cx_cell> d_cells;
  DevBuf<int32_t> d_chunk_start;  // generic eliminator chunks (row ranges), [num_chunks+1]
  int32_t num_chunks = 0;
  std::vector<int64_t> blk_off;   // packed size^2 offsets of the column blocks, [Cb+1]
  DevBuf<int64_t> d_blk_off;
  bool generic_ready = false;
  // gather form of the dynamic-size eliminator (cx_generic.hip: PrepareGenericGather): the cells (b1 <= b2) of S that any
  // row or chunk touches, and per cell the products that update it, in a fixed order -- the role of the static path's
  // pair lists.  0 not built, 1 ready, 2 too many products (the scatter kernels with atomics are used)
  int gather_state = 0;
  int32_t g_num_targets = 0, g_num_fcells = 0;
  DevBuf<int32_t> d_g_target;      // [targets][4] first scalar row / column of the cell in S, its sizes s1, s2
  DevBuf<int32_t> d_g_tuple_begin; // [targets + 1]
  DevBuf<int32_t> d_g_tuples;      // [tuples][4] X position, Y position, inner dimension, kind (0: +F_i'F_j from the values, 1: -B_i'G_j from d_g_bg)
  DevBuf<int32_t> d_g_fcell;       // [fcells][4] row block, cell index, offset of B in d_g_bg, offset of G
  DevBuf<double> d_g_bg;           // per f cell of an e-row: B = E'F and G = (E'E + D^2)^-1 B
  // transposed index of the dynamic-size structure (the reference's transpose block structure, block_sparse_matrix.cc:784-808):
  // the cells of every column block in ascending row order -- what lets A'x be a gather with a fixed summation order instead
  // of atomics (cx_matrix.hip: cxk_build_transpose)
  // per entry: position of the cell's values, first scalar row of its row block, row block size | (1 << 8 for the e cell of
  // an e-row); the column's entries are cut into segments of at most kTransposeSegment, one wavefront each
  DevBuf<int32_t> d_t_pos, d_t_rp, d_t_meta;     // [cells]
  DevBuf<int32_t> d_t_seg_begin, d_t_seg_col;    // [segments + 1], [segments]
  DevBuf<int32_t> d_t_col_seg;                   // [Cb + 1] segments of every column block
  DevBuf<double> d_t_partial;                    // [segments][16] partial sums of the columns that have several segments
  DevBuf<double> d_t_partial_blocks;             // [segments][256] the same for the block diagonal
  int32_t num_t_segments = 0;
  int32_t t_max_col_size = 0;                    // widest column block (wider than 64: A'x takes the scatter kernel)
  int32_t t_stride = 16;                         // doubles per segment in d_t_partial
  bool transpose_ready = false;

  // values
  DevBuf<double> d_values;

  // ---- static <2,3,9> bundle-adjustment layout
  bool is239 = false;
  int64_t O = 0;  // row blocks
  int32_t P = 0, C = 0;
  DevBuf<int32_t> d_row_pt, d_row_cam;   // [O]
  DevBuf<int32_t> d_pt_start;            // [P+1] first row of each point's chunk
  DevBuf<int32_t> d_tile_row, d_tile_pt; // [T+1] chunk-aligned tiles
  int32_t num_tiles = 0;
  bool has_big_tiles = false;            // some chunk is longer than kTileRows
  DevBuf<int32_t> d_cam_rows;            // [O] rows in camera-major order
  DevBuf<int32_t> d_cam_pos;             // [O] inverse: position of row r in camera-major order (d_cam_rows[d_cam_pos[r]] == r)
  DevBuf<int32_t> d_seg_begin, d_seg_cam;  // [S+1], [S] camera-major segments (positions in d_cam_rows)
  DevBuf<int32_t> d_cam_seg_start;       // [C+1] segments of each camera
  int32_t num_segs = 0;
  DevBuf<double> d_Ft;                   // camera-major copy of the F cells [O][18]
  bool ft_valid = false;
  // d_partials holds the per-segment partial sums of the F'F diagonal blocks of the CURRENT values (k_cam_init<false> has just
  // run and nothing has used the buffer since): the explicit assembly then only adds them up (set and cleared by the caller)
  bool ftf_partials_current = false;
  DevBuf<float> d_vals32, d_Ft32;        // fp32 copies for mixed-precision CGNR (cx_matrix_ensure_f32)
  bool f32_valid = false;
  bool use_f32 = false;                  // products read the fp32 copies (fp64 accumulation)
  DevBuf<double> d_partials;             // camera-major partial sums [S][81]
  DevBuf<double> d_partials9;            // [S][9] (fused set-up)
  DevBuf<double> d_elim_blk, d_elim_ete, d_elim_diag, d_elim_rows;  // explicit-S scratch (cx_schur.hip)
  // explicit S without atomics (cx_schur.hip): the non-zero cells of the upper block triangle of S in
  // the order SparseSchurComplementSolver::InitStorage lists them (every (c,c), then co-visible c1 < c2,
  // lexicographic -- schur_complement_solver.cc:224-290) and per cell the co-observing row pairs in chunk order
  std::vector<int32_t> h_cell_c1, h_cell_c2;  // host copy of the cell list (parity tests, sparse consumers)
  std::vector<int32_t> h_cell_item_start;     // host copy of d_cell_item_start
  DevBuf<int32_t> d_cell_c1, d_cell_c2;  // [num_cells]
  DevBuf<int32_t> d_cell_row_start;      // [C + 1] cells of block row c1 (contiguous, sorted by c2)
  DevBuf<int32_t> d_col_cell_start;      // [C + 1] off-diagonal cells of block column c2 ...
  DevBuf<int32_t> d_col_cells;           // ... as cell ids sorted by c1
  DevBuf<int32_t> d_pair_rows;           // [2 * num_pairs] (row of camera c1, row of camera c2)
  DevBuf<int64_t> d_item_begin;          // [num_items + 1] work items: runs of <= kPairItem pairs of one cell
  DevBuf<int32_t> d_cell_item_start;     // [num_cells + 1] items of each cell
  DevBuf<int32_t> d_item_order;          // [num_items] launch order of the items (a permutation, cx_schur.hip)
  DevBuf<double> d_item_partial;         // [num_items][81]
  DevBuf<double> d_S;                    // [num_cells][81] sparse S values (without D_f^2), ExplicitSchur
  int64_t num_items = 0;
  int64_t num_cells = 0;
  DevBuf<double> d_elim_bg0, d_elim_bg1, d_elim_bg2;  // per row B = E'F (3x9) and G = (E'E)^-1 B, 3 x 18 doubles
  DevBuf<float> d_elim_h32;                           // [O][32] the one-table operand H = K'B in single precision (mixed solves)
  // tile-sparse Cholesky of S (cx_sparse_chol.hip)
  cx_sp_plan sp;
  int pairs_state = 0;                   // 0 not built, 1 ready, 2 too many pairs (atomic path)
  // visibility based preconditioners of this structure (cx_visibility.h), one plan per (preconditioner type,
  // clustering type), built on first use
  std::vector<std::shared_ptr<struct cx_vis_plan>> vis_plans;
  int64_t num_pairs = 0;

  // scratch for host-pointer calls
  DevBuf<double> d_x, d_y;
  float last_ms = 0.f;
  // when set, the product kernels return at once if *stop != 0 (CG termination flag)
  const int* stop = nullptr;
  // the visibility based preconditioners count points and rows up to these (-1: all): an inner matrix of an embedding
  // carries dummy points for its trailing rows, which see no point in the reference (visibility.cc:50-85)
  int32_t P_vis = -1;
  int64_t O_vis = -1;
  cx_embed* embed = nullptr;  // dynamic-size matrix with a static <2,3,9> image (cx_embed.hip)

  // ---- front of a matrix on a multi-shard context (cx_multi.hip): the e-blocks (points) are cut into contiguous ranges,
  // parts[i] lives on shard i and holds the rows of its range with columns [its e-blocks | all f-blocks]; the front keeps
  // no device data of its own
  std::vector<cx_matrix*> parts;
  bool parts_owned = true;                  // false: the parts belong to the shards' evaluators
  std::vector<int32_t> part_e0;             // [n + 1] first e-block of each shard
  std::vector<int32_t> part_rowblk0;        // [n + 1] first row block
  std::vector<int64_t> part_row0;           // [n + 1] first scalar row
  std::vector<int64_t> part_ecol0;          // [n + 1] first scalar column of the e part
  struct ValueRun { int64_t global, local, len; };
  std::vector<std::vector<ValueRun>> part_runs;  // where the values of part i sit in the front's (reference layout) value array
};

// ------------------------------------------------ host vectors at the boundary (cx_transfer.cpp)
// Internal memspace value: the pointer is a `const cx_host_slices*` -- a vector handed over as two runs of host memory
// [head | tail].  The fronts of cx_multi.hip describe a shard's part of the caller's vector that way (the shard's own
// e-block range followed by the f part every shard sees), so the copies go straight between the caller's array and the
// shard's device buffer, with no gathered std::vector in between.
constexpr int32_t CX_HOST_SLICES = 0x51;
struct cx_host_slices {
  double* head = nullptr;
  int64_t nhead = 0;
  double* tail = nullptr;
  int64_t ntail = 0;
  bool skip_tail_out = false;  // outputs: do not copy the tail back (another shard delivers the replica)
  const double* as_arg() const { return reinterpret_cast<const double*>(this); }
  double* as_arg() { return reinterpret_cast<double*>(this); }
};
inline bool cx_is_host_space(int32_t memspace) { return memspace == CX_HOST || memspace == CX_HOST_SLICES; }

// Copies between caller (host) memory and device memory, enqueued on `st` (NULL: the context stream).  Arrays a caller
// hands in repeatedly are registered with the HIP runtime (hipHostRegister) by the process-wide registry of
// cx_transfer.cpp, after which the copy is a DMA at PCIe rate instead of a staged pageable copy; bytes, copies and
// device-side durations are counted per context (cx_transfer_stats_get).  An H2D copy from registered memory is
// asynchronous for real: cx_xfer_wait_h2d must run before the call returns to the owner of the source array.
int cx_copy_h2d(cx_context* ctx, void* dst, const void* src, size_t bytes, hipStream_t st = nullptr);
int cx_copy_d2h(cx_context* ctx, void* dst, const void* src, size_t bytes, hipStream_t st = nullptr);
int cx_xfer_wait_h2d(cx_context* ctx);  // blocks until every H2D copy enqueued through cx_copy_h2d has read its source
// a second stream for copies that overlap work on the context stream (created on first use)
hipStream_t cx_copy_stream(cx_context* ctx);
void cx_xfer_destroy(cx_context* ctx);
// dst[0, count) <- the vector `u` of the given memspace (host, host slices or device), enqueued on the context stream;
// the reverse for cx_vector_out (host slices: the tail only unless skip_tail_out).  Host copies are counted; the caller waits
// (cx_xfer_wait_h2d / a stream synchronisation) before it returns
int cx_vector_in(cx_context* ctx, double* dst, const double* u, size_t count, int32_t memspace);
int cx_vector_out(cx_context* ctx, double* u, const double* src, size_t count, int32_t memspace);
// staging buffers: grow-only pool per context, handed out for the duration of one call
DevBuf<double>* cx_stage_acquire(cx_context* ctx, size_t count);
void cx_stage_release(cx_context* ctx, DevBuf<double>* b);
// registry (process-wide): try to make [p, p + bytes) registered memory; true when it is
bool cx_pin_range(const void* p, size_t bytes);

// upload/download helpers for the (memspace) convention
struct HostOrDevice {
  // wraps a user pointer: if host, stages through a device buffer of the context's pool
  cx_context* ctx;
  DevBuf<double>* tmp = nullptr;
  double* dptr = nullptr;
  double* user = nullptr;
  cx_host_slices slices;
  bool sliced = false;
  size_t n = 0;
  bool is_host = false;
  explicit HostOrDevice(cx_context* c) : ctx(c) {}
  HostOrDevice(const HostOrDevice&) = delete;
  HostOrDevice& operator=(const HostOrDevice&) = delete;
  ~HostOrDevice();
  int in(const double* u, size_t count, int memspace);   // read-only input (u may be NULL -> dptr NULL)
  int inout(double* u, size_t count, int memspace, bool copy_in);
  int out();  // copy back if host (and wait for it)
  int out_async(hipStream_t st);  // the same without the wait: the caller synchronises `st` before it returns
 private:
  int bind(double* u, size_t count, int memspace);
  int copy_in();
};

// ------------------------------------------------ kernels (cx_matrix.hip etc.)
int cx_matrix_ensure_ft(cx_matrix* A);
int cxk_build_transpose(cx_matrix* A);  // dynamic-size matrices: the transposed index above
// y += A_sel' x over the transposed index (sel 0 all cells, 1 the e cell of every e-row, 2 the other cells; col_off is
// subtracted from column positions): one wavefront per column block, fixed summation order, no atomics
int cxk_generic_left_multiply(cx_matrix* A, int sel, int col_off, const double* x, double* y);
// blocks[blk_off[c] - off0 ..] = sum over the selected cells of column block c of cell' cell, for c in [first, first + count):
// the block diagonal of A_sel' A_sel over the same transposed index (overwrites; fixed order, no atomics)
int cxk_generic_block_diagonal(cx_matrix* A, int sel, int first, int count, int64_t off0, double* blocks);
int cx_matrix_ensure_f32(cx_matrix* A);

// all of these enqueue on ctx->stream and work on device pointers
int cxk_right_multiply(cx_matrix* A, const double* x, double* y, bool accumulate = true);
int cxk_left_multiply(cx_matrix* A, const double* x, double* y, bool accumulate = true, const double* d = nullptr,
                      const double* dx = nullptr, bool* folded = nullptr);
int cxk_squared_column_norm(cx_matrix* A, double* x);
int cxk_scale_columns(cx_matrix* A, const double* scale);

// y_f (+)= F' t  for the static path; t is row-sized
int cxk_ft_multiply(cx_matrix* A, const double* t, double* y_f, bool accumulate, const double* d_f = nullptr,
                    const double* x_f = nullptr);
// first half of it: the per-segment partial sums only (A->d_partials, 9 per segment)
int cxk_ft_partials(cx_matrix* A, const double* t);

int cx_allreduce_device(cx_context* ctx, double* p, int64_t n);
// recv[0, count) = sum over the ranks q of send_q[rank * count, (rank + 1) * count): every rank contributes nranks * count
// doubles and receives the sum of its own range only (ncclReduceScatter; half the traffic of an all-reduce of the same
// buffer, and counted so in the exchange statistics).  send may be overwritten.
int cx_reduce_scatter_device(cx_context* ctx, double* send, double* recv, int64_t count);
// waits that cannot hang on a lost rank (cx_context.cpp): plain waits on a context without an RCCL communicator of
// several ranks, polling with a deadline + ncclCommAbort otherwise.  st == NULL: the context stream
int cx_stream_sync(cx_context* ctx, hipStream_t st);
int cx_event_sync(cx_context* ctx, hipEvent_t ev);
int cx_comm_abort(cx_context* ctx);
double cx_comm_timeout(const cx_context* ctx);  // seconds a wait on a sharded context may see no progress
// host <- device for the scalars the control flow reads (pageable host memory): a pageable copy blocks inside the runtime
// until the stream has drained, so the bounded wait comes first, then the copy, then the wait for the copy
int cx_read_back(cx_context* ctx, void* host, const void* dev, size_t bytes, hipStream_t st = nullptr);
// all ranks learn whether any of them has failed so far in this call (local_rc != CX_OK); returns local_rc when that is
// an error, CX_ERR_COMM when only another rank failed, CX_OK when all are healthy.  One rank: returns local_rc
int cx_comm_agree(cx_context* ctx, int local_rc);
// statistics of the collectives since the last reset; collect waits for the recorded events (call after the stream
// has been synchronised) and returns the device time, extrapolated when more collectives ran than were timed
void cx_allreduce_reset(cx_context* ctx);
int cx_allreduce_collect(cx_context* ctx, double* device_ms, double* host_ms, double* calls, double* bytes);

// ---------------------------------------------------------------- evaluator (cx_eval.hip)
struct cx_evaluator {
  cx_context* ctx = nullptr;
  cx_matrix* J = nullptr;
  int32_t C = 0, P = 0;
  int64_t O = 0;
  std::vector<int64_t> row_of_obs;   // input observation -> row block
  DevBuf<double> d_obs;              // [2O] in row order
  DevBuf<double> d_partial, d_state, d_res;
  DevBuf<double> d_scratch_values, d_scratch_Ft;  // J of a gradient-only evaluation (the matrix keeps its values)
  DevBuf<double> d_sample;           // 64 sampled entries (cx_evaluator_device_residuals_match)
  DevBuf<double> d_col_scale;        // [3P + 9C] column scales applied by Jacobian evaluations (cx_evaluator_set_column_scale)
  bool has_col_scale = false;
  bool emit_ft = true;               // Jacobian evaluations also write the camera-major copy of F (cx_matrix::d_Ft)
  bool res_valid = false;            // d_res holds the residuals last handed out in host memory
  float last_ms = 0.f;
  int32_t loss_type = CX_LOSS_NONE;
  double loss_a = 0.0, loss_b = 0.0;
  int32_t camera_model = CX_CAMERA_ANGLE_AXIS;
  std::vector<cx_evaluator*> parts;  // front on a multi-shard context (cx_multi.hip): the evaluator of each shard's points
};
// out = Plus(x, sign * delta) on device pointers (Evaluator::Plus), enqueued on the context stream
int cxe_plus(cx_evaluator* e, const double* x, const double* delta, double sign, double* out);


// ------------------------------------------- embedding (cx_embed.hip)
int cxe_try_embed(cx_matrix* A);
void cxe_destroy(cx_matrix* A);
int cxe_sync(cx_matrix* A);
int cxe_matrix_op(cx_matrix* A, int op, const double* x, double* y);  // 0: y += A x, 1: y += A'x, 2: y = diag(A'A)
int cxe_solve(cx_matrix* A, const double* b, const double* D, double* x,
              const std::function<int(cx_matrix*, const double*, const double*, double*)>& solve);

// ------------------------------------------- multi-shard fronts (cx_multi.hip)
// The public entry points hand a front object (ctx->shards / A->parts / S->parts / e->parts non-empty) to these.
struct cx_solver;
bool cxm_is_front(const cx_context* ctx);
void cxm_context_destroy_shards(cx_context* front);
int cxm_matrix_create(cx_context* ctx, const cx_block_structure* bs, int32_t nelim, cx_matrix** out);
void cxm_matrix_destroy(cx_matrix* A);
int cxm_matrix_set_values(cx_matrix* A, const double* src, int32_t memspace);
int cxm_matrix_get_values(const cx_matrix* A, double* dst);
int cxm_matrix_set_zero(cx_matrix* A);
int cxm_matrix_values_changed(cx_matrix* A);
// op 0: y += A x, 1: y += A'x, 2: y = diag(A'A), 3: A <- A diag(x)
int cxm_matrix_op(cx_matrix* A, int op, const double* x, double* y, int32_t memspace);
int cxm_solver_solve(cx_solver* S, cx_matrix* A, const double* b, const cx_per_solve_options* ps, double* x, cx_summary* summary);
void cxm_solver_destroy(cx_solver* S);
int cxm_evaluator_create_bal(cx_context* ctx, int32_t C, int32_t P, int64_t O, const int32_t* cam, const int32_t* pt,
                             const double* obs, cx_evaluator** out);
void cxm_evaluator_destroy(cx_evaluator* e);
int cxm_evaluator_evaluate(cx_evaluator* e, const double* state, double* cost, double* residuals, double* gradient,
                           int32_t evaluate_jacobian, int32_t memspace);
int cxm_evaluator_plus(cx_evaluator* e, const double* x, const double* delta, double* x_plus_delta, int32_t memspace);
int cxm_evaluator_set_column_scale(cx_evaluator* e, const double* scale, int32_t memspace);
int cxm_evaluator_forward_settings(cx_evaluator* e);  // loss, camera model, emit_ft of the front -> its parts
const double* cxm_evaluator_device_residuals(const cx_evaluator* e);
double cxm_evaluator_last_kernel_ms(const cx_evaluator* e);
int cxm_minimize(cx_evaluator* e, cx_solver* s, const cx_minimizer_options* options, double* state, int32_t memspace,
                 cx_minimizer_summary* summary, cx_iteration_summary* iterations, int32_t capacity);

#endif
