// Several shards behind ONE set of handles, in ONE process: cx_context_create_multi and the fronts of
// cx_matrix / cx_solver / cx_evaluator / cx_minimize on such a context.
//
// Why: a Solver::Solve caller is a single process holding a single ContextImpl (context_impl.h:74-83) and calls
// Evaluator::Evaluate / LinearSolver::Solve (linear_solver.h:363-390) once per LM iteration with whole vectors.  The
// sharded solvers of this library are written per rank (points over ranks, camera-space sums through the context's
// exchange step); this file puts N such ranks behind the boundary: the front cuts the points
// into contiguous ranges of about equal non-zeros (cx_partition_points), gives every range to a shard context with
// its own device, stream and worker thread, scatters the caller's vectors, runs the per-rank code of the shards side by
// side and gathers the result.  Nothing numerical lives here.
//
// Exchange step between the shards: RCCL (one communicator per device, ncclAllReduce on each shard's stream) when the
// shards sit on distinct devices -- the production configuration, one shard per GPU of a node; an in-process sum
// (thread barrier + one kernel adding the shards' buffers in rank order) when several logical shards share a device,
// which is how the one-GPU test boxes exercise the whole path.
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <set>
#include <string>
#include <thread>

#include "cx_shard_group.h"
#include "cx_solver_internal.h"

namespace {

constexpr int kMaxShards = 16;

struct ShardPtrs { double* p[kMaxShards]; };

// every shard's buffer <- sum of all of them, added in rank order (the same bits on every shard)
__global__ void k_sum_shards(ShardPtrs ptrs, int n, int64_t len) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= len) return;
  double s = ptrs.p[0][i];
  for (int r = 1; r < n; ++r) s += ptrs.p[r][i];
  for (int r = 0; r < n; ++r) ptrs.p[r][i] = s;
}

// recv_r[k] = sum over the shards q of send_q[r * count + k], added in rank order
__global__ void k_reduce_scatter_shards(ShardPtrs send, ShardPtrs recv, int n, int64_t count) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(n) * count) return;
  const int r = int(i / count);
  double s = send.p[0][i];
  for (int q = 1; q < n; ++q) s += send.p[q][i];
  recv.p[r][i - int64_t(r) * count] = s;
}

}  // namespace

// cx_context::group points at this: the plain threading (cx_shard_group.h, sanitizer-tested on the CPU) plus what the
// device side of a front needs
struct cx_front_group : cx_shard_group {
  std::vector<int> devices;
  bool inproc = false;           // logical shards on one device: the exchange step is the in-process sum below
  hipStream_t stream0 = nullptr;
  struct CbUser { cx_front_group* g; int rank; } users[kMaxShards];
  cx_context* front = nullptr;
  bool broken = false;           // RCCL transport: a shard failed and the communicators were aborted
};
static_assert(cx_shard_group::kOk == CX_OK && cx_shard_group::kCommError == CX_ERR_COMM, "codes of cx_shard_group.h");

static cx_front_group* Group(const cx_context* ctx) { return static_cast<cx_front_group*>(ctx->group); }

// run fn(shard) on every worker of the front's group; the first failing shard's error becomes this thread's last error
static int RunShards(const cx_context* ctx, const std::function<int(int)>& fn) {
  cx_front_group* g = Group(ctx);
  if (g->broken) {
    cx_set_error("this multi-shard context lost a shard in an earlier call and its communicators were aborted: create a new context");
    return CX_ERR_COMM;
  }
  int failed = -1;
  std::string message;
  const int rc = g->run(fn, &failed, &message);
  if (rc != CX_OK) cx_set_error("shard %d: %s", failed, message.c_str());
  return rc;
}

namespace {

// the exchange step of a logical shard (cx_context::allreduce_cb): called with the shard's stream drained
int InprocAllReduce(double* p, int64_t len, void* user) {
  auto* u = static_cast<cx_front_group::CbUser*>(user);
  cx_front_group* g = u->g;
  return g->exchange(u->rank, p, len, [g](double* const* bufs, int n, int64_t count) -> int {
    ShardPtrs ptrs;
    for (int r = 0; r < n; ++r) ptrs.p[r] = bufs[r];
    hipLaunchKernelGGL(k_sum_shards, dim3(unsigned((count + 255) / 256)), dim3(256), 0, g->stream0, ptrs, n, count);
    return (hipGetLastError() != hipSuccess || hipStreamSynchronize(g->stream0) != hipSuccess) ? -1 : 0;
  });
}

int InprocReduceScatter(double* send, double* recv, int64_t count, void* user) {
  auto* u = static_cast<cx_front_group::CbUser*>(user);
  cx_front_group* g = u->g;
  return g->exchange2(u->rank, send, recv, count, [g](double* const* sends, double* const* recvs, int n, int64_t cnt) -> int {
    ShardPtrs a, b;
    for (int r = 0; r < n; ++r) { a.p[r] = sends[r]; b.p[r] = recvs[r]; }
    const int64_t total = int64_t(n) * cnt;
    hipLaunchKernelGGL(k_reduce_scatter_shards, dim3(unsigned((total + 255) / 256)), dim3(256), 0, g->stream0, a, b, n, cnt);
    return (hipGetLastError() != hipSuccess || hipStreamSynchronize(g->stream0) != hipSuccess) ? -1 : 0;
  });
}

std::mutex g_front_evaluators_mutex;
std::set<const cx_evaluator*> g_front_evaluators;  // live evaluator fronts (their address doubles as the residual token)

int NumShards(const cx_context* ctx) { return int(ctx->shards.size()); }

// The part of a column-space host vector `v` that shard i works on: [v[e range of shard i] | tail], tail = the f
// (camera) part every shard sees, tail_len entries starting at tail0.  Nothing is gathered: the descriptor goes to the
// shard's entry point as a CX_HOST_SLICES vector and the copies run between the caller's array and the shard's device
// buffer directly.  For an output, only shard 0 delivers the tail (the shards hold replicas of it).
cx_host_slices ColSlices(const cx_matrix* A, int i, const double* v, int64_t tail0, int64_t tail_len, bool output = false) {
  const int64_t e0 = A->part_ecol0[size_t(i)], e1 = A->part_ecol0[size_t(i) + 1];
  cx_host_slices s;
  s.head = const_cast<double*>(v) + e0;
  s.nhead = e1 - e0;
  s.tail = const_cast<double*>(v) + tail0;
  s.ntail = tail_len;
  s.skip_tail_out = output && i != 0;
  return s;
}
// a shard's rows of a row-space host vector: one contiguous run
cx_host_slices RowSlices(const cx_matrix* A, int i, const double* v) {
  cx_host_slices s;
  s.head = const_cast<double*>(v) + A->part_row0[size_t(i)];
  s.nhead = A->part_row0[size_t(i) + 1] - A->part_row0[size_t(i)];
  return s;
}

int RequireHost(int32_t memspace, const char* what) {
  if (memspace == CX_HOST) return CX_OK;
  cx_set_error("%s on a multi-shard context takes host vectors (a device pointer belongs to one shard only)", what);
  return CX_ERR_UNSUPPORTED;
}

}  // namespace

bool cxm_is_front(const cx_context* ctx) { return ctx != nullptr && !ctx->shards.empty(); }

// ------------------------------------------------------------------ context
extern "C" int cx_context_create_multi(int num_shards, const int* device_ids, cx_context** out) {
  CX_CHECK_ARG(out != nullptr && device_ids != nullptr && num_shards >= 1 && num_shards <= kMaxShards);
  bool distinct = true, same = true;
  for (int i = 0; i < num_shards; ++i)
    for (int j = 0; j < i; ++j) {
      if (device_ids[i] == device_ids[j]) distinct = false;
      else same = false;
    }
  if (!distinct && !same) {
    cx_set_error("device list must name distinct devices (RCCL between them) or one device throughout (logical shards)");
    return CX_ERR_INVALID_ARGUMENT;
  }
  cx_context* front = nullptr;
  CX_TRY(cx_context_create(device_ids[0], &front));
  auto* g = new cx_front_group;
  g->devices.assign(device_ids, device_ids + num_shards);
  g->inproc = num_shards > 1 && same;
  g->front = front;
  g->on_thread_start = [g](int i) { (void)hipSetDevice(g->devices[size_t(i)]); };
  g->last_error = [] { return cx_last_error(); };
  // RCCL between the shards: a shard that fails leaves the others inside (or on their way into) a collective that can
  // never complete.  The failing worker aborts every shard's communicator -- ncclCommAbort is made for exactly this: it
  // ends the collectives the other shards' streams are stuck in -- so their jobs come back with an error instead of
  // never; the front is unusable from then on (RunShards says so).  In-process transport: abort_exchange() has already
  // released the rendezvous and the next call starts clean.
  g->on_failure = [g](int) {
    if (g->inproc || g->n <= 1) return;
    g->broken = true;
    for (cx_context* shard : g->front->shards)
      if (shard) cx_comm_abort(shard);
  };
  front->group = g;
  front->shards.assign(size_t(num_shards), nullptr);
  g->start(num_shards);
  int rc = g->run([&](int i) { return cx_context_create(device_ids[i], &front->shards[size_t(i)]); });
  if (rc == CX_OK && num_shards > 1) {
    if (g->inproc) {
      g->stream0 = front->shards[0]->stream;
      for (int i = 0; i < num_shards; ++i) {
        g->users[i] = {g, i};
        rc = cx_context_set_comm_callback(front->shards[size_t(i)], i, num_shards, InprocAllReduce, &g->users[i]);
        if (rc != CX_OK) break;
        front->shards[size_t(i)]->reduce_scatter_cb = InprocReduceScatter;
      }
    } else {
      char id[128];
      rc = cx_comm_unique_id(id);
      // ncclCommInitRank blocks until every rank has joined: all shards call it at once, each from its own thread
      if (rc == CX_OK) rc = g->run([&](int i) { return cx_context_set_comm(front->shards[size_t(i)], i, num_shards, id); });
    }
  }
  if (rc != CX_OK) {
    std::string keep = cx_last_error();
    cx_context_destroy(front);
    cx_set_error("%s", keep.c_str());
    return rc;
  }
  *out = front;
  return CX_OK;
}

extern "C" int cx_context_num_shards(const cx_context* ctx) { return ctx ? std::max(1, NumShards(ctx)) : 0; }

// how a front matrix is cut: first e-block and first row block of every shard (num_shards + 1 entries each)
extern "C" int cx_matrix_shard_layout(const cx_matrix* A, int32_t* e_block_bounds, int32_t* row_block_bounds, int32_t capacity) {
  CX_CHECK_ARG(A != nullptr && capacity >= 0);
  const int n = int(A->parts.size());
  if (n == 0) {  // a plain matrix is its own single shard
    if (capacity >= 2) {
      if (e_block_bounds) { e_block_bounds[0] = 0; e_block_bounds[1] = A->nelim; }
      if (row_block_bounds) { row_block_bounds[0] = 0; row_block_bounds[1] = A->R; }
    }
    return 1;
  }
  if (capacity >= n + 1)
    for (int i = 0; i <= n; ++i) {
      if (e_block_bounds) e_block_bounds[i] = A->part_e0[size_t(i)];
      if (row_block_bounds) row_block_bounds[i] = A->part_rowblk0[size_t(i)];
    }
  return n;
}

// called by cx_context_destroy for a front, before the front's own stream goes
void cxm_context_destroy_shards(cx_context* front) {
  if (front->group) {
    Group(front)->run([&](int i) {
      cx_context_destroy(front->shards[size_t(i)]);
      return CX_OK;
    });
    delete Group(front);
    front->group = nullptr;
  }
  front->shards.clear();
}

// ------------------------------------------------------------------ matrix
static cx_matrix* NewFrontMatrix(cx_context* ctx, int n) {
  auto* A = new cx_matrix;
  A->ctx = ctx;
  A->parts.assign(size_t(n), nullptr);
  A->part_e0.assign(size_t(n) + 1, 0);
  A->part_rowblk0.assign(size_t(n) + 1, 0);
  A->part_row0.assign(size_t(n) + 1, 0);
  A->part_ecol0.assign(size_t(n) + 1, 0);
  A->part_runs.resize(size_t(n));
  return A;
}

int cxm_matrix_create(cx_context* ctx, const cx_block_structure* bs, int32_t nelim, cx_matrix** out) {
  const int n = NumShards(ctx);
  const int32_t R = bs->num_row_blocks, Cb = bs->num_col_blocks;
  if (nelim <= 0 || R == 0) {
    cx_set_error("a matrix on a multi-shard context is cut by e-blocks: num_eliminate_blocks must be positive");
    return CX_ERR_UNSUPPORTED;
  }
  // the cut needs every row to start with an e-block and the rows to be sorted by it (the order
  // LexicographicallyOrderResidualBlocks leaves, reorder_program.cc:256-338)
  std::vector<int32_t> row_e(static_cast<size_t>(R));
  for (int32_t r = 0; r < R; ++r) {
    const int32_t b = bs->row_cell_begin[r], e = bs->row_cell_begin[r + 1];
    const int32_t first = b < e ? bs->cells[b].block_id : -1;
    if (first < 0 || first >= nelim || (r > 0 && first < row_e[size_t(r) - 1])) {
      cx_set_error("row block %d: a multi-shard matrix needs every row to start with an e-block, rows sorted by e-block", r);
      return CX_ERR_UNSUPPORTED;
    }
    row_e[size_t(r)] = first;
  }
  auto* A = NewFrontMatrix(ctx, n);
  A->R = R;
  A->Cb = Cb;
  A->nelim = nelim;
  A->num_rows = int64_t(bs->row_blocks[R - 1].position) + bs->row_blocks[R - 1].size;
  A->num_cols = int64_t(bs->col_blocks[Cb - 1].position) + bs->col_blocks[Cb - 1].size;
  for (int c = 0; c < Cb; ++c) (c < nelim ? A->num_cols_e : A->num_cols_f) += bs->col_blocks[c].size;
  A->num_row_blocks_e = R;
  int rc = cx_partition_points(bs, nelim, n, A->part_e0.data());
  if (rc != CX_OK) { delete A; return rc; }
  for (int i = 0; i <= n; ++i) {
    const int32_t e0 = A->part_e0[size_t(i)];
    const int32_t r0 = int32_t(std::lower_bound(row_e.begin(), row_e.end(), e0) - row_e.begin());
    A->part_rowblk0[size_t(i)] = r0;
    A->part_row0[size_t(i)] = r0 < R ? bs->row_blocks[r0].position : A->num_rows;
    A->part_ecol0[size_t(i)] = e0 < nelim ? bs->col_blocks[e0].position : A->num_cols_e;
  }
  for (int i = 0; i < n; ++i)
    if (A->part_rowblk0[size_t(i)] == A->part_rowblk0[size_t(i) + 1]) {
      cx_set_error("shard %d of %d would hold no rows (%d row blocks over %d e-blocks): use fewer shards", i, n, R, nelim);
      delete A;
      return CX_ERR_INVALID_ARGUMENT;
    }
  std::vector<int64_t> part_nnz(size_t(n), 0);
  rc = RunShards(ctx, [&](int i) -> int {
    const int32_t r0 = A->part_rowblk0[size_t(i)], r1 = A->part_rowblk0[size_t(i) + 1];
    const int32_t e0 = A->part_e0[size_t(i)], e1 = A->part_e0[size_t(i) + 1], ne = e1 - e0;
    const int32_t c0 = bs->row_cell_begin[r0], c1 = bs->row_cell_begin[r1];
    std::vector<cx_block> rows(size_t(r1 - r0)), cols(size_t(ne + Cb - nelim));
    std::vector<int32_t> rcb(size_t(r1 - r0) + 1);
    std::vector<cx_cell> cells(size_t(c1 - c0));
    for (int32_t r = r0; r < r1; ++r) {
      rows[size_t(r - r0)] = cx_block{bs->row_blocks[r].size, int32_t(bs->row_blocks[r].position - A->part_row0[size_t(i)])};
      rcb[size_t(r - r0)] = bs->row_cell_begin[r] - c0;
    }
    rcb[size_t(r1 - r0)] = c1 - c0;
    const int32_t ecols = int32_t(A->part_ecol0[size_t(i) + 1] - A->part_ecol0[size_t(i)]);
    for (int32_t j = e0; j < e1; ++j)
      cols[size_t(j - e0)] = cx_block{bs->col_blocks[j].size, int32_t(bs->col_blocks[j].position - A->part_ecol0[size_t(i)])};
    for (int32_t j = nelim; j < Cb; ++j)
      cols[size_t(ne + j - nelim)] = cx_block{bs->col_blocks[j].size, int32_t(bs->col_blocks[j].position - A->num_cols_e + ecols)};
    // Value positions: the shard's cells keep the ORDER they have in the front's value array and are packed -- for the
    // BuildJacobianLayout layout (all E cells, then all F cells) that is the shard's slice of E followed by its slice
    // of F, again the reference layout.  First cells of the rows and the other cells are two ascending sequences in
    // that layout (merge); anything else is sorted.
    std::vector<std::pair<int32_t, int32_t>> first, rest, order;  // (front position, cell index)
    first.reserve(size_t(r1 - r0));
    rest.reserve(size_t(c1 - c0) - size_t(r1 - r0));
    for (int32_t r = r0; r < r1; ++r)
      for (int32_t c = bs->row_cell_begin[r]; c < bs->row_cell_begin[r + 1]; ++c)
        (c == bs->row_cell_begin[r] ? first : rest).push_back({bs->cells[c].position, c});
    order.resize(first.size() + rest.size());
    if (std::is_sorted(first.begin(), first.end()) && std::is_sorted(rest.begin(), rest.end())) {
      std::merge(first.begin(), first.end(), rest.begin(), rest.end(), order.begin());
    } else {
      std::copy(first.begin(), first.end(), order.begin());
      std::copy(rest.begin(), rest.end(), order.begin() + int64_t(first.size()));
      std::sort(order.begin(), order.end());
    }
    std::vector<int32_t> row_of_cell(size_t(c1 - c0));
    for (int32_t r = r0; r < r1; ++r)
      for (int32_t c = bs->row_cell_begin[r]; c < bs->row_cell_begin[r + 1]; ++c) row_of_cell[size_t(c - c0)] = r;
    int64_t local = 0;
    auto& runs = A->part_runs[size_t(i)];
    runs.clear();
    for (const auto& pc : order) {
      const cx_cell& g = bs->cells[pc.second];
      const int64_t sz = int64_t(bs->row_blocks[row_of_cell[size_t(pc.second - c0)]].size) * bs->col_blocks[g.block_id].size;
      if (local + sz >= (int64_t(1) << 31)) {
        cx_set_error("shard value positions exceed int32");
        return CX_ERR_UNSUPPORTED;
      }
      cells[size_t(pc.second - c0)] = cx_cell{g.block_id < nelim ? g.block_id - e0 : g.block_id - nelim + ne, int32_t(local)};
      if (!runs.empty() && runs.back().global + runs.back().len == g.position) runs.back().len += sz;
      else runs.push_back(cx_matrix::ValueRun{g.position, local, sz});
      local += sz;
    }
    part_nnz[size_t(i)] = local;
    cx_block_structure sub{r1 - r0, ne + Cb - nelim, rows.data(), cols.data(), rcb.data(), cells.data()};
    return cx_matrix_create(ctx->shards[size_t(i)], &sub, ne, &A->parts[size_t(i)]);
  });
  if (rc != CX_OK) {
    std::string keep = cx_last_error();
    cxm_matrix_destroy(A);
    cx_set_error("%s", keep.c_str());
    return rc;
  }
  A->nnz = std::accumulate(part_nnz.begin(), part_nnz.end(), int64_t(0));
  A->is239 = true;
  for (cx_matrix* p : A->parts) A->is239 = A->is239 && p->is239;
  if (!A->is239) {  // the sharded solvers exist for the native static layout only: say so now, not at the first solve
    cxm_matrix_destroy(A);
    cx_set_error("a multi-shard context takes matrices in the static <2,3,9> layout (BuildJacobianLayout) only");
    return CX_ERR_UNSUPPORTED;
  }
  cx_detect_structure(bs, nelim, &A->row_size, &A->e_size, &A->f_size);
  if (A->is239) {
    A->O = R;
    A->P = nelim;
    A->C = Cb - nelim;
  }
  *out = A;
  return CX_OK;
}

void cxm_matrix_destroy(cx_matrix* A) {
  if (!A) return;
  if (A->parts_owned && A->ctx->group)
    Group(A->ctx)->run([&](int i) {
      if (A->parts[size_t(i)]) cx_matrix_destroy(A->parts[size_t(i)]);
      return CX_OK;
    });
  A->parts.clear();
  delete A;
}

int cxm_matrix_set_values(cx_matrix* A, const double* src, int32_t memspace) {
  CX_TRY(RequireHost(memspace, "cx_matrix_set_values"));
  (void)cx_pin_range(src, size_t(A->nnz) * sizeof(double));
  return RunShards(A->ctx, [&](int i) -> int {
    cx_matrix* p = A->parts[size_t(i)];
    for (const auto& run : A->part_runs[size_t(i)])
      CX_TRY(cx_copy_h2d(p->ctx, p->d_values.p + run.local, src + run.global, size_t(run.len) * sizeof(double)));
    CX_HIP(hipStreamSynchronize(p->ctx->stream));
    return cx_matrix_values_changed(p);
  });
}

int cxm_matrix_get_values(const cx_matrix* A, double* dst) {
  return RunShards(A->ctx, [&](int i) -> int {
    const cx_matrix* p = A->parts[size_t(i)];
    for (const auto& run : A->part_runs[size_t(i)])
      CX_TRY(cx_copy_d2h(p->ctx, dst + run.global, p->d_values.p + run.local, size_t(run.len) * sizeof(double)));
    CX_HIP(hipStreamSynchronize(p->ctx->stream));
    return CX_OK;
  });
}

int cxm_matrix_set_zero(cx_matrix* A) {
  return RunShards(A->ctx, [&](int i) { return cx_matrix_set_zero(A->parts[size_t(i)]); });
}

int cxm_matrix_values_changed(cx_matrix* A) {
  for (cx_matrix* p : A->parts) CX_TRY(cx_matrix_values_changed(p));
  return CX_OK;
}

int cxm_matrix_op(cx_matrix* A, int op, const double* x, double* y, int32_t memspace) {
  CX_TRY(RequireHost(memspace, "a matrix product"));
  const int64_t ne = A->num_cols_e, nf = A->num_cols_f;
  // the caller's whole arrays are registered before the shards copy their slices of them
  if (x) (void)cx_pin_range(x, size_t(op == 1 ? A->num_rows : A->num_cols) * sizeof(double));
  if (y) (void)cx_pin_range(y, size_t(op == 0 || op == 4 ? A->num_rows : A->num_cols) * sizeof(double));
  return RunShards(A->ctx, [&](int i) -> int {
    cx_matrix* p = A->parts[size_t(i)];
    cx_context* c = p->ctx;
    switch (op) {
      case 0: {  // y += A x: the shard's rows are a contiguous slice of y
        cx_host_slices xs = ColSlices(A, i, x, ne, nf), ys = RowSlices(A, i, y);
        return cx_matrix_right_multiply(p, xs.as_arg(), ys.as_arg(), CX_HOST_SLICES);
      }
      case 4: {  // y = A x
        cx_host_slices xs = ColSlices(A, i, x, ne, nf), ys = RowSlices(A, i, y);
        return cx_matrix_right_multiply_overwrite(p, xs.as_arg(), ys.as_arg(), CX_HOST_SLICES);
      }
      case 1:    // y += A'x: the e part is the shard's own; the f parts are partial sums, added in shard order on top of
      case 2: {  // the caller's y_f (shard 0 brings it in).  x = diag(A'A): the same without the caller's values
        CX_HIP(hipSetDevice(c->device));
        cx_host_slices xs = RowSlices(A, i, x), ys = ColSlices(A, i, y, ne, nf, true);
        HostOrDevice hx(c), hy(c);
        if (op == 1) CX_TRY(hx.in(xs.as_arg(), size_t(p->num_rows), CX_HOST_SLICES));
        CX_TRY(hy.inout(ys.as_arg(), size_t(p->num_cols), CX_HOST_SLICES, false));
        CX_TRY(cx_matrix_ensure_ft(p));
        if (op == 1) {
          CX_TRY(cx_copy_h2d(c, hy.dptr, ys.head, size_t(ys.nhead) * sizeof(double)));
          if (i == 0) CX_TRY(cx_copy_h2d(c, hy.dptr + ys.nhead, ys.tail, size_t(nf) * sizeof(double)));
          else CX_HIP(hipMemsetAsync(hy.dptr + ys.nhead, 0, size_t(nf) * sizeof(double), c->stream));
          CX_TRY(cxk_left_multiply(p, hx.dptr, hy.dptr));
        } else {
          CX_TRY(cxk_squared_column_norm(p, hy.dptr));
        }
        CX_TRY(cx_allreduce_device(c, hy.dptr + ys.nhead, nf));
        return hy.out();
      }
      default: {
        cx_host_slices xs = ColSlices(A, i, x, ne, nf);
        return cx_matrix_scale_columns(p, xs.as_arg(), CX_HOST_SLICES);
      }
    }
  });
}

// ------------------------------------------------------------------ solver
static void DestroySolverParts(cx_solver* S) {
  if (S->parts.empty()) return;
  Group(S->ctx)->run([&](int i) {
    if (S->parts[size_t(i)]) cx_solver_destroy(S->parts[size_t(i)]);
    return CX_OK;
  });
  S->parts.clear();
  S->parts_for = nullptr;
}

void cxm_solver_destroy(cx_solver* S) { DestroySolverParts(S); }

static int EnsureSolverParts(cx_solver* S, const cx_matrix* A) {
  if (!S->parts.empty() && S->parts_for == A) {
    // (the same address may belong to a NEW matrix with another cut: the shards' solvers are only good for the e-blocks
    // they were created with)
    bool same_cut = S->parts.size() == A->parts.size();
    for (size_t i = 0; same_cut && i < S->parts.size(); ++i) same_cut = S->parts[i]->opt.num_eliminate_blocks == A->parts[i]->nelim;
    if (same_cut) return CX_OK;
  }
  DestroySolverParts(S);
  const int n = int(A->parts.size());
  S->parts.assign(size_t(n), nullptr);
  int rc = RunShards(S->ctx, [&](int i) {
    cx_solver_options o = S->opt;
    o.num_eliminate_blocks = A->parts[size_t(i)]->nelim;  // the shard's own e-blocks
    return cx_solver_create(S->ctx->shards[size_t(i)], &o, &S->parts[size_t(i)]);
  });
  if (rc != CX_OK) {
    std::string keep = cx_last_error();
    DestroySolverParts(S);
    cx_set_error("%s", keep.c_str());
    return rc;
  }
  S->parts_for = A;
  return CX_OK;
}

int cxm_solver_solve(cx_solver* S, cx_matrix* A, const double* b, const cx_per_solve_options* ps, double* x, cx_summary* summary) {
  std::memset(summary, 0, sizeof(*summary));
  auto fatal = [&](int rc) {
    summary->termination_type = CX_FATAL_ERROR;
    std::snprintf(summary->message, sizeof(summary->message), "%s", cx_last_error());
    return rc;
  };
  if (A->parts.empty() || S->ctx != A->ctx) {
    cx_set_error("solver and matrix are not fronts of the same multi-shard context");
    return fatal(CX_ERR_INVALID_ARGUMENT);
  }
  if (RequireHost(ps->memspace, "cx_solver_solve") != CX_OK) return fatal(CX_ERR_UNSUPPORTED);
  if (S->opt.type != CX_CGNR && S->opt.num_eliminate_blocks != A->nelim) {
    cx_set_error("solver eliminates %d blocks, the matrix has %d e-blocks", S->opt.num_eliminate_blocks, A->nelim);
    return fatal(CX_ERR_INVALID_ARGUMENT);
  }
  const cx_evaluator* residual_owner = nullptr;
  if (ps->b_on_device) {  // the token of cx_evaluator_device_residuals on a front: every shard kept its rows in HBM
    std::lock_guard<std::mutex> lk(g_front_evaluators_mutex);
    const auto* e = reinterpret_cast<const cx_evaluator*>(b);
    if (g_front_evaluators.count(e) && e->J == A) residual_owner = e;
    if (!residual_owner) {
      cx_set_error("b_on_device on a multi-shard context takes the token of cx_evaluator_device_residuals of this matrix' evaluator");
      return fatal(CX_ERR_INVALID_ARGUMENT);
    }
  }
  int rc = EnsureSolverParts(S, A);
  if (rc != CX_OK) return fatal(rc);
  const int n = int(A->parts.size());
  const int64_t ne = A->num_cols_e, nf = A->num_cols_f;
  std::vector<cx_summary> sums(static_cast<size_t>(n));
  // whole arrays registered once, by the caller's thread; the shards then copy their slices of them
  if (!residual_owner) (void)cx_pin_range(b, size_t(A->num_rows) * sizeof(double));
  if (ps->D) (void)cx_pin_range(ps->D, size_t(A->num_cols) * sizeof(double));
  (void)cx_pin_range(x, size_t(A->num_cols) * sizeof(double));
  rc = RunShards(S->ctx, [&](int i) -> int {
    cx_matrix* p = A->parts[size_t(i)];
    cx_per_solve_options psi = *ps;
    psi.memspace = CX_HOST_SLICES;
    cx_host_slices Ds, bs = RowSlices(A, i, b), xs = ColSlices(A, i, x, ne, nf, true);
    if (ps->D) {
      Ds = ColSlices(A, i, ps->D, ne, nf);
      psi.D = Ds.as_arg();
    }
    const double* bi = bs.as_arg();
    if (residual_owner) {
      bi = cx_evaluator_device_residuals(residual_owner->parts[size_t(i)]);
      if (!bi) {
        cx_set_error("the evaluator's device residuals are no longer valid");
        return CX_ERR_INVALID_ARGUMENT;
      }
    }
    return cx_solver_solve(S->parts[size_t(i)], p, bi, &psi, xs.as_arg(), &sums[size_t(i)]);
  });
  if (rc != CX_OK) return fatal(rc);
  // the camera-space vectors of the shards are replicas: the same exchange results, the same arithmetic -- so the
  // shards must agree on how the solve ended
  for (int i = 1; i < n; ++i)
    if (sums[size_t(i)].termination_type != sums[0].termination_type || sums[size_t(i)].num_iterations != sums[0].num_iterations) {
      cx_set_error("shards disagree: shard 0 ended with type %d after %d iterations, shard %d with type %d after %d",
                   sums[0].termination_type, sums[0].num_iterations, i, sums[size_t(i)].termination_type, sums[size_t(i)].num_iterations);
      return fatal(CX_ERR_COMM);
    }
  *summary = sums[0];
  // phase times: the slowest shard of each phase; the exchange counters are the same on every shard
  S->timing = S->parts[0]->timing;
  for (int i = 1; i < n; ++i) {
    const cx_solve_timing& t = S->parts[size_t(i)]->timing;
    S->timing.setup_ms = std::max(S->timing.setup_ms, t.setup_ms);
    S->timing.eliminate_ms = std::max(S->timing.eliminate_ms, t.eliminate_ms);
    S->timing.reduced_solve_ms = std::max(S->timing.reduced_solve_ms, t.reduced_solve_ms);
    S->timing.back_substitute_ms = std::max(S->timing.back_substitute_ms, t.back_substitute_ms);
    S->timing.total_ms = std::max(S->timing.total_ms, t.total_ms);
    S->timing.allreduce_ms = std::max(S->timing.allreduce_ms, t.allreduce_ms);
    S->timing.allreduce_host_ms = std::max(S->timing.allreduce_host_ms, t.allreduce_host_ms);
  }
  return CX_OK;
}

// ------------------------------------------------------------------ evaluator
int cxm_evaluator_create_bal(cx_context* ctx, int32_t C, int32_t P, int64_t O, const int32_t* cam, const int32_t* pt,
                             const double* obs, cx_evaluator** out) {
  const int n = NumShards(ctx);
  // the cut of cx_partition_points (24 non-zeros per residual block, so equal non-zeros = equal observation counts)
  std::vector<int64_t> cum(size_t(P) + 1, 0);
  for (int64_t i = 0; i < O; ++i) cum[size_t(pt[i]) + 1]++;
  std::partial_sum(cum.begin(), cum.end(), cum.begin());
  std::vector<int32_t> bounds(size_t(n) + 1, 0);
  for (int k = 1; k < n; ++k) {
    const int64_t target = O * k / n;
    const int32_t j = int32_t(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
    bounds[size_t(k)] = std::max(bounds[size_t(k) - 1], std::min(j, P));
  }
  bounds[size_t(n)] = P;
  for (int k = 0; k < n; ++k)
    if (cum[size_t(bounds[size_t(k) + 1])] == cum[size_t(bounds[size_t(k)])]) {
      cx_set_error("shard %d of %d would hold no observations: use fewer shards", k, n);
      return CX_ERR_INVALID_ARGUMENT;
    }
  std::vector<std::vector<int32_t>> cams(static_cast<size_t>(n)), pts(static_cast<size_t>(n));
  std::vector<std::vector<double>> xy(static_cast<size_t>(n));
  std::vector<int32_t> shard_of(static_cast<size_t>(O));
  std::vector<int64_t> local_index(static_cast<size_t>(O));
  for (int k = 0; k < n; ++k) {
    const size_t cnt = size_t(cum[size_t(bounds[size_t(k) + 1])] - cum[size_t(bounds[size_t(k)])]);
    cams[size_t(k)].reserve(cnt);
    pts[size_t(k)].reserve(cnt);
    xy[size_t(k)].reserve(2 * cnt);
  }
  for (int64_t i = 0; i < O; ++i) {
    const int k = int(std::upper_bound(bounds.begin() + 1, bounds.end(), pt[i]) - bounds.begin()) - 1;
    shard_of[size_t(i)] = k;
    local_index[size_t(i)] = int64_t(cams[size_t(k)].size());
    cams[size_t(k)].push_back(cam[i]);
    pts[size_t(k)].push_back(pt[i] - bounds[size_t(k)]);
    xy[size_t(k)].push_back(obs[2 * i]);
    xy[size_t(k)].push_back(obs[2 * i + 1]);
  }
  auto* e = new cx_evaluator;
  e->ctx = ctx;
  e->C = C;
  e->P = P;
  e->O = O;
  e->parts.assign(size_t(n), nullptr);
  int rc = RunShards(ctx, [&](int i) {
    return cx_evaluator_create_bal(ctx->shards[size_t(i)], C, bounds[size_t(i) + 1] - bounds[size_t(i)], int64_t(cams[size_t(i)].size()),
                                   cams[size_t(i)].data(), pts[size_t(i)].data(), xy[size_t(i)].data(), &e->parts[size_t(i)]);
  });
  if (rc != CX_OK) {
    std::string keep = cx_last_error();
    cxm_evaluator_destroy(e);
    cx_set_error("%s", keep.c_str());
    return rc;
  }
  // the front of J: rows are sorted by point, so shard i owns the row blocks [sum of the earlier shards' counts, ...)
  cx_matrix* A = NewFrontMatrix(ctx, n);
  A->parts_owned = false;
  A->R = int32_t(O);
  A->Cb = P + C;
  A->nelim = P;
  A->num_rows = 2 * O;
  A->num_cols = 3 * int64_t(P) + 9 * int64_t(C);
  A->nnz = 24 * O;
  A->num_cols_e = 3 * int64_t(P);
  A->num_cols_f = 9 * int64_t(C);
  A->num_row_blocks_e = int32_t(O);
  A->row_size = 2; A->e_size = 3; A->f_size = 9;
  A->is239 = true;
  A->O = O; A->P = P; A->C = C;
  for (int i = 0; i <= n; ++i) {
    const int64_t r0 = cum[size_t(bounds[size_t(i)])];
    A->part_e0[size_t(i)] = bounds[size_t(i)];
    A->part_rowblk0[size_t(i)] = int32_t(r0);
    A->part_row0[size_t(i)] = 2 * r0;
    A->part_ecol0[size_t(i)] = 3 * int64_t(bounds[size_t(i)]);
  }
  for (int i = 0; i < n; ++i) {
    const int64_t r0 = A->part_rowblk0[size_t(i)], Oi = e->parts[size_t(i)]->O;
    A->parts[size_t(i)] = e->parts[size_t(i)]->J;
    A->part_runs[size_t(i)] = {cx_matrix::ValueRun{6 * r0, 0, 6 * Oi}, cx_matrix::ValueRun{6 * O + 18 * r0, 6 * Oi, 18 * Oi}};
  }
  e->J = A;
  e->row_of_obs.resize(size_t(O));
  for (int64_t i = 0; i < O; ++i) {
    const int k = shard_of[size_t(i)];
    e->row_of_obs[size_t(i)] = A->part_rowblk0[size_t(k)] + e->parts[size_t(k)]->row_of_obs[size_t(local_index[size_t(i)])];
  }
  {
    std::lock_guard<std::mutex> lk(g_front_evaluators_mutex);
    g_front_evaluators.insert(e);
  }
  *out = e;
  return CX_OK;
}

void cxm_evaluator_destroy(cx_evaluator* e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(g_front_evaluators_mutex);
    g_front_evaluators.erase(e);
  }
  if (e->J) cxm_matrix_destroy(e->J);  // the front only: the parts belong to the shards' evaluators
  Group(e->ctx)->run([&](int i) {
    if (e->parts[size_t(i)]) cx_evaluator_destroy(e->parts[size_t(i)]);
    return CX_OK;
  });
  delete e;
}

int cxm_evaluator_forward_settings(cx_evaluator* e) {
  for (cx_evaluator* p : e->parts) {
    CX_TRY(cx_evaluator_set_loss(p, e->loss_type, e->loss_a, e->loss_b));
    CX_TRY(cx_evaluator_set_camera_model(p, e->camera_model));
    CX_TRY(cx_evaluator_set_emit_camera_major(p, e->emit_ft ? 1 : 0));
  }
  return CX_OK;
}

int cxm_evaluator_set_column_scale(cx_evaluator* e, const double* scale, int32_t memspace) {
  if (scale == nullptr) {
    for (cx_evaluator* p : e->parts) CX_TRY(cx_evaluator_set_column_scale(p, nullptr, CX_HOST));
    return CX_OK;
  }
  CX_TRY(RequireHost(memspace, "cx_evaluator_set_column_scale"));
  cx_matrix* A = e->J;
  (void)cx_pin_range(scale, size_t(A->num_cols) * sizeof(double));
  return RunShards(e->ctx, [&](int i) -> int {
    cx_host_slices ss = ColSlices(A, i, scale, A->num_cols_e, A->num_cols_f);
    return cx_evaluator_set_column_scale(e->parts[size_t(i)], ss.as_arg(), CX_HOST_SLICES);
  });
}

static int64_t CameraStateSize(const cx_evaluator* e) { return e->camera_model == CX_CAMERA_ANGLE_AXIS ? 9 : 10; }

int cxm_evaluator_evaluate(cx_evaluator* e, const double* state, double* cost, double* residuals, double* gradient,
                           int32_t evaluate_jacobian, int32_t memspace) {
  CX_TRY(RequireHost(memspace, "cx_evaluator_evaluate"));
  cx_matrix* A = e->J;
  const int n = int(e->parts.size());
  const int64_t ne = A->num_cols_e, nf = A->num_cols_f, ncam = CameraStateSize(e) * e->C;
  std::vector<double> costs(size_t(n), 0.0);
  (void)cx_pin_range(state, size_t(ne + ncam) * sizeof(double));
  if (residuals) (void)cx_pin_range(residuals, size_t(A->num_rows) * sizeof(double));
  if (gradient) (void)cx_pin_range(gradient, size_t(A->num_cols) * sizeof(double));
  CX_TRY(RunShards(e->ctx, [&](int i) -> int {
    cx_host_slices ss = ColSlices(A, i, state, ne, ncam), rs, gs;
    if (residuals) rs = RowSlices(A, i, residuals);
    if (gradient) gs = ColSlices(A, i, gradient, ne, nf, true);
    // a shard's context has nranks > 1: cost and the camera part of the gradient come back summed over the shards
    return cx_evaluator_evaluate(e->parts[size_t(i)], ss.as_arg(), cost ? &costs[size_t(i)] : nullptr,
                                 residuals ? rs.as_arg() : nullptr, gradient ? gs.as_arg() : nullptr, evaluate_jacobian,
                                 CX_HOST_SLICES);
  }));
  if (cost) *cost = costs[0];
  return CX_OK;
}

int cxm_evaluator_plus(cx_evaluator* e, const double* x, const double* delta, double* x_plus_delta, int32_t memspace) {
  CX_TRY(RequireHost(memspace, "cx_evaluator_plus"));
  cx_matrix* A = e->J;
  const int n = int(e->parts.size());
  const int64_t ne = A->num_cols_e, nf = A->num_cols_f, ncam = CameraStateSize(e) * e->C;
  (void)n;
  return RunShards(e->ctx, [&](int i) -> int {
    cx_host_slices xs = ColSlices(A, i, x, ne, ncam), ds = ColSlices(A, i, delta, ne, nf), os = ColSlices(A, i, x_plus_delta, ne, ncam, true);
    return cx_evaluator_plus(e->parts[size_t(i)], xs.as_arg(), ds.as_arg(), os.as_arg(), CX_HOST_SLICES);
  });
}

const double* cxm_evaluator_device_residuals(const cx_evaluator* e) {
  for (const cx_evaluator* p : e->parts)
    if (cx_evaluator_device_residuals(p) == nullptr) return nullptr;
  return reinterpret_cast<const double*>(e);  // a token, not an address to read: cx_solver_solve with b_on_device resolves it per shard
}

double cxm_evaluator_last_kernel_ms(const cx_evaluator* e) {
  double ms = 0.0;
  for (const cx_evaluator* p : e->parts) ms = std::max(ms, cx_evaluator_last_kernel_ms(p));
  return ms;
}

// ------------------------------------------------------------------ minimizer
int cxm_minimize(cx_evaluator* e, cx_solver* s, const cx_minimizer_options* options, double* state, int32_t memspace,
                 cx_minimizer_summary* summary, cx_iteration_summary* iterations, int32_t capacity) {
  CX_TRY(RequireHost(memspace, "cx_minimize"));
  if (e->parts.empty() || s->ctx != e->ctx) {
    cx_set_error("solver and evaluator are not fronts of the same multi-shard context");
    return CX_ERR_INVALID_ARGUMENT;
  }
  cx_matrix* A = e->J;
  CX_TRY(EnsureSolverParts(s, A));
  const int n = int(e->parts.size());
  const int64_t ne = A->num_cols_e, ncam = CameraStateSize(e) * e->C;
  std::vector<cx_minimizer_summary> sums(static_cast<size_t>(n));
  std::vector<std::vector<cx_iteration_summary>> its(static_cast<size_t>(n));
  CX_TRY(RunShards(e->ctx, [&](int i) -> int {
    cx_host_slices ss = ColSlices(A, i, state, ne, ncam, true);  // in and out: every shard reads the cameras, shard 0 returns them
    its[size_t(i)].resize(size_t(i == 0 ? capacity : 0));
    return cx_minimize(e->parts[size_t(i)], s->parts[size_t(i)], options, ss.as_arg(), CX_HOST_SLICES, &sums[size_t(i)],
                       i == 0 && capacity > 0 ? its[0].data() : nullptr, i == 0 ? capacity : 0);
  }));
  for (int i = 1; i < n; ++i)
    if (sums[size_t(i)].num_iterations != sums[0].num_iterations || sums[size_t(i)].termination_type != sums[0].termination_type) {
      cx_set_error("shards disagree about the minimisation (%d vs %d iterations)", sums[0].num_iterations, sums[size_t(i)].num_iterations);
      return CX_ERR_COMM;
    }
  *summary = sums[0];
  for (int k = 0; k < capacity && k < sums[0].num_iterations; ++k) iterations[k] = its[0][size_t(k)];
  return CX_OK;
}
