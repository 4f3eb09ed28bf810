// Declarations shared by the solver translation units (cx_solver.hip, cx_generic.hip).
#ifndef CX_SOLVER_INTERNAL_H_
#define CX_SOLVER_INTERNAL_H_

#include "cx_internal.h"

enum CgFlag : int {
  CG_RUNNING = 0,
  CG_CONVERGED_Q = 1,
  CG_CONVERGED_R = 2,
  CG_MAX_ITER = 3,
  CG_FAIL_RHO = 4,
  CG_FAIL_BETA = 5,
  CG_INDEFINITE = 6,
  CG_FAIL_ALPHA = 7,
  // outcomes of the device prologue (before the first iteration)
  CG_ZERO_RHS = 8,
  CG_CONVERGED_AT_START = 9,
  CG_FAIL_PRECONDITIONER = 10
};

struct CgState {
  double rho, last_rho, beta, pq, alpha, Q0, Q1, norm_r, zeta;
  double tol_r, q_tol;
  double s0, s1;  // reduction results
  int flag, iter, min_iter, max_iter;
  int seq, pad;  // seq: iteration this copy was published for (host-pinned ring only)
};

constexpr int kRedBlocks = 2048;  // upper bound of the reduction grids (deterministic for a given n)

// Where a fused "apply + dot product" kernel leaves its sums and which CG scalar step follows
// (cx_solver.hip: dot2_finish).
struct DotTail {
  double* partial;
  unsigned* ticket;
  CgState* st;
  CgState* ring;
  int ring_slots, op, iter;
};

// Small problems (vectors of at most kSmallCgMax entries, one rank): the camera-sized rest of a CG iteration -- the end of
// the operator application with p.q, the x / r update with the Q test, the preconditioner with r.z and the new search
// direction -- runs as ONE single-workgroup kernel (k_cg_small_tail, cx_solver.hip) instead of four launches with ticket
// reductions.  An operator takes part by handing over where its product's partial sums lie.
constexpr int kSmallCgMax = 4096;
struct SmallProduct {
  const double* partial9 = nullptr;       // [segments][9] per-segment partial sums of F't
  const int32_t* cam_seg_start = nullptr;  // [C + 1]
  const double* d = nullptr;               // LM diagonal of the camera part (may be NULL)
};

// The set-up of such a run in one launch too (k_cg_small_setup, vectors of at most kSmallSetupMax entries): the segment
// partial sums cxs_implicit_init(..., defer_reduce) left behind become the block-Jacobi blocks and the reduced right-hand
// side, the blocks are inverted, and the CG prologue and the head of iteration 1 follow.
constexpr int kSmallSetupMax = 1024;
struct SmallSetup {
  const double* partial45 = nullptr;       // [segments][45] packed upper triangles of the cameras' 9x9 blocks
  const double* partial9 = nullptr;        // [segments][9] the reduced right-hand side
  const int32_t* cam_seg_start = nullptr;  // [C + 1]
  const double* Df = nullptr;              // LM diagonal of the camera part (may be NULL)
  double* blocks = nullptr;                // [C][81] out: the inverted blocks
  int C = 0;
};

struct LinOp {
  virtual ~LinOp() = default;
  virtual int64_t size() const = 0;
  virtual int apply(const double* x, double* y) = 0;  // y = A x
  // small-problem form: run the product of x up to its per-segment partial sums and say where they are (false: no such form)
  virtual bool small_partials(const double* x, SmallProduct* out, int* rc) {
    (void)x; (void)out; (void)rc;
    return false;
  }
  // a block-Jacobi preconditioner's inverted 9x9 blocks (NULL: not of that kind)
  virtual const double* block9_inverse() const { return nullptr; }
  // y = A x and the CG dot product x.y with its scalar step in the operator's last kernel;
  // *done = false when the operator has no fused form (the caller then applies and reduces).
  virtual int apply_dot(const double* x, double* y, const DotTail& tail, bool* done) {
    (void)x; (void)y; (void)tail;
    *done = false;
    return CX_OK;
  }
};

// Per-kernel device time of the last solve, sampled with HIP event pairs on the
// context stream (read back after the solve; no synchronisation inside the loop).
struct KernelTimer {
  static constexpr int kSlots = 5, kMaxSamples = 64;
  const char* names[kSlots] = {"k_chunk_pass<0>", "k_cam_ft+k_cam_reduce9_dot", "k_right_239", "k_left_e_239+k_cam_ft", "k_sym_spmv"};
  hipEvent_t ev[kSlots][kMaxSamples][2] = {};
  int count[kSlots] = {};
  int launches[kSlots] = {};
  double total_ms[kSlots] = {};
  bool created = false;
  void reset() { for (int i = 0; i < kSlots; ++i) { count[i] = 0; launches[i] = 0; total_ms[i] = 0.0; } }
  int begin(int slot, hipStream_t st) {
    if (!created) {
      for (auto& a : ev) for (auto& b : a) for (auto& e : b) CX_HIP(hipEventCreate(&e));
      created = true;
    }
    if (!enabled) return CX_OK;
    ++launches[slot];
    if (count[slot] < max_samples) CX_HIP(hipEventRecord(ev[slot][count[slot]][0], st));
    return CX_OK;
  }
  int end(int slot, hipStream_t st) {
    if (!enabled) return CX_OK;
    if (count[slot] < max_samples) { CX_HIP(hipEventRecord(ev[slot][count[slot]][1], st)); ++count[slot]; }
    return CX_OK;
  }
  // false: a solve that takes no samples and keeps the numbers of the last one that did (cx_solver::diag)
  bool enabled = true;
  // how many launches per slot are bracketed by events in a solve (each pair costs the host two enqueues and the queue
  // two markers: a launch-bound small solve samples less)
  int max_samples = kMaxSamples;
  int collect() {  // call after the stream has been synchronised
    if (!enabled) return CX_OK;
    for (int s = 0; s < kSlots; ++s) {
      total_ms[s] = 0.0;
      for (int i = 0; i < count[s]; ++i) {
        float f = 0.f;
        CX_HIP(hipEventElapsedTime(&f, ev[s][i][0], ev[s][i][1]));
        total_ms[s] += f;
      }
    }
    return CX_OK;
  }
  ~KernelTimer() {
    if (created) for (auto& a : ev) for (auto& b : a) for (auto& e : b) (void)hipEventDestroy(e);
  }
};

struct cx_solver {
  cx_context* ctx = nullptr;
  cx_solver_options opt{};
  cx_solve_timing timing{};
  KernelTimer ktimer;
  // persistent device scratch
  DevBuf<double> v_p, v_r, v_z, v_tmp, v_x, v_rhs, v_rows, v_rows2, v_cols;
  DevBuf<double> ete_inv, cam_blocks, pt_blocks, g_e, lhs, partial, v_pack;
  DevBuf<CgState> state, spse_state;
  DevBuf<double> v_spse, v_spse_rows;
  DevBuf<double> lhs_copy;  // dynamic-size DENSE / SPARSE_SCHUR with refinement: S beside its factor
  CgState* ring_h = nullptr;  // host-pinned, device-visible ring of published CG states
  CgState* ring_d = nullptr;
  DevBuf<int> flag;
  DevBuf<unsigned> ticket;  // arrival counter of the fused reductions (zero between kernels)
  // phase timings are read back once, at the end of the solve (no synchronisation between the phases)
  double* pending_ms[4] = {};
  int num_pending = 0;
  // Launch-bound solves (at most kSmallCgMax reduced unknowns on one rank: some twenty enqueues of 2-17 us kernels).  Every
  // event record is a marker in the queue that costs about as much as one of those kernels -- the ten of a solve's phase
  // timings and kernel samples were 40 us of a 0.2 ms Ladybug-49 solve -- so such a solver takes them on its first solve and
  // on every kDiagPeriod-th after it (CX_DIAG_PERIOD=n overrides, 1 = always); in between cx_solver_last_timing() keeps the
  // phases of the last sampled solve, refreshes total_ms from the host clock, and cx_solver_kernel_stats() keeps its samples.
  static constexpr int kDiagPeriod = 16;
  int64_t num_solves = 0;
  bool diag = true;
  bool diag_requested = false;  // cx_solver_sample_next
  // front on a multi-shard context (cx_multi.hip): one solver per shard, created at the first solve of a matrix
  std::vector<cx_solver*> parts;
  const cx_matrix* parts_for = nullptr;
};

// ConjugateGradientsSolver (conjugate_gradients_solver.h:107-305) on device vectors of
// length n; entries [shared0, n) are replicated over the ranks of a sharded CGNR solve
// (shared0 == n otherwise).  x holds the initial guess; zero_initial says it is all zeros.
int cx_cg_run(cx_solver* S, int64_t n, int64_t shared0, LinOp& lhs, LinOp& pre, const double* rhs, double* x,
              bool zero_initial, double r_tol, double q_tol, cx_summary* summary);

// device flag check shared by the solvers: reads S->flag, fills a FAILURE summary when set
int cx_check_flag(cx_solver* S, const char* what, cx_summary* summary, bool* failed);

#endif
