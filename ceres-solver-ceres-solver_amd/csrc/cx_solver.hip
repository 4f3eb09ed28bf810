// LinearSolver implementations on the device: preconditioned conjugate gradients
// (ConjugateGradientsSolver, conjugate_gradients_solver.h:107-305), CgnrSolver
// (cgnr_solver.cc:85-207), IterativeSchurComplementSolver
// (iterative_schur_complement_solver.cc:64-199), Dense/SparseSchurComplementSolver
// (schur_complement_solver.cc:101-203).
//
// All CG scalars (rho, beta, p'q, alpha, Q, |r|) live in one device struct; every
// kernel of an iteration reads its coefficients from it and does nothing once a
// termination flag is set, so the host only polls that struct.  The reference's
// cuBLAS path synchronises on every dot product (cuda_vector.cc:95-182).
#include <chrono>
#include <cmath>
#include <memory>

#include <cstdlib>

#include "cx_internal.h"
#include "cx_kernels.h"
#include "cx_schur.h"
#include "cx_solver_internal.h"
#include "cx_visibility.h"

static int grid_for(int64_t n, int block) { return int((n + block - 1) / block); }

// ------------------------------------------------------------- CG device state

// partial[blockIdx] = sum a.b ; partial[kRedBlocks + blockIdx] = sum c.d  (second pair optional)
__global__ __launch_bounds__(256) void k_dot2_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                      const double* __restrict__ c, const double* __restrict__ d,
                                                      int64_t n, double* __restrict__ partial,
                                                      const CgState* __restrict__ st) {
  __shared__ double red[2 * 4];
  if (st && st->flag) return;
  double s[2] = {0.0, 0.0};
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    s[0] += a[i] * b[i];
    if (c) s[1] += c[i] * d[i];
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s[0];
    partial[kRedBlocks + blockIdx.x] = s[1];
  }
}

// what to do with the finished sums
enum FinOp : int { FIN_STORE = 0, FIN_RHO = 1, FIN_PQ = 2, FIN_Q = 3 };

// Final stage of a two-stage reduction, plus the scalar logic of the CG step it belongs to.
// add_prev: the sums of the rank-local part (already all-reduced into st->s0/s1) are added to
// this launch's sums (sharded CGNR: replicated camera part counted once).
// ring: host-pinned slots the state is published to after FIN_Q, so that the host can follow
// the iteration without synchronising the stream.
__device__ __forceinline__ void cg_scalar_step(double s0, double s1, int op, int iter, int add_prev, CgState* __restrict__ st,
                                               CgState* __restrict__ ring, int ring_slots) {
  double s[2] = {s0, s1};
  if (add_prev) { s[0] += st->s0; s[1] += st->s1; }
  st->s0 = s[0];
  st->s1 = s[1];
  if (op == FIN_RHO) {
    st->last_rho = st->rho;
    st->rho = s[0];
    if (s[0] == 0.0 || isinf(s[0])) {
      st->flag = CG_FAIL_RHO;
    } else if (iter > 1) {
      st->beta = st->rho / st->last_rho;
      if (st->beta == 0.0 || isinf(st->beta)) st->flag = CG_FAIL_BETA;
    }
  } else if (op == FIN_PQ) {
    st->pq = s[0];
    if (s[0] <= 0.0 || isinf(s[0])) {
      st->flag = CG_INDEFINITE;
    } else {
      st->alpha = st->rho / st->pq;
      if (isinf(st->alpha)) st->flag = CG_FAIL_ALPHA;
    }
  } else if (op == FIN_Q) {
    // s0 = x.(rhs + r), s1 = r.r
    st->Q1 = -s[0];
    st->norm_r = sqrt(s[1]);
    st->zeta = iter * (st->Q1 - st->Q0) / st->Q1;
    if (st->zeta < st->q_tol && iter >= st->min_iter) {
      st->flag = CG_CONVERGED_Q;
    } else {
      st->Q0 = st->Q1;
      if (st->norm_r <= st->tol_r && iter >= st->min_iter) st->flag = CG_CONVERGED_R;
      else if (iter >= st->max_iter) st->flag = CG_MAX_ITER;
    }
  }
  if (st->flag != CG_RUNNING || op == FIN_Q) {
    // publish: a failure flag raised in the middle of an iteration is published at once
    st->iter = iter;
    if (ring) {
      CgState* slot = ring + (iter % ring_slots);
      *slot = *st;
      __threadfence_system();
      __hip_atomic_store(&slot->seq, iter, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void k_dot2_final(const double* __restrict__ partial, int nblocks, int op, int iter,
                                                    int add_prev, CgState* __restrict__ st,
                                                    CgState* __restrict__ ring, int ring_slots) {
  __shared__ double red[2 * 4];
  if (op != FIN_STORE && st->flag) return;
  double s[2] = {0.0, 0.0};
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    s[0] += partial[i];
    s[1] += partial[kRedBlocks + i];
  }
  block_sum<2>(s, red);
  if (threadIdx.x != 0) return;
  cg_scalar_step(s[0], s[1], op, iter, add_prev, st, ring, ring_slots);
}

// Prologue of a CG run that starts from x = 0 (conjugate_gradients_solver.h:139-160), on the device so that the
// host does not have to wait for |rhs|: st->s0 = rhs.rhs on entry, st->tol_r holds the relative tolerance.
// Sets tol_r = r_tolerance |rhs|, |r| = |rhs|, and raises a flag when there is nothing to iterate on (zero
// right-hand side, converged at the start, or a preconditioner that could not be formed); a raised flag is
// published as "iteration 1" so that the host's pipeline reads it where it expects the first outcome.
__global__ void k_cg_prologue(CgState* __restrict__ st, CgState* __restrict__ ring, int ring_slots,
                              const int* __restrict__ preconditioner_failed,
                              const int* __restrict__ preconditioner_failed2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double norm_rhs = sqrt(st->s0);
  st->norm_r = norm_rhs;
  if ((preconditioner_failed && *preconditioner_failed) || (preconditioner_failed2 && *preconditioner_failed2)) {
    st->flag = CG_FAIL_PRECONDITIONER;
  } else if (norm_rhs == 0.0) {
    st->flag = CG_ZERO_RHS;
  } else {
    st->tol_r = st->tol_r * norm_rhs;
    if (st->min_iter == 0 && norm_rhs <= st->tol_r) st->flag = CG_CONVERGED_AT_START;
  }
  if (st->flag != CG_RUNNING) {
    st->iter = 0;
    CgState* slot = ring + (1 % ring_slots);
    *slot = *st;
    __threadfence_system();
    __hip_atomic_store(&slot->seq, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------- fused reductions
// One-kernel form of partial + final: every workgroup stores its two sums, takes a ticket, and
// the workgroup that draws the last ticket adds the partials up in index order (so the result
// does not depend on which workgroup that is) and performs the scalar step.  Stores, ticket and
// loads are agent-scope atomics, which keeps them coherent across the XCDs' L2s.
__device__ __forceinline__ void dot2_finish(double a0, double a1, const DotTail& t, double* red) {
  __shared__ int is_last;
  double s[2] = {a0, a1};
  block_sum<2>(s, red);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&t.partial[blockIdx.x], s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&t.partial[kRedBlocks + blockIdx.x], s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The two stores above are agent-scope atomics (write-through, performed at the device's coherence
    // point), so they only have to be COMPLETE before the ticket is taken: partials and ticket are different
    // addresses and may sit in different L2 channels, so program order alone does not order them.  gfx9
    // counts stores on vmcnt; an explicit wait does it (a workgroup-scope fence emits no s_waitcnt vmcnt
    // outside tgsplit mode -- checked in the ISA) without the L2 write-back / invalidate an agent-scope
    // release fence adds (~10 us per kernel here).  The reader side is ordered by data dependence: the last
    // workgroup issues its agent-scope loads only after its own ticket atomic has returned.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (ticket == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  double v[2] = {0.0, 0.0};
  for (int i = threadIdx.x; i < int(gridDim.x); i += 256) {
    v[0] += __hip_atomic_load(&t.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[1] += __hip_atomic_load(&t.partial[kRedBlocks + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_sum<2>(v, red);
  if (threadIdx.x != 0) return;
  __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  cg_scalar_step(v[0], v[1], t.op, t.iter, 0, t.st, t.ring, t.ring_slots);
}

// s0 = a.b, s1 = c.d (second pair optional)
__global__ __launch_bounds__(256) void k_dot2_fused(const double* __restrict__ a, const double* __restrict__ b,
                                                    const double* __restrict__ c, const double* __restrict__ d, int64_t n,
                                                    DotTail t) {
  __shared__ double red[8];
  if (t.op != FIN_STORE && t.st->flag) return;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    s0 += a[i] * b[i];
    if (c) s1 += c[i] * d[i];
  }
  dot2_finish(s0, s1, t, red);
}

// z = blockdiag(M) r for 9x9 blocks and s0 = r.z  (preconditioner + rho)
__global__ __launch_bounds__(256) void k_blockdiag9_dot(const double* __restrict__ blocks, const double* __restrict__ r,
                                                        double* __restrict__ z, int64_t n, DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  double s0 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const int64_t blk = i / 9;
    const int row = int(i - blk * 9);
    const double* m = blocks + blk * 81 + row * 9;
    const double* rv = r + blk * 9;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) s += m[k] * rv[k];
    z[i] = s;
    s0 += rv[row] * s;
  }
  dot2_finish(s0, 0.0, t, red);
}

// CGNR's block Jacobi preconditioner and rho in one launch: z = blockdiag(M) r with 3x3 blocks on the first ne entries and
// 9x9 blocks behind them (k_blockdiag_multiply<3>, <9>), s0 = r.z summed exactly as k_dot2_fused(r, z) sums it (same grid,
// same strided loop), so the fusion does not change a bit of the iteration.
__global__ __launch_bounds__(256) void k_cgnr_jacobi_dot(const double* __restrict__ pt_blocks, const double* __restrict__ cam_blocks,
                                                         const double* __restrict__ r, double* __restrict__ z, int64_t ne, int64_t n,
                                                         DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  double s0 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    double s = 0.0;
    if (i < ne) {
      const int64_t blk = i / 3;
      const int row = int(i - blk * 3);
      const double* m = pt_blocks + blk * 9 + row * 3;
      const double* rv = r + blk * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += m[k] * rv[k];
    } else {
      const int64_t j = i - ne, blk = j / 9;
      const int row = int(j - blk * 9);
      const double* m = cam_blocks + blk * 81 + row * 9;
      const double* rv = r + ne + blk * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) s += m[k] * rv[k];
    }
    z[i] = s;
    s0 += r[i] * s;
  }
  dot2_finish(s0, 0.0, t, red);
}

// y[9c+k] = sum of camera c's segment partials (segment order) + d^2 x ; s0 = x.y
// (second half of F't, the LM diagonal and p.q of the implicit Schur product in one launch)
__global__ __launch_bounds__(256) void k_cam_reduce9_dot(const double* __restrict__ partial9,
                                                         const int32_t* __restrict__ cam_seg_start, double* __restrict__ y,
                                                         int64_t n, const double* __restrict__ d, const double* __restrict__ x,
                                                         DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  double s0 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const int c = int(i / 9), k = int(i - int64_t(c) * 9);
    double s = 0.0;
    for (int sg = cam_seg_start[c]; sg < cam_seg_start[c + 1]; ++sg) s += partial9[int64_t(sg) * 9 + k];
    const double xv = x[i];
    if (d) s += d[i] * d[i] * xv;
    y[i] = s;
    s0 += xv * s;
  }
  dot2_finish(s0, 0.0, t, red);
}

// y += d^2 x (d optional) ; s0 = x.y   (after the all-reduce of a sharded product)
__global__ __launch_bounds__(256) void k_d2x_dot(double* __restrict__ y, const double* __restrict__ d, const double* __restrict__ x,
                                                 int64_t n, DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  double s0 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const double xv = x[i];
    double yv = y[i];
    if (d) {
      yv += d[i] * d[i] * xv;
      y[i] = yv;
    }
    s0 += xv * yv;
  }
  dot2_finish(s0, 0.0, t, red);
}

// x += alpha p ; r -= alpha q ; tmp = rhs + r ; s0 = x.tmp, s1 = r.r
__global__ __launch_bounds__(256) void k_update_xr_dot(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                       const double* __restrict__ q, const double* __restrict__ rhs,
                                                       double* __restrict__ tmp, int64_t n, DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  const double alpha = t.st->alpha;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const double xv = x[i] + alpha * p[i];
    const double rv = r[i] - alpha * q[i];
    const double tv = rhs[i] + rv;
    x[i] = xv;
    r[i] = rv;
    tmp[i] = tv;
    s0 += xv * tv;
    s1 += rv * rv;
  }
  dot2_finish(s0, s1, t, red);
}

// r = rhs - ax ; tmp = rhs + r ; s0 = x.tmp, s1 = r.r   (ax and tmp may alias)
__global__ __launch_bounds__(256) void k_residual_dot(const double* __restrict__ rhs, const double* ax, double* __restrict__ r,
                                                      double* tmp, const double* __restrict__ x, int64_t n, DotTail t) {
  __shared__ double red[8];
  if (t.st->flag) return;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const double rv = rhs[i] - ax[i];
    const double tv = rhs[i] + rv;
    r[i] = rv;
    tmp[i] = tv;
    s0 += x[i] * tv;
    s1 += rv * rv;
  }
  dot2_finish(s0, s1, t, red);
}

// ---- small problems: the rest of CG iteration `iter` after the camera-major pass, and the head of iteration iter + 1, in
// one workgroup of 1024 threads (n <= kSmallCgMax = 4 x 1024 entries).  It replaces k_cam_reduce9_dot, k_update_xr_dot,
// k_blockdiag9_dot and k_update_p -- and reproduces their sums BIT FOR BIT: entry i belongs to "virtual" workgroup i / 256,
// wavefront (i % 256) / 64, lane i % 64 of the kernels it replaces; a real wavefront here holds exactly one virtual
// wavefront per pass (1024 is a multiple of 64), so wave_sum gives the same butterfly, the four wavefront sums of a virtual
// workgroup are added in the same order, and the workgroup sums go through the same wavefront-0 butterfly dot2_finish uses.
// Same iterates, same iteration counts; no tickets, no agent-scope atomics, one launch.
struct SmallTail {
  SmallProduct prod;
  const double* blocks;  // inverted 9x9 blocks of the preconditioner
  double *x, *r, *p, *q;
  const double* rhs;
  double* tmp;
  int n, iter;
  CgState* st;
  CgState* ring;
  int ring_slots;
  long long* dbg;  // CX_SMALL_TAIL_DEBUG: 8 timestamps (s_memtime) of thread 0 at the phase boundaries, else NULL
};
#define CX_TAIL_STAMP(slot) do { if (a.dbg && tid == 0) a.dbg[slot] = (long long)__builtin_readcyclecounter(); } while (0)
__device__ __forceinline__ double small_total(const double* sc, int lane, int G) {
  const double v = lane < G ? ((sc[4 * lane] + sc[4 * lane + 1]) + sc[4 * lane + 2]) + sc[4 * lane + 3] : 0.0;
  return wave_sum(v);
}
// The CG state lives in LDS for the length of the kernel (thread 0 brings it in and writes it back); what the host has to
// see -- the state after the Q test, and a failure raised by the r.z step -- goes to the pinned ring at the END, so that no
// system-scope fence sits in the middle of the dependency chain.  PASSES = 1 (n <= 1024: Ladybug-49 has 441 entries)
// also requests everything that does not depend on this kernel's own results (p, d, x, r, rhs, the segment range, the
// preconditioner's block row) before the first sum.
template <int PASSES>
__global__ __launch_bounds__(1024) void k_cg_small_tail(SmallTail a) {
  __shared__ double sc0[64], sc1[64];
  __shared__ CgState ls, pub_q, pub_rho;
  __shared__ int publish_rho, ran_q;
  __shared__ double rl[PASSES * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    ls = *a.st;
    publish_rho = 0;
    ran_q = 0;
  }
  const int n = a.n, G = (n + 255) / 256;
  CX_TAIL_STAMP(0);
  constexpr bool kPrefetch = PASSES == 1;
  double pv[PASSES], qv[PASSES], dv[PASSES], xv0[PASSES], rv0[PASSES], rhsv[PASSES], mrow[kPrefetch ? 9 : 1];
  int sg0[PASSES], sg1[PASSES];
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    const int i = k * 1024 + tid;
    pv[k] = qv[k] = dv[k] = xv0[k] = rv0[k] = rhsv[k] = 0.0;
    sg0[k] = sg1[k] = 0;
    if (i < n) {
      const int c = i / 9;
      sg0[k] = a.prod.cam_seg_start[c];
      sg1[k] = a.prod.cam_seg_start[c + 1];
      pv[k] = a.p[i];
      if (a.prod.d) dv[k] = a.prod.d[i];
      xv0[k] = a.x[i];
      rv0[k] = a.r[i];
      rhsv[k] = a.rhs[i];
      if (kPrefetch) {
        const double* m = a.blocks + int64_t(c) * 81 + (i - c * 9) * 9;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) mrow[kk] = m[kk];
      }
    }
  }
  __syncthreads();
  CX_TAIL_STAMP(1);
  if (ls.flag) return;  // (an iteration enqueued ahead of the host's look at the ring: nothing to do, nothing to publish)
  // (1) q = sum of the camera's segment partials + d^2 p ; p.q  (k_cam_reduce9_dot).  The segment sums of a thread's PASSES
  // entries advance together: PASSES independent loads in flight per step instead of one (each sum keeps its own order).
  double seg_sum[PASSES];
  {
    const double* sp[PASSES];
    int len[PASSES], maxlen = 0;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int i = k * 1024 + tid;
      seg_sum[k] = 0.0;
      len[k] = (i < n) ? sg1[k] - sg0[k] : 0;
      sp[k] = a.prod.partial9 + int64_t(sg0[k]) * 9 + (i < n ? i - (i / 9) * 9 : 0);
      maxlen = max(maxlen, len[k]);
    }
    if (PASSES == 1) {
      for (int j = 0; j < len[0]; ++j) seg_sum[0] += sp[0][int64_t(j) * 9];
    } else {
      // kChunk segments of every pass are requested together (clamped addresses, the additions stay predicated and in
      // order): a partial sum written by another XCD's workgroup is a trip to memory, 1-2 us, and a camera of a mid-size
      // problem has a dozen of them
      constexpr int kChunk = 6;
      for (int j0 = 0; j0 < maxlen; j0 += kChunk) {
        double v[PASSES][kChunk];
#pragma unroll
        for (int k = 0; k < PASSES; ++k)
#pragma unroll
          for (int u = 0; u < kChunk; ++u) v[k][u] = sp[k][int64_t(max(min(j0 + u, len[k] - 1), 0)) * 9];
#pragma unroll
        for (int k = 0; k < PASSES; ++k)
#pragma unroll
          for (int u = 0; u < kChunk; ++u)
            if (j0 + u < len[k]) seg_sum[k] += v[k][u];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    const int i = k * 1024 + tid;
    double s0 = 0.0;
    if (i < n) {
      double s = seg_sum[k];
      const double xv = pv[k];
      if (a.prod.d) s += dv[k] * dv[k] * xv;
      a.q[i] = s;
      qv[k] = s;
      s0 += xv * s;
    }
    const double ws = wave_sum(s0);
    if (lane == 0) sc0[k * 16 + wave] = ws;
  }
  __syncthreads();
  CX_TAIL_STAMP(2);
  if (wave == 0) {
    const double tot = small_total(sc0, lane, G);
    if (lane == 0) cg_scalar_step(tot, 0.0, FIN_PQ, a.iter, 0, &ls, nullptr, 0);
  }
  __syncthreads();
  CX_TAIL_STAMP(3);
  if (ls.flag == 0) {
    const double alpha = ls.alpha;
    // (2) x += alpha p ; r -= alpha q ; tmp = rhs + r ; x.tmp, r.r  (k_update_xr_dot)
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int i = k * 1024 + tid;
      double s0 = 0.0, s1 = 0.0;
      if (i < n) {
        const double xv = xv0[k] + alpha * pv[k];
        const double rv = rv0[k] - alpha * qv[k];
        const double tv = rhsv[k] + rv;
        a.x[i] = xv;
        a.r[i] = rv;
        a.tmp[i] = tv;
        rl[i] = rv;
        s0 += xv * tv;
        s1 += rv * rv;
      }
      const double w0 = wave_sum(s0), w1 = wave_sum(s1);
      if (lane == 0) {
        sc0[k * 16 + wave] = w0;
        sc1[k * 16 + wave] = w1;
      }
    }
    __syncthreads();
    if (wave == 0) {
      const double t0 = small_total(sc0, lane, G), t1 = small_total(sc1, lane, G);
      if (lane == 0) {
        cg_scalar_step(t0, t1, FIN_Q, a.iter, 0, &ls, nullptr, 0);
        pub_q = ls;  // what the ring gets for this iteration, whatever the head below does to the state
        ran_q = 1;
      }
    }
    __syncthreads();
    CX_TAIL_STAMP(4);
    if (ls.flag == 0) {
      // (3) head of iteration iter + 1: z = blockdiag(M) r ; r.z  (k_blockdiag9_dot).  All block rows of the thread's
      // entries are requested before the first product (PASSES > 1; one pass had them from the top of the kernel).
      double zv[PASSES];
      double mr[kPrefetch ? 1 : PASSES][9];
      if (!kPrefetch) {
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
          const int i = min(k * 1024 + tid, n - 1);
          const int blk = i / 9, row = i - blk * 9;
          const double* m = a.blocks + int64_t(blk) * 81 + row * 9;
#pragma unroll
          for (int kk = 0; kk < 9; ++kk) mr[k][kk] = m[kk];
        }
      }
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const int i = k * 1024 + tid;
        double s0 = 0.0;
        zv[k] = 0.0;
        if (i < n) {
          const int blk = i / 9, row = i - blk * 9;
          const double* rv = rl + blk * 9;
          double s = 0.0;
#pragma unroll
          for (int kk = 0; kk < 9; ++kk) s += (kPrefetch ? mrow[kk] : mr[k][kk]) * rv[kk];
          zv[k] = s;
          s0 += rv[row] * s;
        }
        const double ws = wave_sum(s0);
        if (lane == 0) sc0[k * 16 + wave] = ws;
      }
      __syncthreads();
      if (wave == 0) {
        const double tot = small_total(sc0, lane, G);
        if (lane == 0) {
          cg_scalar_step(tot, 0.0, FIN_RHO, a.iter + 1, 0, &ls, nullptr, 0);
          if (ls.flag != 0) {  // a failure of the r.z step is published at once, as iteration iter + 1 (cg_scalar_step)
            pub_rho = ls;
            publish_rho = 1;
          }
        }
      }
      __syncthreads();
      CX_TAIL_STAMP(5);
      if (ls.flag == 0) {
        const double beta = ls.beta;
        // (4) p = z + beta p  (k_update_p, iteration >= 2)
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
          const int i = k * 1024 + tid;
          if (i < n) a.p[i] = zv[k] + beta * pv[k];
        }
      }
    }
  }
  // state back to global memory (the next kernels on the stream read it), then the ring -- what the general path's
  // cg_scalar_step publishes, in its order: the state after the Q test (or, when the p.q step ended the run, that state) as
  // iteration iter, a failure of the r.z step as iteration iter + 1
  CX_TAIL_STAMP(6);
  if (tid != 0) return;
  *a.st = ls;
  auto publish = [&](const CgState& v, int seq) {
    CgState* slot = a.ring + (seq % a.ring_slots);
    *slot = v;
    __threadfence_system();
    __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (ran_q) publish(pub_q, a.iter);
  else if (ls.flag != 0) publish(ls, a.iter);
  if (publish_rho) publish(pub_rho, a.iter + 1);
  CX_TAIL_STAMP(7);
}

// ---- small problems, the set-up: everything between the two passes over J of ImplicitSchurComplement::Init and the first
// product of the CG run, in one workgroup (n <= kSmallSetupMax entries, so one entry per thread).  It replaces
// k_cam_diag_reduce, k_sum_segments9, k_block9_add_diag_invert, the zeroing of x, the upload of the initial state,
// k_dot2_fused(rhs, rhs), k_cg_prologue, the copy r = rhs, k_blockdiag9_dot and k_update_p of iteration 1 -- with their
// arithmetic, entry by entry and sum by sum (the virtual-workgroup bookkeeping of k_cg_small_tail; rhs.rhs is the one
// reduction whose kernel walks a 256-thread workgroup over 1024 entries, so threads 0..255 replay that loop).
__global__ __launch_bounds__(1024) void k_cg_small_setup(SmallSetup su, CgState init, double* __restrict__ rhs,
                                                         double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                         int n, int* __restrict__ not_pd, const int* __restrict__ failed2,
                                                         CgState* __restrict__ st, CgState* __restrict__ ring, int ring_slots) {
  constexpr int kMaxC = (kSmallSetupMax + 8) / 9;
  __shared__ double slab[kMaxC * 81];  // per camera: the block's upper triangle, then its Cholesky factor in place
  __shared__ double rl[1024];
  __shared__ double sc0[64];
  __shared__ CgState ls;
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = su.C, G = (n + 255) / 256;
  if (tid == 0) {
    ls = init;
    bad = 0;
  }
  // (1) the cameras' blocks: sum of the segment partials  (k_cam_diag_reduce)
  for (int idx = tid; idx < C * 45; idx += 1024) {
    const int c = idx / 45, k = idx - c * 45;
    double s = 0.0;
    for (int sg = su.cam_seg_start[c]; sg < su.cam_seg_start[c + 1]; ++sg) s += su.partial45[int64_t(sg) * 45 + k];
    int a = 0, rem = k;
    while (rem >= 9 - a) { rem -= 9 - a; ++a; }
    slab[c * 81 + a * 9 + a + rem] = s;
  }
  // (2) rhs = sum of the segment partials (k_sum_segments9) ; r = rhs ; x = 0
  {
    double s = 0.0;
    if (tid < n) {
      const int c = tid / 9, k = tid - c * 9;
      for (int sg = su.cam_seg_start[c]; sg < su.cam_seg_start[c + 1]; ++sg) s += su.partial9[int64_t(sg) * 9 + k];
      rhs[tid] = s;
      r[tid] = s;
      x[tid] = 0.0;
    }
    rl[tid] = s;
  }
  __syncthreads();
  // (3a) rhs.rhs  (k_dot2_fused on one workgroup of 256: thread t adds entries t, t + 256, ...)
  if (tid < 256) {
    double s0 = 0.0;
    for (int i = tid; i < n; i += 256) s0 += rl[i] * rl[i];
    const double ws = wave_sum(s0);
    if (lane == 0) sc0[wave] = ws;
  }
  // (3b) blocks <- (blocks + diag(D^2))^-1 through LLT (k_block9_add_diag_invert; the factor overwrites the triangle it was
  // computed from).  Nine threads per camera, counted from the far end of the workgroup: the first of them factors, then
  // each solves U'U x = e_col for its own column -- the arithmetic of every entry is that of the one-thread-per-camera
  // kernel, only the nine columns no longer wait for each other.
  const int inv_t = 1023 - tid, inv_c = inv_t / 9, inv_col = inv_t - inv_c * 9;
  if (inv_c < C && inv_col == 0) {
    double* U = slab + inv_c * 81;
    bool ok = true;
    for (int j = 0; j < 9; ++j) {
      for (int i = 0; i <= j; ++i) {
        double s = U[i * 9 + j];
        if (i == j && su.Df) s += su.Df[9 * int64_t(inv_c) + j] * su.Df[9 * int64_t(inv_c) + j];
        for (int k = 0; k < i; ++k) s -= U[k * 9 + i] * U[k * 9 + j];
        if (i == j) {
          if (!(s > 0.0)) ok = false;
          U[i * 9 + i] = sqrt(s);
        } else {
          U[i * 9 + j] = s / U[i * 9 + i];
        }
      }
    }
    if (!ok) {
      *not_pd = 1;
      bad = 1;
    }
  }
  __syncthreads();
  if (inv_c < C) {
    const double* U = slab + inv_c * 81;
    double* X = su.blocks + int64_t(inv_c) * 81;
    double y[9], xc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double s = (i == inv_col) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s -= U[k * 9 + i] * y[k];
      y[i] = s / U[i * 9 + i];
    }
#pragma unroll
    for (int i = 8; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int k = i + 1; k < 9; ++k) s -= U[i * 9 + k] * xc[k];
      xc[i] = s / U[i * 9 + i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) X[i * 9 + inv_col] = xc[i];
  }
  __syncthreads();
  // (3c) the prologue  (k_cg_prologue)
  if (tid == 0) {
    const double tot = ((sc0[0] + sc0[1]) + sc0[2]) + sc0[3];
    ls.s0 = tot;
    ls.s1 = 0.0;
    const double norm_rhs = sqrt(tot);
    ls.norm_r = norm_rhs;
    if (bad || *not_pd || (failed2 && *failed2)) {
      ls.flag = CG_FAIL_PRECONDITIONER;
    } else if (norm_rhs == 0.0) {
      ls.flag = CG_ZERO_RHS;
    } else {
      ls.tol_r = ls.tol_r * norm_rhs;
      if (ls.min_iter == 0 && norm_rhs <= ls.tol_r) ls.flag = CG_CONVERGED_AT_START;
    }
    if (ls.flag != CG_RUNNING) ls.iter = 0;
  }
  __syncthreads();
  const bool stopped_at_start = ls.flag != CG_RUNNING;
  int published_as = 0;
  if (!stopped_at_start) {
    // (4) head of iteration 1: z = blockdiag(M) r ; r.z  (k_blockdiag9_dot) ; p = z  (k_update_p)
    double zv = 0.0, s0 = 0.0;
    if (tid < n) {
      const int blk = tid / 9, row = tid - blk * 9;
      const double* m = su.blocks + int64_t(blk) * 81 + row * 9;
      const double* rv = rl + blk * 9;
      double s = 0.0;
#pragma unroll
      for (int kk = 0; kk < 9; ++kk) s += m[kk] * rv[kk];
      zv = s;
      s0 += rv[row] * s;
    }
    const double ws = wave_sum(s0);
    if (lane == 0) sc0[wave] = ws;
    __syncthreads();
    if (wave == 0) {
      const double tot = small_total(sc0, lane, G);
      if (lane == 0) cg_scalar_step(tot, 0.0, FIN_RHO, 1, 0, &ls, nullptr, 0);
    }
    __syncthreads();
    if (ls.flag == 0) {
      if (tid < n) p[tid] = zv;
    } else {
      published_as = 1;  // a failure of the r.z step: iteration 1 (cg_scalar_step)
    }
  } else {
    published_as = 1;  // the prologue publishes a run that ended before it began where the host expects the first outcome
  }
  if (tid != 0) return;
  *st = ls;
  if (published_as) {
    CgState* slot = ring + (published_as % ring_slots);
    *slot = ls;
    __threadfence_system();
    __hip_atomic_store(&slot->seq, published_as, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// p = z (first iteration) or z + beta p
__global__ void k_update_p(double* __restrict__ p, const double* __restrict__ z, int64_t n, int iter,
                           const CgState* __restrict__ st) {
  if (st->flag) return;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  p[i] = (iter == 1) ? z[i] : z[i] + st->beta * p[i];
}

// x += alpha p ; r -= alpha q (unless the residual is recomputed) ; tmp = rhs + r
__global__ void k_update_xr(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                            const double* __restrict__ q, const double* __restrict__ rhs,
                            double* __restrict__ tmp, int64_t n, int update_r, const CgState* __restrict__ st) {
  if (st->flag) return;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double alpha = st->alpha;
  x[i] = x[i] + alpha * p[i];
  if (update_r) {
    const double rv = r[i] - alpha * q[i];
    r[i] = rv;
    tmp[i] = rhs[i] + rv;
  }
}

// r = rhs - ax ; tmp = rhs + r
__global__ void k_residual(const double* __restrict__ rhs, const double* __restrict__ ax, double* __restrict__ r,
                           double* __restrict__ tmp, int64_t n, const CgState* __restrict__ st) {
  if (st && st->flag) return;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double rv = rhs[i] - ax[i];
  r[i] = rv;
  tmp[i] = rhs[i] + rv;
}

__global__ void k_add_d2x(double* __restrict__ y, const double* __restrict__ d, const double* __restrict__ x, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] += d[i] * d[i] * x[i];
}

// z = blockdiag(M) r for dense bs x bs row-major blocks (bs = 3 or 9)
template <int BS>
__global__ void k_blockdiag_multiply(const double* __restrict__ blocks, const double* __restrict__ r,
                                     double* __restrict__ z, int64_t nblocks) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nblocks * BS) return;
  const int64_t blk = i / BS;
  const int row = int(i - blk * BS);
  const double* m = blocks + blk * (BS * BS) + row * BS;
  const double* rv = r + blk * BS;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < BS; ++k) s += m[k] * rv[k];
  z[i] = s;
}

// CGNR set-up, the point side in ONE pass over E (chunk-aligned tiles, the cells staged through LDS like k_left_e_239):
// the 3x3 point blocks of J'J + D^2 inverted through LLT (BlockSparseJacobiPreconditioner, block_jacobi_preconditioner.cc:59-115;
// round 3's k_point_jacobi: one thread walking a point's rows) and the point part of J'b (k_left_e_239), both summed over a
// point's rows in row order as those kernels do.
__global__ __launch_bounds__(kBlock) void k_point_jacobi_etb(const double* __restrict__ E, const int32_t* __restrict__ tile_row,
                                                             const int32_t* __restrict__ tile_pt, const int32_t* __restrict__ pt_start,
                                                             const double* __restrict__ D, const double* __restrict__ b,
                                                             double* __restrict__ blocks, double* __restrict__ ye,
                                                             int* __restrict__ not_pd) {
  __shared__ double lds[kBlock * 9];  // staging (6 per row), then 9 products per row
  __shared__ double red[9 * 4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  const int p0 = tile_pt[t], p1 = tile_pt[t + 1];
  double m[9], etb[3];
  bool have = false;
  int p = p0;
  if (r1 - r0 <= kBlock) {
    const int nvalid = r1 - r0;
    double e[6];
    stage_cells<6>(E + 6 * int64_t(r0), nvalid, lds, e);
    __syncthreads();  // (the staging area is reused for the products)
    if (tid < nvalid) {
      const double2 bv = reinterpret_cast<const double2*>(b)[r0 + tid];
      double* w = lds + tid * 9;
      w[0] = e[0] * e[0] + e[3] * e[3];
      w[1] = e[0] * e[1] + e[3] * e[4];
      w[2] = e[0] * e[2] + e[3] * e[5];
      w[3] = e[1] * e[1] + e[4] * e[4];
      w[4] = e[1] * e[2] + e[4] * e[5];
      w[5] = e[2] * e[2] + e[5] * e[5];
      w[6] = e[0] * bv.x + e[3] * bv.y;
      w[7] = e[1] * bv.x + e[4] * bv.y;
      w[8] = e[2] * bv.x + e[5] * bv.y;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      p = p0 + tid;
      have = true;
      double s[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) s[k] = 0.0;
      for (int j = pt_start[p] - r0; j < pt_start[p + 1] - r0; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] += lds[j * 9 + k];
      }
      m[0] = s[0]; m[1] = s[1]; m[2] = s[2]; m[4] = s[3]; m[5] = s[4]; m[8] = s[5];
      etb[0] = s[6]; etb[1] = s[7]; etb[2] = s[8];
    }
  } else {
    // one point whose chunk is longer than a tile: strided loop + block reduction
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const double* e = E + 6 * int64_t(r);
      const double2 bv = reinterpret_cast<const double2*>(b)[r];
      s[0] += e[0] * e[0] + e[3] * e[3];
      s[1] += e[0] * e[1] + e[3] * e[4];
      s[2] += e[0] * e[2] + e[3] * e[5];
      s[3] += e[1] * e[1] + e[4] * e[4];
      s[4] += e[1] * e[2] + e[4] * e[5];
      s[5] += e[2] * e[2] + e[5] * e[5];
      s[6] += e[0] * bv.x + e[3] * bv.y;
      s[7] += e[1] * bv.x + e[4] * bv.y;
      s[8] += e[2] * bv.x + e[5] * bv.y;
    }
    block_sum<9>(s, red);
    if (tid == 0) {
      have = true;
      m[0] = s[0]; m[1] = s[1]; m[2] = s[2]; m[4] = s[3]; m[5] = s[4]; m[8] = s[5];
      etb[0] = s[6]; etb[1] = s[7]; etb[2] = s[8];
    }
  }
  if (!have) return;
  if (D) {
    const double* d = D + 3 * int64_t(p);
    m[0] += d[0] * d[0]; m[4] += d[1] * d[1]; m[8] += d[2] * d[2];
  }
  m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
  double inv[9];
  bool ok;
  inv3_llt(m, inv, ok);
  if (!ok) *not_pd = 1;
#pragma unroll
  for (int k = 0; k < 9; ++k) blocks[9 * int64_t(p) + k] = inv[k];
  ye[3 * int64_t(p)] = etb[0];
  ye[3 * int64_t(p) + 1] = etb[1];
  ye[3 * int64_t(p) + 2] = etb[2];
}

// ------------------------------------------------------------------ operators

namespace {

bool SmallCgAllowed() {
  static const bool allowed = std::getenv("CX_NO_SMALL_CG") == nullptr;
  return allowed;
}
// k_cg_small_setup: one rank, at most kSmallSetupMax reduced unknowns (CX_NO_SMALL_SETUP=1: the separate launches, for A/B
// runs and the bit-equality test)
bool SmallSetupEligible(const cx_context* ctx, int64_t n) {
  static const bool allowed = std::getenv("CX_NO_SMALL_SETUP") == nullptr;
  return allowed && SmallCgAllowed() && ctx->nranks <= 1 && n > 0 && n <= kSmallSetupMax;
}

struct CgDriver {
  cx_solver* S;
  cx_context* ctx;
  hipStream_t st;
  int64_t n;
  // sharded CGNR: entries [shared0, n) are replicated over the ranks
  int64_t shared0;
  static constexpr int kRingSlots = 16;
  // set by solvers whose set-up raises a device flag when the preconditioner cannot be formed: the prologue then
  // ends the run before the first iteration (no host check, no synchronisation, in between)
  const int* preconditioner_failed = nullptr;
  const int* preconditioner_failed2 = nullptr;  // a second, independently raised flag (factorisation of a visibility preconditioner)
  // small problems: the caller stopped its set-up at the segment partial sums; the run's first launch finishes it
  // (k_cg_small_setup).  Only with SmallSetupEligible(); blocks, rhs and x are then OUTPUTS of that launch.
  const SmallSetup* small_setup = nullptr;

  // vectors replicated on every rank (or a single rank): dot products need no exchange
  bool fused() const { return !(ctx->nranks > 1 && shared0 < n); }
  DotTail tail(int op, int iter, CgState* ds) const { return DotTail{S->partial.p, S->ticket.p, ds, S->ring_d, kRingSlots, op, iter}; }
  int vec_grid() const { return int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 255) / 256))); }

  int dot2(const double* a, const double* b, const double* c, const double* d, int op, int iter, CgState* dst) {
    const int nb = int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 1023) / 1024)));
    CgState* ring = S->ring_d;
    if (fused()) {
      hipLaunchKernelGGL(k_dot2_fused, dim3(nb), dim3(256), 0, st, a, b, c, d, n, tail(op, iter, dst));
      CX_HIP(hipGetLastError());
      return CX_OK;
    }
    if (ctx->nranks > 1 && shared0 < n) {
      // rank-local part: sum, all-reduce the two scalars on the device; replicated part added once
      hipLaunchKernelGGL(k_dot2_partial, dim3(nb), dim3(256), 0, st, a, b, c, d, shared0, S->partial.p, (const CgState*)dst);
      hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(256), 0, st, (const double*)S->partial.p, nb, int(FIN_STORE), iter, 0, dst, (CgState*)nullptr, 0);
      CX_TRY(cx_allreduce_device(ctx, &dst->s0, 2));
      hipLaunchKernelGGL(k_dot2_partial, dim3(nb), dim3(256), 0, st, a + shared0, b + shared0,
                         c ? c + shared0 : nullptr, d ? d + shared0 : nullptr, n - shared0, S->partial.p, (const CgState*)dst);
      hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(256), 0, st, (const double*)S->partial.p, nb, op, iter, 1, dst, ring, kRingSlots);
    } else {
      hipLaunchKernelGGL(k_dot2_partial, dim3(nb), dim3(256), 0, st, a, b, c, d, n, S->partial.p,
                         op == FIN_STORE ? (const CgState*)nullptr : (const CgState*)dst);
      hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(256), 0, st, (const double*)S->partial.p, nb, op, iter, 0, dst, ring, kRingSlots);
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }

  int read_state(CgState* h) {
    CX_TRY(cx_read_back(S->ctx, h, S->state.p, sizeof(CgState), st));
    CX_TRY(cx_stream_sync(S->ctx, st));
    return CX_OK;
  }

  // CG iteration `iter` (conjugate_gradients_solver.h:162-301) in two parts.  The head (z = M^-1 r,
  // rho, beta, p) is cheap and is enqueued speculatively; the tail holds the operator application.
  int enqueue_head(int iter, LinOp& pre, double* p, double* r, double* z, CgState* ds) {
    bool done = false;
    if (fused()) CX_TRY(pre.apply_dot(r, z, tail(FIN_RHO, iter, ds), &done));
    if (!done) {
      CX_TRY(pre.apply(r, z));
      CX_TRY(dot2(r, z, nullptr, nullptr, FIN_RHO, iter, ds));
    }
    hipLaunchKernelGGL(k_update_p, dim3(grid_for(n, 256)), dim3(256), 0, st, p, (const double*)z, n, iter, (const CgState*)ds);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int enqueue_tail(int iter, LinOp& lhs, const double* rhs, double* x, double* p, double* r, double* z, double* tmp,
                   CgState* ds) {
    const cx_solver_options& o = S->opt;
    const int g = grid_for(n, 256);
    double* q = z;  // q aliases z, as in the reference
    bool done = false;
    if (fused()) CX_TRY(lhs.apply_dot(p, q, tail(FIN_PQ, iter, ds), &done));
    if (!done) {
      CX_TRY(lhs.apply(p, q));
      CX_TRY(dot2(p, q, nullptr, nullptr, FIN_PQ, iter, ds));
    }
    const bool reset = (iter % o.residual_reset_period) == 0;
    if (fused()) {
      if (!reset) {
        hipLaunchKernelGGL(k_update_xr_dot, dim3(vec_grid()), dim3(256), 0, st, x, r, (const double*)p, (const double*)q, rhs, tmp,
                           n, tail(FIN_Q, iter, ds));
      } else {
        hipLaunchKernelGGL(k_update_xr, dim3(g), dim3(256), 0, st, x, r, (const double*)p, (const double*)q, rhs, tmp, n, 0,
                           (const CgState*)ds);
        CX_TRY(lhs.apply(x, tmp));
        hipLaunchKernelGGL(k_residual_dot, dim3(vec_grid()), dim3(256), 0, st, rhs, (const double*)tmp, r, tmp, (const double*)x, n,
                           tail(FIN_Q, iter, ds));
      }
      CX_HIP(hipGetLastError());
      return CX_OK;
    }
    hipLaunchKernelGGL(k_update_xr, dim3(g), dim3(256), 0, st, x, r, (const double*)p, (const double*)q, rhs, tmp, n,
                       reset ? 0 : 1, (const CgState*)ds);
    if (reset) {
      CX_TRY(lhs.apply(x, tmp));
      hipLaunchKernelGGL(k_residual, dim3(g), dim3(256), 0, st, rhs, (const double*)tmp, r, tmp, n, (const CgState*)ds);
    }
    CX_TRY(dot2(x, tmp, r, r, FIN_Q, iter, ds));
    return CX_OK;
  }

  // Wait until the device has published iteration `iter` in the host-pinned ring.
  int wait_published(int iter, CgState* out) {
    volatile CgState* slot = S->ring_h + (iter % kRingSlots);
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != iter) {
      if ((++spins & 0x3fff) == 0) {
        if (hipStreamQuery(st) == hipSuccess && __atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != iter) {
          // stream drained without the slot being written: a kernel of the iteration failed
          cx_set_error("CG iteration %d was never published (stream idle): %s", iter, hipGetErrorString(hipGetLastError()));
          return CX_ERR_HIP;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > cx_comm_timeout(S->ctx)) {
          // on a sharded context this is a peer that never reached the iteration's exchange step: end the collective
          // this rank's stream is stuck in (cx_comm_abort) instead of leaving it there
          const bool sharded = S->ctx->comm != nullptr && S->ctx->nranks > 1;
          if (sharded) cx_comm_abort(S->ctx);
          cx_set_error("timed out waiting for CG iteration %d%s", iter, sharded ? " (a peer rank never reached the exchange step; communicator aborted)" : "");
          return sharded ? CX_ERR_COMM : CX_ERR_HIP;
        }
      }
    }
    std::memcpy(out, const_cast<CgState*>(slot), sizeof(CgState));
    return CX_OK;
  }

  // conjugate_gradients_solver.h:107-305.  x holds the initial guess (zero_initial tells that
  // it is all zeros so that A x can be skipped).  The loop is software-pipelined: iteration
  // i+1 is enqueued before the host looks at the outcome of iteration i; kernels of an
  // iteration that turns out not to be needed see the termination flag and do nothing.
  int run(LinOp& lhs, LinOp& pre, const double* rhs, double* x, bool zero_initial, double r_tol, double q_tol,
          cx_summary* summary) {
    const cx_solver_options& o = S->opt;
    CX_TRY(S->v_p.alloc(n));
    CX_TRY(S->v_r.alloc(n));
    CX_TRY(S->v_z.alloc(n));
    CX_TRY(S->v_tmp.alloc(n));
    CX_TRY(S->partial.alloc(2 * kRedBlocks));
    CX_TRY(S->state.alloc(1));
    if (!S->ticket.p) {
      CX_TRY(S->ticket.alloc(1));
      CX_HIP(hipMemsetAsync(S->ticket.p, 0, sizeof(unsigned), st));
    }
    if (!S->ring_h) {
      CX_HIP(hipHostMalloc(reinterpret_cast<void**>(&S->ring_h), (kRingSlots + 1) * sizeof(CgState),
                           hipHostMallocMapped | hipHostMallocCoherent));
      CX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&S->ring_d), S->ring_h, 0));
    }
    for (int i = 0; i < kRingSlots; ++i) S->ring_h[i].seq = -1;
    double *p = S->v_p.p, *r = S->v_r.p, *z = S->v_z.p, *tmp = S->v_tmp.p;
    CgState* ds = S->state.p;
    const int g = grid_for(n, 256);

    summary->termination_type = CX_NO_CONVERGENCE;
    std::snprintf(summary->message, sizeof(summary->message), "Maximum number of iterations reached.");
    summary->num_iterations = 0;

    CgState h{};
    CgState* init = S->ring_h + kRingSlots;  // pinned staging slot of the initial device state
    double tol_r = 0.0;
    const bool small_loop = SmallCgAllowed() && ctx->nranks <= 1 && fused() && n <= kSmallCgMax && pre.block9_inverse() != nullptr;  // (one rank: a speculative iteration must not enter a collective)
    if (small_setup && !(zero_initial && small_loop && n <= kSmallSetupMax && pre.block9_inverse() == small_setup->blocks)) {
      cx_set_error("internal: fused small set-up requested for a run that cannot take it");
      return CX_ERR_INVALID_ARGUMENT;
    }
    if (small_setup) {
      CgState first{};
      first.rho = 1.0;
      first.Q0 = 0.0;
      first.tol_r = r_tol;  // relative; the prologue inside the kernel scales it by |rhs|
      first.q_tol = q_tol;
      first.min_iter = o.min_num_iterations;
      first.max_iter = o.max_num_iterations;
      first.flag = CG_RUNNING;
      first.seq = -1;
      hipLaunchKernelGGL(k_cg_small_setup, dim3(1), dim3(1024), 0, st, *small_setup, first, const_cast<double*>(rhs), x, r, p, int(n),
                         const_cast<int*>(preconditioner_failed), preconditioner_failed2, ds, S->ring_d, kRingSlots);
      CX_HIP(hipGetLastError());
    } else if (zero_initial) {
      // x = 0: r = rhs, Q0 = 0; |rhs|, the absolute tolerance and the "nothing to do" cases are the device
      // prologue's business -- the host goes straight on to enqueue the first iteration
      *init = CgState{};
      init->rho = 1.0;
      init->Q0 = 0.0;
      init->tol_r = r_tol;  // relative; k_cg_prologue scales it by |rhs|
      init->q_tol = q_tol;
      init->min_iter = o.min_num_iterations;
      init->max_iter = o.max_num_iterations;
      init->flag = CG_RUNNING;
      init->seq = -1;
      CX_HIP(hipMemcpyAsync(ds, init, sizeof(CgState), hipMemcpyHostToDevice, st));
      CX_TRY(dot2(rhs, rhs, nullptr, nullptr, FIN_STORE, 0, ds));
      hipLaunchKernelGGL(k_cg_prologue, dim3(1), dim3(1), 0, st, ds, S->ring_d, kRingSlots, preconditioner_failed,
                         preconditioner_failed2);
      CX_HIP(hipMemcpyAsync(r, rhs, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
      CX_HIP(hipMemcpyAsync(ds, &h, sizeof(h), hipMemcpyHostToDevice, st));
      CX_TRY(dot2(rhs, rhs, nullptr, nullptr, FIN_STORE, 0, ds));
      int precond_flag = 0, precond_flag2 = 0;
      if (preconditioner_failed)
        CX_TRY(cx_read_back(S->ctx, &precond_flag, preconditioner_failed, sizeof(int), st));
      if (preconditioner_failed2)
        CX_TRY(cx_read_back(S->ctx, &precond_flag2, preconditioner_failed2, sizeof(int), st));
      CX_TRY(read_state(&h));  // this path waits for the device anyway
      if (precond_flag || precond_flag2) {
        summary->termination_type = CX_FAILURE;
        std::snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
        return CX_OK;
      }
      const double norm_rhs = std::sqrt(h.s0);
      if (norm_rhs == 0.0) {
        CX_HIP(hipMemsetAsync(x, 0, n * sizeof(double), st));
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Convergence. |b| = 0.");
        return CX_OK;
      }
      tol_r = r_tol * norm_rhs;
      CX_TRY(lhs.apply(x, tmp));
      hipLaunchKernelGGL(k_residual, dim3(g), dim3(256), 0, st, rhs, (const double*)tmp, r, tmp, n, (const CgState*)nullptr);
      // tmp = rhs + r now;  s0 = x.tmp, s1 = r.r
      CX_TRY(dot2(x, tmp, r, r, FIN_STORE, 0, ds));
      CX_TRY(read_state(&h));
      const double norm_r = std::sqrt(h.s1);
      const double Q0 = -h.s0;  // Q0 = -x.(rhs + r)   (conjugate_gradients_solver.h:155-158)
      if (o.min_num_iterations == 0 && norm_r <= tol_r) {
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Convergence. |r| = %e <= %e.", norm_r, tol_r);
        return CX_OK;
      }
      *init = CgState{};
      init->rho = 1.0;
      init->Q0 = Q0;
      init->tol_r = tol_r;
      init->q_tol = q_tol;
      init->min_iter = o.min_num_iterations;
      init->max_iter = o.max_num_iterations;
      init->flag = CG_RUNNING;
      init->seq = -1;
      CX_HIP(hipMemcpyAsync(ds, init, sizeof(CgState), hipMemcpyHostToDevice, st));
    }

    // Software pipeline: while the device runs the tail of iteration i the host has already
    // enqueued the head of iteration i+1; it then learns the outcome of iteration i from the
    // pinned ring and either enqueues the tail of i+1 or stops (the speculative head sees the
    // termination flag and does nothing).  No stream synchronisation inside the loop.
    // Small problems: see k_cg_small_tail.  One launch after the operator's passes closes iteration i and opens i + 1;
    // every kernel looks at the termination flag, so a whole iteration is enqueued ahead of the host's look at the ring
    // (CX_NO_SMALL_CG=1: the general path, for A/B runs and the bit-equality test).
    SmallProduct probe;
    if (small_loop) {
      if (!small_setup) CX_TRY(enqueue_head(1, pre, p, r, z, ds));  // (k_cg_small_setup has opened iteration 1)
      bool small_ok = true;
      int last_enqueued = 0;
      for (int iter = 1;; ++iter) {
        const bool reset = (iter % o.residual_reset_period) == 0;
        int rc = CX_OK;
        if (!reset && small_ok) {
          small_ok = lhs.small_partials(p, &probe, &rc);
          CX_TRY(rc);
        }
        if (!reset && small_ok) {
          static long long* dbg = [] {
            long long* p = nullptr;
            if (std::getenv("CX_SMALL_TAIL_DEBUG")) (void)hipHostMalloc(reinterpret_cast<void**>(&p), 8 * sizeof(long long), hipHostMallocMapped);
            return p;
          }();
          SmallTail a{probe, pre.block9_inverse(), x, r, p, z, rhs, tmp, int(n), iter, ds, S->ring_d, kRingSlots, dbg};
          if (dbg && iter == 2) {  // (development aid: the stamps of iteration 1's kernel, printed while iteration 2 is enqueued)
            (void)hipStreamSynchronize(st);
            std::fprintf(stderr, "small tail stamps (cycles since kernel start):");
            for (int q = 1; q < 8; ++q) std::fprintf(stderr, " %lld", dbg[q] - dbg[0]);
            std::fprintf(stderr, "\n");
          }
          if (n <= 1024) hipLaunchKernelGGL(k_cg_small_tail<1>, dim3(1), dim3(1024), 0, st, a);
          else hipLaunchKernelGGL(k_cg_small_tail<4>, dim3(1), dim3(1024), 0, st, a);
          CX_HIP(hipGetLastError());
        } else {  // a residual reset (every residual_reset_period-th iteration), or an operator without the small form
          CX_TRY(enqueue_tail(iter, lhs, rhs, x, p, r, z, tmp, ds));
          CX_TRY(enqueue_head(iter + 1, pre, p, r, z, ds));
        }
        last_enqueued = iter;
        if (iter >= 2) {
          CX_TRY(wait_published(iter - 1, &h));
          if (h.flag != CG_RUNNING) break;
        }
      }
      if (h.flag == CG_RUNNING || last_enqueued < 2) {  // (a first iteration that already ended the run)
        CX_TRY(wait_published(last_enqueued, &h));
      }
    } else {
    CX_TRY(enqueue_head(1, pre, p, r, z, ds));
    CX_TRY(enqueue_tail(1, lhs, rhs, x, p, r, z, tmp, ds));
    for (int iter = 2;; ++iter) {
      CX_TRY(enqueue_head(iter, pre, p, r, z, ds));
      CX_TRY(wait_published(iter - 1, &h));
      if (h.flag != CG_RUNNING) break;
      CX_TRY(enqueue_tail(iter, lhs, rhs, x, p, r, z, tmp, ds));
    }
    }
    // the outcome is in h (read from the ring); what is still queued behind it are speculative launches that see the flag and
    // do nothing, so the caller's next launches simply follow them.  A sharded run waits here: this is where a peer that
    // never reached the exchange step is noticed (cx_stream_sync is bounded on a sharded context).
    if (ctx->nranks > 1) CX_TRY(cx_stream_sync(S->ctx, st));
    summary->num_iterations = h.iter;
    switch (h.flag) {
      case CG_CONVERGED_Q:
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Iteration: %d Convergence: zeta = %e < %e. |r| = %e",
                      summary->num_iterations, h.zeta, q_tol, h.norm_r);
        break;
      case CG_CONVERGED_R:
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Iteration: %d Convergence. |r| = %e <= %e.",
                      summary->num_iterations, h.norm_r, h.tol_r);
        break;
      case CG_FAIL_RHO:
        summary->termination_type = CX_FAILURE;
        std::snprintf(summary->message, sizeof(summary->message), "Numerical failure. rho = r'z = %e.", h.rho);
        break;
      case CG_FAIL_BETA:
        summary->termination_type = CX_FAILURE;
        std::snprintf(summary->message, sizeof(summary->message),
                      "Numerical failure. beta = rho_n / rho_{n-1} = %e, rho_n = %e, rho_{n-1} = %e", h.beta, h.rho, h.last_rho);
        break;
      case CG_INDEFINITE:
        summary->termination_type = CX_NO_CONVERGENCE;
        std::snprintf(summary->message, sizeof(summary->message),
                      "Matrix is indefinite, no more progress can be made. p'q = %e.", h.pq);
        break;
      case CG_FAIL_ALPHA:
        summary->termination_type = CX_FAILURE;
        std::snprintf(summary->message, sizeof(summary->message),
                      "Numerical failure. alpha = rho / pq = %e, rho = %e, pq = %e.", h.alpha, h.rho, h.pq);
        break;
      case CG_ZERO_RHS:
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Convergence. |b| = 0.");
        break;
      case CG_CONVERGED_AT_START:
        summary->termination_type = CX_SUCCESS;
        std::snprintf(summary->message, sizeof(summary->message), "Convergence. |r| = %e <= %e.", h.norm_r, h.tol_r);
        break;
      case CG_FAIL_PRECONDITIONER:
        summary->termination_type = CX_FAILURE;
        std::snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
        break;
      default: break;  // CG_MAX_ITER keeps NO_CONVERGENCE
    }
    summary->residual_norm = h.norm_r;
    return CX_OK;
  }
};

// ------------------------------------------------ static <2,3,9> operator pieces
// S x through the chunk pass and the camera-major pass (ImplicitSchurComplement::
// RightMultiplyAndAccumulate, implicit_schur_complement.cc:106-144)
struct ImplicitSchurOp : LinOp {
  cx_solver* S;
  cx_matrix* A;
  const double* D;
  int64_t size() const override { return 9 * int64_t(A->C); }
  int apply(const double* x, double* y) override {
    cx_context* ctx = A->ctx;
    CX_TRY(S->ktimer.begin(0, ctx->stream));
    CX_TRY(cxs_chunk_pass(A, 0, S->ete_inv.p, x, nullptr, S->v_rows.p));
    CX_TRY(S->ktimer.end(0, ctx->stream));
    CX_TRY(S->ktimer.begin(1, ctx->stream));
    CX_TRY(cxk_ft_multiply(A, S->v_rows.p, y, false));
    CX_TRY(S->ktimer.end(1, ctx->stream));
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, y, size()));
    if (D)
      hipLaunchKernelGGL(k_add_d2x, dim3(grid_for(size(), 256)), dim3(256), 0, ctx->stream, y,
                         D + 3 * int64_t(A->P), x, size());
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  // small problems: the two passes over J only; the camera reduction, D^2 x and x.y happen in k_cg_small_tail
  bool small_partials(const double* x, SmallProduct* out, int* rc) override {
    cx_context* ctx = A->ctx;
    if (ctx->nranks > 1) return false;
    *rc = S->ktimer.begin(0, ctx->stream);
    if (*rc == CX_OK) *rc = cxs_chunk_pass(A, 0, S->ete_inv.p, x, nullptr, S->v_rows.p);
    if (*rc == CX_OK) *rc = S->ktimer.end(0, ctx->stream);
    if (*rc == CX_OK) *rc = S->ktimer.begin(1, ctx->stream);
    if (*rc == CX_OK) *rc = cxk_ft_partials(A, S->v_rows.p);
    if (*rc == CX_OK) *rc = S->ktimer.end(1, ctx->stream);
    out->partial9 = A->d_partials.p;
    out->cam_seg_start = A->d_cam_seg_start.p;
    out->d = D ? D + 3 * int64_t(A->P) : nullptr;
    return true;
  }
  // the same product with the LM diagonal and x.y folded into the camera reduction
  int apply_dot(const double* x, double* y, const DotTail& tail, bool* done) override {
    cx_context* ctx = A->ctx;
    const int64_t n = size();
    const int grid = int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 255) / 256)));
    const double* d = D ? D + 3 * int64_t(A->P) : nullptr;
    CX_TRY(S->ktimer.begin(0, ctx->stream));
    CX_TRY(cxs_chunk_pass(A, 0, S->ete_inv.p, x, nullptr, S->v_rows.p));
    CX_TRY(S->ktimer.end(0, ctx->stream));
    CX_TRY(S->ktimer.begin(1, ctx->stream));
    if (ctx->nranks <= 1) {
      CX_TRY(cxk_ft_partials(A, S->v_rows.p));
      hipLaunchKernelGGL(k_cam_reduce9_dot, dim3(grid), dim3(256), 0, ctx->stream, (const double*)A->d_partials.p,
                         (const int32_t*)A->d_cam_seg_start.p, y, n, d, x, tail);
      CX_TRY(S->ktimer.end(1, ctx->stream));
    } else {
      CX_TRY(cxk_ft_multiply(A, S->v_rows.p, y, false));
      CX_TRY(S->ktimer.end(1, ctx->stream));
      CX_TRY(cx_allreduce_device(ctx, y, n));
      hipLaunchKernelGGL(k_d2x_dot, dim3(grid), dim3(256), 0, ctx->stream, y, d, x, n, tail);
    }
    CX_HIP(hipGetLastError());
    *done = true;
    return CX_OK;
  }
};

// S x with the explicitly computed block-sparse Schur complement (BlockRandomAccessSparseMatrixAdapter,
// schur_complement_solver.cc:60-75 -> SymmetricRightMultiplyAndAccumulate); D_f^2 is added after the
// cross-rank sum because every rank holds the S of its own points only.
struct ExplicitSchurOp : LinOp {
  cx_solver* S;
  cx_matrix* A;
  const double* D;
  int64_t size() const override { return 9 * int64_t(A->C); }
  int apply(const double* x, double* y) override {
    cx_context* ctx = A->ctx;
    CX_TRY(S->ktimer.begin(4, ctx->stream));
    CX_TRY(cxs_sparse_multiply(A, x, y));
    CX_TRY(S->ktimer.end(4, ctx->stream));
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, y, size()));
    if (D)
      hipLaunchKernelGGL(k_add_d2x, dim3(grid_for(size(), 256)), dim3(256), 0, ctx->stream, y, D + 3 * int64_t(A->P), x, size());
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int apply_dot(const double* x, double* y, const DotTail& tail, bool* done) override {
    cx_context* ctx = A->ctx;
    const int64_t n = size();
    const int grid = int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 255) / 256)));
    CX_TRY(S->ktimer.begin(4, ctx->stream));
    CX_TRY(cxs_sparse_multiply(A, x, y));
    CX_TRY(S->ktimer.end(4, ctx->stream));
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, y, n));
    hipLaunchKernelGGL(k_d2x_dot, dim3(grid), dim3(256), 0, ctx->stream, y, D ? D + 3 * int64_t(A->P) : (const double*)nullptr, x, n, tail);
    CX_HIP(hipGetLastError());
    *done = true;
    return CX_OK;
  }
};

// residual = rhs - (S z), the D_f^2 z term of S added here (iterative_refiner.cc:60-63)
__global__ void k_refine_residual(const double* __restrict__ rhs, const double* __restrict__ sz, const double* __restrict__ Df,
                                  const double* __restrict__ z, double* __restrict__ out, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = Df ? Df[i] : 0.0;
  out[i] = rhs[i] - (sz[i] + d * d * z[i]);
}

__global__ void k_axpy1(double* __restrict__ y, const double* __restrict__ x, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// PowerSeriesExpansionPreconditioner::RightMultiplyAndAccumulate
// (power_series_expansion_preconditioner.cc:57-84): y = sum_{k=0..} Z^k (F'F)^-1 x with
// Z = (F'F)^-1 F'E (E'E)^-1 E'F  (InversePowerSeriesOperatorRightMultiplyAccumulate,
// implicit_schur_complement.cc:146-177).  tolerance = 0 (the preconditioner setting) runs a
// fixed max_iterations terms and needs no host round trip.
struct SpseOp : LinOp {
  cx_solver* S;
  cx_matrix* A;
  int max_iterations;
  double tolerance;
  int64_t size() const override { return 9 * int64_t(A->C); }
  int norm(const double* v, double* out) {
    const int64_t n = size();
    const int nb = int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 1023) / 1024)));
    hipStream_t st = A->ctx->stream;
    hipLaunchKernelGGL(k_dot2_partial, dim3(nb), dim3(256), 0, st, v, v, (const double*)nullptr, (const double*)nullptr, n,
                       S->partial.p, (const CgState*)nullptr);
    hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(256), 0, st, (const double*)S->partial.p, nb, int(FIN_STORE), 0, 0,
                       S->spse_state.p, (CgState*)nullptr, 0);
    CgState h;
    CX_TRY(cx_read_back(S->ctx, &h, S->spse_state.p, sizeof(h), st));
    CX_TRY(cx_stream_sync(S->ctx, st));
    *out = std::sqrt(h.s0);
    return CX_OK;
  }
  int apply(const double* x, double* y) override {
    cx_context* ctx = A->ctx;
    hipStream_t st = ctx->stream;
    const int64_t n = size();
    const int g = grid_for(n, 256);
    CX_TRY(S->v_spse.alloc(size_t(3 * n)));
    CX_TRY(S->partial.alloc(2 * kRedBlocks));
    CX_TRY(S->spse_state.alloc(1));
    double* series = S->v_spse.p;
    double* previous = S->v_spse.p + n;
    double* tmp_f = S->v_spse.p + 2 * n;
    const int64_t nb9 = A->C;
    // y = (F'F)^-1 x ; previous = y
    hipLaunchKernelGGL(k_blockdiag_multiply<9>, dim3(grid_for(9 * nb9, 256)), dim3(256), 0, st, (const double*)S->cam_blocks.p, x, y, nb9);
    CX_HIP(hipMemcpyAsync(previous, y, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, st));
    double threshold = 0.0;
    if (tolerance > 0.0) {
      double ny = 0.0;
      CX_TRY(norm(y, &ny));
      threshold = tolerance * ny;
    }
    for (int i = 1;; ++i) {
      // series = Z previous
      CX_TRY(cxs_chunk_pass(A, 3, S->ete_inv.p, previous, nullptr, S->v_rows.p));
      CX_TRY(cxk_ft_multiply(A, S->v_rows.p, tmp_f, false));
      if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, tmp_f, n));
      hipLaunchKernelGGL(k_blockdiag_multiply<9>, dim3(grid_for(9 * nb9, 256)), dim3(256), 0, st, (const double*)S->cam_blocks.p,
                         (const double*)tmp_f, series, nb9);
      hipLaunchKernelGGL(k_axpy1, dim3(g), dim3(256), 0, st, y, (const double*)series, n);
      if (i >= max_iterations) break;
      if (tolerance > 0.0) {
        double ns = 0.0;
        CX_TRY(norm(series, &ns));
        if (ns < threshold) break;
      }
      std::swap(previous, series);
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

struct BlockDiag9Op : LinOp {
  cx_context* ctx;
  const double* blocks;
  int64_t nblocks;
  int64_t size() const override { return 9 * nblocks; }
  const double* block9_inverse() const override { return nblocks > 0 ? blocks : nullptr; }
  int apply(const double* x, double* y) override {
    if (nblocks)
      hipLaunchKernelGGL(k_blockdiag_multiply<9>, dim3(grid_for(9 * nblocks, 256)), dim3(256), 0, ctx->stream, blocks, x, y, nblocks);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int apply_dot(const double* x, double* y, const DotTail& tail, bool* done) override {
    *done = nblocks > 0;
    if (!nblocks) return CX_OK;
    const int64_t n = 9 * nblocks;
    const int grid = int(std::min<int64_t>(kRedBlocks, (n + 255) / 256));
    hipLaunchKernelGGL(k_blockdiag9_dot, dim3(grid), dim3(256), 0, ctx->stream, blocks, x, y, n, tail);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

// VisibilityBasedPreconditioner::RightMultiplyAndAccumulate (visibility_based_preconditioner.cc:427-434): the
// banded Cholesky solve of cx_band_chol.hip
struct VisibilityOp : LinOp {
  cx_matrix* A = nullptr;
  cx_vis_plan* plan = nullptr;
  int64_t size() const override { return 9 * int64_t(A->C); }
  int apply(const double* x, double* y) override { return cxv_solve(A, plan, x, y); }
};

struct IdentityOp : LinOp {
  cx_context* ctx;
  int64_t n;
  int64_t size() const override { return n; }
  int apply(const double* x, double* y) override {
    CX_HIP(hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return CX_OK;
  }
};

bool CgnrPlain() {
  static const bool plain = std::getenv("CX_CGNR_PLAIN") != nullptr;
  return plain;
}

// (J'J + D'D) x   (CgnrLinearOperator, cgnr_solver.cc:98-114)
struct CgnrOp : LinOp {
  cx_solver* S;
  cx_matrix* A;
  const double* D;
  int64_t size() const override { return A->num_cols; }
  // t = J x and y = J't + D^2 x, both written outright: no zeroing of t or y beforehand, and on one rank the LM diagonal
  // rides in the kernels that write y (k_left_e_239 for the points, k_cam_reduce9 for the cameras) -- four launches where
  // round 3 had two memsets, the two products and k_add_d2x.  Sharded: the cameras' part of J't is summed over the ranks
  // first, D^2 x is added once after that.
  int apply(const double* x, double* y) override {
    cx_context* ctx = A->ctx;
    hipStream_t st = ctx->stream;
    const bool sharded = ctx->nranks > 1 && A->is239;
    if (CgnrPlain()) {  // round 3's form, for A/B runs and the bit-equality test
      CX_HIP(hipMemsetAsync(S->v_rows.p, 0, size_t(A->num_rows) * sizeof(double), st));
      CX_TRY(cxk_right_multiply(A, x, S->v_rows.p));
      CX_HIP(hipMemsetAsync(y, 0, size_t(A->num_cols) * sizeof(double), st));
      CX_TRY(cxk_left_multiply(A, S->v_rows.p, y));
      if (sharded) CX_TRY(cx_allreduce_device(ctx, y + 3 * int64_t(A->P), 9 * int64_t(A->C)));
      if (D) hipLaunchKernelGGL(k_add_d2x, dim3(grid_for(size(), 256)), dim3(256), 0, st, y, D, x, size());
      CX_HIP(hipGetLastError());
      return CX_OK;
    }
    CX_TRY(S->ktimer.begin(2, st));
    CX_TRY(cxk_right_multiply(A, x, S->v_rows.p, false));
    CX_TRY(S->ktimer.end(2, st));
    bool folded = false;
    CX_TRY(S->ktimer.begin(3, st));
    CX_TRY(cxk_left_multiply(A, S->v_rows.p, y, false, sharded ? nullptr : D, x, &folded));
    CX_TRY(S->ktimer.end(3, st));
    if (sharded) CX_TRY(cx_allreduce_device(ctx, y + 3 * int64_t(A->P), 9 * int64_t(A->C)));
    if (D && !folded) hipLaunchKernelGGL(k_add_d2x, dim3(grid_for(size(), 256)), dim3(256), 0, st, y, D, x, size());
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

// block diagonal (J'J + D'D)^-1: 3x3 point blocks then 9x9 camera blocks
struct CgnrJacobiOp : LinOp {
  cx_solver* S;
  cx_matrix* A;
  int64_t size() const override { return A->num_cols; }
  // z = M r and r.z in one launch (k_cgnr_jacobi_dot: the sum of k_dot2_fused(r, z), term by term)
  int apply_dot(const double* x, double* y, const DotTail& tail, bool* done) override {
    const int64_t n = size();
    *done = n > 0 && !CgnrPlain();
    if (!*done) return CX_OK;
    const int nb = int(std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 1023) / 1024)));
    hipLaunchKernelGGL(k_cgnr_jacobi_dot, dim3(nb), dim3(256), 0, A->ctx->stream, (const double*)S->pt_blocks.p, (const double*)S->cam_blocks.p,
                       x, y, 3 * int64_t(A->P), n, tail);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int apply(const double* x, double* y) override {
    hipStream_t st = A->ctx->stream;
    if (A->P) hipLaunchKernelGGL(k_blockdiag_multiply<3>, dim3(grid_for(3 * int64_t(A->P), 256)), dim3(256), 0, st, S->pt_blocks.p, x, y, int64_t(A->P));
    if (A->C) hipLaunchKernelGGL(k_blockdiag_multiply<9>, dim3(grid_for(9 * int64_t(A->C), 256)), dim3(256), 0, st, S->cam_blocks.p, x + 3 * int64_t(A->P), y + 3 * int64_t(A->P), int64_t(A->C));
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

// Phase timer: an event pair per phase on the context's stream; the elapsed times are read in cx_solver_solve
// after the solve's final synchronisation, so a phase boundary costs two event records and no host wait.
struct Stopwatch {
  cx_solver* S;
  hipStream_t st;
  int start() {
    if (!S->diag || S->num_pending >= 4) return CX_OK;
    return hipEventRecord(S->ctx->ev[8 + 2 * S->num_pending], st) == hipSuccess ? CX_OK : CX_ERR_HIP;
  }
  int stop(double* ms) {
    if (!S->diag || S->num_pending >= 4) return CX_OK;
    if (hipEventRecord(S->ctx->ev[9 + 2 * S->num_pending], st) != hipSuccess) return CX_ERR_HIP;
    S->pending_ms[S->num_pending++] = ms;
    return CX_OK;
  }
};

int CheckFlag(cx_solver* S, const char* what, cx_summary* summary, bool* failed) {
  int h = 0;
  CX_TRY(cx_read_back(S->ctx, &h, S->flag.p, sizeof(int), S->ctx->stream));
  CX_TRY(cx_stream_sync(S->ctx, S->ctx->stream));
  *failed = (h != 0);
  if (h != 0) {
    summary->termination_type = CX_FAILURE;
    summary->num_iterations = 0;
    std::snprintf(summary->message, sizeof(summary->message), "%s", what);
  }
  return CX_OK;
}

// Sharded set-up: the per-camera 9x9 blocks are symmetric, so their 45 distinct values and the 9 entries of the
// reduced right-hand side travel together -- ONE all-reduce of 54 C doubles instead of 81 C and then 9 C.
__global__ void k_pack_blocks_rhs(const double* __restrict__ blocks, const double* __restrict__ rhs, double* __restrict__ packed, int C,
                                  int with_blocks) {
  const int per = with_blocks ? 54 : 9;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(C) * per) return;
  const int c = int(i / per), k = int(i - int64_t(c) * per);
  const int kr = with_blocks ? k - 45 : k;
  if (kr >= 0) { packed[i] = rhs[9 * int64_t(c) + kr]; return; }
  int a = 0, rem = k;
  while (rem >= 9 - a) { rem -= 9 - a; ++a; }
  packed[i] = blocks[81 * int64_t(c) + a * 9 + a + rem];
}
__global__ void k_unpack_blocks_rhs(const double* __restrict__ packed, double* __restrict__ blocks, double* __restrict__ rhs, int C,
                                    int with_blocks) {
  const int per = with_blocks ? 54 : 9;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(C) * per) return;
  const int c = int(i / per), k = int(i - int64_t(c) * per);
  const int kr = with_blocks ? k - 45 : k;
  if (kr >= 0) { rhs[9 * int64_t(c) + kr] = packed[i]; return; }
  int a = 0, rem = k;
  while (rem >= 9 - a) { rem -= 9 - a; ++a; }
  const int bcol = a + rem;
  blocks[81 * int64_t(c) + a * 9 + bcol] = packed[i];
  blocks[81 * int64_t(c) + bcol * 9 + a] = packed[i];
}

// ------------------------------------------------------------------- solvers
int SolveIterativeSchur239(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol,
                           double q_tol, double* x, cx_summary* summary) {
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const cx_solver_options& o = S->opt;
  const int64_t nf = 9 * int64_t(A->C), ne = 3 * int64_t(A->P);
  Stopwatch sw{S, st};
  CX_TRY(S->ete_inv.alloc(9 * size_t(A->P)));
  CX_TRY(S->v_rows.alloc(size_t(A->num_rows)));
  CX_TRY(S->v_rhs.alloc(nf));
  CX_TRY(S->v_x.alloc(nf));
  // flag[0]: set-up (a point's E'E + D^2 or a camera block that cannot be inverted); flag[1]: the factorisation of a
  // visibility based preconditioner.  Kept apart so that the CLUSTER_TRIDIAGONAL retry below, which the reference
  // takes only when Factorize() of the preconditioner itself fails (visibility_based_preconditioner.cc:331-360),
  // neither reacts to nor erases a set-up failure.
  CX_TRY(S->flag.alloc(2));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, 2 * sizeof(int), st));
  int* const vis_flag = S->flag.p + 1;
  CX_TRY(sw.start());
  // ImplicitSchurComplement::Init + UpdateRhs (implicit_schur_complement.cc:49-97, 251-276) fused into
  // one pass over E and one over F (cxs_implicit_init)
  // compute_ftf_inverse_ (implicit_schur_complement.cc:63-67)
  const bool need_ftf = o.preconditioner_type == CX_JACOBI || o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION ||
                        o.use_spse_initialization;
  if (o.use_spse_initialization && o.preconditioner_type == CX_SCHUR_JACOBI) {
    cx_set_error("use_spse_initialization together with SCHUR_JACOBI is not available on the device");
    return CX_ERR_UNSUPPORTED;
  }
  const bool want_blocks = need_ftf || o.preconditioner_type == CX_SCHUR_JACOBI;
  const bool visibility = o.preconditioner_type == CX_CLUSTER_JACOBI || o.preconditioner_type == CX_CLUSTER_TRIDIAGONAL;
  cx_vis_plan* vis_plan = nullptr;
  if (visibility) CX_TRY(cxv_get_plan(A, o.preconditioner_type, o.visibility_clustering_type, &vis_plan));
  if (!want_blocks && !visibility && o.preconditioner_type != CX_IDENTITY) {
    cx_set_error("preconditioner %d is not available for ITERATIVE_SCHUR on the device", o.preconditioner_type);
    return CX_ERR_UNSUPPORTED;
  }
  CX_TRY(S->cam_blocks.alloc(81 * size_t(std::max(A->C, 1))));
  // small problems with a block-Jacobi preconditioner: the two passes stop at their segment partial sums and the CG run's
  // first launch does the rest of the set-up (k_cg_small_setup)
  const bool small_setup = SmallSetupEligible(ctx, nf) && !o.use_spse_initialization && A->num_tiles > 0 && A->num_segs > 0 &&
                           (o.preconditioner_type == CX_JACOBI || o.preconditioner_type == CX_SCHUR_JACOBI);
  CX_TRY(cxs_implicit_init(A, D, b, want_blocks, o.preconditioner_type == CX_SCHUR_JACOBI, S->ete_inv.p, S->v_rows.p,
                           S->cam_blocks.p, S->v_rhs.p, S->flag.p, small_setup));
  if (ctx->nranks > 1 && A->C > 0) {
    const int per = want_blocks ? 54 : 9;
    const int64_t count = int64_t(A->C) * per;
    CX_TRY(S->v_pack.alloc(size_t(count)));
    hipLaunchKernelGGL(k_pack_blocks_rhs, dim3(grid_for(count, 256)), dim3(256), 0, st, (const double*)S->cam_blocks.p,
                       (const double*)S->v_rhs.p, S->v_pack.p, A->C, want_blocks ? 1 : 0);
    CX_TRY(cx_allreduce_device(ctx, S->v_pack.p, count));
    hipLaunchKernelGGL(k_unpack_blocks_rhs, dim3(grid_for(count, 256)), dim3(256), 0, st, (const double*)S->v_pack.p, S->cam_blocks.p,
                       S->v_rhs.p, A->C, want_blocks ? 1 : 0);
    CX_HIP(hipGetLastError());
  }
  if (want_blocks && !small_setup) CX_TRY(cxs_block9_add_diag_invert(ctx, S->cam_blocks.p, D ? D + ne : nullptr, A->C, S->flag.p));
  if (visibility) {
    // VisibilityBasedPreconditioner::UpdateImpl (visibility_based_preconditioner.cc:321-364)
    // (the set-up above ran k_cam_init<false>: the F'F block partial sums the preconditioner's assembly needs are in place)
    struct PartialsInPlace {  // cleared however this block is left
      cx_matrix* A;
      ~PartialsInPlace() { A->ftf_partials_current = false; }
    } in_place{A};
    A->ftf_partials_current = o.preconditioner_type != CX_SCHUR_JACOBI && !small_setup && A->num_tiles > 0 && A->num_segs > 0;
    CX_TRY(cxv_factor(A, vis_plan, D, false, vis_flag));
    if (o.preconditioner_type == CX_CLUSTER_TRIDIAGONAL) {
      // "If it works, great, otherwise we scale all the cells in the preconditioner corresponding to the edges
      // in the degree-2 forest and that guarantees positive definiteness" (:331-360) -- the one host check
      int failed = 0;
      CX_TRY(cx_read_back(S->ctx, &failed, vis_flag, sizeof(int), st));
      CX_TRY(cx_stream_sync(S->ctx, st));
      if (failed) {
        CX_HIP(hipMemsetAsync(vis_flag, 0, sizeof(int), st));
        CX_TRY(cxv_factor(A, vis_plan, D, true, vis_flag));
      }
    }
  }
  CX_TRY(sw.stop(&S->timing.eliminate_ms));
  // a block that could not be inverted raises S->flag on the device; the CG prologue reads it there and ends
  // the run before the first iteration ("Preconditioner update failed.") -- no host check in between

  CX_TRY(sw.start());
  if (!small_setup) CX_HIP(hipMemsetAsync(S->v_x.p, 0, nf * sizeof(double), st));
  ImplicitSchurOp lhs;
  lhs.S = S; lhs.A = A; lhs.D = D;
  BlockDiag9Op bd;
  bd.ctx = ctx; bd.blocks = S->cam_blocks.p; bd.nblocks = A->C;
  IdentityOp id;
  id.ctx = ctx; id.n = nf;
  // the preconditioner ignores spse_tolerance so that it stays fixed during CG
  // (iterative_schur_complement_solver.cc:178-186)
  SpseOp spse_pre;
  spse_pre.S = S; spse_pre.A = A; spse_pre.max_iterations = o.max_num_spse_iterations; spse_pre.tolerance = 0.0;
  bool zero_initial = true;
  if (o.use_spse_initialization) {  // :100-111
    SpseOp init;
    init.S = S; init.A = A; init.max_iterations = o.max_num_spse_iterations; init.tolerance = o.spse_tolerance;
    CX_TRY(init.apply(S->v_rhs.p, S->v_x.p));
    zero_initial = false;
  }
  VisibilityOp vis_pre;
  vis_pre.A = A; vis_pre.plan = vis_plan;
  LinOp& pre = (o.preconditioner_type == CX_IDENTITY) ? static_cast<LinOp&>(id)
               : (o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION) ? static_cast<LinOp&>(spse_pre)
               : visibility ? static_cast<LinOp&>(vis_pre)
                            : static_cast<LinOp&>(bd);
  CgDriver cg{S, ctx, st, nf, nf};
  cg.preconditioner_failed = S->flag.p;
  cg.preconditioner_failed2 = vis_flag;
  SmallSetup setup;
  if (small_setup) {
    setup.partial45 = A->d_partials.p;
    setup.partial9 = A->d_partials9.p;
    setup.cam_seg_start = A->d_cam_seg_start.p;
    setup.Df = D ? D + ne : nullptr;
    setup.blocks = S->cam_blocks.p;
    setup.C = A->C;
    cg.small_setup = &setup;
  }
  CX_TRY(S->state.alloc(1));
  // use_mixed_precision_solves: S x inside CG streams fp32 copies of the cells (fp64 accumulation, fp64
  // vectors); set-up, right-hand side and back substitution stay on the fp64 values.  Not in the reference
  // (its mixed precision is for Cholesky, solver.h:572-590); parity is stated against the fp64 path.
  const bool mixed = o.use_mixed_precision_solves != 0;
  if (mixed) CX_TRY(cx_matrix_ensure_f32(A));
  A->use_f32 = mixed;
  A->stop = &S->state.p->flag;  // product kernels of a speculatively enqueued iteration exit early
  const int cg_rc = cg.run(lhs, pre, S->v_rhs.p, S->v_x.p, zero_initial, r_tol, q_tol, summary);
  A->stop = nullptr;
  A->use_f32 = false;
  CX_TRY(cg_rc);
  CX_TRY(sw.stop(&S->timing.reduced_solve_ms));

  CX_TRY(sw.start());
  if (summary->termination_type != CX_FAILURE && summary->termination_type != CX_FATAL_ERROR) {
    // BackSubstitute (:208-243)
    CX_TRY(cxs_chunk_pass(A, 2, S->ete_inv.p, S->v_x.p, b, x));
    CX_HIP(hipMemcpyAsync(x + ne, S->v_x.p, nf * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  CX_TRY(sw.stop(&S->timing.back_substitute_ms));
  return CX_OK;
}

// ITERATIVE_SCHUR with use_explicit_schur_complement: SparseSchurComplementSolver with the CG reduced solve
// (linear_solver.cc:111-116, schur_complement_solver.cc:101-159 + 337-420): Eliminate into the block-sparse S,
// block-Jacobi preconditioner from its diagonal cells (= SCHUR_JACOBI), CG, BackSubstitute.
int SolveExplicitSchur239(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol, double q_tol,
                          double* x, cx_summary* summary) {
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const int64_t nf = 9 * int64_t(A->C), ne = 3 * int64_t(A->P);
  Stopwatch sw{S, st};
  CX_TRY(S->v_rhs.alloc(nf));
  CX_TRY(S->v_x.alloc(nf));
  CX_TRY(S->cam_blocks.alloc(81 * size_t(std::max(A->C, 1))));
  CX_TRY(S->flag.alloc(1));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, sizeof(int), st));
  CX_TRY(sw.start());
  CX_TRY(cxs_eliminate_sparse(A, b, D, S->v_rhs.p));
  CX_TRY(cxs_sparse_diagonal(A, S->cam_blocks.p));
  if (ctx->nranks > 1) {
    CX_TRY(cx_allreduce_device(ctx, S->cam_blocks.p, 81 * int64_t(A->C)));
    CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p, nf));
  }
  // preconditioner_->Invert() on S(c,c) + D_f^2 (schur_complement_solver.cc:360-383)
  CX_TRY(cxs_block9_add_diag_invert(ctx, S->cam_blocks.p, D ? D + ne : nullptr, A->C, S->flag.p));
  CX_TRY(sw.stop(&S->timing.eliminate_ms));

  CX_TRY(sw.start());
  CX_HIP(hipMemsetAsync(S->v_x.p, 0, nf * sizeof(double), st));
  ExplicitSchurOp lhs;
  lhs.S = S; lhs.A = A; lhs.D = D;
  BlockDiag9Op bd;
  bd.ctx = ctx; bd.blocks = S->cam_blocks.p; bd.nblocks = A->C;
  CgDriver cg{S, ctx, st, nf, nf};
  cg.preconditioner_failed = S->flag.p;
  CX_TRY(S->state.alloc(1));
  A->stop = &S->state.p->flag;
  const int cg_rc = cg.run(lhs, bd, S->v_rhs.p, S->v_x.p, true, r_tol, q_tol, summary);
  A->stop = nullptr;
  CX_TRY(cg_rc);
  CX_TRY(sw.stop(&S->timing.reduced_solve_ms));

  CX_TRY(sw.start());
  if (summary->termination_type != CX_FAILURE && summary->termination_type != CX_FATAL_ERROR) {
    // SchurEliminator::BackSubstitute (schur_eliminator_impl.h:307-373); (E'E + D^2)^-1 is the eliminator's
    CX_TRY(cxs_chunk_pass(A, 2, A->d_elim_ete.p, S->v_x.p, b, x));
    CX_HIP(hipMemcpyAsync(x + ne, S->v_x.p, nf * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  CX_TRY(sw.stop(&S->timing.back_substitute_ms));
  return CX_OK;
}

int SolveCgnr239(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol, double q_tol,
                 double* x, cx_summary* summary) {
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const cx_solver_options& o = S->opt;
  const int64_t n = A->num_cols, ne = 3 * int64_t(A->P);
  Stopwatch sw{S, st};
  CX_TRY(S->v_rows.alloc(size_t(A->num_rows)));
  CX_TRY(S->v_rhs.alloc(n));
  CX_TRY(S->flag.alloc(1));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, sizeof(int), st));
  CX_TRY(sw.start());
  CX_TRY(cx_matrix_ensure_ft(A));
  const bool jacobi = o.preconditioner_type == CX_JACOBI;
  if (jacobi) {
    // BlockSparseJacobiPreconditioner::UpdateImpl (block_jacobi_preconditioner.cc:59-115)
    CX_TRY(S->pt_blocks.alloc(9 * size_t(A->P)));
    CX_TRY(S->cam_blocks.alloc(81 * size_t(A->C)));
    // the points' blocks of J'J and the point part of the right-hand side J'b in one pass over E (round 3: k_point_jacobi, one
    // thread walking a point's rows, then k_left_e_239)
    if (A->P) hipLaunchKernelGGL(k_point_jacobi_etb, dim3(A->num_tiles), dim3(kBlock), 0, st, (const double*)A->d_values.p, A->d_tile_row.p,
                                 A->d_tile_pt.p, A->d_pt_start.p, D, b, S->pt_blocks.p, S->v_rhs.p, S->flag.p);
    // the cameras' blocks of J'J and the camera part of the right-hand side J'b from the SAME pass over the camera-major
    // copy (k_cam_init, the kernel of the implicit Schur set-up; round 3: k_cam_diag, then k_cam_ft)
    CX_TRY(cxs_camera_blocks_and_ft(A, b, S->cam_blocks.p, S->v_rhs.p + ne));
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, S->cam_blocks.p, 81 * int64_t(A->C)));
    CX_TRY(cxs_block9_add_diag_invert(ctx, S->cam_blocks.p, D ? D + ne : nullptr, A->C, S->flag.p));
  } else if (o.preconditioner_type != CX_IDENTITY) {
    cx_set_error("CGNR supports JACOBI and IDENTITY preconditioners (cgnr_solver.cc:125-133)");
    return CX_ERR_UNSUPPORTED;
  } else {
    // rhs = J'b
    CX_TRY(cxk_left_multiply(A, b, S->v_rhs.p, false));
  }
  if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p + ne, n - ne));
  CX_TRY(sw.stop(&S->timing.eliminate_ms));
  CX_TRY(sw.start());
  // (x arrives zeroed: cx_solver_solve and cxe_solve clear the whole solution vector before they come here)
  CgnrOp lhs;
  lhs.S = S; lhs.A = A; lhs.D = D;
  CgnrJacobiOp jac;
  jac.S = S; jac.A = A;
  IdentityOp id;
  id.ctx = ctx; id.n = n;
  LinOp& pre = (o.preconditioner_type == CX_IDENTITY) ? static_cast<LinOp&>(id) : static_cast<LinOp&>(jac);
  CgDriver cg{S, ctx, st, n, ctx->nranks > 1 ? ne : n};
  CX_TRY(S->state.alloc(1));
  // Mixed precision (new; the reference has it for Cholesky only, solver.h:572-590): the CG
  // operator streams fp32 copies of the J values (half the HBM traffic, fp64 accumulation and
  // fp64 vectors); optional refinement steps correct x against the fp64 operator.
  const bool mixed = o.use_mixed_precision_solves != 0;
  if (mixed) CX_TRY(cx_matrix_ensure_f32(A));
  A->stop = &S->state.p->flag;
  A->use_f32 = mixed;
  int cg_rc = cg.run(lhs, pre, S->v_rhs.p, x, true, r_tol, q_tol, summary);
  A->use_f32 = false;
  A->stop = nullptr;
  CX_TRY(cg_rc);
  if (mixed && summary->termination_type != CX_FAILURE) {
    CX_TRY(S->v_cols.alloc(size_t(2 * n)));
    double* resid = S->v_cols.p;
    double* dx = S->v_cols.p + n;
    const int total_iterations = summary->num_iterations;
    int extra = 0;
    for (int it = 0; it < o.max_num_refinement_iterations; ++it) {
      // resid = J'b - (J'J + D'D) x with the fp64 values
      CX_TRY(lhs.apply(x, dx));
      hipLaunchKernelGGL(k_residual, dim3(grid_for(n, 256)), dim3(256), 0, st, (const double*)S->v_rhs.p, (const double*)dx,
                         resid, dx, n, (const CgState*)nullptr);
      CX_HIP(hipMemsetAsync(dx, 0, size_t(n) * sizeof(double), st));
      cx_summary rs;
      std::memset(&rs, 0, sizeof(rs));
      A->stop = &S->state.p->flag;
      A->use_f32 = true;
      cg_rc = cg.run(lhs, pre, resid, dx, true, r_tol, q_tol, &rs);
      A->use_f32 = false;
      A->stop = nullptr;
      CX_TRY(cg_rc);
      if (rs.termination_type == CX_FAILURE) break;
      hipLaunchKernelGGL(k_axpy1, dim3(grid_for(n, 256)), dim3(256), 0, st, x, (const double*)dx, n);
      extra += rs.num_iterations;
    }
    summary->num_iterations = total_iterations + extra;
  }
  CX_TRY(sw.stop(&S->timing.reduced_solve_ms));
  return CX_OK;
}

// SPARSE_SCHUR beyond the size a dense S is sensible for: the block-sparse S goes into 64x64 tiles and is
// factored by the tile-sparse Cholesky (cx_sparse_chol.hip) -- SparseSchurComplementSolver with a sparse direct
// reduced solve (schur_complement_solver.cc:101-159, 292-335).
constexpr int kSparseCholeskyMinCameras = 512;  // below: always the dense path (few, fat steps)
constexpr int kDenseSchurMaxCameras = 6144;     // above: a dense S and its working copy pass 48 GB

int SolveSparseSchur239(cx_solver* S, cx_matrix* A, const double* b, const double* D, double* x, cx_summary* summary) {
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const int64_t nf = 9 * int64_t(A->C), ne = 3 * int64_t(A->P);
  Stopwatch sw{S, st};
  CX_TRY(S->v_rhs.alloc(nf));
  CX_TRY(S->flag.alloc(1));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, sizeof(int), st));
  CX_TRY(sw.start());
  const bool sharded = ctx->nranks > 1;
  // use_mixed_precision_solves: the cells go into a single precision tile pool, so the row operands of their assembly are
  // kept in single precision too (one 128-byte line per operand instead of two; products and sums stay double)
  const bool f32_operands = S->opt.use_mixed_precision_solves != 0;
  if (sharded) {
    // every rank eliminates its own points into its own S cells; the cells (in the order of the ranks' common cell
    // list) and the right-hand side are summed over the ranks, factorisation and triangular solves are replicated
    CX_TRY(cxs_eliminate_sparse(A, b, D, S->v_rhs.p, f32_operands));
    CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p, nf));
  } else {
    CX_TRY(cxs_assemble_pair_items(A, D, nullptr, 0, f32_operands, b, S->v_rhs.p));
  }
  CX_TRY(sw.stop(&S->timing.eliminate_ms));
  CX_TRY(sw.start());
  double* z = x + ne;
  summary->num_iterations = 1;
  summary->termination_type = CX_SUCCESS;
  std::snprintf(summary->message, sizeof(summary->message), "Success.");
  // use_mixed_precision_solves: S is assembled from fp64 values into a SINGLE PRECISION tile pool and factored there
  // (FloatSuiteSparseCholesky / CudaSparseCholesky<float>, sparse_cholesky.cc:53-100; cx_sparse_chol.hip, k_sp_update_f32)
  const bool mixed = S->opt.use_mixed_precision_solves != 0;
  const int refinements = std::max(0, S->opt.max_num_refinement_iterations);
  A->sp.f32 = mixed;
  if (sharded) CX_TRY(cxsp_factor_and_solve_sharded(A, D ? D + ne : nullptr, S->v_rhs.p, z, S->flag.p));
  else CX_TRY(cxsp_factor_and_solve(A, D ? D + ne : nullptr, S->v_rhs.p, z, S->flag.p));
  bool failed = false;
  CX_TRY(CheckFlag(S, "Sparse Cholesky factorization failed: the reduced camera matrix is not positive definite.", summary, &failed));
  if (failed) {
    summary->num_iterations = 1;
    CX_HIP(hipMemsetAsync(z, 0, size_t(nf) * sizeof(double), st));  // not the NaNs of the failed factorisation
  } else if (refinements > 0 && nf > 0) {
    // RefinedSparseCholesky::Solve -> SparseIterativeRefiner::Refine (sparse_cholesky.cc:148-160, iterative_refiner.cc:53-72),
    // with either factor, as the reference wraps either: residual = rhs - S z in fp64, z += factor^-1 residual, a fixed
    // number of times.  S z is the implicit product of ITERATIVE_SCHUR (chunk pass + camera-major pass on the fp64 J and the
    // (E'E + D_e^2)^-1 the elimination left behind): S itself is not kept beside its factor.
    CX_TRY(S->v_rows.alloc(size_t(A->num_rows)));
    CX_TRY(S->v_p.alloc(nf));
    CX_TRY(S->v_tmp.alloc(nf));
    const double* Df = D ? D + ne : nullptr;
    for (int it = 0; it < refinements; ++it) {
      CX_TRY(cxs_chunk_pass(A, 0, A->d_elim_ete.p, z, nullptr, S->v_rows.p));
      CX_TRY(cxk_ft_multiply(A, S->v_rows.p, S->v_tmp.p, false));
      if (sharded) CX_TRY(cx_allreduce_device(ctx, S->v_tmp.p, nf));
      hipLaunchKernelGGL(k_refine_residual, dim3(grid_for(nf, 256)), dim3(256), 0, st, (const double*)S->v_rhs.p, (const double*)S->v_tmp.p, Df,
                         (const double*)z, S->v_p.p, nf);
      CX_HIP(hipGetLastError());
      CX_TRY(cxsp_solve(ctx, &A->sp, S->v_p.p, S->v_tmp.p));
      hipLaunchKernelGGL(k_axpy1, dim3(grid_for(nf, 256)), dim3(256), 0, st, z, (const double*)S->v_tmp.p, nf);
    }
    CX_HIP(hipGetLastError());
  }
  if (!failed && (mixed || refinements > 0))
    std::snprintf(summary->message, sizeof(summary->message), "Success. (%s factorisation, %d refinement step%s)",
                  mixed ? "single precision" : "double precision", refinements, refinements == 1 ? "" : "s");
  CX_TRY(sw.stop(&S->timing.reduced_solve_ms));
  CX_TRY(sw.start());
  if (summary->termination_type == CX_SUCCESS) CX_TRY(cxs_chunk_pass(A, 2, A->d_elim_ete.p, z, b, x));
  CX_TRY(sw.stop(&S->timing.back_substitute_ms));
  return CX_OK;
}

int SolveDenseSchur239(cx_solver* S, cx_matrix* A, const double* b, const double* D, double* x, cx_summary* summary) {
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const int64_t nf = 9 * int64_t(A->C), ne = 3 * int64_t(A->P);
  // use_mixed_precision_solves / max_num_refinement_iterations (solver.h:572-590; DenseCholesky::Create, dense_cholesky.cc:84-136,
  // SparseCholesky::Create, sparse_cholesky.cc:45-118): the single precision factorisation and the solves with a stored factor
  // that refinement needs live in the tile code (cx_sparse_chol.hip), so either option sends DENSE_SCHUR and SPARSE_SCHUR of any
  // size there -- for a dense S the plan simply holds every tile.
  const bool wants_tiles = S->opt.use_mixed_precision_solves != 0 || S->opt.max_num_refinement_iterations > 0;
  if (S->opt.type == CX_SPARSE_SCHUR || wants_tiles) {
    // tile-sparse factorisation when it stores less than half of the dense upper triangle, or when the dense
    // matrix (and its working copy) would not be reasonable any more; small problems stay dense (fewer steps).
    // On a sharded matrix the plan comes from the union of the ranks' cells, so every rank decides the same.
    const bool forced = std::getenv("CX_SPARSE_CHOLESKY") != nullptr || wants_tiles;
    if (forced || A->C >= kSparseCholeskyMinCameras) {
      // (round 3 kept the factor whole on every rank when refinement was asked for; round 4: the stored-factor sweeps run on
      // the distributed factor, CX_SPARSE_REFINE_REPLICATED=1 keeps the old form for A/B runs)
      static const bool refine_replicated = std::getenv("CX_SPARSE_REFINE_REPLICATED") != nullptr;
      A->sp.replicate = refine_replicated && S->opt.max_num_refinement_iterations > 0;
      if (ctx->nranks > 1) CX_TRY(cxsp_build_plan_sharded(A));
      else CX_TRY(cxsp_build_plan(A));
      if (A->sp.state == 1) {
        const int64_t T = A->sp.T, dense_tiles = T * (T + 1) / 2 + T;
        if (forced || 2 * A->sp.num_tiles <= dense_tiles || A->C >= kDenseSchurMaxCameras) return SolveSparseSchur239(S, A, b, D, x, summary);
      }
    }
  }
  if (A->C >= 2 * kDenseSchurMaxCameras || (ctx->nranks > 1 && A->C >= kDenseSchurMaxCameras)) {
    // a dense S of this size (> 190 GB with its working copy; on several ranks every rank would hold and all-reduce
    // its own 24+ GB copy) cannot be meant: DENSE_SCHUR, or a structure whose tile-sparse plan could not be built
    cx_set_error("%s with %d cameras: the dense reduced matrix does not fit and the tile-sparse Cholesky is not available "
                 "here (DENSE_SCHUR, or too much fill); use ITERATIVE_SCHUR", S->opt.type == CX_SPARSE_SCHUR ? "SPARSE_SCHUR" : "DENSE_SCHUR", A->C);
    return CX_ERR_UNSUPPORTED;
  }
  Stopwatch sw{S, st};
  CX_TRY(S->lhs.alloc(size_t(nf) * nf));
  CX_TRY(S->v_rhs.alloc(nf));
  CX_TRY(S->flag.alloc(1));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, sizeof(int), st));
  CX_TRY(sw.start());
  // Eliminate (schur_eliminator_impl.h:177-304).  Sharded: D_f^2 is added once after the sum.
  const bool sharded = ctx->nranks > 1;
  CX_TRY(cxs_eliminate_dense(A, b, D, !sharded || ctx->rank == 0, S->lhs.p, S->v_rhs.p));
  if (sharded) {
    CX_TRY(cx_allreduce_device(ctx, S->lhs.p, nf * nf));
    CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p, nf));
  }
  CX_TRY(sw.stop(&S->timing.eliminate_ms));
  CX_TRY(sw.start());
  // DenseSchurComplementSolver::SolveReducedLinearSystem (schur_complement_solver.cc:182-203)
  double* z = x + ne;
  summary->num_iterations = 0;
  summary->termination_type = CX_SUCCESS;
  std::snprintf(summary->message, sizeof(summary->message), "Success.");
  if (nf > 0) {
    CX_TRY(cxd_cholesky_solve(ctx, int(nf), S->lhs.p, S->v_rhs.p, z, S->flag.p));
    summary->num_iterations = 1;
    bool failed = false;
    CX_TRY(CheckFlag(S, "Dense Cholesky factorization failed: the reduced camera matrix is not positive definite.", summary, &failed));
    if (failed) {
      summary->num_iterations = 1;
      CX_HIP(hipMemsetAsync(z, 0, size_t(nf) * sizeof(double), st));  // not the NaNs of the failed factorisation
    }
  }
  CX_TRY(sw.stop(&S->timing.reduced_solve_ms));
  CX_TRY(sw.start());
  if (summary->termination_type == CX_SUCCESS) {
    // SchurEliminator::BackSubstitute (schur_eliminator_impl.h:307-373): the cofactor inverses of E'E + D^2 are the
    // eliminator's own (A->d_elim_ete, filled by cxs_eliminate_dense above on either of its paths)
    CX_TRY(cxs_chunk_pass(A, 2, A->d_elim_ete.p, z, b, x));
  }
  CX_TRY(sw.stop(&S->timing.back_substitute_ms));
  // (reached with use_mixed_precision_solves / refinement only when no tile plan could be built for this structure)
  if (wants_tiles && summary->termination_type == CX_SUCCESS && S->opt.use_mixed_precision_solves) summary->notes |= CX_NOTE_DOUBLE_PRECISION_FACTOR;
  if (wants_tiles && summary->termination_type == CX_SUCCESS)
    std::snprintf(summary->message, sizeof(summary->message),
                  "Success. (no tile plan for this structure: double precision dense factorisation, no refinement)");
  return CX_OK;
}

}  // namespace

int cx_cg_run(cx_solver* S, int64_t n, int64_t shared0, LinOp& lhs, LinOp& pre, const double* rhs, double* x,
              bool zero_initial, double r_tol, double q_tol, cx_summary* summary) {
  CgDriver cg{S, S->ctx, S->ctx->stream, n, shared0};
  return cg.run(lhs, pre, rhs, x, zero_initial, r_tol, q_tol, summary);
}

int cx_check_flag(cx_solver* S, const char* what, cx_summary* summary, bool* failed) {
  return CheckFlag(S, what, summary, failed);
}

// generic (dynamic block size) solvers live in cx_generic.hip
int cxg_solve(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol, double q_tol, double* x,
              cx_summary* summary);
int cxg_eliminate_dense(cx_solver* S, cx_matrix* A, const double* b, const double* D, double* lhs, double* rhs);
int cxg_back_substitute(cx_solver* S, cx_matrix* A, const double* b, const double* D, const double* z, double* x);
int cxg_implicit_schur_multiply(cx_solver* S, cx_matrix* A, const double* D, const double* b, const double* x,
                                double* y, double* rhs);

extern "C" {

void cx_solver_default_options(cx_solver_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->type = CX_ITERATIVE_SCHUR;
  o->preconditioner_type = CX_JACOBI;
  o->min_num_iterations = 0;
  o->max_num_iterations = 500;
  o->residual_reset_period = 10;
  o->max_num_refinement_iterations = 0;
  o->max_num_spse_iterations = 5;
  o->spse_tolerance = 0.1;
  o->deterministic = 1;
}

int cx_solver_create(cx_context* ctx, const cx_solver_options* options, cx_solver** out) {
  CX_CHECK_ARG(ctx && options && out);
  CX_CHECK_ARG(options->type >= CX_DENSE_SCHUR && options->type <= CX_CGNR);
  CX_CHECK_ARG(options->residual_reset_period > 0 && options->max_num_iterations >= 0);
  if (options->type != CX_CGNR) CX_CHECK_ARG(options->num_eliminate_blocks > 0);
  const bool cluster_pre = options->preconditioner_type == CX_CLUSTER_JACOBI || options->preconditioner_type == CX_CLUSTER_TRIDIAGONAL;
  if (cluster_pre && options->type == CX_CGNR) {
    // solver.cc:237-260 / solver_test.cc:1082-1092
    cx_set_error("Can only use CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL with ITERATIVE_SCHUR, not with CGNR.");
    return CX_ERR_INVALID_ARGUMENT;
  }
  if (cluster_pre && options->visibility_clustering_type != CX_CANONICAL_VIEWS && options->visibility_clustering_type != CX_SINGLE_LINKAGE) {
    cx_set_error("Unknown visibility clustering algorithm.");
    return CX_ERR_INVALID_ARGUMENT;
  }
  if (options->type == CX_ITERATIVE_SCHUR && options->use_explicit_schur_complement) {
    // the reference's option validation (solver.cc:277-290)
    if (options->preconditioner_type != CX_SCHUR_JACOBI) {
      cx_set_error("use_explicit_schur_complement only supports SCHUR_JACOBI as the preconditioner.");
      return CX_ERR_INVALID_ARGUMENT;
    }
    if (options->use_spse_initialization) {
      cx_set_error("use_explicit_schur_complement does not support use_spse_initialization.");
      return CX_ERR_INVALID_ARGUMENT;
    }
  }
  auto* s = new cx_solver;
  s->ctx = ctx;
  s->opt = *options;
  *out = s;
  return CX_OK;
}

void cx_solver_destroy(cx_solver* s) {
  if (!s) return;
  if (!s->parts.empty()) cxm_solver_destroy(s);
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->ring_h) (void)hipHostFree(s->ring_h);
  delete s;
}

int cx_solver_kernel_stats(const cx_solver* s, cx_kernel_stat* out, int32_t capacity, int32_t* count) {
  CX_CHECK_ARG(s && out && count && capacity >= 0);
  if (!s->parts.empty()) return cx_solver_kernel_stats(s->parts[0], out, capacity, count);  // shard 0 of a multi-shard front
  int n = 0;
  for (int i = 0; i < KernelTimer::kSlots && n < capacity; ++i) {
    if (s->ktimer.launches[i] == 0) continue;
    std::memset(&out[n], 0, sizeof(out[n]));
    std::snprintf(out[n].name, sizeof(out[n].name), "%s", s->ktimer.names[i]);
    out[n].sampled_ms = s->ktimer.total_ms[i];
    out[n].sampled_launches = s->ktimer.count[i];
    out[n].launches = s->ktimer.launches[i];
    ++n;
  }
  *count = n;
  return CX_OK;
}

int cx_solver_sample_next(cx_solver* s) {
  CX_CHECK_ARG(s);
  s->diag_requested = true;
  for (cx_solver* part : s->parts)
    if (part) part->diag_requested = true;
  return CX_OK;
}

int cx_solver_last_timing(const cx_solver* s, cx_solve_timing* out) {
  CX_CHECK_ARG(s && out);
  *out = s->timing;
  return CX_OK;
}

int cx_solver_solve(cx_solver* S, cx_matrix* A, const double* b, const cx_per_solve_options* ps, double* x,
                    cx_summary* summary) {
  CX_CHECK_ARG(S && A && b && ps && x && summary);
  if (cxm_is_front(S->ctx) || !A->parts.empty()) return cxm_solver_solve(S, A, b, ps, x, summary);
  CX_CHECK_ARG(S->ctx == A->ctx);
  cx_context* ctx = S->ctx;
  CX_HIP(hipSetDevice(ctx->device));
  const cx_solver_options& o = S->opt;
  if (o.type != CX_CGNR) CX_CHECK_ARG(o.num_eliminate_blocks == A->nelim);
  std::memset(summary, 0, sizeof(*summary));
  const bool launch_bound = ctx->nranks <= 1 && A->num_cols_f <= kSmallCgMax;
  {
    static const int period = std::getenv("CX_DIAG_PERIOD") ? std::max(1, std::atoi(std::getenv("CX_DIAG_PERIOD"))) : int(cx_solver::kDiagPeriod);
    S->diag = !launch_bound || S->diag_requested || (S->num_solves % period) == 0;  // see cx_solver::diag
    S->diag_requested = false;
    ++S->num_solves;
  }
  S->num_pending = 0;
  S->ktimer.enabled = S->diag;
  S->timing.sampled = S->diag ? 1.0 : 0.0;
  if (S->diag) {
    S->timing = cx_solve_timing{};
    S->timing.sampled = 1.0;
    S->ktimer.reset();
    static const int forced = std::getenv("CX_KTIMER_SAMPLES") ? std::atoi(std::getenv("CX_KTIMER_SAMPLES")) : -1;  // A/B switch
    // small problems are bound by their launches: one sampled launch per kernel reports its time
    S->ktimer.max_samples = forced >= 0 ? std::min(forced, int(KernelTimer::kMaxSamples)) : (launch_bound ? 1 : int(KernelTimer::kMaxSamples));
  }
  const auto host_t0 = std::chrono::steady_clock::now();
  cx_allreduce_reset(ctx);
  HostOrDevice hb(ctx), hD(ctx), hx(ctx);
  int staged = hb.in(b, size_t(A->num_rows), ps->b_on_device ? CX_DEVICE : ps->memspace);
  if (staged == CX_OK) staged = hD.in(ps->D, size_t(A->num_cols), ps->memspace);
  if (staged == CX_OK) staged = hx.inout(x, size_t(A->num_cols), ps->memspace, false);
  // sharded: a rank that could not stage its inputs says so before anybody enters the solve's first collective
  CX_TRY(cx_comm_agree(ctx, staged));
  if (S->diag) CX_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
  // "std::fill(x, x + A->num_cols(), 0.0)" (schur_complement_solver.cc:137): whatever a failed solve leaves
  // unwritten is zero, in the caller's host buffer as well (the staging copy is not initialised otherwise)
  CX_HIP(hipMemsetAsync(hx.dptr, 0, size_t(A->num_cols) * sizeof(double), ctx->stream));
  int rc;
  // the static solvers, on a <2,3,9> matrix: the caller's, or the inner matrix of an embedding (cx_embed.hip)
  auto solve_static = [&](cx_matrix* M, const double* bd, const double* Dd, double* xd) -> int {
    switch (o.type) {
      case CX_ITERATIVE_SCHUR:
        return o.use_explicit_schur_complement ? SolveExplicitSchur239(S, M, bd, Dd, ps->r_tolerance, ps->q_tolerance, xd, summary)
                                               : SolveIterativeSchur239(S, M, bd, Dd, ps->r_tolerance, ps->q_tolerance, xd, summary);
      case CX_CGNR: return SolveCgnr239(S, M, bd, Dd, ps->r_tolerance, ps->q_tolerance, xd, summary);
      default: return SolveDenseSchur239(S, M, bd, Dd, xd, summary);
    }
  };
  if (A->is239) {
    rc = solve_static(A, hb.dptr, hD.dptr, hx.dptr);
  } else if (ctx->nranks > 1) {
    // the sharded solvers (points over ranks, camera-space sums exchanged) exist for the native <2,3,9> layout only
    cx_set_error("a sharded context solves matrices in the static <2,3,9> layout only (this one takes the %s path)",
                 A->embed ? "embedded" : "dynamic-size");
    rc = CX_ERR_UNSUPPORTED;
  } else if (A->embed) {
    rc = cxe_solve(A, hb.dptr, hD.dptr, hx.dptr, solve_static);
  } else if (o.type == CX_ITERATIVE_SCHUR && o.use_explicit_schur_complement) {
    cx_set_error("use_explicit_schur_complement needs the static <2,3,9> layout");
    rc = CX_ERR_UNSUPPORTED;
  } else {
    rc = cxg_solve(S, A, hb.dptr, hD.dptr, ps->r_tolerance, ps->q_tolerance, hx.dptr, summary);
  }
  if (rc != CX_OK) {
    summary->termination_type = CX_FATAL_ERROR;
    std::snprintf(summary->message, sizeof(summary->message), "%s", cx_last_error());
    return rc;
  }
  if (!S->diag) {
    CX_TRY(cx_stream_sync(ctx, ctx->stream));
    S->timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    return hx.out();
  }
  CX_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
  CX_TRY(cx_event_sync(ctx, ctx->ev[5]));
  float ms = 0.f;
  CX_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
  S->timing.total_ms = ms;
  for (int i = 0; i < S->num_pending; ++i) {
    float f = 0.f;
    CX_HIP(hipEventElapsedTime(&f, ctx->ev[8 + 2 * i], ctx->ev[9 + 2 * i]));
    *S->pending_ms[i] = f;
  }
  CX_TRY(cx_allreduce_collect(ctx, &S->timing.allreduce_ms, &S->timing.allreduce_host_ms, &S->timing.allreduce_calls,
                              &S->timing.allreduce_bytes));
  CX_TRY(S->ktimer.collect());
  return hx.out();
}

// ---- Schur pieces on their own (parity tests)
int cx_schur_sparse_structure(cx_matrix* A, int64_t* num_cells, int32_t* cell_row, int32_t* cell_col, int64_t capacity) {
  CX_CHECK_ARG(A && num_cells && capacity >= 0);
  if (A->embed) A = A->embed->inner;  // the f-blocks of an embedding keep their numbers
  if (!A->is239) {
    cx_set_error("cx_schur_sparse_structure needs the static <2,3,9> layout");
    return CX_ERR_UNSUPPORTED;
  }
  CX_HIP(hipSetDevice(A->ctx->device));
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state != 1) {
    cx_set_error("the pair list of this structure is too large to build");
    return CX_ERR_UNSUPPORTED;
  }
  *num_cells = A->num_cells;
  const int64_t m = std::min<int64_t>(capacity, A->num_cells);
  for (int64_t i = 0; i < m; ++i) {
    if (cell_row) cell_row[i] = A->h_cell_c1[size_t(i)];
    if (cell_col) cell_col[i] = A->h_cell_c2[size_t(i)];
  }
  return CX_OK;
}
int cx_schur_eliminate_dense(cx_context* ctx, cx_matrix* A, const double* b, const double* D, double* lhs,
                             double* rhs, int32_t memspace) {
  CX_CHECK_ARG(ctx && A && lhs);
  CX_CHECK_ARG(A->nelim > 0);
  const int64_t nf = A->num_cols_f;
  HostOrDevice hb(ctx), hD(ctx), hl(ctx), hr(ctx);
  CX_TRY(hb.in(b, size_t(A->num_rows), memspace));
  CX_TRY(hD.in(D, size_t(A->num_cols), memspace));
  CX_TRY(hl.inout(lhs, size_t(nf) * nf, memspace, false));
  CX_TRY(hr.inout(rhs, size_t(nf), memspace, false));
  if (A->is239) {
    CX_TRY(cxs_eliminate_dense(A, hb.dptr, hD.dptr, true, hl.dptr, hb.dptr ? hr.dptr : nullptr));
  } else {
    cx_solver tmp;
    tmp.ctx = ctx;
    CX_TRY(cxg_eliminate_dense(&tmp, A, hb.dptr, hD.dptr, hl.dptr, hb.dptr ? hr.dptr : nullptr));
    CX_TRY(cx_stream_sync(ctx, ctx->stream));
  }
  CX_TRY(hl.out());
  return hr.out();
}

int cx_schur_back_substitute(cx_context* ctx, cx_matrix* A, const double* b, const double* D, const double* z,
                             double* x, int32_t memspace) {
  CX_CHECK_ARG(ctx && A && b && z && x);
  CX_CHECK_ARG(A->nelim > 0);
  HostOrDevice hb(ctx), hD(ctx), hz(ctx), hx(ctx);
  if (!A->is239) {
    CX_TRY(hb.in(b, size_t(A->num_rows), memspace));
    CX_TRY(hD.in(D, size_t(A->num_cols), memspace));
    CX_TRY(hz.in(z, size_t(A->num_cols_f), memspace));
    CX_TRY(hx.inout(x, size_t(A->num_cols), memspace, true));
    cx_solver tmp;
    tmp.ctx = ctx;
    CX_TRY(cxg_back_substitute(&tmp, A, hb.dptr, hD.dptr, hz.dptr, hx.dptr));
    CX_TRY(cx_stream_sync(ctx, ctx->stream));
    return hx.out();
  }
  DevBuf<double> ete;
  DevBuf<int> flag;
  CX_TRY(ete.alloc(9 * size_t(A->P)));
  CX_TRY(flag.alloc(1));
  CX_TRY(hb.in(b, size_t(A->num_rows), memspace));
  CX_TRY(hD.in(D, size_t(A->num_cols), memspace));
  CX_TRY(hz.in(z, 9 * size_t(A->C), memspace));
  CX_TRY(hx.inout(x, size_t(A->num_cols), memspace, true));
  CX_TRY(cxs_compute_ete_inverse(A, hD.dptr, nullptr, ete.p, nullptr, false, flag.p));
  CX_TRY(cxs_chunk_pass(A, 2, ete.p, hz.dptr, hb.dptr, hx.dptr));
  return hx.out();
}

int cx_implicit_schur_multiply(cx_context* ctx, cx_matrix* A, const double* D, const double* b, const double* x,
                               double* y, double* rhs, int32_t memspace) {
  CX_CHECK_ARG(ctx && A);
  CX_CHECK_ARG(A->nelim > 0);
  const int64_t nf = A->num_cols_f;
  HostOrDevice hb(ctx), hD(ctx), hx(ctx), hy(ctx), hr(ctx);
  if (!A->is239) {
    CX_TRY(hb.in(b, size_t(A->num_rows), memspace));
    CX_TRY(hD.in(D, size_t(A->num_cols), memspace));
    CX_TRY(hx.in(x, size_t(nf), memspace));
    CX_TRY(hy.inout(y, size_t(nf), memspace, false));
    CX_TRY(hr.inout(rhs, size_t(nf), memspace, false));
    cx_solver tmp;
    tmp.ctx = ctx;
    CX_TRY(cxg_implicit_schur_multiply(&tmp, A, hD.dptr, hb.dptr, hx.dptr, hy.dptr, hb.dptr ? hr.dptr : nullptr));
    CX_TRY(cx_stream_sync(ctx, ctx->stream));
    if (hx.dptr && hy.dptr) CX_TRY(hy.out());
    if (hb.dptr) CX_TRY(hr.out());
    return CX_OK;
  }
  DevBuf<double> ete, rows;
  DevBuf<int> flag;
  CX_TRY(ete.alloc(9 * size_t(A->P)));
  CX_TRY(rows.alloc(size_t(A->num_rows)));
  CX_TRY(flag.alloc(1));
  CX_TRY(hb.in(b, size_t(A->num_rows), memspace));
  CX_TRY(hD.in(D, size_t(A->num_cols), memspace));
  CX_TRY(hx.in(x, size_t(nf), memspace));
  CX_TRY(hy.inout(y, size_t(nf), memspace, false));
  CX_TRY(hr.inout(rhs, size_t(nf), memspace, false));
  CX_TRY(cxs_compute_ete_inverse(A, hD.dptr, nullptr, ete.p, nullptr, true, flag.p));
  if (hx.dptr && hy.dptr) {
    CX_TRY(cxs_chunk_pass(A, 0, ete.p, hx.dptr, nullptr, rows.p));
    CX_TRY(cxk_ft_multiply(A, rows.p, hy.dptr, false));
    if (hD.dptr)
      hipLaunchKernelGGL(k_add_d2x, dim3(grid_for(nf, 256)), dim3(256), 0, ctx->stream, hy.dptr,
                         hD.dptr + 3 * int64_t(A->P), (const double*)hx.dptr, nf);
    CX_TRY(hy.out());
  }
  if (hr.dptr && hb.dptr) {
    CX_TRY(cxs_chunk_pass(A, 1, ete.p, nullptr, hb.dptr, rows.p));
    CX_TRY(cxk_ft_multiply(A, rows.p, hr.dptr, false));
    CX_TRY(hr.out());
  }
  CX_TRY(cx_stream_sync(ctx, ctx->stream));
  return CX_OK;
}

int cx_dense_cholesky_solve(cx_context* ctx, int32_t n, double* lhs, const double* rhs, double* x, int32_t memspace,
                            cx_summary* summary) {
  CX_CHECK_ARG(ctx && n >= 0 && (n == 0 || (lhs && rhs && x)));
  HostOrDevice hl(ctx), hr(ctx), hx(ctx);
  DevBuf<int> flag;
  CX_TRY(flag.alloc(1));
  CX_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  CX_TRY(hl.inout(lhs, size_t(n) * n, memspace, true));
  CX_TRY(hr.in(rhs, size_t(n), memspace));
  CX_TRY(hx.inout(x, size_t(n), memspace, false));
  if (n > 0) CX_TRY(cxd_cholesky_solve(ctx, n, hl.dptr, hr.dptr, hx.dptr, flag.p));
  int h = 0;
  CX_TRY(cx_read_back(ctx, &h, flag.p, sizeof(int), ctx->stream));
  CX_TRY(cx_stream_sync(ctx, ctx->stream));
  if (summary) {
    std::memset(summary, 0, sizeof(*summary));
    summary->num_iterations = 1;
    summary->termination_type = h ? CX_FAILURE : CX_SUCCESS;
    std::snprintf(summary->message, sizeof(summary->message), "%s", h ? "Matrix is not positive definite." : "Success.");
  }
  return hx.out();
}

}  // extern "C"
