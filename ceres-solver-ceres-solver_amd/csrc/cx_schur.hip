// Schur-complement kernels for the static <2,3,9> bundle-adjustment layout (gfx950).
//
// Reference functions replaced:
//   ImplicitSchurComplement::{Init,RightMultiplyAndAccumulate,UpdateRhs,BackSubstitute,
//     AddDiagonalAndInvert}                      implicit_schur_complement.cc:49-276
//   PartitionedMatrixView::UpdateBlockDiagonal{EtE,FtF}   partitioned_matrix_view_impl.h:420-658
//   SchurEliminator<2,3,9>::{Eliminate,BackSubstitute}    schur_eliminator_impl.h:177-561
//   SchurJacobiPreconditioner::UpdateImpl        schur_jacobi_preconditioner.cc:88-98
//
// Work decomposition.  Point-major ("chunk") kernels run one workgroup per tile of
// whole chunks (<= 256 row blocks): E / F cells are streamed once through LDS into
// registers (coalesced 16-byte loads), per-row partial results go to LDS, one
// thread per point folds the rows of its chunk in row order (deterministic), and
// the rows read the per-point result back.  Camera-space sums run camera-major over
// a second copy of the F cells (cx_matrix.hip) in fixed-size segments whose partial
// sums are added in segment order, so every result is bitwise reproducible.
#include "cx_internal.h"
#include "cx_kernels.h"
#include "cx_schur.h"

#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <cstdlib>

static int grid_for(int64_t n, int block) { return int((n + block - 1) / block); }

// (i, j), i <= j < k, of the q-th pair in row-major order of the upper triangle
__device__ __forceinline__ void tri_decode(int64_t q, int k, int& i, int& j) {
  const double kk = 2.0 * k + 1.0;
  int ii = int((kk - sqrt(kk * kk - 8.0 * double(q))) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > k - 1) ii = k - 1;
  // first pair index of row ii is ii*k - ii*(ii-1)/2
  while (ii > 0 && int64_t(ii) * k - int64_t(ii) * (ii - 1) / 2 > q) --ii;
  while (int64_t(ii + 1) * k - int64_t(ii + 1) * ii / 2 <= q) ++ii;
  i = ii;
  j = ii + int(q - (int64_t(ii) * k - int64_t(ii) * (ii - 1) / 2));
}

// ------------------------------------------------------------ (E'E + D^2)^-1, E'b
// USE_LLT = true : LLT inverse, as AddDiagonalAndInvert (implicit_schur_complement.cc:179-204)
// USE_LLT = false: cofactor inverse, as InvertPSDMatrix<3> (invert_psd_matrix.h:60-63)
template <bool USE_LLT>
__global__ __launch_bounds__(kBlock) void k_chunk_ete(const double* __restrict__ E,
                                                      const int32_t* __restrict__ tile_row,
                                                      const int32_t* __restrict__ tile_pt,
                                                      const int32_t* __restrict__ pt_start,
                                                      const double* __restrict__ De,   // [3P] or null
                                                      const double* __restrict__ b,    // [2O] or null
                                                      double* __restrict__ ete_inv,    // [9P]
                                                      double* __restrict__ g,          // [3P] or null: E'b
                                                      int* __restrict__ not_pd) {
  __shared__ double lds[kBlock * 6];
  __shared__ double w[kBlock * 9];
  __shared__ double red[9 * 4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  const int p0 = tile_pt[t], p1 = tile_pt[t + 1];
  double m[9], gg[3] = {0.0, 0.0, 0.0};
  bool have = false;
  int p = -1;
  if (r1 - r0 <= kBlock) {
    const int nvalid = r1 - r0;
    double e[6];
    stage_cells<6>(E + 6 * int64_t(r0), nvalid, lds, e);
    if (tid < nvalid) {
      double2 bv = make_double2(0.0, 0.0);
      if (b) bv = reinterpret_cast<const double2*>(b)[r0 + tid];
      double* wr = w + tid * 9;
      wr[0] = e[0] * e[0] + e[3] * e[3];
      wr[1] = e[0] * e[1] + e[3] * e[4];
      wr[2] = e[0] * e[2] + e[3] * e[5];
      wr[3] = e[1] * e[1] + e[4] * e[4];
      wr[4] = e[1] * e[2] + e[4] * e[5];
      wr[5] = e[2] * e[2] + e[5] * e[5];
      wr[6] = e[0] * bv.x + e[3] * bv.y;
      wr[7] = e[1] * bv.x + e[4] * bv.y;
      wr[8] = e[2] * bv.x + e[5] * bv.y;
    }
    int jb = 0, je = 0;
    if (tid < p1 - p0) {
      jb = pt_start[p0 + tid] - r0;
      je = pt_start[p0 + tid + 1] - r0;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      p = p0 + tid;
      have = true;
      double s[9], s2[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { s[k] = 0.0; s2[k] = 0.0; }
      // two independent partial sums over alternating rows, combined at the end
      int j = jb;
      for (; j + 2 <= je; j += 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { s[k] += w[j * 9 + k]; s2[k] += w[j * 9 + 9 + k]; }
      }
      if (j < je) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] += w[j * 9 + k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) s[k] += s2[k];
      m[0] = s[0]; m[1] = s[1]; m[2] = s[2]; m[4] = s[3]; m[5] = s[4]; m[8] = s[5];
      gg[0] = s[6]; gg[1] = s[7]; gg[2] = s[8];
    }
  } else {
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const double* e = E + 6 * int64_t(r);
      double2 bv = make_double2(0.0, 0.0);
      if (b) bv = reinterpret_cast<const double2*>(b)[r];
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
      p = p0;
      have = true;
      m[0] = s[0]; m[1] = s[1]; m[2] = s[2]; m[4] = s[3]; m[5] = s[4]; m[8] = s[5];
      gg[0] = s[6]; gg[1] = s[7]; gg[2] = s[8];
    }
  }
  if (have) {
    if (De) {
      const double* d = De + 3 * int64_t(p);
      m[0] += d[0] * d[0];
      m[4] += d[1] * d[1];
      m[8] += d[2] * d[2];
    }
    m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
    double inv[9];
    if (USE_LLT) {
      bool ok;
      inv3_llt(m, inv, ok);
      if (!ok) *not_pd = 1;
    } else {
      inv3_cofactor(m, inv);
    }
    double* o = ete_inv + 9 * int64_t(p);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = inv[k];
    if (g) { g[3 * int64_t(p)] = gg[0]; g[3 * int64_t(p) + 1] = gg[1]; g[3 * int64_t(p) + 2] = gg[2]; }
  }
}

// ------------------------------------------------- chunk pass of the implicit S
// MODE 0 (SX)     : t = F x_f ;  t' = t - E (E'E)^-1 E' t ; write t'        (S x, first half)
// MODE 1 (RHS)    : t = b     ;  t' = t - E (E'E)^-1 E' t ; write t'        (UpdateRhs)
// MODE 2 (BACKSUB): s = b - F z;  x_pt = (E'E)^-1 E' s                       (BackSubstitute)
// MODE 3 (SPSE)   : t = F x_f ;  out = E (E'E)^-1 E' t                      (power series operator)
// T = float reads the fp32 copies of the cells (mixed-precision CG, modes 0 and 3); arithmetic is fp64.
#ifndef CX_CHUNK_PASS_OCCUPANCY
#define CX_CHUNK_PASS_OCCUPANCY 5
#endif
#ifndef CX_CHUNK_PASS_OCCUPANCY_F32
#define CX_CHUNK_PASS_OCCUPANCY_F32 6
#endif
template <int MODE, typename T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? CX_CHUNK_PASS_OCCUPANCY_F32 : CX_CHUNK_PASS_OCCUPANCY) void k_chunk_pass(const T* __restrict__ E,
                                                       const T* __restrict__ F,
                                                       const int32_t* __restrict__ tile_row,
                                                       const int32_t* __restrict__ tile_pt,
                                                       const int32_t* __restrict__ pt_start,
                                                       const int32_t* __restrict__ row_cam,
                                                       const int32_t* __restrict__ row_pt,
                                                       const double* __restrict__ ete_inv,
                                                       const double* __restrict__ xf,   // camera vector (SX: x, BACKSUB: z)
                                                       const double* __restrict__ b,    // row vector (RHS, BACKSUB)
                                                       double* __restrict__ out,        // t' [2O]  or x_e [3P]
                                                       const int* __restrict__ stop, int big_only) {
  // 18 KB of LDS per workgroup (fp64: F goes through the buffer in two halves) -> the register count, not
  // LDS, sets the occupancy (5 workgroups per CU); measured 3 % faster than the 36 KB version.  The staging
  // buffer is dead once the cells are in registers (the staging ends with a barrier), so the per-row /
  // per-point scratch lives in it.
  __shared__ double lds[FStage<T>::kLdsDoubles];
  double* const w = lds;                 // [kBlock][3] per-row E' t
  double* const u = lds + kBlock * 3;    // [kBlock][3] per-point (E'E)^-1 sum
  double* const red = lds + kBlock * 6;  // block reduction scratch (long chunks only)
  if (stop && *stop) return;
  const int tl = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[tl], r1 = tile_row[tl + 1];
  const int p0 = tile_pt[tl], p1 = tile_pt[tl + 1];
  if (big_only && r1 - r0 <= kBlock) return;
  if (r1 - r0 <= kBlock) {
    const int nvalid = r1 - r0;
    const int r = r0 + tid;
    // F is staged and consumed before E is requested: measured 12 % faster on MI355X than
    // requesting everything up front (fewer live registers, see DESIGN.md "A/B notes").
    // Round 2 A/B: the cooperative camera gather that sped up k_bal_evaluate (gather_by_row, cx_kernels.h; issued while
    // the F cells are in flight, transposed through the staging buffer) made this kernel no faster -- 1.133 against
    // 1.111 ms on the same box; it reads 212 bytes per row where the evaluator reads 24, the gather's requests hide
    // behind the cell stream -- and is not kept here.
    const bool live = tid < nvalid;
    const int lp = (MODE != 2 && live) ? row_pt[r] - p0 : 0;
    double e[6];
    double t0 = 0.0, t1 = 0.0;
    if (MODE != 1) {
      double f[18];
      FStage<T>::run(F + 18 * int64_t(r0), nvalid, lds, f);
      if (live) {
        const double* xc = xf + 9 * int64_t(row_cam[r]);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double xv = xc[k];
          t0 += f[k] * xv;
          t1 += f[9 + k] * xv;
        }
      }
    }
    stage_cells<6>(E + 6 * int64_t(r0), nvalid, lds, e);
    if (live) {
      if (MODE == 1 || MODE == 2) {
        const double2 bv = reinterpret_cast<const double2*>(b)[r];
        if (MODE == 1) { t0 = bv.x; t1 = bv.y; }
        else { t0 = bv.x - t0; t1 = bv.y - t1; }
      }
      w[tid * 3 + 0] = e[0] * t0 + e[3] * t1;
      w[tid * 3 + 1] = e[1] * t0 + e[4] * t1;
      w[tid * 3 + 2] = e[2] * t0 + e[5] * t1;
    }
    // one thread per point: fetch its (E'E)^-1 and chunk bounds before the barrier so the
    // loads overlap the row phase of the other wavefronts
    double m[9];
    int jb = 0, je = 0;
    if (tid < p1 - p0) {
      const double* mp = ete_inv + 9 * int64_t(p0 + tid);
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = mp[k];
      jb = pt_start[p0 + tid] - r0;
      je = pt_start[p0 + tid + 1] - r0;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      const int p = p0 + tid;
      // rows of the chunk, four independent partial sums (fixed combination order)
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
      int j = jb;
      for (; j + 4 <= je; j += 4) {
        a0 += w[j * 3]; a1 += w[j * 3 + 1]; a2 += w[j * 3 + 2];
        b0 += w[j * 3 + 3]; b1 += w[j * 3 + 4]; b2 += w[j * 3 + 5];
        c0 += w[j * 3 + 6]; c1 += w[j * 3 + 7]; c2 += w[j * 3 + 8];
        d0 += w[j * 3 + 9]; d1 += w[j * 3 + 10]; d2 += w[j * 3 + 11];
      }
      for (; j < je; ++j) { a0 += w[j * 3]; a1 += w[j * 3 + 1]; a2 += w[j * 3 + 2]; }
      const double s0 = (a0 + b0) + (c0 + d0), s1 = (a1 + b1) + (c1 + d1), s2 = (a2 + b2) + (c2 + d2);
      const double u0 = m[0] * s0 + m[1] * s1 + m[2] * s2;
      const double u1 = m[3] * s0 + m[4] * s1 + m[5] * s2;
      const double u2 = m[6] * s0 + m[7] * s1 + m[8] * s2;
      if (MODE == 2) {
        double* o = out + 3 * int64_t(p);
        o[0] = u0; o[1] = u1; o[2] = u2;
      } else {
        u[tid * 3] = u0; u[tid * 3 + 1] = u1; u[tid * 3 + 2] = u2;
      }
    }
    if (MODE != 2) {
      __syncthreads();
      if (tid < nvalid) {
        const int lo = lp;  // point of this row, local to the tile
        const double u0 = u[lo * 3], u1 = u[lo * 3 + 1], u2 = u[lo * 3 + 2];
        const double eu0 = e[0] * u0 + e[1] * u1 + e[2] * u2;
        const double eu1 = e[3] * u0 + e[4] * u1 + e[5] * u2;
        if (MODE == 3) { t0 = eu0; t1 = eu1; }
        else { t0 -= eu0; t1 -= eu1; }
        reinterpret_cast<double2*>(out)[r] = make_double2(t0, t1);
      }
    }
  } else {
    // a single point with more than kBlock rows: strided loops and a block reduction
    const int p = p0;
    double s[3] = {0.0, 0.0, 0.0};
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const T* e = E + 6 * int64_t(r);
      double t0 = 0.0, t1 = 0.0;
      if (MODE != 1) {
        const T* f = F + 18 * int64_t(r);
        const double* xc = xf + 9 * int64_t(row_cam[r]);
        for (int k = 0; k < 9; ++k) { t0 += f[k] * xc[k]; t1 += f[9 + k] * xc[k]; }
      }
      if (MODE == 1 || MODE == 2) {
        const double2 bv = reinterpret_cast<const double2*>(b)[r];
        if (MODE == 1) { t0 = bv.x; t1 = bv.y; }
        else { t0 = bv.x - t0; t1 = bv.y - t1; }
      }
      s[0] += e[0] * t0 + e[3] * t1;
      s[1] += e[1] * t0 + e[4] * t1;
      s[2] += e[2] * t0 + e[5] * t1;
    }
    block_sum<3>(s, red);
    const double* m = ete_inv + 9 * int64_t(p);
    const double u0 = m[0] * s[0] + m[1] * s[1] + m[2] * s[2];
    const double u1 = m[3] * s[0] + m[4] * s[1] + m[5] * s[2];
    const double u2 = m[6] * s[0] + m[7] * s[1] + m[8] * s[2];
    if (MODE == 2) {
      if (tid == 0) { double* o = out + 3 * int64_t(p); o[0] = u0; o[1] = u1; o[2] = u2; }
    } else {
      for (int r = r0 + tid; r < r1; r += kBlock) {
        const T* e = E + 6 * int64_t(r);
        double t0 = 0.0, t1 = 0.0;
        if (MODE == 0) {
          const T* f = F + 18 * int64_t(r);
          const double* xc = xf + 9 * int64_t(row_cam[r]);
          for (int k = 0; k < 9; ++k) { t0 += f[k] * xc[k]; t1 += f[9 + k] * xc[k]; }
        } else if (MODE == 1) {
          const double2 bv = reinterpret_cast<const double2*>(b)[r];
          t0 = bv.x; t1 = bv.y;
        }
        const double eu0 = e[0] * u0 + e[1] * u1 + e[2] * u2;
        const double eu1 = e[3] * u0 + e[4] * u1 + e[5] * u2;
        if (MODE == 3) { t0 = eu0; t1 = eu1; }
        else { t0 -= eu0; t1 -= eu1; }
        reinterpret_cast<double2*>(out)[r] = make_double2(t0, t1);
      }
    }
  }
}

// -------------------------------------------------- fused set-up of the implicit S
// ImplicitSchurComplement::Init in two passes over J instead of five:
//   k_chunk_init : one pass over E:  (E'E + D_e^2)^-1 per point (stored) and
//                  t' = b - E (E'E)^-1 E'b per row (first half of UpdateRhs)
//   k_cam_init   : one pass over the camera-major Ft: the per-camera blocks of F'F (or of S) and
//                  F't' (the reduced rhs) from the same registers
// (gathering F inside k_cam_init, to save the separate k_permute_ft, was measured 1.6x slower: two
// dependent loads per piece at 2 workgroups per CU do not hide their latency)
// Same per-point / per-segment summation orders as the separate kernels.
#ifndef CX_CHUNK_INIT_OCCUPANCY
#define CX_CHUNK_INIT_OCCUPANCY 5
#endif
template <bool USE_LLT>
__global__ __launch_bounds__(kBlock, CX_CHUNK_INIT_OCCUPANCY) void k_chunk_init(const double* __restrict__ E,
                                                       const int32_t* __restrict__ tile_row,
                                                       const int32_t* __restrict__ tile_pt,
                                                       const int32_t* __restrict__ pt_start,
                                                       const int32_t* __restrict__ row_pt,
                                                       const double* __restrict__ De,
                                                       const double* __restrict__ b,
                                                       double* __restrict__ ete_inv,
                                                       double* __restrict__ tprime,
                                                       int* __restrict__ not_pd) {
  __shared__ double lds[kBlock * 9];     // staging (6 per row) first, then 9 partials per row
  __shared__ double u[kBlock * 3];
  __shared__ double red[9 * 4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  const int p0 = tile_pt[t], p1 = tile_pt[t + 1];
  if (r1 - r0 <= kBlock) {
    const int nvalid = r1 - r0;
    const bool live = tid < nvalid;
    double e[6];
    stage_cells<6>(E + 6 * int64_t(r0), nvalid, lds, e);
    double2 bv = make_double2(0.0, 0.0);
    int lp = 0;
    if (live) {
      bv = reinterpret_cast<const double2*>(b)[r0 + tid];
      lp = row_pt[r0 + tid] - p0;
      double* wr = lds + tid * 9;
      wr[0] = e[0] * e[0] + e[3] * e[3];
      wr[1] = e[0] * e[1] + e[3] * e[4];
      wr[2] = e[0] * e[2] + e[3] * e[5];
      wr[3] = e[1] * e[1] + e[4] * e[4];
      wr[4] = e[1] * e[2] + e[4] * e[5];
      wr[5] = e[2] * e[2] + e[5] * e[5];
      wr[6] = e[0] * bv.x + e[3] * bv.y;
      wr[7] = e[1] * bv.x + e[4] * bv.y;
      wr[8] = e[2] * bv.x + e[5] * bv.y;
    }
    int jb = 0, je = 0;
    if (tid < p1 - p0) {
      jb = pt_start[p0 + tid] - r0;
      je = pt_start[p0 + tid + 1] - r0;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      const int p = p0 + tid;
      double sacc[9], s2[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { sacc[k] = 0.0; s2[k] = 0.0; }
      int j = jb;
      for (; j + 2 <= je; j += 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { sacc[k] += lds[j * 9 + k]; s2[k] += lds[j * 9 + 9 + k]; }
      }
      if (j < je) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sacc[k] += lds[j * 9 + k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) sacc[k] += s2[k];
      double m[9], inv[9];
      m[0] = sacc[0]; m[1] = sacc[1]; m[2] = sacc[2]; m[4] = sacc[3]; m[5] = sacc[4]; m[8] = sacc[5];
      if (De) {
        const double* d = De + 3 * int64_t(p);
        m[0] += d[0] * d[0]; m[4] += d[1] * d[1]; m[8] += d[2] * d[2];
      }
      m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
      if (USE_LLT) {
        bool ok;
        inv3_llt(m, inv, ok);
        if (!ok) *not_pd = 1;
      } else {
        inv3_cofactor(m, inv);
      }
      double* o = ete_inv + 9 * int64_t(p);
#pragma unroll
      for (int k = 0; k < 9; ++k) o[k] = inv[k];
      u[tid * 3] = inv[0] * sacc[6] + inv[1] * sacc[7] + inv[2] * sacc[8];
      u[tid * 3 + 1] = inv[3] * sacc[6] + inv[4] * sacc[7] + inv[5] * sacc[8];
      u[tid * 3 + 2] = inv[6] * sacc[6] + inv[7] * sacc[7] + inv[8] * sacc[8];
    }
    __syncthreads();
    if (live) {
      const double u0 = u[lp * 3], u1 = u[lp * 3 + 1], u2 = u[lp * 3 + 2];
      reinterpret_cast<double2*>(tprime)[r0 + tid] =
          make_double2(bv.x - (e[0] * u0 + e[1] * u1 + e[2] * u2), bv.y - (e[3] * u0 + e[4] * u1 + e[5] * u2));
    }
  } else {
    // one long chunk: strided loops and a block reduction
    double sacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) sacc[k] = 0.0;
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const double* e = E + 6 * int64_t(r);
      const double2 bv = reinterpret_cast<const double2*>(b)[r];
      sacc[0] += e[0] * e[0] + e[3] * e[3];
      sacc[1] += e[0] * e[1] + e[3] * e[4];
      sacc[2] += e[0] * e[2] + e[3] * e[5];
      sacc[3] += e[1] * e[1] + e[4] * e[4];
      sacc[4] += e[1] * e[2] + e[4] * e[5];
      sacc[5] += e[2] * e[2] + e[5] * e[5];
      sacc[6] += e[0] * bv.x + e[3] * bv.y;
      sacc[7] += e[1] * bv.x + e[4] * bv.y;
      sacc[8] += e[2] * bv.x + e[5] * bv.y;
    }
    block_sum<9>(sacc, red);
    double m[9], inv[9];
    m[0] = sacc[0]; m[1] = sacc[1]; m[2] = sacc[2]; m[4] = sacc[3]; m[5] = sacc[4]; m[8] = sacc[5];
    if (De) {
      const double* d = De + 3 * int64_t(p0);
      m[0] += d[0] * d[0]; m[4] += d[1] * d[1]; m[8] += d[2] * d[2];
    }
    m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
    if (USE_LLT) {
      bool ok;
      inv3_llt(m, inv, ok);
      if (!ok && tid == 0) *not_pd = 1;
    } else {
      inv3_cofactor(m, inv);
    }
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) ete_inv[9 * int64_t(p0) + k] = inv[k];
    }
    const double u0 = inv[0] * sacc[6] + inv[1] * sacc[7] + inv[2] * sacc[8];
    const double u1 = inv[3] * sacc[6] + inv[4] * sacc[7] + inv[5] * sacc[8];
    const double u2 = inv[6] * sacc[6] + inv[7] * sacc[7] + inv[8] * sacc[8];
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const double* e = E + 6 * int64_t(r);
      const double2 bv = reinterpret_cast<const double2*>(b)[r];
      reinterpret_cast<double2*>(tprime)[r] =
          make_double2(bv.x - (e[0] * u0 + e[1] * u1 + e[2] * u2), bv.y - (e[3] * u0 + e[4] * u1 + e[5] * u2));
    }
  }
}

// One workgroup per camera-major segment of Ft.
#ifndef CX_CAM_INIT_OCCUPANCY
#define CX_CAM_INIT_OCCUPANCY 2
#endif
template <bool WITH_SCHUR>
__global__ __launch_bounds__(kBlock, CX_CAM_INIT_OCCUPANCY) void k_cam_init(const double* __restrict__ F, const double* __restrict__ E,
                                                     const int32_t* __restrict__ cam_rows,
                                                     const int32_t* __restrict__ row_pt,
                                                     const int32_t* __restrict__ seg_begin,
                                                     const double* __restrict__ ete_inv,
                                                     const double* __restrict__ tprime,   // may be null
                                                     const double* __restrict__ Ft,
                                                     double* __restrict__ partial45, double* __restrict__ partial9,
                                                     int num_segs) {
  __shared__ double lds[kBlock * 18];
  __shared__ double red[45 * 4];
  const int sgm = xcd_segment(num_segs), tid = threadIdx.x;
  if (sgm < 0) return;
  const int b0 = seg_begin[sgm], e0 = seg_begin[sgm + 1];
  double acc[45], acc9[9];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc9[k] = 0.0;
  // Software pipeline over the segment's passes of kBlock rows (round 2): the cells, the row ids and the gathered t' of pass
  // k + 1 are requested before the arithmetic of pass k.  With 178 registers for the 54 accumulators only two workgroups
  // fit on a CU, too few to hide a pass's load -> gather chain by themselves; the 40 registers of the prefetch are free
  // (the kernel stays under 256).
  // (the variant with the Schur terms has 234 registers already and keeps the plain loop)
  constexpr bool kPrefetch = !WITH_SCHUR;
  double2 v[9];
  int r_next = 0;
  double2 tv_next = make_double2(0.0, 0.0);
  if (kPrefetch) {
    load_cells<18>(Ft + 18 * int64_t(b0), min(kBlock, e0 - b0), v);
    if (b0 + tid < e0) {
      r_next = cam_rows[b0 + tid];
      if (tprime) tv_next = reinterpret_cast<const double2*>(tprime)[r_next];
    }
  }
  for (int k0 = b0; k0 < e0; k0 += kBlock) {
    const int nvalid = min(kBlock, e0 - k0);
    double f[18];
    if (!kPrefetch) {
      load_cells<18>(Ft + 18 * int64_t(k0), nvalid, v);
      if (tid < nvalid) {
        r_next = cam_rows[k0 + tid];
        if (tprime) tv_next = reinterpret_cast<const double2*>(tprime)[r_next];
      }
    }
    exchange_cells<18>(v, lds, f);
    const int r = r_next;
    const double2 tv = tv_next;
    if (kPrefetch && k0 + kBlock < e0) {
      load_cells<18>(Ft + 18 * int64_t(k0 + kBlock), min(kBlock, e0 - k0 - kBlock), v);
      if (k0 + kBlock + tid < e0) {
        r_next = cam_rows[k0 + kBlock + tid];
        if (tprime) tv_next = reinterpret_cast<const double2*>(tprime)[r_next];
      }
    }
    if (tid < nvalid) {
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int c = a; c < 9; ++c) acc[a * 9 - a * (a - 1) / 2 + c - a] += f[a] * f[c] + f[9 + a] * f[9 + c];
      if (WITH_SCHUR) {
        const double* e = E + 6 * int64_t(r);
        const double* m = ete_inv + 9 * int64_t(row_pt[r]);
        double e6[6], mi[9];
#pragma unroll
        for (int k = 0; k < 6; ++k) e6[k] = e[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) mi[k] = m[k];
        double B[27], G[27];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int a = 0; a < 9; ++a) B[q * 9 + a] = e6[q] * f[a] + e6[3 + q] * f[9 + a];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int a = 0; a < 9; ++a) G[q * 9 + a] = mi[q * 3] * B[a] + mi[q * 3 + 1] * B[9 + a] + mi[q * 3 + 2] * B[18 + a];
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
          for (int c = a; c < 9; ++c)
            acc[a * 9 - a * (a - 1) / 2 + c - a] -= B[a] * G[c] + B[9 + a] * G[9 + c] + B[18 + a] * G[18 + c];
      }
      if (tprime) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc9[k] += f[k] * tv.x + f[9 + k] * tv.y;
      }
    }
  }
  static_assert(kBlock == 256, "block_sum_store_multi adds four wavefronts");
  block_sum_store_multi<45>(acc, red, partial45 + int64_t(sgm) * 45);
  if (tprime) block_sum_store_multi<9>(acc9, red, partial9 + int64_t(sgm) * 9);
}

// ---------------------------------------------- camera-major 9x9 block diagonals
// WITH_SCHUR = false: partial[seg] = sum F_r' F_r                      (block diagonal of F'F)
// WITH_SCHUR = true : partial[seg] = sum F_r'F_r - (E_r'F_r)'(E'E)^-1(E_r'F_r)   (block diagonal of S)
// 45 upper-triangle entries per segment, packed row-major (a <= b).
#ifndef CX_CAM_DIAG_OCCUPANCY
#define CX_CAM_DIAG_OCCUPANCY 3
#endif
template <bool WITH_SCHUR>
__global__ __launch_bounds__(kBlock, CX_CAM_DIAG_OCCUPANCY) void k_cam_diag(const double* __restrict__ Ft,
                                                     const double* __restrict__ E,
                                                     const int32_t* __restrict__ cam_rows,
                                                     const int32_t* __restrict__ row_pt,
                                                     const int32_t* __restrict__ seg_begin,
                                                     const double* __restrict__ ete_inv,
                                                     double* __restrict__ partial) {  // [S][45]
  __shared__ double lds[kBlock * 18];
  __shared__ double red[45 * 4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int b = seg_begin[s], e_ = seg_begin[s + 1];
  double acc[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
  for (int k0 = b; k0 < e_; k0 += kBlock) {
    const int nvalid = min(kBlock, e_ - k0);
    double f[18];
    stage_cells<18>(Ft + 18 * int64_t(k0), nvalid, lds, f);
    if (tid < nvalid) {
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int c = a; c < 9; ++c) acc[a * 9 - a * (a - 1) / 2 + c - a] += f[a] * f[c] + f[9 + a] * f[9 + c];
      if (WITH_SCHUR) {
        const int r = cam_rows[k0 + tid];
        const double* e = E + 6 * int64_t(r);
        const double* m = ete_inv + 9 * int64_t(row_pt[r]);
        double e6[6], mi[9];
#pragma unroll
        for (int k = 0; k < 6; ++k) e6[k] = e[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) mi[k] = m[k];
        // B = E'F (3x9), G = inv B (3x9)
        double B[27], G[27];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int a = 0; a < 9; ++a) B[q * 9 + a] = e6[q] * f[a] + e6[3 + q] * f[9 + a];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int a = 0; a < 9; ++a) G[q * 9 + a] = mi[q * 3] * B[a] + mi[q * 3 + 1] * B[9 + a] + mi[q * 3 + 2] * B[18 + a];
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
          for (int c = a; c < 9; ++c)
            acc[a * 9 - a * (a - 1) / 2 + c - a] -= B[a] * G[c] + B[9 + a] * G[9 + c] + B[18 + a] * G[18 + c];
      }
    }
  }
  block_sum_store_multi<45>(acc, red, partial + int64_t(s) * 45);
}

// out[9c + k] = sum of the camera's 9-wide segment partials, in segment order
__global__ void k_sum_segments9(const double* __restrict__ partial, const int32_t* __restrict__ cam_seg_start,
                                double* __restrict__ out, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 9) return;
  const int c = i / 9, k = i - c * 9;
  double sum = 0.0;
  for (int sg = cam_seg_start[c]; sg < cam_seg_start[c + 1]; ++sg) sum += partial[int64_t(sg) * 9 + k];
  out[i] = sum;
}

// blocks[c] (81, row-major, full) = sum of the camera's packed segment partials
__global__ void k_cam_diag_reduce(const double* __restrict__ partial, const int32_t* __restrict__ cam_seg_start,
                                  double* __restrict__ blocks, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 45) return;
  const int c = i / 45, k = i - c * 45;
  double s = 0.0;
  for (int sg = cam_seg_start[c]; sg < cam_seg_start[c + 1]; ++sg) s += partial[int64_t(sg) * 45 + k];
  // unpack k -> (a, b), a <= b
  int a = 0, rem = k;
  while (rem >= 9 - a) { rem -= 9 - a; ++a; }
  const int bcol = a + rem;
  blocks[int64_t(c) * 81 + a * 9 + bcol] = s;
  blocks[int64_t(c) * 81 + bcol * 9 + a] = s;
}

// blocks[i] <- (blocks[i] + diag(D_i^2))^-1 through LLT on the upper triangle
// (BlockRandomAccessDiagonalMatrix::Invert, block_random_access_diagonal_matrix.cc:90-100;
// AddDiagonalAndInvert).  Nine threads per 9x9 block, 64 blocks per workgroup: the block goes into its LDS slab, the first
// of its threads factors it in place, then each thread solves U'U x = e_col for its own column -- entry by entry the
// arithmetic of one thread per block (round 3: 11.6 us for 49 blocks, most of it the nine column solves one after another).
constexpr int kInvertBlocks = 64;
__global__ __launch_bounds__(kInvertBlocks * 9) void k_block9_add_diag_invert(double* __restrict__ blocks,
                                                                              const double* __restrict__ Df, int C,
                                                                              int* __restrict__ not_pd) {
  __shared__ double slab[kInvertBlocks * 81];  // stride 81 doubles: odd in 8-byte words
  const int c0 = blockIdx.x * kInvertBlocks;
  const int nb = min(kInvertBlocks, C - c0);
  for (int k = threadIdx.x; k < nb * 81; k += kInvertBlocks * 9) slab[k] = blocks[int64_t(c0) * 81 + k];
  __syncthreads();
  const int cl = threadIdx.x / 9, col = threadIdx.x - cl * 9;
  const int c = c0 + cl;
  double* U = slab + cl * 81;
  if (cl < nb && col == 0) {
    bool ok = true;
    for (int j = 0; j < 9; ++j) {
      for (int i = 0; i <= j; ++i) {
        double s = U[i * 9 + j];
        if (i == j && Df) s += Df[9 * int64_t(c) + j] * Df[9 * int64_t(c) + j];
        for (int k = 0; k < i; ++k) s -= U[k * 9 + i] * U[k * 9 + j];
        if (i == j) {
          if (!(s > 0.0)) ok = false;
          U[i * 9 + i] = sqrt(s);
        } else {
          U[i * 9 + j] = s / U[i * 9 + i];
        }
      }
    }
    if (!ok) *not_pd = 1;
  }
  __syncthreads();
  if (cl >= nb) return;
  // solve U'U x = e_col
  double y[9], xc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double s = (i == col) ? 1.0 : 0.0;
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
  double* A = blocks + int64_t(c) * 81;
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i * 9 + col] = xc[i];
}

// ------------------------------------------------------- explicit S (dense lhs)
// SchurEliminator<2,3,9>::Eliminate (schur_eliminator_impl.h:177-561) for a dense reduced
// matrix (BlockRandomAccessDenseMatrix): upper block triangle, full diagonal blocks.
//   S(c,c)  += F'F                 EBlockRowOuterProduct :665-714  -> camera-major k_cam_diag (deterministic)
//   rhs      = F'(b - E inv E'b)   UpdateRhs :379-420              -> k_chunk_pass<1> + k_cam_ft (deterministic)
//   S(c1,c2) -= B1' inv B2         ChunkOuterProduct :512-561      -> here, fp64 atomics
// The pair products of different chunks meet in the same 9x9 cell; the reference serialises
// them with a mutex per cell (:550), here they are global_atomic_add_f64 into a BLOCK-MAJOR
// copy of S (cell (c1,c2) = 81 contiguous doubles), so that one wavefront instruction adds 512
// contiguous bytes -- the shape float atomics run at full rate on gfx950.  Like the reference
// with num_threads > 1 the sum order across chunks is not fixed.
constexpr int kMaxPtsPerTile = kBlock;
__global__ __launch_bounds__(kBlock) void k_chunk_eliminate(const double* __restrict__ E,
                                                            const double* __restrict__ F,
                                                            const int32_t* __restrict__ tile_row,
                                                            const int32_t* __restrict__ tile_pt,
                                                            const int32_t* __restrict__ pt_start,
                                                            const int32_t* __restrict__ row_cam,
                                                            const double* __restrict__ ete_inv,
                                                            double* __restrict__ blk, int C) {
  __shared__ double Bs[kBlock * 27];     // staging first, then B_r = E_r' F_r (3x9) per row
  __shared__ double inv_s[kMaxPtsPerTile * 9];
  __shared__ int cam_s[kBlock];
  __shared__ int pair_start[kMaxPtsPerTile + 1];
  __shared__ int start_s[kMaxPtsPerTile + 1];
  const int tl = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[tl], r1 = tile_row[tl + 1];
  const int p0 = tile_pt[tl], p1 = tile_pt[tl + 1];
  const int npts = p1 - p0;
  if (r1 - r0 > kBlock) return;  // long chunks: k_big_chunk_eliminate
  const int nvalid = r1 - r0;
  double2 fv[9], ev[3];
  load_cells<18>(F + 18 * int64_t(r0), nvalid, fv);
  load_cells<6>(E + 6 * int64_t(r0), nvalid, ev);
  if (tid < nvalid) cam_s[tid] = row_cam[r0 + tid];
  if (tid < npts) {
    const double* m = ete_inv + 9 * int64_t(p0 + tid);
#pragma unroll
    for (int k = 0; k < 9; ++k) inv_s[tid * 9 + k] = m[k];
  }
  if (tid <= npts) start_s[tid] = pt_start[p0 + tid] - r0;
  double f[18], e[6];
  exchange_cells<18>(fv, Bs, f);
  exchange_cells<6>(ev, Bs, e);
  if (tid < nvalid) {
    double* Br = Bs + tid * 27;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int a = 0; a < 9; ++a) Br[q * 9 + a] = e[q] * f[a] + e[3 + q] * f[9 + a];
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < npts; ++i) {
      pair_start[i] = acc;
      const int k = start_s[i + 1] - start_s[i];
      acc += k * (k + 1) / 2;
    }
    pair_start[npts] = acc;
  }
  __syncthreads();
  const int total = pair_start[npts] * 81;
  for (int item = tid; item < total; item += kBlock) {
    const int pair = item / 81, el = item - pair * 81;
    int lo = 0, hi = npts - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pair_start[mid] <= pair) lo = mid; else hi = mid - 1;
    }
    const int base = start_s[lo];
    const int k = start_s[lo + 1] - base;
    int i, j;
    tri_decode(pair - pair_start[lo], k, i, j);
    int ri = base + i, rj = base + j;
    int c1 = cam_s[ri], c2 = cam_s[rj];
    if (c1 > c2) { const int tq = c1; c1 = c2; c2 = tq; const int tr = ri; ri = rj; rj = tr; }
    const int a = el / 9, c = el - a * 9;
    const double* B1 = Bs + ri * 27;
    const double* B2 = Bs + rj * 27;
    const double* iv = inv_s + lo * 9;
    double sum = 0.0;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      const double g = iv[pp * 3] * B2[c] + iv[pp * 3 + 1] * B2[9 + c] + iv[pp * 3 + 2] * B2[18 + c];
      sum += B1[pp * 9 + a] * g;
    }
    atomicAdd(&blk[(int64_t(c1) * C + c2) * 81 + el], -sum);
  }
}

// Long chunks (> kBlock rows): one workgroup per such chunk, pairs straight from global memory
// (rare; correctness path).
__global__ __launch_bounds__(kBlock) void k_big_chunk_eliminate(const double* __restrict__ E,
                                                                const double* __restrict__ F,
                                                                const int32_t* __restrict__ tile_row,
                                                                const int32_t* __restrict__ tile_pt,
                                                                const int32_t* __restrict__ row_cam,
                                                                const double* __restrict__ ete_inv,
                                                                double* __restrict__ blk, int C) {
  const int tl = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[tl], r1 = tile_row[tl + 1];
  if (r1 - r0 <= kBlock) return;
  const double* inv_s = ete_inv + 9 * int64_t(tile_pt[tl]);
  const int k = r1 - r0;
  const int64_t total = int64_t(k) * (k + 1) / 2;
  for (int64_t pair = tid; pair < total; pair += kBlock) {
    int i, j;
    tri_decode(pair, k, i, j);
    int ri = r0 + i, rj = r0 + j;
    int c1 = row_cam[ri], c2 = row_cam[rj];
    if (c1 > c2) { const int tq = c1; c1 = c2; c2 = tq; const int tr = ri; ri = rj; rj = tr; }
    double B1[27], G2[27];
    {
      const double* e = E + 6 * int64_t(ri);
      const double* f = F + 18 * int64_t(ri);
      for (int qq = 0; qq < 3; ++qq)
        for (int a = 0; a < 9; ++a) B1[qq * 9 + a] = e[qq] * f[a] + e[3 + qq] * f[9 + a];
    }
    {
      const double* e = E + 6 * int64_t(rj);
      const double* f = F + 18 * int64_t(rj);
      double B2[27];
      for (int qq = 0; qq < 3; ++qq)
        for (int a = 0; a < 9; ++a) B2[qq * 9 + a] = e[qq] * f[a] + e[3 + qq] * f[9 + a];
      for (int qq = 0; qq < 3; ++qq)
        for (int a = 0; a < 9; ++a)
          G2[qq * 9 + a] = inv_s[qq * 3] * B2[a] + inv_s[qq * 3 + 1] * B2[9 + a] + inv_s[qq * 3 + 2] * B2[18 + a];
    }
    double* dst = blk + (int64_t(c1) * C + c2) * 81;
    for (int a = 0; a < 9; ++a)
      for (int c = 0; c < 9; ++c)
        atomicAdd(&dst[a * 9 + c], -(B1[a] * G2[c] + B1[9 + a] * G2[9 + c] + B1[18 + a] * G2[18 + c]));
  }
}

// ---- explicit S without atomics (the default): gather instead of scatter.
// The structure is fixed over the LM iterations, so the list of row pairs (ri, rj) of one chunk whose
// cameras are (c1, c2) is built once per matrix, grouped by the S cell they update (the role of the
// reference's cell map + mutex, block_random_access_dense_matrix.cc:41-72, schur_eliminator_impl.h:550).
// Per solve: k_row_bg writes B_r = E_r' F_r and G_r = (E'E + D^2)^-1 B_r for every row (one streaming
// pass over J), then one workgroup per cell adds its pairs' -B_ri' G_rj in list order and writes the cell
// of the dense matrix.  Sums are in a fixed order: bitwise reproducible, unlike the reference with
// num_threads > 1.
constexpr int64_t kMaxPairs = int64_t(1) << 28;
constexpr int kPairGroups = 28;             // 28 groups of 9 threads, each thread a 3x3 piece of the 9x9 product
constexpr int kPairItem = kPairGroups * 8;  // pairs per work item

// Host half of the gather assembly's structure: everything cxs_build_pair_lists derives from the block structure alone.
// No device is touched (the sanitizer builds of tools/sanitize run it on the CPU), threads: up to 16.
struct PairListsHost {
  int64_t total = 0;            // pairs
  bool too_many = false;        // more than kMaxPairs (or CX_ELIM_ATOMICS=1): the atomics path is used instead
  std::unique_ptr<int32_t[]> pairs;  // [2 * total] (row of camera c1, row of camera c2) per pair, cells in order, chunk order inside
  std::vector<int32_t> cell_c1, cell_c2, cell_item_start, row_cells, col_count, col_cells, item_order;
  std::vector<int64_t> item_begin;
  int64_t num_items = 0, num_cells = 0;
};

static int BuildPairListsHost(const cx_cell* cells, int32_t P, int32_t C, int64_t O, PairListsHost* out) {
  const bool verbose = std::getenv("CX_SPARSE_CHOLESKY_VERBOSE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (verbose) std::fprintf(stderr, "[cxschur] pair lists: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  int64_t total = 0;
  std::vector<int32_t> start(size_t(P) + 1, 0);
  {
    // rows are sorted by point: chunk of point p = rows with cells[2r].block_id == p
    int64_t r = 0;
    for (int p = 0; p < P; ++p) {
      start[p] = int32_t(r);
      while (r < O && cells[2 * r].block_id == p) ++r;
      const int64_t k = r - start[p];
      total += k * (k + 1) / 2;
    }
    start[P] = int32_t(r);
  }
  out->total = total;
  // CX_ELIM_ATOMICS=1 selects the scatter (fp64 atomics) path, otherwise only used when the list would be huge
  const char* force = std::getenv("CX_ELIM_ATOMICS");
  if (total > kMaxPairs || (force && force[0] == '1')) {
    out->too_many = true;
    return CX_OK;
  }
  auto cam_of = [&](int64_t r) { return cells[2 * r + 1].block_id - P; };
  // 1. bucket the pairs by their smaller camera c1 (stable: chunk order inside a bucket).  Worker threads take
  //    contiguous ranges of points balanced by pair count; thread t's pairs of a bucket go behind those of threads
  //    < t, which is the chunk order (one thread walking 164 M pairs of the Final shape twice took 1.1 s of the
  //    1.9 s one-time set-up of SPARSE_SCHUR).
  struct Pair { int32_t c2, ri, rj; };
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<int> range(size_t(hw) + 1, P);
  {
    range[0] = 0;
    int64_t acc = 0;
    unsigned t = 1;
    for (int p = 0; p < P && t < hw; ++p) {
      const int64_t k = start[p + 1] - start[p];
      acc += k * (k + 1) / 2;
      if (acc * int64_t(hw) >= total * int64_t(t)) range[t++] = p + 1;
    }
  }
  std::vector<std::vector<int64_t>> count(hw, std::vector<int64_t>(size_t(C), 0));  // then: fill cursors
  {
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < hw; ++t)
      workers.emplace_back([&, t]() {
        std::vector<int64_t>& cnt = count[t];
        for (int p = range[t]; p < range[t + 1]; ++p)
          for (int64_t i = start[p]; i < start[p + 1]; ++i)
            for (int64_t j = i; j < start[p + 1]; ++j) cnt[size_t(std::min(cam_of(i), cam_of(j)))]++;
      });
    for (auto& w : workers) w.join();
  }
  std::vector<int64_t> bucket(size_t(C) + 1, 0);
  for (int c = 0; c < C; ++c) {
    int64_t acc = bucket[c];
    for (unsigned t = 0; t < hw; ++t) {
      const int64_t n = count[t][size_t(c)];
      count[t][size_t(c)] = acc;
      acc += n;
    }
    bucket[c + 1] = acc;
  }
  // (not a std::vector: its value-initialisation would touch the 2 GB from one thread before the workers fill them)
  std::unique_ptr<Pair[]> tmp(new Pair[static_cast<size_t>(std::max<int64_t>(total, 1))]);
  {
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < hw; ++t)
      workers.emplace_back([&, t]() {
        std::vector<int64_t>& fill = count[t];
        for (int p = range[t]; p < range[t + 1]; ++p)
          for (int64_t i = start[p]; i < start[p + 1]; ++i)
            for (int64_t j = i; j < start[p + 1]; ++j) {
              const int ci = cam_of(i), cj = cam_of(j);
              Pair& q = tmp[size_t(fill[size_t(std::min(ci, cj))]++)];
              q.c2 = std::max(ci, cj);
              q.ri = int32_t(ci <= cj ? i : j);
              q.rj = int32_t(ci <= cj ? j : i);
            }
      });
    for (auto& w : workers) w.join();
  }
  lap("bucket by first camera");
  // 2. inside every bucket a stable counting sort by c2 gives the cells of block row c1 in order, each with
  //    its pairs in chunk order; the diagonal cell (c1, c1) exists even without pairs.  Buckets are
  //    independent: worker threads take them round-robin.
  std::vector<int32_t>& row_cells = out->row_cells;  // cells per block row, then prefix sums
  row_cells.assign(size_t(C) + 1, 0);
  std::unique_ptr<int32_t[]> pairs(new int32_t[static_cast<size_t>(std::max<int64_t>(2 * total, 1))]);
  std::vector<std::vector<int32_t>> row_c2(static_cast<size_t>(C));     // c2 of the row's cells
  std::vector<std::vector<int64_t>> row_begin(static_cast<size_t>(C));  // first pair of each of them
  {
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < hw; ++t)
      workers.emplace_back([&, t]() {
        std::vector<int64_t> cnt(size_t(C) + 1);
        for (int c1 = int(t); c1 < C; c1 += int(hw)) {
          const int64_t b0 = bucket[c1], b1 = bucket[c1 + 1];
          std::fill(cnt.begin() + c1, cnt.end(), 0);
          for (int64_t k = b0; k < b1; ++k) cnt[size_t(tmp[size_t(k)].c2) + 1]++;
          std::vector<int32_t>& c2s = row_c2[size_t(c1)];
          std::vector<int64_t>& begins = row_begin[size_t(c1)];
          int64_t acc = b0;
          for (int c2 = c1; c2 < C; ++c2) {
            const int64_t n = cnt[size_t(c2) + 1];
            cnt[size_t(c2) + 1] = acc;  // becomes the fill cursor of c2
            if (n > 0 || c2 == c1) {
              c2s.push_back(c2);
              begins.push_back(acc);
            }
            acc += n;
          }
          for (int64_t k = b0; k < b1; ++k) {
            const Pair& q = tmp[size_t(k)];
            const int64_t slot = cnt[size_t(q.c2) + 1]++;
            pairs[size_t(2 * slot)] = q.ri;
            pairs[size_t(2 * slot + 1)] = q.rj;
          }
        }
      });
    for (auto& w : workers) w.join();
  }
  tmp.reset();
  lap("sort by second camera");
  for (int c = 0; c < C; ++c) row_cells[size_t(c) + 1] = row_cells[size_t(c)] + int32_t(row_c2[size_t(c)].size());
  const int64_t ncell = row_cells[size_t(C)];
  out->cell_c1.resize(size_t(ncell));
  out->cell_c2.resize(size_t(ncell));
  std::vector<int64_t>& item_begin = out->item_begin;
  item_begin.clear();
  std::vector<int32_t>& cell_item_start = out->cell_item_start;
  cell_item_start.assign(size_t(ncell) + 1, 0);
  std::vector<int32_t>& col_count = out->col_count;
  col_count.assign(size_t(C) + 1, 0);
  for (int c1 = 0; c1 < C; ++c1) {
    const auto& c2s = row_c2[size_t(c1)];
    const auto& begins = row_begin[size_t(c1)];
    for (size_t k = 0; k < c2s.size(); ++k) {
      const int64_t cell = row_cells[size_t(c1)] + int64_t(k);
      out->cell_c1[size_t(cell)] = c1;
      out->cell_c2[size_t(cell)] = c2s[k];
      if (c2s[k] != c1) col_count[size_t(c2s[k]) + 1]++;
      // work items: the cell's pair run cut into pieces of at most kPairItem pairs
      const int64_t b0 = begins[k];
      const int64_t b1 = (k + 1 < c2s.size()) ? begins[k + 1] : bucket[size_t(c1) + 1];
      cell_item_start[size_t(cell)] = int32_t(item_begin.size());
      for (int64_t q = b0; q < b1; q += kPairItem) item_begin.push_back(q);
    }
  }
  cell_item_start[size_t(ncell)] = int32_t(item_begin.size());
  out->num_items = int64_t(item_begin.size());
  item_begin.push_back(total);  // consecutive items meet at a cell boundary, so item i ends where item i + 1 begins
  // transposed index: the off-diagonal cells of block column c2, by ascending c1
  for (int c = 0; c < C; ++c) col_count[size_t(c) + 1] += col_count[size_t(c)];
  std::vector<int32_t>& col_cells = out->col_cells;
  col_cells.assign(static_cast<size_t>(col_count[size_t(C)]), 0);
  {
    std::vector<int32_t> fill(col_count.begin(), col_count.end() - 1);
    for (int64_t cell = 0; cell < ncell; ++cell)
      if (out->cell_c1[size_t(cell)] != out->cell_c2[size_t(cell)]) col_cells[size_t(fill[size_t(out->cell_c2[size_t(cell)])]++)] = int32_t(cell);
  }
  lap("cell and item index");
  out->pairs = std::move(pairs);
  out->num_cells = ncell;
  // Launch order of the items (a permutation; sums do not depend on it): kItemRowGroup consecutive block rows of S at a
  // time, inside a group by block column -- the cells (c1 .. c1 + g - 1, c2) follow each other, so the right operands (rows
  // of camera c2, largely the same points for neighbouring c1) are fetched once per group instead of once per block row,
  // while the left operands of the group's cameras still fit an XCD's L2.  CX_PAIR_ROW_GROUP=1 restores cell order.
  {
    static const int group = std::max(1, std::getenv("CX_PAIR_ROW_GROUP") ? std::atoi(std::getenv("CX_PAIR_ROW_GROUP")) : 4);
    struct Key { int32_t g, c2, c1, item; };
    std::vector<Key> keys(static_cast<size_t>(out->num_items));
    for (int64_t cell = 0; cell < ncell; ++cell)
      for (int32_t it = out->cell_item_start[size_t(cell)]; it < out->cell_item_start[size_t(cell) + 1]; ++it)
        keys[size_t(it)] = Key{out->cell_c1[size_t(cell)] / group, out->cell_c2[size_t(cell)], out->cell_c1[size_t(cell)], it};
    if (group > 1)
      std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        return a.g != b.g ? a.g < b.g : (a.c2 != b.c2 ? a.c2 < b.c2 : (a.c1 != b.c1 ? a.c1 < b.c1 : a.item < b.item));
      });
    out->item_order.resize(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) out->item_order[i] = keys[i].item;
  }
  return CX_OK;
}

int cxs_build_pair_lists(cx_matrix* A) {
  if (A->pairs_state != 0) return CX_OK;
  PairListsHost h;
  CX_TRY(BuildPairListsHost(A->cells.data(), A->P, A->C, A->O, &h));
  if (h.too_many) {
    A->pairs_state = 2;
    return CX_OK;
  }
  const int64_t total = h.total;
  A->h_cell_c1 = h.cell_c1;
  A->h_cell_c2 = h.cell_c2;
  A->num_items = h.num_items;
  hipStream_t st = A->ctx->stream;
  CX_TRY(A->d_pair_rows.upload(h.pairs.get(), static_cast<size_t>(2 * total), st));
  CX_TRY(A->d_item_begin.upload(h.item_begin, st));
  CX_TRY(A->d_cell_item_start.upload(h.cell_item_start, st));
  A->h_cell_item_start = h.cell_item_start;
  CX_TRY(A->d_cell_c1.upload(A->h_cell_c1, st));
  CX_TRY(A->d_cell_c2.upload(A->h_cell_c2, st));
  CX_TRY(A->d_cell_row_start.upload(h.row_cells, st));
  CX_TRY(A->d_col_cell_start.upload(h.col_count, st));
  CX_TRY(A->d_col_cells.upload(h.col_cells, st));
  CX_TRY(A->d_item_partial.alloc(size_t(std::max<int64_t>(A->num_items, 1)) * 81));
  CX_TRY(A->d_item_order.upload(h.item_order, st));
  A->num_pairs = total;
  A->num_cells = h.num_cells;
  A->pairs_state = 1;
  return CX_OK;
}

// The same structure from the flat block structure alone, for callers without a device (cxschur.h)
extern "C" int cx_schur_pair_lists_host(const cx_block_structure* bs, int32_t num_eliminate_blocks, int64_t* num_cells,
                                        int64_t* num_pairs, int64_t* num_items, int32_t* cell_row, int32_t* cell_col,
                                        int64_t cell_capacity, int32_t* pair_rows, int64_t pair_capacity) {
  CX_CHECK_ARG(bs != nullptr && num_cells != nullptr && num_pairs != nullptr && num_items != nullptr);
  const int32_t P = num_eliminate_blocks, C = bs->num_col_blocks - P;
  const int64_t O = bs->num_row_blocks;
  CX_CHECK_ARG(P > 0 && C > 0 && cell_capacity >= 0 && pair_capacity >= 0);
  for (int64_t r = 0; r < O; ++r) {  // the static layout: two cells per row, e-block first, rows sorted by e-block
    const bool ok = bs->row_cell_begin[r] == 2 * r && bs->row_cell_begin[r + 1] == 2 * r + 2 && bs->cells[2 * r].block_id < P &&
                    bs->cells[2 * r + 1].block_id >= P && (r == 0 || bs->cells[2 * r].block_id >= bs->cells[2 * r - 2].block_id);
    if (!ok) {
      cx_set_error("row block %lld does not have the static two-cell layout", (long long)r);
      return CX_ERR_UNSUPPORTED;
    }
  }
  PairListsHost h;
  CX_TRY(BuildPairListsHost(bs->cells, P, C, O, &h));
  *num_pairs = h.total;
  *num_cells = h.too_many ? -1 : h.num_cells;
  *num_items = h.too_many ? -1 : h.num_items;
  if (h.too_many) return CX_OK;
  if (cell_row && cell_col && cell_capacity >= h.num_cells) {
    std::copy(h.cell_c1.begin(), h.cell_c1.end(), cell_row);
    std::copy(h.cell_c2.begin(), h.cell_c2.end(), cell_col);
  }
  if (pair_rows && pair_capacity >= 2 * h.total) std::copy(h.pairs.get(), h.pairs.get() + 2 * h.total, pair_rows);
  return CX_OK;
}

// B_r and G_r of every row, 54 doubles per row in three [O][18] arrays:
//   bg0 = B rows 0,1 ; bg1 = B row 2 | G row 0 ; bg2 = G rows 1,2
__global__ __launch_bounds__(kBlock) void k_row_bg(const double* __restrict__ E, const double* __restrict__ F,
                                                   const int32_t* __restrict__ row_pt, const double* __restrict__ ete_inv,
                                                   int64_t O, double* __restrict__ bg0, double* __restrict__ bg1,
                                                   double* __restrict__ bg2) {
  __shared__ double lds[kBlock * 18];
  const int64_t r0 = int64_t(blockIdx.x) * kBlock;
  const int nvalid = int(min(int64_t(kBlock), O - r0));
  const int tid = threadIdx.x;
  double f[18], e[6];
  stage_cells<18>(F + 18 * r0, nvalid, lds, f);
  stage_cells<6>(E + 6 * r0, nvalid, lds, e);
  double o0[18], o1[18], o2[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) { o0[k] = 0.0; o1[k] = 0.0; o2[k] = 0.0; }
  if (tid < nvalid) {
    double B[27], iv[9];
    const double* m = ete_inv + 9 * int64_t(row_pt[r0 + tid]);
#pragma unroll
    for (int k = 0; k < 9; ++k) iv[k] = m[k];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int a = 0; a < 9; ++a) B[q * 9 + a] = e[q] * f[a] + e[3 + q] * f[9 + a];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      o0[a] = B[a];
      o0[9 + a] = B[9 + a];
      o1[a] = B[18 + a];
      o1[9 + a] = iv[0] * B[a] + iv[1] * B[9 + a] + iv[2] * B[18 + a];
      o2[a] = iv[3] * B[a] + iv[4] * B[9 + a] + iv[5] * B[18 + a];
      o2[9 + a] = iv[6] * B[a] + iv[7] * B[9 + a] + iv[8] * B[18 + a];
    }
  }
  unstage_cells<18>(bg0 + 18 * r0, nvalid, lds, o0);
  unstage_cells<18>(bg1 + 18 * r0, nvalid, lds, o1);
  unstage_cells<18>(bg2 + 18 * r0, nvalid, lds, o2);
}

// Stage 1, one workgroup per work item: sum of B_ri' G_rj over the item's pairs.  Thread t of the
// first 252 belongs to group t / 9 and owns the 3x3 piece (a0.., c0..) of the 9x9 product, so a pair
// costs it 18 loads for 27 FMAs; group g takes pairs g, g + 28, ... of the item and the 28 group
// sums are added in group order.
__global__ __launch_bounds__(kBlock) void k_pair_items(const int32_t* __restrict__ pair_rows,
                                                       const int64_t* __restrict__ item_begin,
                                                       const int32_t* __restrict__ item_ids,
                                                       const double* __restrict__ bg0, const double* __restrict__ bg1,
                                                       const double* __restrict__ bg2, double* __restrict__ item_partial,
                                                       int num_launch_items) {
  __shared__ double part[kPairGroups * 81];
  const int tid = threadIdx.x;
  // num_launch_items > 0: XCD-aware map (xcd_segment) -- the items of one block row of S (pairs whose first rows all belong
  // to one camera) run on ONE XCD, whose L2 then serves the B halves of those rows to every cell of the block row
  const int slot = num_launch_items > 0 ? xcd_segment(num_launch_items) : int(blockIdx.x);
  if (slot < 0) return;
  const int64_t item = item_ids ? int64_t(item_ids[slot]) : int64_t(slot);  // a selection, or all items
  const int64_t p0 = item_begin[item], p1 = item_begin[item + 1];
  const int g = tid / 9, sub = tid - g * 9;
  const int a0 = 3 * (sub / 3), c0 = 3 * (sub - 3 * (sub / 3));
  if (g < kPairGroups) {
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int64_t k = p0 + g; k < p1; k += kPairGroups) {
      const int64_t ri = pair_rows[2 * k], rj = pair_rows[2 * k + 1];
      double B[9], G[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        B[i] = bg0[18 * ri + a0 + i];
        B[3 + i] = bg0[18 * ri + 9 + a0 + i];
        B[6 + i] = bg1[18 * ri + a0 + i];
        G[i] = bg1[18 * rj + 9 + c0 + i];
        G[3 + i] = bg2[18 * rj + c0 + i];
        G[6 + i] = bg2[18 * rj + 9 + c0 + i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i * 3 + j] += (B[i] * G[j] + B[3 + i] * G[3 + j]) + B[6 + i] * G[6 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) part[g * 81 + (a0 + i) * 9 + c0 + j] = acc[i * 3 + j];
  }
  __syncthreads();
  if (tid < 81) {
    double v = 0.0;
#pragma unroll 4
    for (int q = 0; q < kPairGroups; ++q) v += part[q * 81 + tid];
    item_partial[item * 81 + tid] = v;
  }
}

// Stage 1 with the operands staged through LDS (round 2, the default).  k_pair_items above issues 18 eight-byte loads
// per thread and pair -- 4 536 lane loads for the 28 pairs of a step, every value fetched by three threads.  Here the
// workgroup fetches the 27 + 27 doubles of a pair once, as 28 sixteen-byte pieces (the two operands start on 16-byte
// boundaries when ten doubles of the shared middle record are taken instead of nine), one step ahead of the products,
// and the threads read their 3 + 3 triples from LDS.  Same pairs per group in the same order: bitwise the same sums.
// Measured (Final shape, 164 M pairs, same box): elimination 20.4 -> 19.1 ms, i.e. this kernel 14.4 -> 13.1 ms -- the
// load instructions were not the limit either (nor the XCD placement of the items: 2 %); what remains is 32.8 GB
// (FETCH_SIZE) of 216-byte records read from random places of a 12.5 GB table, one record per pair, at 2.5 TB/s.
// More A/B: the items launched in 8 x 8 blocks of the cell grid (both operands shared by eight cells close together on one
// XCD): 19.05 against 19.12 ms; two, three, four steps of operands in flight instead of one: 22.6 / 23.4 / 23.9 ms.
constexpr int kPairOperand = 28;  // staged doubles per operand: B = bg0[0..18) | bg1[0..10), G = bg1[8..18) | bg2[0..18)
constexpr int kPairPieces = kPairGroups * 28;  // 16-byte pieces per step
__global__ __launch_bounds__(kBlock) void k_pair_items_staged(const int32_t* __restrict__ pair_rows,
                                                              const int64_t* __restrict__ item_begin,
                                                              const int32_t* __restrict__ item_ids,
                                                              const double* __restrict__ bg0, const double* __restrict__ bg1,
                                                              const double* __restrict__ bg2, double* __restrict__ item_partial,
                                                              int num_launch_items) {
  constexpr int kStage = kPairGroups * 2 * kPairOperand;  // doubles per buffer
  __shared__ double lds[(2 * kStage > kPairGroups * 81) ? 2 * kStage : kPairGroups * 81];
  const int tid = threadIdx.x;
  const int slot = num_launch_items > 0 ? xcd_segment(num_launch_items) : int(blockIdx.x);
  if (slot < 0) return;
  const int64_t item = item_ids ? int64_t(item_ids[slot]) : int64_t(slot);
  const int64_t p0 = item_begin[item], p1 = item_begin[item + 1];
  const int steps = int((p1 - p0 + kPairGroups - 1) / kPairGroups);
  // the (up to four) pieces this thread moves per step: pair slot, source array and offset, place in the buffer
  int pg[4], dst[4], off[4], sel[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + i * kBlock;
    const int g = q / 28, w = q - g * 28;
    const int operand = w / 14, pc = w - operand * 14;
    pg[i] = q < kPairPieces ? g : -1;
    // B: pieces 0..8 of bg0, then 0..4 of bg1;  G: pieces 4..8 of bg1 (from double 8), then 0..8 of bg2
    const int first = operand == 0 ? 9 : 5;
    sel[i] = 2 * operand + (pc < first ? 0 : 1);                          // 0 bg0, 1 bg1 (front), 2 bg1 (back), 3 bg2
    off[i] = pc < first ? (operand == 0 ? 2 * pc : 8 + 2 * pc) : 2 * (pc - first);
    dst[i] = (g * 2 + operand) * kPairOperand + 2 * pc;
  }
  // (macros, not lambdas: arrays captured by reference went to scratch memory)
#define CX_PAIR_ROW(i, step, out)                                                                     \
  do {                                                                                                \
    const int64_t k_ = p0 + int64_t(step) * kPairGroups + pg[i];                                      \
    out = (pg[i] >= 0 && k_ < p1) ? pair_rows[2 * k_ + (sel[i] >> 1)] : -1;                           \
  } while (0)
#define CX_PAIR_FETCH(i, row, out)                                                                    \
  do {                                                                                                \
    const double* base_ = sel[i] == 0 ? bg0 : (sel[i] == 3 ? bg2 : bg1);                              \
    out = (row) < 0 ? make_double2(0.0, 0.0)                                                          \
                    : *reinterpret_cast<const double2*>(base_ + 18 * int64_t(row) + off[i]);         \
  } while (0)
  int32_t rows_next[4];
  double2 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_ROW(i, 0, rows_next[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (steps > 1) CX_PAIR_ROW(i, 1, rows_next[i]);
    else rows_next[i] = -1;
  }
  const int g = tid / 9, sub = tid - g * 9;
  const int a0 = 3 * (sub / 3), c0 = 3 * (sub - 3 * (sub / 3));
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int s = 0; s < steps; ++s) {
    double* buf = lds + (s & 1) * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (pg[i] >= 0) *reinterpret_cast<double2*>(buf + dst[i]) = v[i];
    __syncthreads();
    if (s + 1 < steps) {
#pragma unroll
      for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s + 2 < steps) CX_PAIR_ROW(i, s + 2, rows_next[i]);
        else rows_next[i] = -1;
      }
    }
    if (g < kPairGroups && p0 + int64_t(s) * kPairGroups + g < p1) {
      const double* Bop = buf + (g * 2) * kPairOperand;
      const double* Gop = Bop + kPairOperand;
      double B[9], G[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        B[i] = Bop[a0 + i];
        B[3 + i] = Bop[9 + a0 + i];
        B[6 + i] = Bop[18 + a0 + i];
        G[i] = Gop[1 + c0 + i];
        G[3 + i] = Gop[10 + c0 + i];
        G[6 + i] = Gop[19 + c0 + i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i * 3 + j] += (B[i] * G[j] + B[3 + i] * G[3 + j]) + B[6 + i] * G[6 + j];
    }
  }
  __syncthreads();  // the staging buffers become the groups' partial sums
  double* part = lds;
  if (g < kPairGroups) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) part[g * 81 + (a0 + i) * 9 + c0 + j] = acc[i * 3 + j];
  }
  __syncthreads();
  if (tid < 81) {
    double v81 = 0.0;
#pragma unroll 4
    for (int q = 0; q < kPairGroups; ++q) v81 += part[q * 81 + tid];
    item_partial[item * 81 + tid] = v81;
  }
#undef CX_PAIR_ROW
#undef CX_PAIR_FETCH
}

// ---- Round 3: ONE operand table instead of two.  B_i'(E'E + D^2)^-1 B_j = (K'B_i)'(K'B_j) with the Cholesky factor K of the
// 3 x 3 inverse ((E'E + D^2)^-1 = K K'), so every row r carries ONE 3 x 9 block H_r = K'B_r that serves as the left and
// as the right operand of its pairs.  Two [O][16] arrays (H0 = the first 16 entries of H, H1 = the other 11): a row's
// halves are one aligned 128-byte line each -- an operand is exactly two lines instead of 216 bytes at an arbitrary
// 8-byte boundary (2.7 lines on average) -- and the table is 7.4 GB instead of 12.5 GB on the Final shape, read in both
// roles, which is what the caches see.  The sums differ from the B'G form in rounding only (and make S symmetric by
// construction).  Measured on the Final shape (tools/pmc_sparse.sh, same box): FETCH_SIZE of the item kernel 58.3 -> 32.9 GB
// (4.4 -> 2.5 x the algorithmic 13.1 GB), 13.6 -> 11.5 ms; the row kernel 3.5 -> 2.5 ms; Eliminate 18.8 -> 16.2 ms.
// CX_PAIR_BG=1 keeps the B'G form for A/B runs.  Tried on top, same box: two / three steps of operands in flight instead of
// one -- SLOWER again (Eliminate + 1.1 / + 6.1 ms), so it is not the latency of the fetches; launching the items in groups
// of 2 / 4 / 8 block rows interleaved by block column (CX_PAIR_ROW_GROUP; right operands shared inside a group) -- 16.4 ->
// 16.3 / 16.2 / 16.2 ms, so it is not the reuse of the right operands either.  What is left is 32.9 GB of random 128-byte
// line fetches at 2.9 TB/s.  Later in round 3 (rocprofv3 kernel stats, same box): the table in CAMERA-MAJOR order (a cell's
// pairs then read ascending subsets of two contiguous 0.5 MB blocks instead of rows strewn over 7.4 GB; pair list translated
// once) -- 10.11 against 10.22 ms per launch, so it is not the locality (or the TLB reach) of the fetches either: not kept.
// Little's law fits what is seen: a CU holds ~280 pairs in flight (five workgroups x 28 staged + 28 requested, by registers and
// LDS alike -- LDS-DMA would hold the same), 448 B each = 32 MB chip-wide, at ~3 us of loaded latency = the ~9.4 TB/s the
// kernel pulls on the CU side (65 % L2 hits).
#ifndef CX_ROW_H_OCCUPANCY
#define CX_ROW_H_OCCUPANCY 4
#endif
__global__ __launch_bounds__(kBlock, CX_ROW_H_OCCUPANCY) void k_row_h(const double* __restrict__ E, const double* __restrict__ F,
                                                  const int32_t* __restrict__ row_pt, const double* __restrict__ ete_inv,
                                                  int64_t O, double* __restrict__ h0, double* __restrict__ h1) {
  __shared__ double lds[kBlock * 18];
  const int64_t r0 = int64_t(blockIdx.x) * kBlock;
  const int nvalid = int(min(int64_t(kBlock), O - r0));
  const int tid = threadIdx.x;
  double f[18], e[6];
  stage_cells<18>(F + 18 * r0, nvalid, lds, f);
  stage_cells<6>(E + 6 * r0, nvalid, lds, e);
  double o0[16], o1[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { o0[k] = 0.0; o1[k] = 0.0; }
  if (tid < nvalid) {
    const double* m = ete_inv + 9 * int64_t(row_pt[r0 + tid]);
    // lower Cholesky factor of the symmetric positive definite 3 x 3 inverse: M^-1 = K K'
    const double k00 = sqrt(m[0]);
    const double k10 = m[3] / k00, k20 = m[6] / k00;
    const double k11 = sqrt(m[4] - k10 * k10);
    const double k21 = (m[7] - k20 * k10) / k11;
    const double k22 = sqrt(m[8] - k20 * k20 - k21 * k21);
    double H[27];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double b0 = e[0] * f[a] + e[3] * f[9 + a];  // B = E'F, rows q = 0, 1, 2
      const double b1 = e[1] * f[a] + e[4] * f[9 + a];
      const double b2 = e[2] * f[a] + e[5] * f[9 + a];
      H[a] = k00 * b0 + k10 * b1 + k20 * b2;            // H = K'B
      H[9 + a] = k11 * b1 + k21 * b2;
      H[18 + a] = k22 * b2;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) o0[k] = H[k];
#pragma unroll
    for (int k = 16; k < 27; ++k) o1[k - 16] = H[k];
  }
  unstage_cells<16>(h0 + 16 * r0, nvalid, lds, o0);
  unstage_cells<16>(h1 + 16 * r0, nvalid, lds, o1);
}

// k_pair_items_staged on the one-table operands: per pair 14 + 14 sixteen-byte pieces (8 of the row's H0 line, 6 of its H1
// line, for the left row and for the right row), staged one step ahead; sum of H_ri' H_rj over the item's pairs.
#ifndef CX_PAIR_H_OCCUPANCY
#define CX_PAIR_H_OCCUPANCY 5
#endif
__global__ __launch_bounds__(kBlock, CX_PAIR_H_OCCUPANCY) void k_pair_items_h(const int32_t* __restrict__ pair_rows, const int64_t* __restrict__ item_begin,
                                                         const int32_t* __restrict__ item_ids, const double* __restrict__ h0,
                                                         const double* __restrict__ h1, double* __restrict__ item_partial,
                                                         int num_launch_items) {
  constexpr int kStage = kPairGroups * 2 * kPairOperand;  // doubles per buffer
  __shared__ double lds[(2 * kStage > kPairGroups * 81) ? 2 * kStage : kPairGroups * 81];
  const int tid = threadIdx.x;
  const int slot = num_launch_items > 0 ? xcd_segment(num_launch_items) : int(blockIdx.x);
  if (slot < 0) return;
  const int64_t item = item_ids ? int64_t(item_ids[slot]) : int64_t(slot);
  const int64_t p0 = item_begin[item], p1 = item_begin[item + 1];
  const int steps = int((p1 - p0 + kPairGroups - 1) / kPairGroups);
  // the (up to four) pieces this thread moves per step, packed: pair slot g (bits 0-4, 31 = none), offset in the source line
  // in doubles (5-8), source array H1 (9), right operand (10), place in the staging buffer in doubles (11-22)
  int piece[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + i * kBlock;
    const int g = q / 28, w = q - g * 28;
    const int operand = w / 14, pc = w - operand * 14;
    const int off = pc < 8 ? 2 * pc : 2 * (pc - 8);
    const int dst = (g * 2 + operand) * kPairOperand + 2 * pc;
    piece[i] = (q < kPairPieces ? g : 31) | (off << 5) | ((pc < 8 ? 0 : 1) << 9) | (operand << 10) | (dst << 11);
  }
#define CX_PAIR_ROW(i, step, out)                                                                     \
  do {                                                                                                \
    const int g_ = piece[i] & 31;                                                                     \
    const int64_t k_ = p0 + int64_t(step) * kPairGroups + g_;                                         \
    out = (g_ != 31 && k_ < p1) ? pair_rows[2 * k_ + ((piece[i] >> 10) & 1)] : -1;                    \
  } while (0)
#define CX_PAIR_FETCH(i, row, out)                                                                    \
  do {                                                                                                \
    const double* base_ = ((piece[i] >> 9) & 1) ? h1 : h0;                                            \
    out = (row) < 0 ? make_double2(0.0, 0.0)                                                          \
                    : *reinterpret_cast<const double2*>(base_ + 16 * int64_t(row) + ((piece[i] >> 5) & 15)); \
  } while (0)
  int32_t rows_next[4];
  double2 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_ROW(i, 0, rows_next[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (steps > 1) CX_PAIR_ROW(i, 1, rows_next[i]);
    else rows_next[i] = -1;
  }
  const int g = tid / 9, sub = tid - g * 9;
  const int a0 = 3 * (sub / 3), c0 = 3 * (sub - 3 * (sub / 3));
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int s = 0; s < steps; ++s) {
    double* buf = lds + (s & 1) * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((piece[i] & 31) != 31) *reinterpret_cast<double2*>(buf + (piece[i] >> 11)) = v[i];
    __syncthreads();
    if (s + 1 < steps) {
#pragma unroll
      for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s + 2 < steps) CX_PAIR_ROW(i, s + 2, rows_next[i]);
        else rows_next[i] = -1;
      }
    }
    if (g < kPairGroups && p0 + int64_t(s) * kPairGroups + g < p1) {
      const double* Hl = buf + (g * 2) * kPairOperand;
      const double* Hr = Hl + kPairOperand;
      // one row of the two 3 x 9 operands at a time (six values live instead of eighteen: the kernel's occupancy is set by
      // its registers, and it lives on the number of scattered line fetches it keeps in flight)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double b0 = Hl[9 * q + a0], b1 = Hl[9 * q + a0 + 1], b2 = Hl[9 * q + a0 + 2];
        const double g0 = Hr[9 * q + c0], g1 = Hr[9 * q + c0 + 1], g2 = Hr[9 * q + c0 + 2];
        acc[0] += b0 * g0; acc[1] += b0 * g1; acc[2] += b0 * g2;
        acc[3] += b1 * g0; acc[4] += b1 * g1; acc[5] += b1 * g2;
        acc[6] += b2 * g0; acc[7] += b2 * g1; acc[8] += b2 * g2;
      }
    }
  }
  __syncthreads();  // the staging buffers become the groups' partial sums
  double* part = lds;
  if (g < kPairGroups) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) part[g * 81 + (a0 + i) * 9 + c0 + j] = acc[i * 3 + j];
  }
  __syncthreads();
  if (tid < 81) {
    double v81 = 0.0;
#pragma unroll 4
    for (int q = 0; q < kPairGroups; ++q) v81 += part[q * 81 + tid];
    item_partial[item * 81 + tid] = v81;
  }
#undef CX_PAIR_ROW
#undef CX_PAIR_FETCH
}

// ---- Round 4, use_mixed_precision_solves: the operand table in SINGLE precision.  The factor these cells feed is fp32 anyway
// (cx_sparse_chol.hip, float tile pool), so H = K'B is rounded to float when it is written: 27 floats = 108 bytes, padded to
// one aligned 128-byte line per row -- an operand is ONE line instead of two, the table 3.7 GB instead of 7.4 GB on the Final
// shape.  Products of two floats are exact in double, and the sums stay double: S differs from the fp64 assembly by the
// rounding of H (relative 6e-8 per entry), below what the single precision factorisation does to it afterwards.
__global__ __launch_bounds__(kBlock) void k_row_h32(const double* __restrict__ E, const double* __restrict__ F,
                                                    const int32_t* __restrict__ row_pt, const double* __restrict__ ete_inv,
                                                    int64_t O, float* __restrict__ hf) {
  __shared__ double lds[kBlock * 18];
  const int64_t r0 = int64_t(blockIdx.x) * kBlock;
  const int nvalid = int(min(int64_t(kBlock), O - r0));
  const int tid = threadIdx.x;
  double f[18], e[6];
  stage_cells<18>(F + 18 * r0, nvalid, lds, f);
  stage_cells<6>(E + 6 * r0, nvalid, lds, e);
  double packed[16];  // 32 floats as 16 eight-byte pieces (the row's 128-byte line; entries 27..31 are zero)
#pragma unroll
  for (int k = 0; k < 16; ++k) packed[k] = 0.0;
  if (tid < nvalid) {
    const double* m = ete_inv + 9 * int64_t(row_pt[r0 + tid]);
    const double k00 = sqrt(m[0]);
    const double k10 = m[3] / k00, k20 = m[6] / k00;
    const double k11 = sqrt(m[4] - k10 * k10);
    const double k21 = (m[7] - k20 * k10) / k11;
    const double k22 = sqrt(m[8] - k20 * k20 - k21 * k21);
    float H[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) H[k] = 0.f;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double b0 = e[0] * f[a] + e[3] * f[9 + a];  // B = E'F, as k_row_h
      const double b1 = e[1] * f[a] + e[4] * f[9 + a];
      const double b2 = e[2] * f[a] + e[5] * f[9 + a];
      H[a] = float(k00 * b0 + k10 * b1 + k20 * b2);     // H = K'B, rounded once
      H[9 + a] = float(k11 * b1 + k21 * b2);
      H[18 + a] = float(k22 * b2);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float2 two = make_float2(H[2 * k], H[2 * k + 1]);
      packed[k] = __builtin_bit_cast(double, two);
    }
  }
  unstage_cells<16>(reinterpret_cast<double*>(hf) + 16 * r0, nvalid, lds, packed);
}

// k_pair_items_h on the float table: per pair 7 + 7 sixteen-byte pieces (one line per row), 56 pairs per step -- the same
// number of lines in flight per workgroup as the fp64 kernel's 28 pairs of four lines -- staged one step ahead; every
// group of 9 threads takes two pairs of a step.  Sum of H_ri' H_rj in double.
constexpr int kPairGroups32 = 2 * kPairGroups;
constexpr int kPairOperand32 = 28;                    // staged floats per operand (7 pieces of 4)
constexpr int kPairPieces32 = kPairGroups32 * 14;     // 16-byte pieces per step
template <int OCC>
__global__ __launch_bounds__(kBlock, OCC) void k_pair_items_h32(const int32_t* __restrict__ pair_rows, const int64_t* __restrict__ item_begin,
                                                         const int32_t* __restrict__ item_ids, const float* __restrict__ hf,
                                                         double* __restrict__ item_partial, int num_launch_items) {
  constexpr int kStage = kPairGroups32 * 2 * kPairOperand32;  // floats per buffer
  constexpr int kLdsBytes = (2 * kStage * 4 > kPairGroups * 81 * 8) ? 2 * kStage * 4 : kPairGroups * 81 * 8;
  __shared__ double lds[kLdsBytes / 8];
  float* ldsf = reinterpret_cast<float*>(lds);
  const int tid = threadIdx.x;
  const int slot = num_launch_items > 0 ? xcd_segment(num_launch_items) : int(blockIdx.x);
  if (slot < 0) return;
  const int64_t item = item_ids ? int64_t(item_ids[slot]) : int64_t(slot);
  const int64_t p0 = item_begin[item], p1 = item_begin[item + 1];
  const int steps = int((p1 - p0 + kPairGroups32 - 1) / kPairGroups32);
  // the (up to four) pieces this thread moves per step, packed: pair slot g (bits 0-5, 63 = none), piece of the row's line
  // (6-8), right operand (9), place in the staging buffer in floats (10-21)
  int piece[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + i * kBlock;
    const int g = q / 14, w = q - g * 14;
    const int operand = w / 7, pc = w - operand * 7;
    const int dst = (g * 2 + operand) * kPairOperand32 + 4 * pc;
    piece[i] = (q < kPairPieces32 ? g : 63) | (pc << 6) | (operand << 9) | (dst << 10);
  }
#define CX_PAIR_ROW(i, step, out)                                                                     \
  do {                                                                                                \
    const int g_ = piece[i] & 63;                                                                     \
    const int64_t k_ = p0 + int64_t(step) * kPairGroups32 + g_;                                       \
    out = (g_ != 63 && k_ < p1) ? pair_rows[2 * k_ + ((piece[i] >> 9) & 1)] : -1;                     \
  } while (0)
#define CX_PAIR_FETCH(i, row, out)                                                                    \
  do {                                                                                                \
    out = (row) < 0 ? make_float4(0.f, 0.f, 0.f, 0.f)                                                 \
                    : *reinterpret_cast<const float4*>(hf + 32 * int64_t(row) + 4 * ((piece[i] >> 6) & 7)); \
  } while (0)
  int32_t rows_next[4];
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_ROW(i, 0, rows_next[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (steps > 1) CX_PAIR_ROW(i, 1, rows_next[i]);
    else rows_next[i] = -1;
  }
  const int g = tid / 9, sub = tid - g * 9;
  const int a0 = 3 * (sub / 3), c0 = 3 * (sub - 3 * (sub / 3));
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int s = 0; s < steps; ++s) {
    float* buf = ldsf + (s & 1) * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((piece[i] & 63) != 63) *reinterpret_cast<float4*>(buf + (piece[i] >> 10)) = v[i];
    __syncthreads();
    if (s + 1 < steps) {
#pragma unroll
      for (int i = 0; i < 4; ++i) CX_PAIR_FETCH(i, rows_next[i], v[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s + 2 < steps) CX_PAIR_ROW(i, s + 2, rows_next[i]);
        else rows_next[i] = -1;
      }
    }
    if (g < kPairGroups) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pair = g + h * kPairGroups;  // ascending pair order inside a group's sum: g, then g + 28
        if (p0 + int64_t(s) * kPairGroups32 + pair < p1) {
          const float* Hl = buf + (pair * 2) * kPairOperand32;
          const float* Hr = Hl + kPairOperand32;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const double b0 = double(Hl[9 * q + a0]), b1 = double(Hl[9 * q + a0 + 1]), b2 = double(Hl[9 * q + a0 + 2]);
            const double g0 = double(Hr[9 * q + c0]), g1 = double(Hr[9 * q + c0 + 1]), g2 = double(Hr[9 * q + c0 + 2]);
            acc[0] += b0 * g0; acc[1] += b0 * g1; acc[2] += b0 * g2;
            acc[3] += b1 * g0; acc[4] += b1 * g1; acc[5] += b1 * g2;
            acc[6] += b2 * g0; acc[7] += b2 * g1; acc[8] += b2 * g2;
          }
        }
      }
    }
  }
  __syncthreads();  // the staging buffers become the groups' partial sums
  double* part = lds;
  if (g < kPairGroups) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) part[g * 81 + (a0 + i) * 9 + c0 + j] = acc[i * 3 + j];
  }
  __syncthreads();
  if (tid < 81) {
    double v81 = 0.0;
#pragma unroll 4
    for (int q = 0; q < kPairGroups; ++q) v81 += part[q * 81 + tid];
    item_partial[item * 81 + tid] = v81;
  }
#undef CX_PAIR_ROW
#undef CX_PAIR_FETCH
}

// Stage 2, 81 threads per non-zero cell (c1 <= c2): [c1 == c2] F'F - sum of the cell's items, written either
// into the dense row-major lhs (pre-zeroed; D_f^2 added on the diagonal -- the reference's dense S) or
// into the cell-major sparse value array (81 contiguous doubles per cell, D_f^2 NOT added: the sparse
// consumers add it after the cross-rank sum).
__global__ __launch_bounds__(3 * 81) void k_pair_cells(const int32_t* __restrict__ cell_c1, const int32_t* __restrict__ cell_c2,
                                                       const int32_t* __restrict__ cell_item_start,
                                                       const double* __restrict__ item_partial,
                                                       const double* __restrict__ diag, const double* __restrict__ Df,
                                                       double* __restrict__ lhs, double* __restrict__ sparse, int C,
                                                       int64_t num_cells) {
  const int64_t cell = int64_t(blockIdx.x) * 3 + threadIdx.x / 81;
  if (cell >= num_cells) return;
  const int el = threadIdx.x % 81;
  const int c1 = cell_c1[cell], c2 = cell_c2[cell];
  const int a = el / 9, c = el - a * 9;
  double v = 0.0;
  for (int it = cell_item_start[cell]; it < cell_item_start[cell + 1]; ++it) v -= item_partial[int64_t(it) * 81 + el];
  if (c1 == c2) v += diag[int64_t(c1) * 81 + el];
  if (sparse) {
    sparse[cell * 81 + el] = v;
    return;
  }
  if (c1 == c2 && Df && a == c) {
    const double d = Df[9 * int64_t(c1) + a];
    v += d * d;
  }
  const int64_t n = 9 * int64_t(C);
  lhs[(9 * int64_t(c1) + a) * n + 9 * int64_t(c2) + c] = v;
}

// BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate (block_random_access_sparse_matrix.cc:
// 124-163) for the cell-major upper-stored S: y_c = sum_{(c, c2)} S_cell x_c2 + sum_{(c1 < c, c)} S_cell' x_c1.
// One wavefront per block row c: 7 groups of 9 lanes (lane = one entry of y_c) take every 7th cell of the
// row list, then of the column list; the 7 group sums are added in group order, so y is reproducible
// (the reference walks the cells serially and scatters into y).
__global__ __launch_bounds__(256) void k_sym_spmv(const double* __restrict__ S, const int32_t* __restrict__ cell_c2,
                                                  const int32_t* __restrict__ cell_c1,
                                                  const int32_t* __restrict__ row_start,
                                                  const int32_t* __restrict__ col_start,
                                                  const int32_t* __restrict__ col_cells, const double* __restrict__ x,
                                                  double* __restrict__ y, int C, const int* __restrict__ stop) {
  __shared__ double part[4][63];
  if (stop && *stop) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + wave;
  if (c >= C) return;
  const int g = lane / 9, a = lane - g * 9;
  double sum = 0.0;
  if (g < 7) {
    for (int k = row_start[c] + g; k < row_start[c + 1]; k += 7) {
      const double* __restrict__ B = S + int64_t(k) * 81 + a * 9;   // row a of the cell
      const double* __restrict__ xv = x + 9 * int64_t(cell_c2[k]);
#pragma unroll
      for (int j = 0; j < 9; ++j) sum += B[j] * xv[j];
    }
    for (int k = col_start[c] + g; k < col_start[c + 1]; k += 7) {
      const int cell = col_cells[k];
      const double* __restrict__ B = S + int64_t(cell) * 81 + a;    // column a of the cell
      const double* __restrict__ xv = x + 9 * int64_t(cell_c1[cell]);
#pragma unroll
      for (int j = 0; j < 9; ++j) sum += B[j * 9] * xv[j];
    }
    part[wave][lane] = sum;
  }
  __syncthreads();
  if (lane < 9) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) v += part[wave][q * 9 + lane];
    y[9 * int64_t(c) + lane] = v;
  }
}

// blocks[c] = S(c, c) (first cell of every block row)
__global__ void k_extract_diag_cells(const double* __restrict__ S, const int32_t* __restrict__ row_start,
                                     double* __restrict__ blocks, int C) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(C) * 81) return;
  const int c = int(i / 81);
  blocks[i] = S[int64_t(row_start[c]) * 81 + (i - int64_t(c) * 81)];
}

// dense row-major lhs (n = 9C) from the block-major pair sums, the F'F diagonal blocks and D_f^2;
// blocks below the diagonal stay zero (the reference never touches them)
__global__ void k_blocks_to_dense(const double* __restrict__ blk, const double* __restrict__ diag,
                                  const double* __restrict__ Df, double* __restrict__ lhs, int C) {
  const int64_t n = 9 * int64_t(C);
  const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int row = int(idx / n), col = int(idx - int64_t(row) * n);
  const int c1 = row / 9, a = row - c1 * 9, c2 = col / 9, c = col - c2 * 9;
  double v = 0.0;
  if (c1 <= c2) {
    v = blk[(int64_t(c1) * C + c2) * 81 + a * 9 + c];
    if (c1 == c2) {
      v += diag[int64_t(c1) * 81 + a * 9 + c];
      if (Df && a == c) v += Df[row] * Df[row];
    }
  }
  lhs[idx] = v;
}

// ================================================================ host drivers

int cxs_compute_ete_inverse(cx_matrix* A, const double* D, const double* b, double* ete_inv, double* g,
                            bool llt, int* d_flag) {
  hipStream_t st = A->ctx->stream;
  if (A->num_tiles == 0) return CX_OK;
  if (llt)
    hipLaunchKernelGGL(k_chunk_ete<true>, dim3(A->num_tiles), dim3(kBlock), 0, st, A->d_values.p, A->d_tile_row.p,
                       A->d_tile_pt.p, A->d_pt_start.p, D, b, ete_inv, g, d_flag);
  else
    hipLaunchKernelGGL(k_chunk_ete<false>, dim3(A->num_tiles), dim3(kBlock), 0, st, A->d_values.p, A->d_tile_row.p,
                       A->d_tile_pt.p, A->d_pt_start.p, D, b, ete_inv, g, d_flag);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxs_chunk_pass(cx_matrix* A, int mode, const double* ete_inv, const double* xf, const double* b, double* out) {
  hipStream_t st = A->ctx->stream;
  if (A->num_tiles == 0) return CX_OK;
#define CX_LAUNCH_PASS(M, T, EP, FP)                                                                           \
  hipLaunchKernelGGL((k_chunk_pass<M, T>), dim3(A->num_tiles), dim3(kBlock), 0, st, EP, FP, A->d_tile_row.p,   \
                     A->d_tile_pt.p, A->d_pt_start.p, A->d_row_cam.p, A->d_row_pt.p, ete_inv, xf, b, out, A->stop, 0)
  if (A->use_f32 && (mode == 0 || mode == 3)) {
    // products inside a mixed-precision CG: fp32 copies of the cells (cx_matrix_ensure_f32)
    const float* E = A->d_vals32.p;
    const float* F = A->d_vals32.p + 6 * A->O;
    if (mode == 0) CX_LAUNCH_PASS(0, float, E, F);
    else CX_LAUNCH_PASS(3, float, E, F);
  } else {
    const double* E = A->d_values.p;
    const double* F = A->d_values.p + 6 * A->O;
    if (mode == 0) CX_LAUNCH_PASS(0, double, E, F);
    else if (mode == 1) CX_LAUNCH_PASS(1, double, E, F);
    else if (mode == 3) CX_LAUNCH_PASS(3, double, E, F);
    else CX_LAUNCH_PASS(2, double, E, F);
  }
#undef CX_LAUNCH_PASS
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxs_camera_block_diagonal(cx_matrix* A, double* blocks) {
  hipStream_t st = A->ctx->stream;
  CX_TRY(cx_matrix_ensure_ft(A));
  if (A->num_segs > 0)
    hipLaunchKernelGGL(k_cam_diag<false>, dim3(A->num_segs), dim3(kBlock), 0, st, A->d_Ft.p, A->d_values.p,
                       A->d_cam_rows.p, A->d_row_pt.p, A->d_seg_begin.p, (const double*)nullptr, A->d_partials.p);
  hipLaunchKernelGGL(k_cam_diag_reduce, dim3(grid_for(int64_t(A->C) * 45, 256)), dim3(256), 0, st,
                     A->d_partials.p, A->d_cam_seg_start.p, blocks, A->C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// blocks[81C] = block diagonal of F'F and ftb[9C] = F't in ONE pass over the camera-major copy (k_cam_init without the Schur
// terms: the kernel of the implicit set-up, here for CGNR's Jacobi blocks and the camera part of J'b).
int cxs_camera_blocks_and_ft(cx_matrix* A, const double* t, double* blocks, double* ftb) {
  hipStream_t st = A->ctx->stream;
  if (A->num_segs == 0) {
    if (A->C > 0) {
      CX_HIP(hipMemsetAsync(blocks, 0, 81 * size_t(A->C) * sizeof(double), st));
      CX_HIP(hipMemsetAsync(ftb, 0, 9 * size_t(A->C) * sizeof(double), st));
    }
    return CX_OK;
  }
  CX_TRY(cx_matrix_ensure_ft(A));
  CX_TRY(A->d_partials9.alloc(size_t(A->num_segs) * 9));
  hipLaunchKernelGGL(k_cam_init<false>, dim3(xcd_grid(A->num_segs)), dim3(kBlock), 0, st, (const double*)nullptr, (const double*)nullptr,
                     A->d_cam_rows.p, A->d_row_pt.p, A->d_seg_begin.p, (const double*)nullptr, t, (const double*)A->d_Ft.p,
                     A->d_partials.p, A->d_partials9.p, A->num_segs);
  hipLaunchKernelGGL(k_cam_diag_reduce, dim3(grid_for(int64_t(A->C) * 45, 256)), dim3(256), 0, st,
                     (const double*)A->d_partials.p, A->d_cam_seg_start.p, blocks, A->C);
  hipLaunchKernelGGL(k_sum_segments9, dim3(grid_for(int64_t(A->C) * 9, 256)), dim3(256), 0, st,
                     (const double*)A->d_partials9.p, A->d_cam_seg_start.p, ftb, A->C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxs_block9_add_diag_invert(cx_context* ctx, double* blocks, const double* Df, int C, int* d_flag) {
  if (C == 0) return CX_OK;
  hipLaunchKernelGGL(k_block9_add_diag_invert, dim3(grid_for(C, kInvertBlocks)), dim3(kInvertBlocks * 9), 0, ctx->stream, blocks, Df, C, d_flag);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// Gather assembly shared by the dense and the block-sparse explicit S: (E'E + D_e^2)^-1 (closed-form inverse
// of InvertPSDMatrix<3>), B / G of every row, F'F diagonal blocks, per-item pair sums.
// rhs != nullptr: the reduced right-hand side F'(b - E (E'E)^-1 E'b) is delivered too, and from the SAME two passes that
// produce (E'E + D^2)^-1 and the F'F diagonal blocks -- k_chunk_init<cofactor> writes t' = b - E (E'E)^-1 E'b next to the
// inverses, k_cam_init sums F't' next to the blocks (the set-up kernels of the implicit Schur complement; round 3 ran
// k_chunk_ete, k_cam_diag and then k_chunk_pass<1>, k_cam_ft: two passes over J more).
int cxs_assemble_pair_items(cx_matrix* A, const double* D, const int32_t* item_ids, int64_t num_selected, bool f32_operands,
                            const double* b, double* rhs) {
  hipStream_t st = A->ctx->stream;
  CX_TRY(A->d_elim_ete.alloc(size_t(std::max<int64_t>(9 * int64_t(A->P), 1))));
  CX_TRY(A->d_elim_diag.alloc(size_t(std::max<int64_t>(81 * int64_t(A->C), 1))));
  static const bool fuse_allowed = std::getenv("CX_ELIMINATE_RHS_SEPARATE") == nullptr;  // A/B switch: round 3's four passes
  const bool fused_rhs = fuse_allowed && rhs != nullptr && b != nullptr && A->num_tiles > 0 && A->num_segs > 0;
  // the cofactor inverse reports nothing, exactly as InvertPSDMatrix<3> (invert_psd_matrix.h:60-63): a singular
  // E'E + D^2 shows as Inf/NaN in S and ends the solve in the Cholesky factorisation, as in the reference
  if (fused_rhs) {
    CX_TRY(A->d_elim_rows.alloc(size_t(std::max<int64_t>(A->num_rows, 1))));
    hipLaunchKernelGGL(k_chunk_init<false>, dim3(A->num_tiles), dim3(kBlock), 0, st, (const double*)A->d_values.p, A->d_tile_row.p,
                       A->d_tile_pt.p, A->d_pt_start.p, A->d_row_pt.p, D, b, A->d_elim_ete.p, A->d_elim_rows.p, (int*)nullptr);
    CX_HIP(hipGetLastError());
  } else {
    CX_TRY(cxs_compute_ete_inverse(A, D, nullptr, A->d_elim_ete.p, nullptr, false, nullptr));
  }
  // the F'F diagonal blocks (and, fused, the right-hand side) -- called where the variants below had the block diagonal
  auto camera_pass = [&]() -> int {
    if (fused_rhs) return cxs_camera_blocks_and_ft(A, A->d_elim_rows.p, A->d_elim_diag.p, rhs);
    if (A->ftf_partials_current && A->num_segs > 0) {  // (the implicit set-up of this solve has summed them already)
      hipLaunchKernelGGL(k_cam_diag_reduce, dim3(grid_for(int64_t(A->C) * 45, 256)), dim3(256), 0, st,
                         (const double*)A->d_partials.p, A->d_cam_seg_start.p, A->d_elim_diag.p, A->C);
      CX_HIP(hipGetLastError());
    } else {
      CX_TRY(cxs_camera_block_diagonal(A, A->d_elim_diag.p));
    }
    return rhs ? cxs_eliminate_rhs(A, b, rhs) : CX_OK;
  };
  static const bool one_table = std::getenv("CX_PAIR_BG") == nullptr && std::getenv("CX_PAIR_ITEMS_DIRECT") == nullptr;  // A/B switch
  static const bool allow_f32 = std::getenv("CX_PAIR_H64") == nullptr;  // A/B switch: fp64 operands under a float factor
  if (one_table && f32_operands && allow_f32) {  // round 4: the same on a float table, one 128-byte line per row
    CX_TRY(A->d_elim_h32.alloc(size_t(std::max<int64_t>(32 * A->O, 1))));
    if (A->O > 0)
      hipLaunchKernelGGL(k_row_h32, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, (const double*)A->d_values.p,
                         (const double*)(A->d_values.p + 6 * A->O), (const int32_t*)A->d_row_pt.p, (const double*)A->d_elim_ete.p,
                         A->O, A->d_elim_h32.p);
    CX_TRY(camera_pass());
    const int32_t* ids = item_ids ? item_ids : (A->d_item_order.n > 0 && A->d_item_order.p ? (const int32_t*)A->d_item_order.p : nullptr);
    const int64_t n_items = item_ids ? num_selected : A->num_items;
    static const bool xcd = std::getenv("CX_NO_XCD_ITEMS") == nullptr;
    static const int occ = std::getenv("CX_PAIR_H32_OCC") ? std::atoi(std::getenv("CX_PAIR_H32_OCC")) : 5;  // A/B switch
    if (n_items > 0) {
      if (occ == 4)
        hipLaunchKernelGGL(k_pair_items_h32<4>, dim3(unsigned(xcd ? xcd_grid(int(n_items)) : n_items)), dim3(kBlock), 0, st,
                           (const int32_t*)A->d_pair_rows.p, (const int64_t*)A->d_item_begin.p, ids, (const float*)A->d_elim_h32.p,
                           A->d_item_partial.p, xcd ? int(n_items) : 0);
      else
        hipLaunchKernelGGL(k_pair_items_h32<5>, dim3(unsigned(xcd ? xcd_grid(int(n_items)) : n_items)), dim3(kBlock), 0, st,
                           (const int32_t*)A->d_pair_rows.p, (const int64_t*)A->d_item_begin.p, ids, (const float*)A->d_elim_h32.p,
                           A->d_item_partial.p, xcd ? int(n_items) : 0);
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  if (one_table) {  // round 3: H = K'B, one operand table (k_row_h / k_pair_items_h)
    const size_t rows16 = size_t(std::max<int64_t>(16 * A->O, 1));
    CX_TRY(A->d_elim_bg0.alloc(rows16));
    CX_TRY(A->d_elim_bg1.alloc(rows16));
    if (A->O > 0)
      hipLaunchKernelGGL(k_row_h, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, (const double*)A->d_values.p,
                         (const double*)(A->d_values.p + 6 * A->O), (const int32_t*)A->d_row_pt.p, (const double*)A->d_elim_ete.p,
                         A->O, A->d_elim_bg0.p, A->d_elim_bg1.p);
    CX_TRY(camera_pass());
    // all items: in the launch order of the pair-list builder (groups of block rows interleaved by block column, so that the
    // right operands of neighbouring block rows meet in one XCD's L2)
    const int32_t* ids = item_ids ? item_ids : (A->d_item_order.n > 0 && A->d_item_order.p ? (const int32_t*)A->d_item_order.p : nullptr);
    const int64_t n_items = item_ids ? num_selected : A->num_items;
    static const bool xcd = std::getenv("CX_NO_XCD_ITEMS") == nullptr;
    if (n_items > 0)
      hipLaunchKernelGGL(k_pair_items_h, dim3(unsigned(xcd ? xcd_grid(int(n_items)) : n_items)), dim3(kBlock), 0, st,
                         (const int32_t*)A->d_pair_rows.p, (const int64_t*)A->d_item_begin.p, ids, (const double*)A->d_elim_bg0.p,
                         (const double*)A->d_elim_bg1.p, A->d_item_partial.p, xcd ? int(n_items) : 0);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  const size_t rows18 = size_t(std::max<int64_t>(18 * A->O, 1));
  CX_TRY(A->d_elim_bg0.alloc(rows18));
  CX_TRY(A->d_elim_bg1.alloc(rows18));
  CX_TRY(A->d_elim_bg2.alloc(rows18));
  if (A->O > 0)
    hipLaunchKernelGGL(k_row_bg, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, (const double*)A->d_values.p,
                       (const double*)(A->d_values.p + 6 * A->O), (const int32_t*)A->d_row_pt.p,
                       (const double*)A->d_elim_ete.p, A->O, A->d_elim_bg0.p, A->d_elim_bg1.p, A->d_elim_bg2.p);
  CX_TRY(camera_pass());
  const int64_t launch_items = item_ids ? num_selected : A->num_items;
  static const bool xcd_items = std::getenv("CX_NO_XCD_ITEMS") == nullptr;  // A/B switch
  static const bool staged = std::getenv("CX_PAIR_ITEMS_DIRECT") == nullptr;  // A/B switch: round 1's k_pair_items
  if (launch_items > 0)
    hipLaunchKernelGGL(staged ? k_pair_items_staged : k_pair_items, dim3(unsigned(xcd_items ? xcd_grid(int(launch_items)) : launch_items)), dim3(kBlock), 0, st,
                       (const int32_t*)A->d_pair_rows.p, (const int64_t*)A->d_item_begin.p, item_ids, (const double*)A->d_elim_bg0.p,
                       (const double*)A->d_elim_bg1.p, (const double*)A->d_elim_bg2.p, A->d_item_partial.p,
                       xcd_items ? int(launch_items) : 0);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// rhs = F'(b - E (E'E)^-1 E'b)  (UpdateRhs, schur_eliminator_impl.h:379-420); A->d_elim_ete must be current
int cxs_eliminate_rhs(cx_matrix* A, const double* b, double* rhs) {
  if (!rhs) return CX_OK;
  if (!b) {
    CX_HIP(hipMemsetAsync(rhs, 0, size_t(9) * A->C * sizeof(double), A->ctx->stream));
    return CX_OK;
  }
  CX_TRY(A->d_elim_rows.alloc(size_t(std::max<int64_t>(A->num_rows, 1))));
  CX_TRY(cxs_chunk_pass(A, 1, A->d_elim_ete.p, nullptr, b, A->d_elim_rows.p));
  return cxk_ft_multiply(A, A->d_elim_rows.p, rhs, false);
}

int cxs_eliminate_dense(cx_matrix* A, const double* b, const double* D, bool add_df, double* lhs, double* rhs) {
  hipStream_t st = A->ctx->stream;
  const int C = A->C;
  const int64_t n = 9 * int64_t(C);
  const double* Df = (D && add_df) ? D + 3 * int64_t(A->P) : nullptr;
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state == 1) {
    CX_TRY(cxs_assemble_pair_items(A, D, nullptr, 0, false, b, rhs));
    if (n > 0) CX_HIP(hipMemsetAsync(lhs, 0, size_t(n) * n * sizeof(double), st));
    if (A->num_cells > 0)
      hipLaunchKernelGGL(k_pair_cells, dim3(unsigned((A->num_cells + 2) / 3)), dim3(3 * 81), 0, st,
                         (const int32_t*)A->d_cell_c1.p, (const int32_t*)A->d_cell_c2.p, (const int32_t*)A->d_cell_item_start.p,
                         (const double*)A->d_item_partial.p, (const double*)A->d_elim_diag.p, Df, lhs, (double*)nullptr, C,
                         A->num_cells);
    CX_HIP(hipGetLastError());
    return CX_OK;  // (the right-hand side came out of the assembly's own passes)
  } else {
    // scatter path: fp64 atomics into a block-major copy of S
    CX_TRY(A->d_elim_ete.alloc(size_t(std::max<int64_t>(9 * int64_t(A->P), 1))));
    CX_TRY(A->d_elim_diag.alloc(size_t(std::max<int64_t>(81 * int64_t(C), 1))));
    CX_TRY(cxs_compute_ete_inverse(A, D, nullptr, A->d_elim_ete.p, nullptr, false, nullptr));
    CX_TRY(A->d_elim_blk.alloc(size_t(std::max<int64_t>(int64_t(C) * C * 81, 1))));
    CX_HIP(hipMemsetAsync(A->d_elim_blk.p, 0, size_t(C) * C * 81 * sizeof(double), st));
    if (A->num_tiles > 0) {
      const double* E = A->d_values.p;
      const double* F = A->d_values.p + 6 * A->O;
      hipLaunchKernelGGL(k_chunk_eliminate, dim3(A->num_tiles), dim3(kBlock), 0, st, E, F, A->d_tile_row.p,
                         A->d_tile_pt.p, A->d_pt_start.p, A->d_row_cam.p, (const double*)A->d_elim_ete.p, A->d_elim_blk.p, C);
      if (A->has_big_tiles)
        hipLaunchKernelGGL(k_big_chunk_eliminate, dim3(A->num_tiles), dim3(kBlock), 0, st, E, F, A->d_tile_row.p,
                           A->d_tile_pt.p, A->d_row_cam.p, (const double*)A->d_elim_ete.p, A->d_elim_blk.p, C);
    }
    CX_TRY(cxs_camera_block_diagonal(A, A->d_elim_diag.p));
    if (n > 0)
      hipLaunchKernelGGL(k_blocks_to_dense, dim3(grid_for(n * n, 256)), dim3(256), 0, st, (const double*)A->d_elim_blk.p,
                         (const double*)A->d_elim_diag.p, Df, lhs, C);
  }
  CX_HIP(hipGetLastError());
  return cxs_eliminate_rhs(A, b, rhs);
}

int cxs_eliminate_sparse(cx_matrix* A, const double* b, const double* D, double* rhs, bool f32_operands) {
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state != 1) {
    cx_set_error("the explicit Schur complement of this structure needs more than 2^28 row pairs");
    return CX_ERR_UNSUPPORTED;
  }
  CX_TRY(A->d_S.alloc(size_t(std::max<int64_t>(A->num_cells, 1)) * 81));
  CX_TRY(cxs_assemble_pair_items(A, D, nullptr, 0, f32_operands, b, rhs));
  if (A->num_cells > 0)
    hipLaunchKernelGGL(k_pair_cells, dim3(unsigned((A->num_cells + 2) / 3)), dim3(3 * 81), 0, A->ctx->stream,
                       (const int32_t*)A->d_cell_c1.p, (const int32_t*)A->d_cell_c2.p, (const int32_t*)A->d_cell_item_start.p,
                       (const double*)A->d_item_partial.p, (const double*)A->d_elim_diag.p, (const double*)nullptr,
                       (double*)nullptr, A->d_S.p, A->C, A->num_cells);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxs_sparse_multiply(cx_matrix* A, const double* x, double* y) {
  if (A->C == 0) return CX_OK;
  hipLaunchKernelGGL(k_sym_spmv, dim3((A->C + 3) / 4), dim3(256), 0, A->ctx->stream, (const double*)A->d_S.p,
                     (const int32_t*)A->d_cell_c2.p, (const int32_t*)A->d_cell_c1.p, (const int32_t*)A->d_cell_row_start.p,
                     (const int32_t*)A->d_col_cell_start.p, (const int32_t*)A->d_col_cells.p, x, y, A->C, A->stop);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxs_sparse_diagonal(cx_matrix* A, double* blocks) {
  if (A->C == 0) return CX_OK;
  hipLaunchKernelGGL(k_extract_diag_cells, dim3(grid_for(int64_t(A->C) * 81, 256)), dim3(256), 0, A->ctx->stream,
                     (const double*)A->d_S.p, (const int32_t*)A->d_cell_row_start.p, blocks, A->C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// Fused ImplicitSchurComplement::Init for the static layout: ete_inv[9P], the inverse of the
// per-camera preconditioner blocks is left to the caller (blocks[81C] = F'F or S diagonal blocks,
// D_f^2 not yet added), rhs_out[9C] = F'(b - E (E'E)^-1 E'b).
// defer_reduce: stop at the per-segment partial sums (A->d_partials: 45 per segment, A->d_partials9: 9 per segment); the
// caller's next kernel adds them up (small problems: k_cg_small_setup, cx_solver.hip).
int cxs_implicit_init(cx_matrix* A, const double* D, const double* b, bool want_blocks, bool with_schur,
                      double* ete_inv, double* rows_scratch, double* blocks, double* rhs_out, int* d_flag, bool defer_reduce) {
  hipStream_t st = A->ctx->stream;
  if (A->num_tiles == 0 || A->num_segs == 0) {
    if (want_blocks) CX_HIP(hipMemsetAsync(blocks, 0, 81 * size_t(A->C) * sizeof(double), st));
    CX_HIP(hipMemsetAsync(rhs_out, 0, 9 * size_t(A->C) * sizeof(double), st));
    return CX_OK;
  }
  const double* E = A->d_values.p;
  const double* F = A->d_values.p + 6 * A->O;
  hipLaunchKernelGGL(k_chunk_init<true>, dim3(A->num_tiles), dim3(kBlock), 0, st, E, A->d_tile_row.p, A->d_tile_pt.p,
                     A->d_pt_start.p, A->d_row_pt.p, D, b, ete_inv, rows_scratch, d_flag);
  CX_TRY(cx_matrix_ensure_ft(A));
  CX_TRY(A->d_partials9.alloc(size_t(A->num_segs) * 9));
  if (with_schur)
    hipLaunchKernelGGL(k_cam_init<true>, dim3(xcd_grid(A->num_segs)), dim3(kBlock), 0, st, F, E, A->d_cam_rows.p,
                       A->d_row_pt.p, A->d_seg_begin.p, (const double*)ete_inv, (const double*)rows_scratch, (const double*)A->d_Ft.p,
                       A->d_partials.p, A->d_partials9.p, A->num_segs);
  else
    hipLaunchKernelGGL(k_cam_init<false>, dim3(xcd_grid(A->num_segs)), dim3(kBlock), 0, st, F, E, A->d_cam_rows.p,
                       A->d_row_pt.p, A->d_seg_begin.p, (const double*)ete_inv, (const double*)rows_scratch, (const double*)A->d_Ft.p,
                       A->d_partials.p, A->d_partials9.p, A->num_segs);
  if (!defer_reduce) {
    if (want_blocks)
      hipLaunchKernelGGL(k_cam_diag_reduce, dim3(grid_for(int64_t(A->C) * 45, 256)), dim3(256), 0, st,
                         (const double*)A->d_partials.p, A->d_cam_seg_start.p, blocks, A->C);
    hipLaunchKernelGGL(k_sum_segments9, dim3(grid_for(int64_t(A->C) * 9, 256)), dim3(256), 0, st,
                       (const double*)A->d_partials9.p, A->d_cam_seg_start.p, rhs_out, A->C);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
