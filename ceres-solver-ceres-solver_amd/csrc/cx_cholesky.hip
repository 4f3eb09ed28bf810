// Dense Cholesky of the reduced camera matrix on gfx950 (DenseCholesky::
// FactorAndSolve, dense_cholesky.cc:139-207; the reference's device variant is
// cusolverDnDpotrf/Dpotrs, dense_cholesky.cc:360-444).
//
// A (n x n, row-major, upper triangle valid) = U'U, blocked right-looking with 32-wide panels on a
// working copy W that carries the right-hand side as column n (so the forward substitution happens
// inside the factorisation).  The factorisation is a chain of n / 32 dependent steps, each far too
// short to amortise several kernel launches (a dependent kernel boundary costs 1.5-1.9 us and every launch
// re-reads its operands from L2/HBM), so a step is ONE launch, k_chol_step:
//   * every workgroup owns a 64x64 tile of the trailing matrix (32x32 per wavefront) and first forms
//     the panel rows it needs itself, X = U_kk^-T W(k, .), as an MFMA product with the explicit
//     inverse of the 32x32 diagonal block (v_mfma_f64_16x16x4_f64; the result registers are already
//     in the operand layout of the next product), then subtracts X_i' X_j -- the one genuinely
//     dense, MFMA-bound piece;
//   * tiles of the first tile row store X: together they write rows k.. of the factor;
//   * the workgroup of tile (0,0) goes on to factor and invert the next diagonal block (look-ahead): its updated
//     quadrant stays in registers and feeds the MFMA potrf (cxchol::potrf_inverse_regs) directly.
// Backward substitution follows, one launch per two blocks (k_trsv_bwd64).
//
// Measured alternatives (n = 3204, Dubrovnik-356): five kernels per step (potrf, per-column trsm,
// syrk, blocked forward/backward substitution) 8.2 ms; the whole solve as one cooperative kernel with
// grid barriers 10.2 ms (one workgroup per CU cannot hide the memory latency of its tiles after every
// barrier's L2 invalidation); this file 2.3 ms (2.0 ms factor + 0.3 ms backward substitution; 4.6 ms while an
// agent-scope fence and a barrier stood in front of the look-ahead and the backward substitution took one launch
// per 32-row block; 2.9 ms with the scalar diagonal-block kernel and a store/reload between update and look-ahead).
#include <cstdlib>

#include "cx_chol_blocks.h"
#include "cx_internal.h"
#include "cx_schur.h"

namespace {

using cxchol::NB;
using cxchol::double4_t;
using cxchol::readlane_f64;

// absolute-index wrappers of the shared blocks (cx_chol_blocks.h) for the dense working copy W (ld ldw) and factor F (ld n)
__device__ __forceinline__ void potrf_inverse_block(const double* __restrict__ W, int ldw, double* __restrict__ F, int n,
                                                    int k0, int kb, double* __restrict__ uinv, int* __restrict__ not_pd,
                                                    double* __restrict__ lds) {
  cxchol::potrf_inverse_block(W + size_t(k0) * ldw + k0, ldw, F + size_t(k0) * n + k0, n, kb, uinv, not_pd, lds);
}
__device__ __forceinline__ void panel_x(const double* __restrict__ W, int ldw, const double* __restrict__ uinv, int k0, int kb,
                                        int c0, int cmax, double4_t (&X)[2][2]) {
  cxchol::panel_x(W + size_t(k0) * ldw + c0, ldw, uinv, kb, cmax - c0 + 1, X);
}

// store X (see panel_x) as rows k0.. of the factor: columns < n into F, column n (the right-hand side) into y
__device__ __forceinline__ void store_x(const double4_t (&X)[2][2], double* __restrict__ F, int n, double* __restrict__ y,
                                        int k0, int kb, int c0) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = 16 * mt + lk + 4 * g;
        const int c = c0 + 16 * nt + li;
        if (m < kb) {
          if (c < n) F[size_t(k0 + m) * n + c] = X[mt][nt][g];
          else if (c == n) y[k0 + m] = X[mt][nt][g];
        }
      }
}

// Tile (ti, tj) of step k0 by the whole workgroup: wavefront w owns the 32 x 32 quadrant
// rows i0 = rest + 64 ti + 32 (w >> 1), columns j0 = rest + 64 tj + 32 (w & 1) of the trailing matrix
// (column n = right-hand side).  It forms X_i and X_j itself (the triangular solve of the panel, redone
// per tile on the matrix cores instead of a separate step with its own barrier) and subtracts X_i' X_j.
// Tiles of the first tile row also store X_j: together they write rows k0.. of the factor.
// keep != nullptr (the look-ahead wavefront): the updated quadrant is also returned in registers, padded for
// cxchol::potrf_inverse_regs (unit diagonal outside the next diagonal block, zeros below the diagonal).
template <bool SHARE>
__device__ __forceinline__ void fused_tile(double* __restrict__ W, int ldw, double* __restrict__ F, int n, double* __restrict__ y,
                                           const double* __restrict__ uinv, int k0, int kb, int ti, int tj, bool update,
                                           double4_t (*keep)[2][2] = nullptr, double* __restrict__ xshare = nullptr) {
  const int rest = k0 + kb;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int i0 = rest + ti * 64 + (wave >> 1) * 32;
  const int j0 = rest + tj * 64 + (wave & 1) * 32;
  // the quadrant this wavefront updates is requested first, so that its latency overlaps the panel's (one memory
  // round trip instead of two dependent ones)
  double4_t Wq[2][2];
  if (update) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = i0 + a * 16 + lk + 4 * g;
          const int j = j0 + b * 16 + li;
          Wq[a][b][g] = (i < n && j <= n && j >= i) ? W[size_t(i) * ldw + j] : 0.0;
        }
  }
  double4_t Xj[2][2], Xi[2][2];
  if constexpr (SHARE) {
    // The panel solve X = U_kk^-T W(k, .) of the tile's four 32-column blocks -- two of its columns, two of its rows --
    // is done ONCE per workgroup: wavefront w solves block w (0, 1: the column blocks, which the first tile row also
    // stores as rows of the factor; 2, 3: the row blocks, equal to the column blocks on a diagonal tile) and the
    // others pick it up from LDS, where the result registers ARE the operand layout.  64 instead of 96 matrix
    // instructions per wavefront and step: at 105 cycles each (tools/mfma_peak.hip) they were two thirds of an early
    // step's time.
    const bool diag_tile = ti == tj;
    const int c0 = wave < 2 ? rest + tj * 64 + 32 * wave : rest + ti * 64 + 32 * (wave - 2);
    double* mine = xshare + wave * 1024;
    if (wave < 2 || !diag_tile) {
      double4_t X[2][2];
      panel_x(W, ldw, uinv, k0, kb, c0, n, X);
      if (ti == 0 && wave < 2) store_x(X, F, n, y, k0, kb, c0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) mine[((mt * 2 + nt) * 4 + g) * 64 + lane] = X[mt][nt][g];
    }
    __syncthreads();
    const double* pj = xshare + (wave & 1) * 1024;
    const double* pi = xshare + ((diag_tile ? 0 : 2) + (wave >> 1)) * 1024;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Xj[mt][nt][g] = pj[((mt * 2 + nt) * 4 + g) * 64 + lane];
          Xi[mt][nt][g] = pi[((mt * 2 + nt) * 4 + g) * 64 + lane];
        }
  } else {
    panel_x(W, ldw, uinv, k0, kb, j0, n, Xj);
    if (ti == 0 && (wave >> 1) == 0) store_x(Xj, F, n, y, k0, kb, j0);
    if (!update) return;
    if (i0 == j0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) Xi[mt][nt] = Xj[mt][nt];
    } else {
      panel_x(W, ldw, uinv, k0, kb, i0, n, Xi);
    }
  }
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xi[mt][a][g], Xj[mt][b][g], acc[a][b], 0, 0, 0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = i0 + a * 16 + lk + 4 * g;
        const int j = j0 + b * 16 + li;
        const double nv = Wq[a][b][g] - acc[a][b][g];
        if (i < n && j <= n && j >= i) W[size_t(i) * ldw + j] = nv;
        if (keep) {
          const int kbn = min(NB, n - i0), r = a * 16 + lk + 4 * g, c = b * 16 + li;
          (*keep)[a][b][g] = (r < kbn && c < kbn && c >= r) ? nv : ((r == c) ? 1.0 : 0.0);
        }
      }
}

// W (n x ldw, ldw > n) = upper triangle of A with the right-hand side as column n
__global__ __launch_bounds__(256) void k_chol_augment(const double* __restrict__ A, const double* __restrict__ rhs, int n, int ldw,
                                                      double* __restrict__ W) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) { if (j >= i) W[size_t(i) * ldw + j] = A[size_t(i) * n + j]; }
  else if (j == n) W[size_t(i) * ldw + n] = rhs[i];
}

// first diagonal block
__global__ __launch_bounds__(64) void k_chol_first(const double* __restrict__ W, int ldw, double* __restrict__ F, int n,
                                                   double* __restrict__ uinv, int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  potrf_inverse_block(W, ldw, F, n, 0, min(NB, n), uinv, not_pd, lds);
}

// One block step (see the file header).  Block b > 0 owns tile number b of the upper block trapezoid of
// the trailing matrix (row ti has Tc - ti tiles; the last column tile holds the right-hand side); block
// 0 owns tile (0,0) and the look-ahead.  Tr == 0 (last block step): only the right-hand side is left.
template <bool SHARE>
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ W, int ldw, double* __restrict__ F, int n,
                                                   double* __restrict__ y, double* __restrict__ uinv, int k0,
                                                   int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  __shared__ double xshare[4 * 1024];  // the four 32 x 32 panel blocks of the tile (fused_tile)
  const int kb = min(NB, n - k0);
  const int rest = k0 + kb;
  const int rem = n - rest;
  const double* ui = uinv + size_t(k0 / NB) * NB * NB;  // every block's inverse is kept: the backward substitution uses them
  const int Tr = (rem + 63) / 64;
  const int Tc = (rem + 1 + 63) / 64;
  if (Tr == 0) {
    fused_tile<false>(W, ldw, F, n, y, ui, k0, kb, 0, 0, false);
    return;
  }
  int ti = 0, first = 0;
  const int t = blockIdx.x;
  while (t >= first + (Tc - ti)) { first += Tc - ti; ++ti; }
  const bool lookahead = t == 0 && threadIdx.x < 64;
  double4_t next[2][2];
  fused_tile<SHARE>(W, ldw, F, n, y, ui, k0, kb, ti, ti + (t - first), true, lookahead ? &next : nullptr, xshare);
  if (lookahead) {
    // look-ahead: the next diagonal block is the quadrant this very wavefront has just updated -- it goes on to the
    // factorisation in registers (round 1 stored it, waited for the stores and loaded it again: two dependent memory
    // round trips of ~2 us each on the critical path of every step)
    cxchol::potrf_inverse_regs(next, F + size_t(rest) * n + rest, n, min(NB, n - rest), uinv + size_t(rest / NB) * NB * NB, not_pd, lds);
  }
}

// Backward substitution, two blocks (64 rows k0 .. k0 + 63) per launch, right-looking: every workgroup first solves
// the 64 x 64 triangular system of the group with the stored inverses of its two diagonal blocks,
//   x2 = U22^-1 y2,  x1 = U11^-1 (y1 - U12 x2)
// (three 32 x 32 matrix-vector products; workgroup 0 stores x), then workgroup w subtracts U(i, group) x from the
// 64 rows i = 64 w .. above the group.  Half as many dependent launches as one block per launch (a launch of this
// size is ~10 us of dispatch and drain around ~5 us of work) and no 32-step substitution chain.
// out[m] = sum_c M[m][c] v[c]; 8 threads per row, 4 columns each.  In two halves: the matrix entries do not depend on the
// vectors, so a kernel requests those of all its products up front and only LDS-resident vectors are left on its chain of
// dependent steps.
__device__ __forceinline__ void gemv32_load(const double* __restrict__ M, int ldm, bool upper_only, int rows_valid, int cols_valid,
                                            double (&a)[4]) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * part + q;
    const bool in = m < rows_valid && c < cols_valid && (!upper_only || c >= m);
    a[q] = in ? M[size_t(m) * ldm + c] : 0.0;
  }
}
__device__ __forceinline__ void gemv32_apply(const double (&a)[4], const double* __restrict__ v, double* __restrict__ out) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += a[q] * v[4 * part + q];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (part == 0) out[m] = s;
}

__global__ __launch_bounds__(256) void k_trsv_bwd64(const double* __restrict__ A, int n, int k0, const double* __restrict__ uinv,
                                                    double* __restrict__ x, double* __restrict__ sol) {
  __shared__ double y1[NB], y2[NB], x1[NB], x2[NB], tmp[NB];
  const int t = threadIdx.x;
  const int kb1 = min(NB, n - k0), kb2 = max(0, min(NB, n - k0 - NB));
  double a22[4], a12[4], a11[4];  // (requested before the right-hand side is read: three dependent steps of memory latency less)
  {
    const double* __restrict__ ui = uinv + size_t(k0 / NB) * NB * NB;
    gemv32_load(ui + NB * NB, NB, true, kb2 > 0 ? NB : 0, NB, a22);
    gemv32_load(A + size_t(k0) * n + k0 + NB, n, false, kb1, kb2, a12);
    gemv32_load(ui, NB, true, NB, NB, a11);
  }
  if (t < NB) {
    y1[t] = t < kb1 ? x[k0 + t] : 0.0;
    y2[t] = t < kb2 ? x[k0 + NB + t] : 0.0;
  }
  __syncthreads();
  if (kb2 > 0) {
    gemv32_apply(a22, y2, x2);                                              // x2 = U22^-1 y2 (identity-padded inverse)
    __syncthreads();
    gemv32_apply(a12, x2, tmp);                                             // U12 x2
    __syncthreads();
    if (t < NB) y1[t] -= tmp[t];
  } else if (t < NB) {
    x2[t] = 0.0;
  }
  __syncthreads();
  gemv32_apply(a11, y1, x1);                                                // x1 = U11^-1 (y1 - U12 x2)
  __syncthreads();
  if (blockIdx.x == 0 && t < 64) {
    const double v = t < NB ? x1[t] : x2[t - NB];
    if (t < kb1 + kb2) sol[k0 + t] = v;   // not into x: other workgroups still read the group's y
  }
  // 4 threads per row above the group, 16 columns of the group each
  const int gi = blockIdx.x * 64 + (t >> 2);
  const int part = t & 3;
  double s = 0.0;
  if (gi < k0) {
    const double* __restrict__ row = A + size_t(gi) * n + k0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = 16 * part + q;
      const double xv = c < NB ? x1[c] : x2[c - NB];
      s += c < kb1 + kb2 ? row[c] * xv : 0.0;
    }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (gi < k0 && part == 0) x[gi] -= s;
}

}  // namespace

int cxd_cholesky_solve(cx_context* ctx, int n, double* a, const double* rhs, double* x, int* d_flag) {
  if (n <= 0) return CX_OK;
  hipStream_t st = ctx->stream;
  const int ldw = ((n + 1 + 15) / 16) * 16;
  const size_t wsize = size_t(n) * ldw;
  CX_TRY(ctx->chol_scratch.alloc(wsize + size_t(n) + size_t((n + NB - 1) / NB) * NB * NB));
  double* W = ctx->chol_scratch.p;
  double* y = W + wsize;  // U^-T rhs, then updated in place by the backward substitution
  double* uinv = y + n;  // one inverted diagonal block per 32 rows
  hipLaunchKernelGGL(k_chol_augment, dim3((n + 1 + 255) / 256, n), dim3(256), 0, st, (const double*)a, rhs, n, ldw, W);
  hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(64), 0, st, (const double*)W, ldw, a, n, uinv, d_flag);
  static const bool share_panel = std::getenv("CX_CHOL_NO_PANEL_SHARING") == nullptr;  // A/B switch
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int rem = n - std::min(n, k0 + NB);
    const int Tr = (rem + 63) / 64, Tc = (rem + 1 + 63) / 64;
    const int NT = Tr == 0 ? 1 : Tr * Tc - Tr * (Tr - 1) / 2;
    // (A/B: capping the kernel at 168 registers -- three instead of two workgroups per CU, no spills -- changed nothing:
    // reduced solve 2.13 ms either way on Dubrovnik-356)
    if (share_panel) hipLaunchKernelGGL(k_chol_step<true>, dim3(NT), dim3(256), 0, st, W, ldw, a, n, y, uinv, k0, d_flag);
    else hipLaunchKernelGGL(k_chol_step<false>, dim3(NT), dim3(256), 0, st, W, ldw, a, n, y, uinv, k0, d_flag);
  }
  CX_HIP(hipGetLastError());
  const int last = ((n - 1) / 64) * 64;
  for (int k0 = last; k0 >= 0; k0 -= 64)
    hipLaunchKernelGGL(k_trsv_bwd64, dim3(std::max(1, (k0 + 63) / 64)), dim3(256), 0, st, (const double*)a, n, k0, (const double*)uinv, y, x);
  CX_HIP(hipGetLastError());
  return CX_OK;
}
