// Banded Cholesky for the visibility based preconditioners (VisibilityBasedPreconditioner::UpdateImpl / Factorize /
// RightMultiplyAndAccumulate, visibility_based_preconditioner.cc:321-434; the reference hands a CRS copy of the
// matrix to a sparse Cholesky library that is not part of its tree -- this factorisation is this library's own
// and parity is stated on M^-1 r).
//
// With the cameras ordered along the paths of the cluster forest (cx_visibility.cpp) the preconditioner matrix M
// is block tridiagonal in cluster blocks, its Cholesky factor has no fill outside that profile, and independent
// paths (every cluster, for CLUSTER_JACOBI) factor independently:
//   storage   diagonal-aligned band, entry (r, c) at r * ld + c: the dense kernels' addressing with leading
//             dimension ld, valid while only r <= c < column end of the block row is touched
//   assembly  the gather assembly of the explicit S (cx_schur.hip) restricted to the selected cells' work items,
//             scattered into the band (k_band_assemble)
//   factor    the dense solver's step (cx_cholesky.hip): one launch per 32-row block step, taken by ALL paths at
//             once (blockIdx.y = path, paths sorted by length so the active ones are a prefix); panel solve and
//             trailing update on fp64 MFMA inside the block row's column window, look-ahead factorisation and
//             inversion of the next diagonal block; the 32x32 inverses are kept for the solves
//   solve     per CG iteration ONE launch, a workgroup per path walking its blocks forward (U' y = r) and
//             backward (U x = y) with the stored inverses; clusters / paths run in parallel
// The forward/backward walk is a chain of (rows / 32) dependent steps per path: ~50 us for a cluster of 30
// cameras, but several ms when the forest is one long path through thousands of cameras.
#include <algorithm>
#include <cstdlib>

#include "cx_chol_blocks.h"
#include "cx_schur.h"
#include "cx_visibility.h"

using cxchol::NB;
using cxchol::double4_t;

namespace {

// selected S cells -> band; 81 threads per cell
__global__ __launch_bounds__(3 * 81) void k_band_assemble(const int32_t* __restrict__ sel_cells, const int32_t* __restrict__ sel_offdiag,
                                                          const int32_t* __restrict__ cell_c1, const int32_t* __restrict__ cell_c2,
                                                          const int32_t* __restrict__ cell_item_start,
                                                          const double* __restrict__ item_partial, const double* __restrict__ diag,
                                                          const double* __restrict__ Df, const int32_t* __restrict__ cam_row,
                                                          double* __restrict__ W, int ld, int64_t num_sel, double offdiag_scale) {
  const int64_t k = int64_t(blockIdx.x) * 3 + threadIdx.x / 81;
  if (k >= num_sel) return;
  const int el = threadIdx.x % 81;
  const int64_t cell = sel_cells[k];
  const int c1 = cell_c1[cell], c2 = cell_c2[cell];
  const int a = el / 9, c = el - a * 9;
  double v = 0.0;
  for (int it = cell_item_start[cell]; it < cell_item_start[cell + 1]; ++it) v -= item_partial[int64_t(it) * 81 + el];
  if (c1 == c2) {
    v += diag[int64_t(c1) * 81 + el];
    if (Df && a == c) {
      const double d = Df[9 * int64_t(c1) + a];
      v += d * d;
    }
  }
  if (sel_offdiag[k]) v *= offdiag_scale;
  int row = cam_row[c1] + a, col = cam_row[c2] + c;
  if (c1 != c2 && row > col) { const int t = row; row = col; col = t; }  // the cell lands transposed
  if (row > col) return;                                                 // lower half of a diagonal cell
  W[size_t(row) * ld + col] = v;
}

__global__ void k_band_pad(const int32_t* __restrict__ row_src, double* __restrict__ W, int ld, int N) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < N && row_src[r] < 0) W[size_t(r) * ld + r] = 1.0;
}

// first diagonal block of every path
__global__ __launch_bounds__(64) void k_band_first(const double* __restrict__ W, double* __restrict__ F, int ld,
                                                   const int32_t* __restrict__ path_first_blk, double* __restrict__ uinv,
                                                   int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  const int b = path_first_blk[blockIdx.x];
  const size_t off = size_t(NB * b) * ld + NB * b;
  cxchol::potrf_inverse_block(W + off, ld, F + off, ld, NB, uinv + size_t(b) * NB * NB, not_pd, lds);
}

// Block step s of path blockIdx.y: block row b = first block of the path + s, trailing window
// [rest, cend) x [rest, cend), workgroup blockIdx.x owns one 64x64 tile of its upper triangle (a 32x32 quadrant per
// wavefront), forms the panel pieces it needs as MFMA products with the inverse of the diagonal block and
// subtracts X_i' X_j; first-row tiles store X = rows of the factor; tile (0, 0) goes on to the next diagonal block.
__global__ __launch_bounds__(256) void k_band_step(double* __restrict__ W, double* __restrict__ F, int ld,
                                                   const int32_t* __restrict__ path_first_blk, const int32_t* __restrict__ path_num_blk,
                                                   const int32_t* __restrict__ blk_cend, double* __restrict__ uinv, int s,
                                                   int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  const int path = blockIdx.y;
  const int nblk = path_num_blk[path];
  if (s >= nblk) return;
  const int b = path_first_blk[path] + s;
  const int k0 = NB * b, rest = k0 + NB, cend = blk_cend[b];
  const int T = (cend - rest + 63) / 64;
  const int t = blockIdx.x;
  const double* ui = uinv + size_t(b) * NB * NB;
  if (t < T * (T + 1) / 2) {
    int ti = 0, first = 0;
    while (t >= first + (T - ti)) { first += T - ti; ++ti; }
    const int tj = ti + (t - first);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = rest + ti * 64 + (wave >> 1) * 32;
    const int j0 = rest + tj * 64 + (wave & 1) * 32;
    double4_t Xj[2][2];
    cxchol::panel_x(W + size_t(k0) * ld + j0, ld, ui, NB, cend - j0, Xj);
    if (ti == 0 && (wave >> 1) == 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int m = 16 * mt + lk + 4 * g;
            const int c = j0 + 16 * nt + li;
            if (c < cend) F[size_t(k0 + m) * ld + c] = Xj[mt][nt][g];
          }
    }
    double4_t Xi[2][2];
    if (i0 == j0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) Xi[mt][nt] = Xj[mt][nt];
    } else {
      cxchol::panel_x(W + size_t(k0) * ld + i0, ld, ui, NB, cend - i0, Xi);
    }
    double4_t acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xi[mt][x][g], Xj[mt][y][g], acc[x][y], 0, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = i0 + x * 16 + lk + 4 * g;
          const int j = j0 + y * 16 + li;
          if (i < cend && j < cend && j >= i) W[size_t(i) * ld + j] -= acc[x][y][g];
        }
  }
  if (t == 0 && s + 1 < nblk && threadIdx.x < 64) {
    // the next diagonal block is the quadrant this wavefront has just updated (see k_chol_step)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const size_t off = size_t(rest) * ld + rest;
    cxchol::potrf_inverse_block(W + off, ld, F + off, ld, NB, uinv + size_t(b + 1) * NB * NB, not_pd, lds);
  }
}

constexpr int kUiStride = NB + 1;  // padded rows: row and column walks of the staged inverse are both conflict free

__device__ __forceinline__ void stage_inverse(const double* __restrict__ uinv_b, double* __restrict__ ui) {
  for (int i = threadIdx.x; i < NB * NB; i += 256) ui[(i / NB) * kUiStride + (i % NB)] = uinv_b[i];
}

// z = M^-1 r: one workgroup per path.  Forward, right-looking: y_b = U_bb^-T t_b, then t(c) -= U(b, c)' y_b for the
// columns of the block row's window; backward, left-looking: x_b = U_bb^-1 (y_b - U(b, window) x(window)).
__global__ __launch_bounds__(256) void k_band_solve(const double* __restrict__ F, int ld, const double* __restrict__ uinv,
                                                    const int32_t* __restrict__ path_first_blk, const int32_t* __restrict__ path_num_blk,
                                                    const int32_t* __restrict__ blk_cend, const int32_t* __restrict__ row_src,
                                                    const double* __restrict__ r_in, double* __restrict__ y, double* __restrict__ z_out) {
  __shared__ double ui[NB * kUiStride];
  __shared__ double tb[NB], yb[NB];
  const int tid = threadIdx.x;
  const int b0 = path_first_blk[blockIdx.x], nb = path_num_blk[blockIdx.x];
  const int r0 = NB * b0, r1 = NB * (b0 + nb);
  for (int r = r0 + tid; r < r1; r += 256) {
    const int src = row_src[r];
    y[r] = src >= 0 ? r_in[src] : 0.0;
  }
  __syncthreads();
  for (int s = 0; s < nb; ++s) {
    const int b = b0 + s, k0 = NB * b, rest = k0 + NB, cend = blk_cend[b];
    stage_inverse(uinv + size_t(b) * NB * NB, ui);
    if (tid < NB) tb[tid] = y[k0 + tid];
    __syncthreads();
    if (tid < 64) {
      const int m = tid & 31, half = tid >> 5;
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += ui[(16 * half + q) * kUiStride + m] * tb[16 * half + q];
      sum += __shfl_xor(sum, 32, 64);
      if (half == 0) {
        yb[m] = sum;
        y[k0 + m] = sum;
      }
    }
    __syncthreads();
    for (int c = rest + tid; c < cend; c += 256) {
      const double* __restrict__ col = F + size_t(k0) * ld + c;
      double acc = 0.0;
#pragma unroll 8
      for (int m = 0; m < NB; ++m) acc = fma(col[size_t(m) * ld], yb[m], acc);
      y[c] -= acc;
    }
    __syncthreads();
  }
  for (int s = nb - 1; s >= 0; --s) {
    const int b = b0 + s, k0 = NB * b, rest = k0 + NB, cend = blk_cend[b];
    stage_inverse(uinv + size_t(b) * NB * NB, ui);
    {
      const int m = tid >> 3, part = tid & 7;
      const double* __restrict__ row = F + size_t(k0 + m) * ld;
      double acc = 0.0;
      for (int c = rest + part; c < cend; c += 8) acc = fma(row[c], y[c], acc);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (part == 0) tb[m] = y[k0 + m] - acc;
    }
    __syncthreads();
    if (tid < 64) {
      const int m = tid & 31, half = tid >> 5;
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += ui[m * kUiStride + 16 * half + q] * tb[16 * half + q];
      sum += __shfl_xor(sum, 32, 64);
      if (half == 0) y[k0 + m] = sum;
    }
    __syncthreads();
  }
  for (int r = r0 + tid; r < r1; r += 256) {
    const int src = row_src[r];
    if (src >= 0) z_out[src] = y[r];
  }
}

// The same walk with the active part of the vector in LDS: at block step b only y(c), c in [k0, column end of b),
// is live, a ring of kWin doubles indexed by c mod kWin.  The dependency chain then runs through LDS only; what
// comes from HBM in a step (the block row of the factor, the next inverse, the columns that enter the window) does
// not depend on it, and the inverse and the new columns are fetched one step ahead.  Two barriers per step instead
// of three and no global read-after-write inside the chain: 6.5 -> see DESIGN.md us per step on the long path of
// the Final shape.  Needs band + 64 <= kWin (otherwise k_band_solve).
constexpr int kWin = 4096;

// NT threads per workgroup: the walk is bound by the instruction issue of its one workgroup (per step ~100 KB of the
// factor as 8-byte loads plus as many FMAs), so more wavefronts per path mean fewer instructions per wavefront.
template <int NT>
__global__ __launch_bounds__(NT) void k_band_solve_lds(const double* __restrict__ F, int ld, const double* __restrict__ uinv,
                                                       const int32_t* __restrict__ path_first_blk, const int32_t* __restrict__ path_num_blk,
                                                       const int32_t* __restrict__ blk_cend, const int32_t* __restrict__ row_src,
                                                       const double* __restrict__ r_in, double* __restrict__ y, double* __restrict__ z_out) {
  constexpr int KCF = NT >= 512 ? 1 : 512 / NT;       // chunks of new window columns fetched a step ahead
  constexpr int PU = NB * NB / NT;                    // pieces of the next inverse per thread
  constexpr int PART = NT / NB;                       // threads per row in the backward dot products
  constexpr int JB = 512 / PART;                      // loads in flight per thread and pass (512 columns of the row)
  __shared__ double yw[kWin];
  __shared__ double ui[2][NB * kUiStride];
  __shared__ double tb[NB], yb[NB];
  const int tid = threadIdx.x;
  const int b0 = path_first_blk[blockIdx.x], nb = path_num_blk[blockIdx.x];
  const int r0 = NB * b0, r1 = NB * (b0 + nb);
  // r in band order
  for (int r = r0 + tid; r < r1; r += NT) {
    const int src = row_src[r];
    y[r] = src >= 0 ? r_in[src] : 0.0;
  }
  __syncthreads();
  {
    const int hi = blk_cend[b0];
    for (int c = r0 + tid; c < hi; c += NT) yw[c & (kWin - 1)] = y[c];
    for (int i = tid; i < NB * NB; i += NT) ui[0][(i / NB) * kUiStride + (i % NB)] = uinv[size_t(b0) * NB * NB + i];
  }
  __syncthreads();
  // ---- forward: U' y = r
  for (int s = 0; s < nb; ++s) {
    const int b = b0 + s, k0 = NB * b, rest = k0 + NB, cend = blk_cend[b];
    const bool more = s + 1 < nb;
    const int next_cend = more ? blk_cend[b + 1] : cend;
    double pu[PU], pn[KCF];
    if (more) {
#pragma unroll
      for (int q = 0; q < PU; ++q) pu[q] = uinv[size_t(b + 1) * NB * NB + tid + NT * q];
#pragma unroll
      for (int q = 0; q < KCF; ++q) {
        const int c = cend + tid + NT * q;
        pn[q] = c < next_cend ? y[c] : 0.0;
      }
    }
    // this step's block row of the factor: all loads in flight at once, before the barrier (they do not depend
    // on y_b).  A thread owns two adjacent columns (16-byte loads: rows start at multiples of 16 doubles and the
    // window at a multiple of 32); the first 2 NT columns of the window, wider windows take the loop below
    double2 f[NB];
    {
      const double* __restrict__ col = F + size_t(k0) * ld;
      const int c = rest + 2 * tid;
#pragma unroll
      for (int m = 0; m < NB; ++m) f[m] = c < cend ? *reinterpret_cast<const double2*>(col + size_t(m) * ld + c) : double2{0.0, 0.0};
    }
    if (tid < 64) {
      const int m = tid & 31, half = tid >> 5;
      const double* __restrict__ u = ui[s & 1];
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += u[(16 * half + q) * kUiStride + m] * yw[(k0 + 16 * half + q) & (kWin - 1)];
      sum += __shfl_xor(sum, 32, 64);
      if (half == 0) {
        yb[m] = sum;
        y[k0 + m] = sum;
      }
    }
    __syncthreads();
    {
      const int c = rest + 2 * tid;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        a0 = fma(f[m].x, yb[m], a0);
        a1 = fma(f[m].y, yb[m], a1);
      }
      // (the column end is even only when the band's last row block is: guard the odd column separately)
      if (c < cend) yw[c & (kWin - 1)] -= a0;
      if (c + 1 < cend) yw[(c + 1) & (kWin - 1)] -= a1;
    }
    for (int c = rest + 2 * NT + tid; c < cend; c += NT) {
      const double* __restrict__ col = F + size_t(k0) * ld + c;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < NB; ++m) acc = fma(col[size_t(m) * ld], yb[m], acc);
      yw[c & (kWin - 1)] -= acc;
    }
    if (more) {
      double* __restrict__ un = ui[(s + 1) & 1];
#pragma unroll
      for (int q = 0; q < PU; ++q) {
        const int i = tid + NT * q;
        un[(i / NB) * kUiStride + (i % NB)] = pu[q];
      }
#pragma unroll
      for (int q = 0; q < KCF; ++q) {
        const int c = cend + tid + NT * q;
        if (c < next_cend) yw[c & (kWin - 1)] = pn[q];
      }
      for (int c = cend + NT * KCF + tid; c < next_cend; c += NT) yw[c & (kWin - 1)] = y[c];
    }
    __syncthreads();
  }
  // ---- backward: U x = y.  The window now holds x(c), c in [rest, column end); ui[(nb - 1) & 1] is still staged.
  if (tid < NB) tb[tid] = y[r1 - NB + tid];  // y of the last block (tb doubles as the "current y block")
  __syncthreads();
  for (int s = nb - 1; s >= 0; --s) {
    const int b = b0 + s, k0 = NB * b, rest = k0 + NB, cend = blk_cend[b];
    const bool more = s > 0;
    double pu[PU], py = 0.0;
    if (more) {
#pragma unroll
      for (int q = 0; q < PU; ++q) pu[q] = uinv[size_t(b - 1) * NB * NB + tid + NT * q];
      if (tid < NB) py = y[k0 - NB + tid];
    }
    {
      const int m = tid / PART, part = tid % PART;
      const double* __restrict__ row = F + size_t(k0 + m) * ld;
      double acc = 0.0;
      // pairs of adjacent columns (16-byte loads), JB / 2 loads in flight per pass over 512 columns of the row
      for (int base = rest + 2 * part; base < cend; base += 512) {
        double2 fr[JB / 2];
#pragma unroll
        for (int j = 0; j < JB / 2; ++j) fr[j] = base + 2 * PART * j < cend ? *reinterpret_cast<const double2*>(row + base + 2 * PART * j) : double2{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < JB / 2; ++j) {
          const int c = base + 2 * PART * j;
          // (stale slots of the ring may hold NaN: no multiplication by zero instead of the guards)
          acc = c < cend ? fma(fr[j].x, yw[c & (kWin - 1)], acc) : acc;
          acc = c + 1 < cend ? fma(fr[j].y, yw[(c + 1) & (kWin - 1)], acc) : acc;
        }
      }
#pragma unroll
      for (int d = 1; d < PART; d <<= 1) acc += __shfl_xor(acc, d, 64);
      if (part == 0) yb[m] = tb[m] - acc;
    }
    __syncthreads();
    if (tid < 64) {
      const int m = tid & 31, half = tid >> 5;
      const double* __restrict__ u = ui[s & 1];
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += u[m * kUiStride + 16 * half + q] * yb[16 * half + q];
      sum += __shfl_xor(sum, 32, 64);
      if (half == 0) {
        yw[(k0 + m) & (kWin - 1)] = sum;
        y[k0 + m] = sum;
      }
    }
    if (more) {
      double* __restrict__ un = ui[(s - 1) & 1];
#pragma unroll
      for (int q = 0; q < PU; ++q) {
        const int i = tid + NT * q;
        un[(i / NB) * kUiStride + (i % NB)] = pu[q];
      }
      if (tid < NB) tb[tid] = py;
    }
    __syncthreads();
  }
  for (int r = r0 + tid; r < r1; r += NT) {
    const int src = row_src[r];
    if (src >= 0) z_out[src] = y[r];
  }
}

}  // namespace

// band or tile-sparse factorisation for this plan (see cx_vis_plan::use_sparse); CX_VISIBILITY_SPARSE=0/1 forces one
static int cxv_use_sparse(cx_matrix* A, cx_vis_plan* plan) {
  if (plan->use_sparse >= 0) return CX_OK;
  const char* env = std::getenv("CX_VISIBILITY_SPARSE");
  const int longest = plan->num_paths ? plan->path_num_blk[0] : 0;
  bool sparse = env ? std::atoi(env) != 0 : (plan->preconditioner_type == CX_CLUSTER_TRIDIAGONAL && longest > 64);
  if (sparse) {
    std::vector<int32_t> c1(plan->sel_cells.size()), c2(plan->sel_cells.size());
    for (size_t k = 0; k < plan->sel_cells.size(); ++k) {
      c1[k] = A->h_cell_c1[size_t(plan->sel_cells[k])];
      c2[k] = A->h_cell_c2[size_t(plan->sel_cells[k])];
    }
    if (A->ctx->nranks > 1) {
      // Sharded matrix: the tile pool is summed over the ranks entry by entry (cxv_factor), so every rank must hold the
      // SAME dissection order and tile list -- planned from the block pairs of all ranks, not from this rank's cells
      // (which come from its own points only and differ from rank to rank).
      c1 = plan->global_pair_c1;
      c2 = plan->global_pair_c2;
    }
    CX_TRY(cxsp_plan_from_cells(A->ctx, A->C, c1.data(), c2.data(), int64_t(c1.size()), &plan->sp));
    if (plan->sp.state != 1) sparse = false;
  }
  plan->use_sparse = sparse ? 1 : 0;
  return CX_OK;
}

int cxv_factor(cx_matrix* A, cx_vis_plan* plan, const double* D, bool halve_offdiag, int* d_flag) {
  hipStream_t st = A->ctx->stream;
  CX_TRY(cxv_use_sparse(A, plan));
  if (plan->use_sparse == 1) {
    CX_TRY(cxs_assemble_pair_items(A, D, plan->d_sel_items.p, plan->num_sel_items));
    const bool shard = A->ctx->nranks > 1;
    const double* Dfs = (D && (!shard || A->ctx->rank == 0)) ? D + 3 * int64_t(A->P) : nullptr;
    CX_TRY(cxsp_assemble(A, &plan->sp, Dfs, plan->d_sel_cells.p, plan->d_sel_offdiag.p, int64_t(plan->sel_cells.size()),
                         halve_offdiag ? 0.5 : 1.0));
    // sharded matrix: every rank has assembled the contributions of its points into the same tile layout
    if (shard) CX_TRY(cx_allreduce_device(A->ctx, plan->sp.d_W.p, plan->sp.num_tiles * 4096));
    return cxsp_factor(A->ctx, &plan->sp, d_flag);
  }
  const int N = plan->N, ld = plan->ld;
  const size_t band = size_t(N) * size_t(ld + 1);
  CX_TRY(plan->d_W.alloc(band));
  CX_TRY(plan->d_F.alloc(band));
  CX_TRY(plan->d_uinv.alloc(size_t(N / NB) * NB * NB));
  CX_TRY(plan->d_y.alloc(size_t(N)));
  CX_HIP(hipMemsetAsync(plan->d_W.p, 0, band * sizeof(double), st));
  CX_TRY(cxs_assemble_pair_items(A, D, plan->d_sel_items.p, plan->num_sel_items));
  const int64_t num_sel = int64_t(plan->sel_cells.size());
  // sharded matrix: every rank assembles the contributions of its points into the same band layout, the band is
  // summed over the ranks, D_f^2 enters once (rank 0) and the unit diagonal of the padding rows after the sum
  const bool sharded = A->ctx->nranks > 1;
  const double* Df = (D && (!sharded || A->ctx->rank == 0)) ? D + 3 * int64_t(A->P) : nullptr;
  hipLaunchKernelGGL(k_band_assemble, dim3(unsigned((num_sel + 2) / 3)), dim3(3 * 81), 0, st, (const int32_t*)plan->d_sel_cells.p,
                     (const int32_t*)plan->d_sel_offdiag.p, (const int32_t*)A->d_cell_c1.p, (const int32_t*)A->d_cell_c2.p,
                     (const int32_t*)A->d_cell_item_start.p, (const double*)A->d_item_partial.p, (const double*)A->d_elim_diag.p, Df,
                     (const int32_t*)plan->d_cam_row.p, plan->d_W.p, ld, num_sel, halve_offdiag ? 0.5 : 1.0);
  if (sharded) CX_TRY(cx_allreduce_device(A->ctx, plan->d_W.p, int64_t(band)));
  hipLaunchKernelGGL(k_band_pad, dim3((N + 255) / 256), dim3(256), 0, st, (const int32_t*)plan->d_row_src.p, plan->d_W.p, ld, N);
  hipLaunchKernelGGL(k_band_first, dim3(unsigned(plan->num_paths)), dim3(64), 0, st, (const double*)plan->d_W.p, plan->d_F.p, ld,
                     (const int32_t*)plan->d_path_first_blk.p, plan->d_uinv.p, d_flag);
  for (size_t s = 0; s < plan->step_paths.size(); ++s)
    hipLaunchKernelGGL(k_band_step, dim3(unsigned(std::max(1, plan->step_tiles[s])), unsigned(plan->step_paths[s])), dim3(256), 0, st,
                       plan->d_W.p, plan->d_F.p, ld, (const int32_t*)plan->d_path_first_blk.p, (const int32_t*)plan->d_path_num_blk.p,
                       (const int32_t*)plan->d_blk_cend.p, plan->d_uinv.p, int(s), d_flag);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxv_solve(cx_matrix* A, cx_vis_plan* plan, const double* r, double* z) {
  if (plan->use_sparse == 1) return cxsp_solve(A->ctx, &plan->sp, r, z);
  const bool force_global = std::getenv("CX_BAND_SOLVE_GLOBAL") != nullptr;  // A/B switch, read per call so tests can flip it
  // 512 threads per path pay off on long paths (instruction issue of the one workgroup); the short walks of
  // CLUSTER_JACOBI (a few blocks per cluster) are quicker with 256 (72 vs 78 us on the Final shape)
  static const int forced = std::getenv("CX_BAND_SOLVE_THREADS") ? std::atoi(std::getenv("CX_BAND_SOLVE_THREADS")) : 0;
  const int threads = forced ? forced : (plan->step_paths.size() > 32 ? 512 : 256);
  if (plan->ld + 64 <= kWin && !force_global) {
#define CX_LAUNCH_BAND_SOLVE(NT)                                                                                                   \
  hipLaunchKernelGGL(k_band_solve_lds<NT>, dim3(unsigned(plan->num_paths)), dim3(NT), 0, A->ctx->stream, (const double*)plan->d_F.p, \
                     plan->ld, (const double*)plan->d_uinv.p, (const int32_t*)plan->d_path_first_blk.p,                              \
                     (const int32_t*)plan->d_path_num_blk.p, (const int32_t*)plan->d_blk_cend.p, (const int32_t*)plan->d_row_src.p, \
                     r, plan->d_y.p, z)
    if (threads == 256) CX_LAUNCH_BAND_SOLVE(256);
    else CX_LAUNCH_BAND_SOLVE(512);  // 1 024 threads leave 128 registers per thread: the row block spills (6 s per solve)
#undef CX_LAUNCH_BAND_SOLVE
  }
  else
    hipLaunchKernelGGL(k_band_solve, dim3(unsigned(plan->num_paths)), dim3(256), 0, A->ctx->stream, (const double*)plan->d_F.p, plan->ld,
                       (const double*)plan->d_uinv.p, (const int32_t*)plan->d_path_first_blk.p, (const int32_t*)plan->d_path_num_blk.p,
                       (const int32_t*)plan->d_blk_cend.p, (const int32_t*)plan->d_row_src.p, r, plan->d_y.p, z);
  CX_HIP(hipGetLastError());
  return CX_OK;
}
