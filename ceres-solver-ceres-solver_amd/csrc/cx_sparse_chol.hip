// Tile-sparse Cholesky of the reduced camera matrix: the reduced solve of SPARSE_SCHUR when 9C is too
// large for a dense S (SparseSchurComplementSolver::SolveReducedLinearSystem, schur_complement_solver.cc:
// 292-335, which hands a CRS copy of S to CHOLMOD, suitesparse.cc:397-469, after a fill-reducing ordering
// of the cameras, reorder_program.cc:341-382).
//
// CHOLMOD is third-party code that is not part of the reference tree: the factorisation here is this
// library's own and parity is stated on the solution (it must equal the dense solve), not on the ordering
// ("parity unpinned" for CAMD / AMD).
//
//   ordering   nested dissection of the camera graph of S (recursive bisection of a reverse Cuthill-McKee
//              arrangement: the smaller of the two boundary sets of a cut is the separator), host, once per
//              structure -- the role of CHOLMOD's ordering step (suitesparse.cc:218-279).  CX_SPARSE_ORDERING=rcm
//              keeps the band ordering of round 1 (reverse Cuthill-McKee, optionally minimum degree on groups of 64)
//   structure  S in 64x64 tiles (upper triangle), symbolic fill at tile level; every tile row ends with one
//              extra tile that carries the right-hand side as its column 0, so the forward substitution
//              is part of the factorisation (as in the dense solver)
//   assembly   the gather assembly of cx_schur.hip (k_pair_items) scattered into the tile pool at the
//              permuted positions
//   numeric    by LEVELS of the tile elimination tree (tile rows of equal height are independent): per level
//              k_sp_diag (the diagonal tiles: two 32x32 factor + inverse blocks each), k_sp_panel (the rows of
//              the factor, F(I, J) = U_II^-T W(I, J) on fp64 MFMA, in place) and the update (every tile whose
//              window of contributions closes at this level subtracts F(I, Ja)' F(I, Jb) for its sources in
//              ascending I: a gather, no atomics, bitwise reproducible; k_sp_update_f64_lds / k_sp_update_f32_lds:
//              both source tiles staged once per workgroup through LDS).  The pool is double, or float under
//              use_mixed_precision_solves (every kernel is a template on it).  The dependent chain is the tree
//              height (about a hundred to a few hundred levels on the Final-13682 shape) instead of n / 32 = 3 848
//              block steps.
//   solve      backward substitution by the same levels, top down: one workgroup per tile row gathers
//              F(I, J) x_J over its row and applies the kept inverses; then the inverse permutation
// CX_SPARSE_CHOLESKY_STEPS=1 runs round 1's numeric phase instead (one launch per 32-column block step in
// elimination order, look-ahead of the next diagonal block), kept for A/B runs.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <queue>
#include <string>
#include <type_traits>

#include "cx_chol_blocks.h"
#include "cx_internal.h"
#include "cx_schur.h"

using cxchol::NB;
using cxchol::double4_t;
using cxchol::readlane_f64;

namespace {

constexpr int kTile = 64;
constexpr int kTileDoubles = kTile * kTile;

// position of column tile J in the sorted list of tile row I (must be present)
__device__ __forceinline__ int tile_find(const int32_t* __restrict__ row_tiles, int lo, int hi, int J) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row_tiles[mid] < J) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// The kernels below are templates on TW, the scalar type of the tile pool: double, or float for the single precision
// factorisation of use_mixed_precision_solves (see k_sp_update_f32).
// S cells (gather assembly of cx_schur.hip) -> tile pool at the permuted positions; 81 threads per cell
template <typename TW>
__global__ __launch_bounds__(3 * 81) void k_sp_assemble(const int32_t* __restrict__ cell_c1, const int32_t* __restrict__ cell_c2,
                                                        const int32_t* __restrict__ cell_item_start,
                                                        const double* __restrict__ item_partial, const double* __restrict__ diag,
                                                        const double* __restrict__ Df, const int32_t* __restrict__ cam_pos,
                                                        const int32_t* __restrict__ row_start, const int32_t* __restrict__ row_tiles,
                                                        TW* __restrict__ W, int64_t num_cells,
                                                        const int32_t* __restrict__ sel_cells, const int32_t* __restrict__ sel_offdiag,
                                                        double offdiag_scale) {
  // sel_cells != NULL: only these cells of the list (a visibility based preconditioner keeps a subset of S), the ones
  // flagged in sel_offdiag scaled (ScaleOffDiagonalCells, visibility_based_preconditioner.cc:366-393)
  const int64_t k = int64_t(blockIdx.x) * 3 + threadIdx.x / 81;
  if (k >= num_cells) return;
  const int64_t cell = sel_cells ? int64_t(sel_cells[k]) : k;
  const int el = threadIdx.x % 81;
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
  if (sel_offdiag && sel_offdiag[k]) v *= offdiag_scale;
  int row = cam_pos[c1] + a, col = cam_pos[c2] + c;  // cam_pos: first row of the camera in the padded elimination order
  if (c1 != c2 && row > col) { const int t = row; row = col; col = t; }  // the cell lands transposed
  if (row > col) return;                                                 // lower half of a diagonal cell
  const int I = row >> 6, J = col >> 6;
  const int idx = tile_find(row_tiles, row_start[I], row_start[I + 1], J);
  W[size_t(idx) * kTileDoubles + (row & 63) * kTile + (col & 63)] = TW(v);
}

// right-hand side (camera order) -> column 0 of every tile row's last tile
template <typename TW>
__global__ void k_sp_rhs(const double* __restrict__ rhs, const int32_t* __restrict__ cam_pos, const int32_t* __restrict__ row_start,
                         TW* __restrict__ W, int C, const int32_t* __restrict__ row_keep = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, a = i - 9 * c;
  const int row = cam_pos[c] + a;
  const int I = row >> 6;
  if (row_keep && !row_keep[I]) return;  // (distributed: a replicated row's right-hand side enters the sum at the split once, on rank 0)
  W[size_t(row_start[I + 1] - 1) * kTileDoubles + (row & 63) * kTile] = TW(rhs[i]);
}

// out[m] = sum_c M[m][c] v[c] for a 32 x 32 block, 8 threads per row.  In two halves, so that a kernel can request the
// matrix entries of all its products up front (they do not depend on the vectors) and only the LDS-resident vectors are
// left on its chain of dependent steps.
template <typename TM>
__device__ __forceinline__ void sp_gemv32_load(const TM* __restrict__ M, int ldm, int rows_valid, int cols_valid, double (&a)[4]) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * part + q;
    a[q] = (m < rows_valid && c < cols_valid) ? M[size_t(m) * ldm + c] : 0.0;
  }
}
__device__ __forceinline__ void sp_gemv32_apply(const double (&a)[4], const double* __restrict__ v, double* __restrict__ out) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += a[q] * v[4 * part + q];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (part == 0) out[m] = s;
}
__device__ __forceinline__ void sp_gemv32(const double* __restrict__ M, int ldm, const double* __restrict__ v, double* __restrict__ out,
                                          int rows_valid, int cols_valid) {
  double a[4];
  sp_gemv32_load(M, ldm, rows_valid, cols_valid, a);
  sp_gemv32_apply(a, v, out);
}

// ---------------------------------------------------------------------------------------------------------
// Level-scheduled factorisation (see the file header).  Everything is in place in ONE tile pool W: after its
// level a tile row holds rows of the factor U (diagonal tile: U_II, upper; other tiles: F(I, J); last tile: the
// forward-substituted right-hand side in column 0).

// X (see cxchol::panel_x) -> rows of a tile: X[m][c], m < kb, c < ncols
template <typename TW>
__device__ __forceinline__ void store_rows(const double4_t (&X)[2][2], TW* __restrict__ dst, int kb, int ncols) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = 16 * mt + lk + 4 * g, c = 16 * nt + li;
        if (m < kb && c < ncols) dst[size_t(m) * kTile + c] = TW(X[mt][nt][g]);
      }
}

// a 32 x 32 block of a tile in the operand layout of the trailing update (register g of tile (mt, nt) of lane l is
// element [16 mt + (l >> 4) + 4 g][16 nt + (l & 15)]); rows >= kb and columns >= ncols read as zero
template <typename TW>
__device__ __forceinline__ void load_operand(const TW* __restrict__ src, int kb, int ncols, double4_t (&X)[2][2]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // Float pool: an unconditional load and the padding selected afterwards -- a load inside a branch, followed by the
        // float -> double conversion, made every one of these loads a memory round trip of its own (k_sp_panel: 69 s_waitcnt
        // against 8; 3.9 -> 2.1 ms per Final factorisation, k_sp_diag 3.9 -> 3.5).  Double pool: the conditional load, which the
        // compiler keeps in flight as it is; the unconditional form costs it 2.5 -> 3.7 ms (panel) and 3.5 -> 3.9 (diag).
        // Measured per level (tools/sparse_levels.sh), both ways for both pools.
        const int m = 16 * mt + lk + 4 * g, c = 16 * nt + li;
        if constexpr (std::is_same_v<TW, double>) {
          X[mt][nt][g] = (m < kb && c < ncols) ? src[size_t(m) * kTile + c] : 0.0;
        } else {
          const TW v = src[size_t(m) * kTile + c];  // (the 32 x 32 block lies inside its 64 x 64 tile whatever kb and ncols are)
          X[mt][nt][g] = (m < kb && c < ncols) ? double(v) : 0.0;
        }
      }
}

// acc[a][b] (+)= A' B for two operands in that layout (32 rows of K): 32 x v_mfma_f64_16x16x4_f64
__device__ __forceinline__ void mfma_atb(const double4_t (&A)[2][2], const double4_t (&B)[2][2], double4_t (&acc)[2][2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mt][a][g], B[mt][b][g], acc[a][b], 0, 0, 0);
}

// dst[i][j] -= acc[i][j] for the 32 x 32 block at dst (C layout of mfma_atb), i < rows, j < cols, optionally j >= i only
__device__ __forceinline__ void subtract_block(const double4_t (&acc)[2][2], double* __restrict__ dst, int rows, int cols, bool upper_only) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = 16 * a + lk + 4 * g, j = 16 * b + li;
        if (i < rows && j < cols && (!upper_only || j >= i)) dst[size_t(i) * kTile + j] -= acc[a][b][g];
      }
}

// operands of the panel solve X = Uinv' W (cxchol::panel_x) held in registers: A(m, r) = Uinv[r][m], B(r, c) = W[r][c]
__device__ __forceinline__ void load_aop(const double* __restrict__ uinv, double (&aop)[8][2]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) aop[s][mt] = uinv[(4 * s + lk) * NB + 16 * mt + li];
}
template <typename TW>
__device__ __forceinline__ void load_bop(const TW* __restrict__ Wrow, int kb, int ncols, double (&bop)[8][2]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int r = 4 * s + lk, c = 16 * nt + li;
      if constexpr (std::is_same_v<TW, double>) {  // (see load_operand)
        bop[s][nt] = (r < kb && c < ncols) ? Wrow[size_t(r) * kTile + c] : 0.0;
      } else {
        const TW v = Wrow[size_t(r) * kTile + c];
        bop[s][nt] = (r < kb && c < ncols) ? double(v) : 0.0;
      }
    }
}
__device__ __forceinline__ void solve_x(const double (&aop)[8][2], const double (&bop)[8][2], double4_t (&X)[2][2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) X[mt][nt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) X[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s][mt], bop[s][nt], X[mt][nt], 0, 0, 0);
}

// Level, part 1: the diagonal tiles of the level's tile rows, one wavefront each.  U11 = chol(A11) and its inverse;
// U12 = U11^-T A12; A22 -= U12' U12; U22 = chol(A22) and its inverse.  The whole tile is requested up front and the chain
// runs in registers (the 32 x 32 routine takes and leaves its block in the MFMA C layout, U11^-1 stays in LDS for the
// panel product): one memory round trip instead of the four of a store / wait / reload between the stages (27 -> ... us).
template <typename TW>
__global__ __launch_bounds__(64) void k_sp_diag(TW* __restrict__ W, const int32_t* __restrict__ row_start,
                                                const int32_t* __restrict__ level_rows, const int32_t* __restrict__ valid,
                                                double* __restrict__ uinv, int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  __shared__ double inv1[NB * NB];
  const int I = level_rows[blockIdx.x];
  TW* D = W + size_t(row_start[I]) * kTileDoubles;
  const int kb1 = min(NB, valid[I]), kb2 = max(0, min(NB, valid[I] - NB));
  double* ui1 = uinv + size_t(2 * I) * NB * NB;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  // A11 and A22 in the C layout, padded (unit diagonal outside the valid block, zeros below the diagonal); A12 as the
  // B operand of the panel product
  double4_t T1[2][2], T2[2][2];
  double a12[8][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = a; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * a + 4 * g + lk, c = 16 * b + li;
        const double v1 = D[size_t(min(r, kb1 - 1)) * kTile + min(c, kb1 - 1)];
        T1[a][b][g] = (r < kb1 && c < kb1 && c >= r) ? v1 : ((r == c) ? 1.0 : 0.0);
        const int k2 = max(kb2, 1);
        const double v2 = D[size_t(NB + min(r, k2 - 1)) * kTile + NB + min(c, k2 - 1)];
        T2[a][b][g] = (r < kb2 && c < kb2 && c >= r) ? v2 : ((r == c) ? 1.0 : 0.0);
      }
  T1[1][0] = double4_t{0.0, 0.0, 0.0, 0.0};
  T2[1][0] = double4_t{0.0, 0.0, 0.0, 0.0};
  load_bop(D + NB, kb1, kb2, a12);
  cxchol::potrf_inverse_regs(T1, D, kTile, kb1, ui1, not_pd, lds, inv1);
  if (kb2 <= 0) return;
  double a1[8][2];
  load_aop(inv1, a1);  // this wavefront's own LDS writes: ordered before its LDS reads
  double4_t X[2][2];
  solve_x(a1, a12, X);
  store_rows(X, D + NB, kb1, kb2);
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  mfma_atb(X, X, acc);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = a; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * a + 4 * g + lk, c = 16 * b + li;
        if (r < kb2 && c < kb2 && c >= r) T2[a][b][g] -= acc[a][b][g];
      }
  TW* D22 = D + size_t(NB) * kTile + NB;
  cxchol::potrf_inverse_regs(T2, D22, kTile, kb2, ui1 + NB * NB, not_pd, lds);
}

// Level, part 2: the other tiles of those rows become rows of the factor, F(I, J) = U_II^-T W(I, J), in place:
// X1 = U11^-T W1; X2 = U22^-T (W2 - U12' X1).  One workgroup of two wavefronts per tile, wavefront q = columns 32 q ...
// Everything the chain needs is requested up front (both inverses, U12, both halves of the tile): the result registers
// of U12' X1 have the layout of the next product's B operand, so W2 - U12' X1 never leaves the registers -- one
// memory round trip per tile.
template <typename TW>
__global__ __launch_bounds__(128) void k_sp_panel(TW* __restrict__ W, const int32_t* __restrict__ row_start,
                                                  const int32_t* __restrict__ row_tiles, const int32_t* __restrict__ panel_row,
                                                  const int32_t* __restrict__ panel_pool, const int32_t* __restrict__ valid, int T,
                                                  const double* __restrict__ uinv) {
  const int I = panel_row[blockIdx.x], q = panel_pool[blockIdx.x];
  const int J = row_tiles[q];
  const int wave = threadIdx.x >> 6;
  const int ncols = (J < T) ? max(0, min(32, valid[J] - 32 * wave)) : (wave == 0 ? 1 : 0);
  if (ncols <= 0) return;
  TW* Wt = W + size_t(q) * kTileDoubles + 32 * wave;
  const TW* D = W + size_t(row_start[I]) * kTileDoubles;
  const int kb1 = min(NB, valid[I]), kb2 = max(0, min(NB, valid[I] - NB));
  const double* ui1 = uinv + size_t(2 * I) * NB * NB;
  double a1[8][2], a2[8][2], w1[8][2], w2[8][2];
  double4_t U12[2][2];
  load_aop(ui1, a1);
  load_bop(Wt, kb1, ncols, w1);
  if (kb2 > 0) {
    load_aop(ui1 + NB * NB, a2);
    load_bop(Wt + size_t(NB) * kTile, kb2, ncols, w2);
    load_operand(D + NB, kb1, kb2, U12);
  }
  double4_t X1[2][2];
  solve_x(a1, w1, X1);
  store_rows(X1, Wt, kb1, ncols);
  if (kb2 <= 0) return;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  mfma_atb(U12, X1, acc);
  // register g of acc[a][nt] holds row 16 a + (lane >> 4) + 4 g = 4 (4 a + g) + (lane >> 4): operand step s = 4 a + g
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) w2[4 * a + g][nt] -= acc[a][nt][g];
  double4_t X2[2][2];
  solve_x(a2, w2, X2);
  store_rows(X2, Wt + size_t(NB) * kTile, kb2, ncols);
}

// Level, part 3: every tile that receives contributions from this level's rows: W(Ja, Jb) -= sum_I F(I, Ja)' F(I, Jb)
// over its sources I of the level, ascending I.  Workgroup = target tile, wavefront = 32 x 32 quadrant.
// flags: 1 = diagonal target (the lower-left quadrant is not needed), 2 = right-hand-side target (one column).
template <int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void k_sp_update(double* __restrict__ W, const int32_t* __restrict__ tgt_pool,
                                                   const int32_t* __restrict__ tgt_flags, const int32_t* __restrict__ src_begin,
                                                   const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b) {
  const int t = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int qi = wave >> 1, qj = wave & 1;
  const int flags = tgt_flags[t];
  if ((flags & 1) && qi == 1 && qj == 0) return;
  if ((flags & 2) && qj == 1) return;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  // Round 2 measurements on the Final shape (rocprofv3 per launch, FETCH_SIZE per launch, tools/pmc_sparse.sh): the first 70 of
  // the 190 levels hold 15-22 k targets each (1.06 sources per target) and take 28 of the kernel's 34 ms at 24 ns per
  // target, i.e. 18 us per workgroup with three on a CU, for 1.7 us of products; HBM delivers 1.9 TB/s, the matrix cores run
  // at 29 %.  A/B, same box: both source tiles staged once per workgroup through LDS with 16-byte loads, the target
  // requested first and the next source's tiles requested before the current products (73 KB of LDS, two workgroups
  // per CU): reduced solve 60.0 against 45.0 ms, bitwise the same result -- not kept; an XCD-aware target map: 45.1
  // against 45.0 ms -- not kept; a probe without the operand loads: 33.2 ms -- two thirds of the time is what a
  // one-source workgroup costs around its products (dispatch, index chain, the target's read-modify-write and drain).
  // Last A/B: the one-source targets (94 %) grouped into tasks by (source row, left tile), the left operand kept in
  // registers over 2 / 4 / 8 / 16 targets (190 registers, two workgroups per CU): 48.5 / 49.6 / 53.3 / 60.4 ms against
  // 45.0 -- a target costs the same inside a loop as in a workgroup of its own, and long tasks add a serial tail.
  // Most targets have one or two sources per level, so a workgroup is a short chain of memory latency -> 32 products
  // -> memory latency; the latency is hidden by running four workgroups per CU (128 registers each: one 32-row half
  // of the two operands at a time), not by software pipelining (a two-stage prefetch needed 300 registers and left one
  // workgroup per CU: 45 ns per tile pair against ...).
  const int s0 = src_begin[t], s1 = src_begin[t + 1];
  for (int s = s0; s < s1; ++s) {
    const double* Fa = W + size_t(src_a[s]) * kTileDoubles + 32 * qi;
    const double* Fb = W + size_t(src_b[s]) * kTileDoubles + 32 * qj;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double4_t A[2][2], B[2][2];
      load_operand(Fb + size_t(32 * half) * kTile, 32, 32, B);
      if (Fa == Fb) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) A[mt][nt] = B[mt][nt];
      } else {
        load_operand(Fa + size_t(32 * half) * kTile, 32, 32, A);
      }
      mfma_atb(A, B, acc);
    }
  }
  double* dst = W + size_t(tgt_pool[t]) * kTileDoubles + size_t(32 * qi) * kTile + 32 * qj;
  subtract_block(acc, dst, 32, 32, (flags & 1) && qi == qj);
}

// k_sp_update with the operands in SLICES of the 64-row K dimension (kSteps x 4 rows: 4 kSteps registers per operand), the
// next slice requested before the current one's 4 kSteps products (round 2, the default with 8-row slices).  The kernel is
// bound by what one target costs at the occupancy its registers allow (see k_sp_update): 76 instead of 162 registers put
// five workgroups on a CU instead of three, and every product has the next slice's loads in flight behind it.  Same
// products in the same order: bitwise the results of k_sp_update.  Final shape, same box, reduced solve: 45.0 ms
// (k_sp_update, halves of 32 rows) -> 41.1 (16-row slices, 4 per CU) -> 38.6 (8 rows, 5 per CU; 6 per CU the same) ->
// 39.1 (4 rows); requesting two or three slices ahead: 40.2-41.2 (the register shuffle costs more than it hides); one
// 32-byte record per target (pool slot, flags, source range, first source's tiles: one scalar load instead of a chain of
// two): 38.6, no change.
template <int kSteps>
__device__ __forceinline__ void load_slice(const double* __restrict__ src, double (&X)[2][kSteps]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < kSteps; ++g) X[nt][g] = src[size_t(lk + 4 * g) * kTile + 16 * nt + li];
}
template <int kSteps>
__device__ __forceinline__ void mfma_slice(const double (&A)[2][kSteps], const double (&B)[2][kSteps], double4_t (&acc)[2][2]) {
#pragma unroll
  for (int g = 0; g < kSteps; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[a][g], B[b][g], acc[a][b], 0, 0, 0);
}
template <int kSteps, int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void k_sp_update_slices(double* __restrict__ W, const int32_t* __restrict__ tgt_pool,
                                                                    const int32_t* __restrict__ tgt_flags, const int32_t* __restrict__ src_begin,
                                                                    const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b) {
  constexpr int kSlices = 16 / kSteps;  // per source: 64 rows of K
  const int t = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int qi = wave >> 1, qj = wave & 1;
  const int flags = tgt_flags[t];
  if ((flags & 1) && qi == 1 && qj == 0) return;
  if ((flags & 2) && qj == 1) return;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int s0 = src_begin[t], s1 = src_begin[t + 1];
  const int nq = kSlices * (s1 - s0);
  double A[2][kSteps], B[2][kSteps], An[2][kSteps], Bn[2][kSteps];
#define CX_SP_REQUEST(q, a, b)                                                                     \
  do {                                                                                             \
    const int s_ = s0 + (q) / kSlices;                                                             \
    const size_t row_ = size_t(4 * kSteps * ((q) % kSlices)) * kTile;                              \
    load_slice<kSteps>(W + size_t(src_b[s_]) * kTileDoubles + 32 * qj + row_, b);                 \
    load_slice<kSteps>(W + size_t(src_a[s_]) * kTileDoubles + 32 * qi + row_, a);                 \
  } while (0)
  if (nq > 0) CX_SP_REQUEST(0, A, B);
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) CX_SP_REQUEST(q + 1, An, Bn);
    mfma_slice<kSteps>(A, B, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < kSteps; ++g) { A[i][g] = An[i][g]; B[i][g] = Bn[i][g]; }
  }
#undef CX_SP_REQUEST
  double* dst = W + size_t(tgt_pool[t]) * kTileDoubles + size_t(32 * qi) * kTile + 32 * qj;
  subtract_block(acc, dst, 32, 32, (flags & 1) && qi == qj);
}

// The update on a SINGLE PRECISION pool (use_mixed_precision_solves: "the Gauss-Newton matrix is computed in double precision,
// but its factorization is computed in single precision", solver.h:572-585; FloatSuiteSparseCholesky / CudaSparseCholesky<float>,
// sparse_cholesky.cc:53-100): v_mfma_f32_16x16x4_f32 on float operands, float accumulators, the target read and written as
// floats.  Half the bytes of every operand and target, and the instruction occupies its SIMD for 32 cycles instead of the 64
// nominal / 105 sustained of the fp64 one.  Same workgroup shape as k_sp_update_slices (wavefront = 32 x 32 quadrant, slices
// of the 64-row K dimension requested one ahead).  A lane loads two ADJACENT columns of a source row as one 8-byte load
// (16 lanes = one 128-byte line), so MFMA tile `a` of the quadrant holds the columns 2 i + a, not 16 a + i: the result
// register g of tile (a, b) of lane (li, lk) is the target element [2 (4 lk + g) + a][2 li + b] of the quadrant, and the two
// column tiles of a row go out as one 8-byte store.  The diagonal and panel kernels keep their fp64 register chains on the
// float pool (they are latency chains of 32 x 32 blocks, not flops; their results are rounded once, on the store).
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int kSteps, int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void k_sp_update_f32(float* __restrict__ W, const int32_t* __restrict__ tgt_pool,
                                                                 const int32_t* __restrict__ tgt_flags, const int32_t* __restrict__ src_begin,
                                                                 const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b) {
  constexpr int kSlices = 16 / kSteps;  // per source: 64 rows of K
  const int t = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int qi = wave >> 1, qj = wave & 1;
  const int flags = tgt_flags[t];
  if ((flags & 1) && qi == 1 && qj == 0) return;
  if ((flags & 2) && qj == 1) return;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  float4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};
  const int s0 = src_begin[t], s1 = src_begin[t + 1];
  const int nq = kSlices * (s1 - s0);
  float2_t A[kSteps], B[kSteps], An[kSteps], Bn[kSteps];
#define CX_SP_REQUEST(q, a, b)                                                                                         \
  do {                                                                                                                 \
    const int s_ = s0 + (q) / kSlices;                                                                                 \
    const size_t row_ = size_t(4 * kSteps * ((q) % kSlices) + lk) * kTile + 2 * li;                                   \
    const float* pb_ = W + size_t(src_b[s_]) * kTileDoubles + 32 * qj + row_;                                         \
    const float* pa_ = W + size_t(src_a[s_]) * kTileDoubles + 32 * qi + row_;                                         \
    _Pragma("unroll") for (int g = 0; g < kSteps; ++g) b[g] = *reinterpret_cast<const float2_t*>(pb_ + size_t(4 * g) * kTile); \
    _Pragma("unroll") for (int g = 0; g < kSteps; ++g) a[g] = *reinterpret_cast<const float2_t*>(pa_ + size_t(4 * g) * kTile); \
  } while (0)
  if (nq > 0) CX_SP_REQUEST(0, A, B);
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) CX_SP_REQUEST(q + 1, An, Bn);
#pragma unroll
    for (int g = 0; g < kSteps; ++g) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B[g].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B[g].y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B[g].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B[g].y, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < kSteps; ++g) { A[g] = An[g]; B[g] = Bn[g]; }
  }
#undef CX_SP_REQUEST
  float* dst = W + size_t(tgt_pool[t]) * kTileDoubles + size_t(32 * qi) * kTile + 32 * qj;
  const bool upper_only = (flags & 1) && qi == qj;
  float2_t cur[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[a][g] = *reinterpret_cast<const float2_t*>(dst + size_t(8 * lk + 2 * g + a) * kTile + 2 * li);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = 8 * lk + 2 * g + a, j = 2 * li;
      float2_t v = cur[a][g];
      if (!upper_only || j >= i) v.x -= acc[a][0][g];
      if (!upper_only || j + 1 >= i) v.y -= acc[a][1][g];
      *reinterpret_cast<float2_t*>(dst + size_t(i) * kTile + j) = v;
    }
}

// k_sp_update_f32 with both source tiles staged ONCE per workgroup through LDS.  On the float pool the update is no longer
// bound by the matrix cores (64 v_mfma_f32_16x16x4_f32 per wavefront and source = 2 048 cycles) but by the bytes its four
// wavefronts pull through the caches: each loads a 32-column half of both tiles, 64 KB per product, 9-10 ns per product on the
// wide levels = ~7 TB/s on the CU side.  Here the workgroup's 256 threads fetch the two 16 KB tiles once (32 KB per product,
// 16-byte loads, a wavefront = 1 KB contiguous), in parts of kRows rows: the next part's pieces are requested into registers
// before the current products and written to LDS behind them (whole tiles, 32 KB, four workgroups per CU: reduced solve 19.2 /
// 45.4 ms on the Final shape / the second scene; **16-row parts, 8 KB, eight per CU: 18.3 / 42.9**).  LDS image: row-major 64 x 64 floats with
// the two 32-column halves swapped on odd rows, so the four rows a k-step reads (one per 16-lane group) fall on both halves
// of the 64 banks: a ds_read_b64 of 64 lanes takes its minimum of two passes.  Same products in the same order as
// k_sp_update_f32: bitwise the same result.
template <int kRows, int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void k_sp_update_f32_lds(float* __restrict__ W, const int32_t* __restrict__ tgt_pool,
                                                                     const int32_t* __restrict__ tgt_flags, const int32_t* __restrict__ src_begin,
                                                                     const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b) {
  constexpr int kPart = kRows * kTile;  // floats of a kRows-row part of a tile (64: the whole tile)
  constexpr int kParts = kTile / kRows, kPieces = kRows / 16;  // parts per tile; 16-byte pieces per thread and part
  __shared__ float lds[2 * kPart];  // A image | B image
  typedef float float4v __attribute__((ext_vector_type(4)));
  const int t = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int qi = wave >> 1, qj = wave & 1;
  const int flags = tgt_flags[t];
  const bool computes = !((flags & 1) && qi == 1 && qj == 0) && !((flags & 2) && qj == 1);  // (every wavefront loads)
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  float4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};
  const int s0 = src_begin[t], s1 = src_begin[t + 1];
  const int nq = kParts * (s1 - s0);
  // piece j of this thread: float4 number tid + 256 j of a part = row (tid + 256 j) / 16, columns 4 ((tid + 256 j) % 16) ...
  float4v ra[kPieces], rb[kPieces];
#define CX_SP_REQUEST(q_)                                                                             \
  do {                                                                                                \
    const int s_ = s0 + (q_) / kParts;                                                                \
    const float* pa_ = W + size_t(src_a[s_]) * kTileDoubles + ((q_) % kParts) * kPart + 4 * tid;      \
    const float* pb_ = W + size_t(src_b[s_]) * kTileDoubles + ((q_) % kParts) * kPart + 4 * tid;      \
    _Pragma("unroll") for (int j = 0; j < kPieces; ++j) ra[j] = *reinterpret_cast<const float4v*>(pa_ + 1024 * j); \
    _Pragma("unroll") for (int j = 0; j < kPieces; ++j) rb[j] = *reinterpret_cast<const float4v*>(pb_ + 1024 * j); \
  } while (0)
  if (nq > 0) CX_SP_REQUEST(0);
  for (int q = 0; q < nq; ++q) {
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
      const int idx = tid + 256 * j, k = idx >> 4, c = (idx & 15) << 2;
      const int at = k * kTile + (c ^ ((k & 1) << 5));
      *reinterpret_cast<float4v*>(lds + at) = ra[j];
      *reinterpret_cast<float4v*>(lds + kPart + at) = rb[j];
    }
    __syncthreads();
    if (q + 1 < nq) CX_SP_REQUEST(q + 1);
    if (computes) {
      const int sw = (lk & 1) << 5;  // (k & 1) == (lk & 1)
#pragma unroll
      for (int g = 0; g < kRows / 4; ++g) {
        const int k = 4 * g + lk;
        const float2_t av = *reinterpret_cast<const float2_t*>(lds + k * kTile + ((32 * qi + 2 * li) ^ sw));
        const float2_t bv = *reinterpret_cast<const float2_t*>(lds + kPart + k * kTile + ((32 * qj + 2 * li) ^ sw));
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();  // the images are free for the next part
  }
#undef CX_SP_REQUEST
  if (!computes) return;
  float* dst = W + size_t(tgt_pool[t]) * kTileDoubles + size_t(32 * qi) * kTile + 32 * qj;
  const bool upper_only = (flags & 1) && qi == qj;
  float2_t cur[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[a][g] = *reinterpret_cast<const float2_t*>(dst + size_t(8 * lk + 2 * g + a) * kTile + 2 * li);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = 8 * lk + 2 * g + a, j = 2 * li;
      float2_t v = cur[a][g];
      if (!upper_only || j >= i) v.x -= acc[a][0][g];
      if (!upper_only || j + 1 >= i) v.y -= acc[a][1][g];
      *reinterpret_cast<float2_t*>(dst + size_t(i) * kTile + j) = v;
    }
}

// The double precision update with both source tiles staged once per workgroup through LDS, as k_sp_update_f32_lds.  The
// counters (tools/pmc_mfma.sh) said that k_sp_update_slices is NOT bound by the matrix cores: 61-63 busy cycles per
// v_mfma_f64_16x16x4_f64 (the nominal 64) at 2.2-2.5 GHz, but the pipes busy only 55-60 % of the SIMD cycles on the widest
// levels (11.3 ns per product; 100 % would be 6.9) -- the wavefronts wait for operands, 128 KB per product through the
// caches.  Here: 64 KB per product (two 32 KB tiles, 16-byte loads), in parts of kRows rows: the next part's pieces in
// registers behind the current products, one LDS buffer; LDS image row-major with the 16-column blocks swapped in pairs on odd
// rows, so that the four rows of a k-step spread over all 64 banks (ds_read_b64: two passes, the minimum).  Same products in
// the same order: bitwise the results of k_sp_update_slices.  Reduced solve, Final / second scene (k_sp_update_slices: 31.4 /
// 81.4 ms): whole tiles, 64 KB, two workgroups per CU 33.4 / 87.9; 32-row parts, 32 KB, four per CU 28.9 / 73.6; **16-row
// parts, 16 KB, six per CU 27.5 / 68.6**; seven per CU (20 B of scratch) or 8-row parts 27.9 / 70.0; two alternating images
// with one barrier per part instead of two (16 rows, five per CU / 8 rows, six per CU) 29.5 / 75.1 and 28.7 / 72.2 -- what
// pays is workgroups in flight, not fewer barriers.
template <int kRows, int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void k_sp_update_f64_lds(double* __restrict__ W, const int32_t* __restrict__ tgt_pool,
                                                                     const int32_t* __restrict__ tgt_flags, const int32_t* __restrict__ src_begin,
                                                                     const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b) {
  constexpr int kHalf = kRows * kTile;  // doubles of a kRows-row part of a tile (32: half)
  constexpr int kParts = kTile / kRows, kPieces = kRows / 8;  // parts per tile; 16-byte pieces per thread and part
  __shared__ double lds[2 * kHalf];     // A part | B part: 16 KB at 16 rows
  typedef double double2v __attribute__((ext_vector_type(2)));
  const int t = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int qi = wave >> 1, qj = wave & 1;
  const int flags = tgt_flags[t];
  const bool computes = !((flags & 1) && qi == 1 && qj == 0) && !((flags & 2) && qj == 1);  // (every wavefront loads)
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int s0 = src_begin[t], s1 = src_begin[t + 1];
  const int nq = kParts * (s1 - s0);  // parts of the sources
  // piece j of this thread: double2 number tid + 256 j of a half tile = row (tid + 256 j) / 32, columns 2 ((tid + 256 j) % 32), + 1
  double2v ra[kPieces], rb[kPieces];
#define CX_SP_REQUEST(q_)                                                                              \
  do {                                                                                                 \
    const int s_ = s0 + (q_) / kParts;                                                                 \
    const double* pa_ = W + size_t(src_a[s_]) * kTileDoubles + ((q_) % kParts) * kHalf + 2 * tid;      \
    const double* pb_ = W + size_t(src_b[s_]) * kTileDoubles + ((q_) % kParts) * kHalf + 2 * tid;      \
    _Pragma("unroll") for (int j = 0; j < kPieces; ++j) ra[j] = *reinterpret_cast<const double2v*>(pa_ + 512 * j); \
    _Pragma("unroll") for (int j = 0; j < kPieces; ++j) rb[j] = *reinterpret_cast<const double2v*>(pb_ + 512 * j); \
  } while (0)
  if (nq > 0) CX_SP_REQUEST(0);
  for (int q = 0; q < nq; ++q) {
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
      const int idx = tid + 256 * j, k = idx >> 5, c = (idx & 31) << 1;
      const int at = k * kTile + (c ^ ((k & 1) << 4));
      *reinterpret_cast<double2v*>(lds + at) = ra[j];
      *reinterpret_cast<double2v*>(lds + kHalf + at) = rb[j];
    }
    __syncthreads();
    if (q + 1 < nq) CX_SP_REQUEST(q + 1);
    if (computes) {
      const int sw = (lk & 1) << 4;  // (k & 1) == (lk & 1)
#pragma unroll
      for (int g = 0; g < kRows / 4; ++g) {
        const double* Ak = lds + (4 * g + lk) * kTile;
        const double* Bk = Ak + kHalf;
        const double a0 = Ak[(32 * qi + li) ^ sw], a1 = Ak[(32 * qi + 16 + li) ^ sw];
        const double b0 = Bk[(32 * qj + li) ^ sw], b1 = Bk[(32 * qj + 16 + li) ^ sw];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();  // the images are free for the next part
  }
#undef CX_SP_REQUEST
  if (!computes) return;
  double* dst = W + size_t(tgt_pool[t]) * kTileDoubles + size_t(32 * qi) * kTile + 32 * qj;
  subtract_block(acc, dst, 32, 32, (flags & 1) && qi == qj);
}

// (Round 3 A/B, not kept: ONE WAVEFRONT per target tile -- the whole 64 x 64 in 128 accumulator registers, each source tile
// loaded once per product (64 instead of 128 KB), four targets of one tile row per workgroup, two wavefronts per SIMD.  Per
// level, tools/sparse_levels.sh: 24 ns per product on the wide levels against 18.5 (reduced solve 46.3 against 38.2 ms at
// window 1, 35.3 against 32.2 at window 4) and 25 instead of 11 us for a lone target: with 32 MFMAs behind one slice of
// loads and two wavefronts on a SIMD the loads are not hidden.  And the wide levels are no longer where the time is: with
// chains (CX_SPARSE_WINDOW) they run at 12-15 ns per product, and what the chip SUSTAINS in fp64 MFMAs is 10.9 ns
// (48 TFLOP/s, tools/mfma_peak.hip).)

// Backward substitution of a level (top down), part 1: workgroup = one tile F(I, J) right of the diagonal of a row of
// the level: partial[tile][r] = sum_c F(I, J)[r][c] x_J[c].
// Lanes run along the COLUMNS of the tile (a wavefront's load is 512 contiguous bytes, as in k_sp_fwd_partial); wavefront w owns
// rows 16 w ... 16 w + 15, and the 16 x 64 products are summed over the lanes by a halving exchange (8 + 4 + 2 + 1 values
// swapped with lane ^ 32, 16, 8, 4, then two plain steps): 17 double shuffles instead of 96.  Round 2 had every thread read
// 16 consecutive doubles of a row -- each load instruction touched 64 lines, and the kernel took twice as long as the forward
// one on the same tiles (12.6 against 6.2 us per launch in a CLUSTER_TRIDIAGONAL solve).
template <typename TW>
__global__ __launch_bounds__(256) void k_sp_bwd_partial(const TW* __restrict__ W, const int32_t* __restrict__ row_tiles,
                                                        const int32_t* __restrict__ panel_pool, const int32_t* __restrict__ valid, int T,
                                                        const double* __restrict__ x, double* __restrict__ partial) {
  const int q = panel_pool[blockIdx.x];
  const int J = row_tiles[q];
  if (J >= T) return;  // the right-hand-side tile
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  const TW* __restrict__ F = W + size_t(q) * kTileDoubles + size_t(16 * w) * kTile + c;
  TW f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = F[size_t(i) * kTile];
  const double xc = (c < valid[J]) ? x[size_t(kTile) * J + c] : 0.0;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = double(f[i]) * xc;
  // halving exchange: after the step with mask m a lane keeps the rows whose bit (in the order 8, 4, 2, 1 of the row index)
  // equals its own bit m
#define CX_SP_HALVE(n, mask)                                              \
  _Pragma("unroll") for (int i = 0; i < n; ++i) {                          \
    const bool hi = (c & mask) != 0;                                       \
    const double keep = hi ? v[i + n] : v[i], send = hi ? v[i] : v[i + n]; \
    v[i] = keep + __shfl_xor(send, mask, 64);                              \
  }
  CX_SP_HALVE(8, 32)
  CX_SP_HALVE(4, 16)
  CX_SP_HALVE(2, 8)
  CX_SP_HALVE(1, 4)
#undef CX_SP_HALVE
  double s = v[0];
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  // this lane's row: bit 3 of the row index from lane bit 5, bit 2 from bit 4, bit 1 from bit 3, bit 0 from bit 2
  const int row = 16 * w + (((c >> 5) & 1) << 3 | ((c >> 4) & 1) << 2 | ((c >> 3) & 1) << 1 | ((c >> 2) & 1));
  if ((c & 3) == 0) partial[size_t(q) * kTile + row] = s;
}

// ... part 2: workgroup = tile row I: y_I (column 0 of the row's last tile) minus the partial sums of its tiles in
// ascending order, then the 64 x 64 triangular solve with the kept inverses (x2 = U22^-1 y2, x1 = U11^-1 (y1 - U12 x2)).
template <typename TW>
__global__ __launch_bounds__(256) void k_sp_bwd_level(const TW* __restrict__ W, const int32_t* __restrict__ row_start,
                                                      const int32_t* __restrict__ level_rows, const int32_t* __restrict__ valid,
                                                      const double* __restrict__ uinv, const double* __restrict__ partial,
                                                      double* __restrict__ x, int y_in_x) {
  // y_in_x: the right-hand side of the sweep is x itself (a solve with a stored factor); otherwise column 0 of the
  // row's last tile (the right-hand side that was forward-substituted along with the factorisation)
  __shared__ double y1[NB], y2[NB], x1[NB], x2[NB], tmp[NB];
  const int I = level_rows[blockIdx.x];
  const int t = threadIdx.x;
  const int k0 = kTile * I;
  const int kb1 = min(NB, valid[I]), kb2 = max(0, min(NB, valid[I] - NB));
  const int q0 = row_start[I], q1 = row_start[I + 1] - 1;  // [q0] diagonal tile, [q1] right-hand-side tile
  const TW* __restrict__ D = W + size_t(q0) * kTileDoubles;
  const double* __restrict__ ui1 = uinv + size_t(2 * I) * NB * NB;
  // the entries of the three 32 x 32 products are requested before the row's partial sums are gathered: five dependent
  // steps of global-memory latency per level become two
  double a22[4], a12[4], a11[4];
  sp_gemv32_load(ui1 + NB * NB, NB, NB, NB, a22);
  sp_gemv32_load(D + NB, kTile, kb1, kb2, a12);
  sp_gemv32_load(ui1, NB, NB, NB, a11);
  if (t < kTile) {
    double s = 0.0;
#pragma unroll 8
    for (int q = q0 + 1; q < q1; ++q) s += partial[size_t(q) * kTile + t];
    const double yv = (y_in_x ? x[k0 + t] : W[size_t(q1) * kTileDoubles + size_t(t) * kTile]) - s;
    if (t < NB) y1[t] = t < kb1 ? yv : 0.0;
    else y2[t - NB] = (t - NB) < kb2 ? yv : 0.0;
  }
  __syncthreads();
  if (kb2 > 0) {
    sp_gemv32_apply(a22, y2, x2);
    __syncthreads();
    sp_gemv32_apply(a12, x2, tmp);  // U12 x2
    __syncthreads();
    if (t < NB) y1[t] -= tmp[t];
  } else if (t < NB) {
    x2[t] = 0.0;
  }
  __syncthreads();
  sp_gemv32_apply(a11, y1, x1);
  __syncthreads();
  if (t < 64) {
    const double v = t < NB ? x1[t] : x2[t - NB];
    x[k0 + t] = (t < kb1 + kb2) ? v : 0.0;  // padding rows of the tile row carry zeros
  }
}

// out[m] = sum_c M[c][m] v[c] (the transposed 32 x 32 block), 8 threads per output: the load half (the apply half is
// sp_gemv32_apply)
template <typename TM>
__device__ __forceinline__ void sp_gemv32_t_load(const TM* __restrict__ M, int ldm, int rows_valid, int cols_valid, double (&a)[4]) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * part + q;
    a[q] = (c < rows_valid && m < cols_valid) ? M[size_t(c) * ldm + m] : 0.0;
  }
}

// Forward substitution U' y = r with a stored factor, by levels bottom up.  Part 1, workgroup = tile row I of the level:
// r_I minus the partial products of the tiles of COLUMN I (rows K < I, all of lower levels, ascending K), then
// y1 = U11^-T v1, y2 = U22^-T (v2 - U12' y1) with the kept inverses.  y holds r on entry and y on exit.
template <typename TW>
__global__ __launch_bounds__(256) void k_sp_fwd_level(const TW* __restrict__ W, const int32_t* __restrict__ row_start,
                                                      const int32_t* __restrict__ level_rows, const int32_t* __restrict__ valid,
                                                      const double* __restrict__ uinv, const int32_t* __restrict__ col_start,
                                                      const double* __restrict__ partial,
                                                      double* __restrict__ y) {
  __shared__ double v1[NB], v2[NB], y1[NB], y2[NB], tmp[NB];
  const int I = level_rows[blockIdx.x];
  const int t = threadIdx.x;
  const int k0 = kTile * I;
  const int kb1 = min(NB, valid[I]), kb2 = max(0, min(NB, valid[I] - NB));
  const TW* __restrict__ D = W + size_t(row_start[I]) * kTileDoubles;
  const double* __restrict__ ui1 = uinv + size_t(2 * I) * NB * NB;
  double a11[4], a12[4], a22[4];  // (requested before the partial sums are gathered, see k_sp_bwd_level)
  sp_gemv32_t_load(ui1, NB, NB, NB, a11);
  sp_gemv32_t_load(D + NB, kTile, kb1, kb2, a12);
  sp_gemv32_t_load(ui1 + NB * NB, NB, NB, NB, a22);
  if (t < kTile) {
    double s = 0.0;
#pragma unroll 8
    for (int p = col_start[I]; p < col_start[I + 1]; ++p) s += partial[size_t(p) * kTile + t];  // (slots in column order: k_sp_fwd_partial)
    const double v = y[k0 + t] - s;
    if (t < NB) v1[t] = t < kb1 ? v : 0.0;
    else v2[t - NB] = (t - NB) < kb2 ? v : 0.0;
  }
  __syncthreads();
  sp_gemv32_apply(a11, v1, y1);  // y1 = U11^-T v1 (identity-padded inverse)
  __syncthreads();
  if (kb2 > 0) {
    sp_gemv32_apply(a12, y1, tmp);  // U12' y1
    __syncthreads();
    if (t < NB) v2[t] -= tmp[t];
    __syncthreads();
    sp_gemv32_apply(a22, v2, y2);
    __syncthreads();
  } else if (t < NB) {
    y2[t] = 0.0;
  }
  __syncthreads();
  if (t < 64) {
    const double v = t < NB ? y1[t] : y2[t - NB];
    y[k0 + t] = (t < kb1 + kb2) ? v : 0.0;
  }
}

// ... part 2, workgroup = one tile F(K, J) right of the diagonal of a row K of the level: partial[slot][c] =
// sum_r F(K, J)[r][c] y_K[r], gathered later by tile row J -- slot = the tile's place in the list of tile column J (ascending
// K), so that row J reads its incoming sums as one contiguous run instead of through an index.
template <typename TW>
__global__ __launch_bounds__(256) void k_sp_fwd_partial(const TW* __restrict__ W, const int32_t* __restrict__ row_tiles,
                                                        const int32_t* __restrict__ panel_row, const int32_t* __restrict__ panel_pool,
                                                        const int32_t* __restrict__ col_slot, int T, const double* __restrict__ y,
                                                        double* __restrict__ partial) {
  __shared__ double red[4][kTile];
  const int q = panel_pool[blockIdx.x];
  if (row_tiles[q] >= T) return;  // the right-hand-side tile
  const int slot = col_slot[q];
  const int K = panel_row[blockIdx.x];
  const int t = threadIdx.x, c = t & 63, part = t >> 6;
  const TW* __restrict__ F = W + size_t(q) * kTileDoubles + size_t(16 * part) * kTile + c;
  const double* __restrict__ yk = y + size_t(kTile) * K + 16 * part;
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += F[size_t(r) * kTile] * yk[r];  // padding rows of y are zero
  red[part][c] = s;
  __syncthreads();
  if (part == 0) partial[size_t(slot) * kTile + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// camera order -> padded elimination order (padding rows zero: memset first)
__global__ void k_sp_permute(const double* __restrict__ r, const int32_t* __restrict__ cam_pos, double* __restrict__ yp, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, a = i - 9 * c;
  yp[cam_pos[c] + a] = r[i];
}

__global__ void k_sp_unpermute(const double* __restrict__ xp, const int32_t* __restrict__ cam_pos, double* __restrict__ x, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, a = i - 9 * c;
  x[i] = xp[cam_pos[c] + a];
}

// reverse Cuthill-McKee on the camera graph; returns position[camera]
std::vector<int32_t> ReverseCuthillMcKee(int C, const std::vector<std::vector<int32_t>>& adj) {
  std::vector<int32_t> order;
  order.reserve(size_t(C));
  std::vector<char> seen(size_t(C), 0);
  std::vector<int32_t> by_degree(static_cast<size_t>(C));
  for (int i = 0; i < C; ++i) by_degree[size_t(i)] = i;
  std::stable_sort(by_degree.begin(), by_degree.end(), [&](int32_t u, int32_t v) { return adj[size_t(u)].size() < adj[size_t(v)].size(); });
  for (int32_t start : by_degree) {
    if (seen[size_t(start)]) continue;
    // pseudo-peripheral start: two BFS sweeps from the component's minimum-degree vertex
    int32_t root = start;
    for (int sweep = 0; sweep < 2; ++sweep) {
      std::vector<int32_t> frontier{root}, last;
      std::vector<char> mark(size_t(C), 0);
      mark[size_t(root)] = 1;
      while (!frontier.empty()) {
        last = frontier;
        std::vector<int32_t> next;
        for (int32_t u : frontier)
          for (int32_t v : adj[size_t(u)])
            if (!mark[size_t(v)] && !seen[size_t(v)]) { mark[size_t(v)] = 1; next.push_back(v); }
        frontier.swap(next);
      }
      root = *std::min_element(last.begin(), last.end(), [&](int32_t u, int32_t v) { return adj[size_t(u)].size() < adj[size_t(v)].size(); });
    }
    std::queue<int32_t> q;
    q.push(root);
    seen[size_t(root)] = 1;
    while (!q.empty()) {
      const int32_t u = q.front();
      q.pop();
      order.push_back(u);
      std::vector<int32_t> nb;
      for (int32_t v : adj[size_t(u)]) if (!seen[size_t(v)]) { seen[size_t(v)] = 1; nb.push_back(v); }
      std::stable_sort(nb.begin(), nb.end(), [&](int32_t s, int32_t t) { return adj[size_t(s)].size() < adj[size_t(t)].size(); });
      for (int32_t v : nb) q.push(v);
    }
  }
  std::reverse(order.begin(), order.end());
  std::vector<int32_t> pos(static_cast<size_t>(C));
  for (int k = 0; k < C; ++k) pos[size_t(order[size_t(k)])] = k;
  return pos;
}

// Minimum degree on the quotient graph of groups of 64 consecutive cameras of a locality-preserving order
// (64 cameras = 576 rows = 9 whole tiles, so the groups stay tile-aligned): hub groups that see everything move to
// the end, where their fill is unavoidable, instead of sitting wherever the breadth-first order met them.
std::vector<int32_t> GroupMinimumDegree(int C, const std::vector<std::vector<int32_t>>& adj, const std::vector<int32_t>& pos) {
  const int G = (C + 63) / 64;
  std::vector<std::vector<char>> g(static_cast<size_t>(G), std::vector<char>(static_cast<size_t>(G), 0));
  for (int c = 0; c < C; ++c)
    for (int32_t d : adj[size_t(c)]) g[size_t(pos[size_t(c)] / 64)][size_t(pos[size_t(d)] / 64)] = 1;
  std::vector<char> done(static_cast<size_t>(G), 0);
  std::vector<int32_t> rank(static_cast<size_t>(G), 0);
  for (int step = 0; step < G; ++step) {
    int best = -1, best_deg = 1 << 30;
    for (int u = 0; u < G; ++u) {
      if (done[size_t(u)]) continue;
      int deg = 0;
      for (int v = 0; v < G; ++v) deg += (!done[size_t(v)] && v != u && g[size_t(u)][size_t(v)]) ? 1 : 0;
      if (deg < best_deg) { best_deg = deg; best = u; }
    }
    done[size_t(best)] = 1;
    rank[size_t(best)] = step;
    for (int v = 0; v < G; ++v) {
      if (done[size_t(v)] || !g[size_t(best)][size_t(v)]) continue;
      for (int w = 0; w < G; ++w)
        if (!done[size_t(w)] && g[size_t(best)][size_t(w)]) g[size_t(v)][size_t(w)] = 1;
    }
  }
  // new position = start of the group's slot + old offset inside the group
  std::vector<int32_t> size(static_cast<size_t>(G), 0), start(static_cast<size_t>(G), 0), by_rank(static_cast<size_t>(G), 0);
  for (int c = 0; c < C; ++c) size[size_t(pos[size_t(c)] / 64)]++;
  for (int u = 0; u < G; ++u) by_rank[size_t(rank[size_t(u)])] = u;
  int32_t acc = 0;
  for (int r = 0; r < G; ++r) { start[size_t(by_rank[size_t(r)])] = acc; acc += size[size_t(by_rank[size_t(r)])]; }
  std::vector<int32_t> out(static_cast<size_t>(C));
  for (int c = 0; c < C; ++c) out[size_t(c)] = start[size_t(pos[size_t(c)] / 64)] + pos[size_t(c)] % 64;
  return out;
}

// Nested dissection of the camera graph (George's automatic nested dissection): the level structure of a breadth-
// first search from a pseudo-peripheral camera cuts a connected piece into "before", one level, and "after"; the level
// is a vertex separator.  Both sides are ordered first (recursively), the separator last, so the two sides are
// independent subtrees of the elimination tree -- the parallelism the level schedule of the factorisation exploits --
// and whatever a separator eliminates couples only with its own subtree and the separators above it.  A cut is taken
// only while the piece is at least `ratio` times larger than the separator: below that (a band as wide as the piece is
// long) dissecting further only adds fill, and the piece is laid out as a band instead (reverse breadth-first order).
// Work stack instead of recursion; pieces are told apart by an owner id per camera.
struct Dissection {
  const std::vector<std::vector<int32_t>>& adj;
  int ratio, leaf;
  std::vector<int32_t> owner;  // piece id of a camera, -1 once ordered
  std::vector<int32_t> dist;   // breadth-first level inside the current search, -1 otherwise
  std::vector<int32_t> bfs;    // visit order of the last search
  std::vector<int32_t> order;  // result: cameras in elimination order
  std::vector<int32_t> piece_end;  // ... cut into pieces (bands and separators): end of each piece in `order`
  int next_piece = 1;

  Dissection(const std::vector<std::vector<int32_t>>& a, int leaf_size, int min_ratio)
      : adj(a), ratio(min_ratio), leaf(leaf_size), owner(a.size(), 0), dist(a.size(), -1) { order.reserve(a.size()); }

  int Search(int32_t start, int32_t piece) {  // fills bfs / dist, returns the number of levels
    bfs.clear();
    bfs.push_back(start);
    dist[size_t(start)] = 0;
    for (size_t h = 0; h < bfs.size(); ++h) {
      const int32_t u = bfs[h];
      for (int32_t v : adj[size_t(u)])
        if (owner[size_t(v)] == piece && dist[size_t(v)] < 0) { dist[size_t(v)] = dist[size_t(u)] + 1; bfs.push_back(v); }
    }
    return dist[size_t(bfs.back())] + 1;
  }
  void Forget() { for (int32_t u : bfs) dist[size_t(u)] = -1; }
  int32_t FarLowDegree(int levels) const {  // a camera of the last level with the fewest neighbours
    int32_t best = bfs.back();
    for (size_t i = bfs.size(); i-- > 0 && dist[size_t(bfs[i])] == levels - 1;)
      if (adj[size_t(bfs[i])].size() < adj[size_t(best)].size()) best = bfs[i];
    return best;
  }
  void EmitBand() {  // the last search's cameras as a band: reverse breadth-first order
    for (size_t i = bfs.size(); i-- > 0;) { order.push_back(bfs[i]); owner[size_t(bfs[i])] = -1; }
    piece_end.push_back(int32_t(order.size()));
  }

  // entries of the work stack: a piece to dissect (seed >= 0: any camera of it) or a separator to emit (list)
  struct Job { int32_t piece, seed; std::vector<int32_t> emit; };

  void Run() {
    std::vector<Job> stack;
    // whole graph: every connected component is a piece of its own
    const int32_t n = int32_t(adj.size());
    for (int32_t c = n - 1; c >= 0; --c) {
      if (owner[size_t(c)] != 0) continue;
      const int32_t piece = next_piece++;
      // label the component
      owner[size_t(c)] = piece;
      std::vector<int32_t> comp{c};
      for (size_t h = 0; h < comp.size(); ++h)
        for (int32_t v : adj[size_t(comp[h])]) if (owner[size_t(v)] == 0) { owner[size_t(v)] = piece; comp.push_back(v); }
      stack.push_back(Job{piece, c, {}});
    }
    while (!stack.empty()) {
      Job job = std::move(stack.back());
      stack.pop_back();
      if (job.seed < 0) {
        for (int32_t u : job.emit) { order.push_back(u); owner[size_t(u)] = -1; }
        piece_end.push_back(int32_t(order.size()));
        continue;
      }
      // pseudo-peripheral start: search again from a far, low-degree camera while the structure gets deeper
      int levels = Search(job.seed, job.piece);
      for (int pass = 0; pass < 3; ++pass) {
        const int32_t far = FarLowDegree(levels);
        Forget();
        const int deeper = Search(far, job.piece);
        const bool improved = deeper > levels;
        levels = deeper;
        if (!improved) break;
      }
      const int32_t size = int32_t(bfs.size());
      if (size <= leaf || levels < 3) { EmitBand(); Forget(); continue; }
      std::vector<int32_t> count(size_t(levels), 0);
      for (int32_t u : bfs) count[size_t(dist[size_t(u)])]++;
      // the thinnest level that leaves at least a quarter of the piece on either side
      int cut = -1;
      int32_t below = 0;
      for (int l = 0; l < levels; ++l) {
        const int32_t above = size - below - count[size_t(l)];
        if (l > 0 && l < levels - 1 && 4 * below >= size && 4 * above >= size && (cut < 0 || count[size_t(l)] < count[size_t(cut)])) cut = l;
        below += count[size_t(l)];
      }
      if (cut < 0 || int64_t(count[size_t(cut)]) * ratio > size) { EmitBand(); Forget(); continue; }
      const int32_t lo = next_piece++, hi = next_piece++;
      std::vector<int32_t> sep;
      int32_t seed_lo = -1, seed_hi = -1;
      for (int32_t u : bfs) {
        const int d = dist[size_t(u)];
        if (d < cut) { owner[size_t(u)] = lo; seed_lo = u; }
        else if (d > cut) { owner[size_t(u)] = hi; seed_hi = u; }
        else { owner[size_t(u)] = -2; sep.push_back(u); }
      }
      Forget();
      // stack order: the separator is emitted after both sides (pushed first, popped last).  A side may fall apart into
      // several components once the separator is gone: each becomes a piece of its own.
      stack.push_back(Job{-1, -1, std::move(sep)});
      const int32_t seeds[2] = {seed_lo, seed_hi};
      const int32_t sides[2] = {lo, hi};
      for (int k = 0; k < 2; ++k) PushComponents(sides[k], seeds[k], &stack);
    }
  }

  // split the cameras labelled `piece` into connected components (each relabelled) and push them
  void PushComponents(int32_t piece, int32_t any, std::vector<Job>* stack) {
    if (any < 0) return;
    // collect the members by a search over the piece, restarting until all are seen
    std::vector<int32_t> members;
    for (int32_t u = 0; u < int32_t(owner.size()); ++u) if (owner[size_t(u)] == piece) members.push_back(u);
    for (int32_t m : members) {
      if (owner[size_t(m)] != piece) continue;
      const int32_t comp = next_piece++;
      owner[size_t(m)] = comp;
      std::vector<int32_t> q{m};
      for (size_t h = 0; h < q.size(); ++h)
        for (int32_t v : adj[size_t(q[h])]) if (owner[size_t(v)] == piece) { owner[size_t(v)] = comp; q.push_back(v); }
      stack->push_back(Job{comp, m, {}});
    }
  }
};

// Padded layout of an elimination order cut into pieces: every piece starts at a multiple of 64 rows, so that no
// 64-row tile row holds cameras of two pieces.  (A tile row shared by the end of one subtree and the start of its
// sibling would tie the two subtrees into one chain of the TILE elimination tree, which is the tree the level schedule
// works on.)  The rows between the end of a piece and the next tile boundary are padding: unit diagonal, zero
// right-hand side; valid[I] = rows of tile row I that belong to cameras (always its first rows).
struct PaddedLayout {
  std::vector<int32_t> cam_row;  // first row of every camera
  std::vector<int32_t> valid;    // [T]
  int T = 0;
};

PaddedLayout LayOut(int C, const std::vector<int32_t>& order, const std::vector<int32_t>& piece_end) {
  PaddedLayout out;
  out.cam_row.assign(size_t(C), 0);
  int32_t row = 0, k = 0;
  for (int32_t end : piece_end) {
    row = (row + kTile - 1) / kTile * kTile;
    for (; k < end; ++k) { out.cam_row[size_t(order[size_t(k)])] = row; row += 9; }
  }
  out.T = (row + kTile - 1) / kTile;
  out.valid.assign(size_t(out.T), 0);
  for (int c = 0; c < C; ++c)
    for (int a = 0; a < 9; ++a) out.valid[size_t((out.cam_row[size_t(c)] + a) >> 6)]++;
  return out;
}

PaddedLayout NestedDissection(int C, const std::vector<std::vector<int32_t>>& adj, int leaf) {
  static const int ratio = [] { const char* v = std::getenv("CX_SPARSE_ND_RATIO"); return v ? std::max(2, atoi(v)) : 6; }();
  Dissection d(adj, leaf, ratio);
  // hub cameras (co-visible with a large part of all cameras) would glue every level structure into two or three
  // levels: they are taken out first and eliminated last, where their fill is unavoidable anyway
  std::vector<int32_t> hubs;
  if (C >= 64)
    for (int c = 0; c < C; ++c)
      if (adj[size_t(c)].size() * 5 > size_t(C) * 2) { d.owner[size_t(c)] = -3; hubs.push_back(c); }
  std::stable_sort(hubs.begin(), hubs.end(), [&](int32_t u, int32_t v) { return adj[size_t(u)].size() < adj[size_t(v)].size(); });
  d.Run();
  if (!hubs.empty()) {
    d.order.insert(d.order.end(), hubs.begin(), hubs.end());
    d.piece_end.push_back(int32_t(d.order.size()));
  }
  return LayOut(C, d.order, d.piece_end);
}

// tile-level structure of the permuted S (upper) and its symbolic fill (eliminating tile row k connects every pair
// of its later column tiles); rows as sorted lists, each closed by the right-hand-side tile T
void TileStructure(const int32_t* cell_c1, const int32_t* cell_c2, int64_t num_cells, const std::vector<int32_t>& cam_row, int T,
                   std::vector<int32_t>* row_start, std::vector<int32_t>* row_tiles) {
  std::vector<std::vector<char>> nz(static_cast<size_t>(T), std::vector<char>(static_cast<size_t>(T), 0));
  auto mark = [&](int r1, int r2) {  // first rows of the two cameras, r1 <= r2
    for (int rt = r1 >> 6; rt <= (r1 + 8) >> 6; ++rt)
      for (int ct = r2 >> 6; ct <= (r2 + 8) >> 6; ++ct) nz[size_t(std::min(rt, ct))][size_t(std::max(rt, ct))] = 1;
  };
  for (int64_t k = 0; k < num_cells; ++k) {
    const int r1 = cam_row[size_t(cell_c1[k])], r2 = cam_row[size_t(cell_c2[k])];
    mark(std::min(r1, r2), std::max(r1, r2));
  }
  for (int I = 0; I < T; ++I) nz[size_t(I)][size_t(I)] = 1;
  std::vector<int32_t> later;
  for (int k = 0; k < T; ++k) {
    later.clear();
    for (int J = k + 1; J < T; ++J) if (nz[size_t(k)][size_t(J)]) later.push_back(J);
    for (size_t x = 0; x < later.size(); ++x)
      for (size_t y = x; y < later.size(); ++y) nz[size_t(later[x])][size_t(later[y])] = 1;
  }
  row_start->assign(size_t(T) + 1, 0);
  row_tiles->clear();
  for (int I = 0; I < T; ++I) {
    (*row_start)[size_t(I)] = int32_t(row_tiles->size());
    for (int J = I; J < T; ++J) if (nz[size_t(I)][size_t(J)]) row_tiles->push_back(J);
    row_tiles->push_back(T);  // the right-hand side rides along
  }
  (*row_start)[size_t(T)] = int32_t(row_tiles->size());
}

}  // namespace

namespace {
// Everything of the plan that needs no device: layout, tile structure after fill, level schedule, update lists.
struct HostPlan {
  PaddedLayout layout;
  std::vector<int32_t> row_start, row_tiles, height, level_rows, lrb, lpb, ltb, panel_row, panel_pool;
  std::vector<int32_t> tgt_pool, tgt_flags, src_begin, src_a, src_b, col_start, col_pool;
  int T = 0, L = 0;
  int64_t num_tiles = 0;
  bool fits = true;
  // Distributed factorisation (nranks > 1): owner[I] = the rank that factors tile row I alone, -1 = a row every rank
  // factors (the top of the elimination tree).  The schedule then holds this rank's rows in levels [0, L_split) and the
  // replicated rows in levels [L_split, L): everything a rank's own subtrees need lies below the split, everything the
  // top needs from the subtrees arrives with ONE exchange at the split.
  std::vector<int32_t> owner;
  int L_split = -1;
  std::vector<int64_t> work_per_rank;  // tile-pair updates of every rank's own rows
  int64_t work_shared = 0;
  // the schedule of the updates: window <= 0 = the default (CX_SPARSE_WINDOW, 4); keep_schedule: per product (in the order of
  // src_a / src_b) its source row, and per tile row the level of the schedule it is factored in (-1: another rank's)
  int window = 0;
  bool keep_schedule = false;
  std::vector<int32_t> src_row, row_level;
};

// Which rank factors which tile row.  Proportional mapping by splitting: the candidate subtrees start as the roots of
// the elimination forest; the heaviest candidate is replaced by its children (and becomes a row of the replicated top)
// until no candidate carries more than 1 / (4 nranks) of the work that is still distributable; the candidates then go to
// the ranks largest first, each to the least loaded rank.  Deterministic, the same on every rank.
void AssignOwners(const std::vector<int32_t>& row_start, const std::vector<int32_t>& row_tiles, int T, int nranks, HostPlan* H) {
  std::vector<int32_t> parent(static_cast<size_t>(T), -1);
  std::vector<std::vector<int32_t>> children(static_cast<size_t>(T));
  std::vector<int64_t> work(static_cast<size_t>(T), 0), subtree(static_cast<size_t>(T), 0);
  for (int I = 0; I < T; ++I) {
    const int64_t m = row_start[size_t(I) + 1] - row_start[size_t(I)] - 1;  // tiles right of the diagonal incl. the right-hand side
    work[size_t(I)] = m * (m - 1) / 2 + m;                                  // its tile-pair updates, about
    const int32_t p = row_tiles[size_t(row_start[size_t(I)]) + 1];
    if (p < T) { parent[size_t(I)] = p; children[size_t(p)].push_back(I); }
  }
  for (int I = 0; I < T; ++I) {  // children have smaller indices
    subtree[size_t(I)] += work[size_t(I)];
    if (parent[size_t(I)] >= 0) subtree[size_t(parent[size_t(I)])] += subtree[size_t(I)];
  }
  H->owner.assign(size_t(T), -1);
  std::vector<int32_t> cand, root_shared;
  int64_t distributable = 0;
  for (int I = 0; I < T; ++I)
    if (parent[size_t(I)] < 0) { cand.push_back(I); distributable += subtree[size_t(I)]; }
  auto heavier = [&](int32_t a, int32_t b) { return subtree[size_t(a)] != subtree[size_t(b)] ? subtree[size_t(a)] < subtree[size_t(b)] : a > b; };
  std::make_heap(cand.begin(), cand.end(), heavier);
  H->work_shared = 0;
  const int64_t total = distributable;
  std::vector<int32_t> atomic;  // candidates that cannot be split: chains of tile rows (the band pieces of the dissection)
  while (!cand.empty()) {
    const int32_t top = cand.front();
    // small enough: so is everything else.  (And the replicated top may not grow beyond a quarter of all the work.)
    if (subtree[size_t(top)] * 4 * nranks <= total || H->work_shared * 4 > total) break;
    std::pop_heap(cand.begin(), cand.end(), heavier);
    cand.pop_back();
    // a split only helps where the tree branches: walk down the chain below `top` to the first row with several children
    int32_t branch = top;
    int64_t chain_work = 0;
    while (children[size_t(branch)].size() == 1) { chain_work += work[size_t(branch)]; branch = children[size_t(branch)][0]; }
    if (children[size_t(branch)].empty() || (H->work_shared + chain_work + work[size_t(branch)]) * 4 > total) {
      atomic.push_back(top);
      continue;
    }
    for (int32_t I = top;; I = children[size_t(I)][0]) {  // the chain and the branching row join the replicated top
      H->work_shared += work[size_t(I)];
      distributable -= work[size_t(I)];
      root_shared.push_back(I);
      if (I == branch) break;
    }
    for (int32_t c : children[size_t(branch)]) { cand.push_back(c); std::push_heap(cand.begin(), cand.end(), heavier); }
  }
  cand.insert(cand.end(), atomic.begin(), atomic.end());
  std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return heavier(b, a); });  // largest first
  H->work_per_rank.assign(size_t(nranks), 0);
  std::vector<int32_t> root_owner(static_cast<size_t>(T), -1);
  for (int32_t c : cand) {
    int best = 0;
    for (int r = 1; r < nranks; ++r)
      if (H->work_per_rank[size_t(r)] < H->work_per_rank[size_t(best)]) best = r;
    H->work_per_rank[size_t(best)] += subtree[size_t(c)];
    root_owner[size_t(c)] = best;
  }
  for (int I = T - 1; I >= 0; --I) {  // parents have larger indices: owners flow down the subtrees
    if (root_owner[size_t(I)] >= 0) H->owner[size_t(I)] = root_owner[size_t(I)];
    else if (parent[size_t(I)] >= 0 && H->owner[size_t(parent[size_t(I)])] >= 0) H->owner[size_t(I)] = H->owner[size_t(parent[size_t(I)])];
  }
}

// rank < 0: the whole factorisation on one device; otherwise the schedule of `rank` of `nranks` (see HostPlan::owner)
void BuildHostPlan(int C, const int32_t* cell_c1, const int32_t* cell_c2, int64_t num_cells, HostPlan* H, int rank = -1, int nranks = 1) {
  std::vector<std::vector<int32_t>> adj(static_cast<size_t>(C));
  for (int64_t k = 0; k < num_cells; ++k) {
    const int c1 = cell_c1[k], c2 = cell_c2[k];
    if (c1 != c2) { adj[size_t(c1)].push_back(c2); adj[size_t(c2)].push_back(c1); }
  }
  const char* ordering_env = std::getenv("CX_SPARSE_ORDERING");
  const bool band_ordering = ordering_env != nullptr && std::string(ordering_env) == "rcm";
  PaddedLayout& layout = H->layout;
  std::vector<int32_t>& row_start = H->row_start;
  std::vector<int32_t>& row_tiles = H->row_tiles;
  if (band_ordering) {
    // round 1's orderings: one band (reverse Cuthill-McKee), optionally minimum degree on groups of 64 cameras,
    // whichever leaves fewer tiles after fill; one piece, no padding, the tile elimination tree is (close to) a chain
    std::vector<int32_t> pos = ReverseCuthillMcKee(C, adj);
    auto as_layout = [&](const std::vector<int32_t>& p) {
      std::vector<int32_t> order(static_cast<size_t>(C));
      for (int c = 0; c < C; ++c) order[size_t(p[size_t(c)])] = c;
      return LayOut(C, order, std::vector<int32_t>{int32_t(C)});
    };
    layout = as_layout(pos);
    TileStructure(cell_c1, cell_c2, num_cells, layout.cam_row, layout.T, &row_start, &row_tiles);
    std::vector<int32_t> rs2, rt2;
    PaddedLayout layout2 = as_layout(GroupMinimumDegree(C, adj, pos));
    TileStructure(cell_c1, cell_c2, num_cells, layout2.cam_row, layout2.T, &rs2, &rt2);
    if (rt2.size() < row_tiles.size()) {
      layout = layout2;
      row_start.swap(rs2);
      row_tiles.swap(rt2);
    }
  } else {
    // leaves of about 5 tile rows: below that a subtree is a chain of tile rows anyway
    layout = NestedDissection(C, adj, 36);
    TileStructure(cell_c1, cell_c2, num_cells, layout.cam_row, layout.T, &row_start, &row_tiles);
  }
  const int T = layout.T;
  const int64_t num_tiles = int64_t(row_tiles.size());
  H->T = T;
  H->num_tiles = num_tiles;
  // one pool, factored in place: refuse structures that would not fit comfortably
  if (double(num_tiles) * kTileDoubles * 8.0 > 160e9) { H->fits = false; return; }
  // Level schedule: the elimination tree of the TILE rows (parent = first tile right of the diagonal), tile rows
  // grouped by height; per level the diagonal tiles, the panel tiles and, for every tile that a row of the level
  // updates, its sources (pool indices of F(I, Ja), F(I, Jb)) in ascending I.
  std::vector<int32_t>& height = H->height;
  height.assign(static_cast<size_t>(T), 0);
  int max_h = 0;
  for (int I = 0; I < T; ++I) {  // children have smaller indices: one ascending sweep pushes heights up
    const int32_t q = row_start[size_t(I)] + 1;
    const int32_t parent = row_tiles[size_t(q)];  // the list always ends with T, so q is valid
    if (parent < T) height[size_t(parent)] = std::max(height[size_t(parent)], height[size_t(I)] + 1);
    max_h = std::max(max_h, height[size_t(I)]);
  }
  const int tree_levels = T > 0 ? max_h + 1 : 0;
  // the level of a row in the schedule: its height -- for a distributed plan the replicated rows come after ALL levels of
  // the rank's own rows (their heights shifted by the number of levels), rows of other ranks are left out (-1)
  const bool distributed = rank >= 0 && nranks > 1 && T > 0;
  std::vector<int32_t> vlevel(height);
  if (distributed) {
    AssignOwners(row_start, row_tiles, T, nranks, H);
    for (int I = 0; I < T; ++I) {
      const int32_t o = H->owner[size_t(I)];
      vlevel[size_t(I)] = o < 0 ? tree_levels + height[size_t(I)] : (o == rank ? height[size_t(I)] : -1);
    }
    H->L_split = tree_levels;
  }
  const int L = distributed ? 2 * tree_levels : tree_levels;
  H->L = L;
  std::vector<int32_t>&lrb = H->lrb, &lpb = H->lpb, &ltb = H->ltb;
  lrb.assign(size_t(L) + 1, 0);
  lpb.assign(size_t(L) + 1, 0);
  ltb.assign(size_t(L) + 1, 0);
  for (int I = 0; I < T; ++I)
    if (vlevel[size_t(I)] >= 0) lrb[size_t(vlevel[size_t(I)]) + 1]++;
  for (int l = 0; l < L; ++l) lrb[size_t(l) + 1] += lrb[size_t(l)];
  std::vector<int32_t>& level_rows = H->level_rows;
  level_rows.assign(static_cast<size_t>(lrb[size_t(L)]), 0);
  {
    std::vector<int32_t> cur(lrb.begin(), lrb.end() - 1);
    for (int I = 0; I < T; ++I)
      if (vlevel[size_t(I)] >= 0) level_rows[size_t(cur[size_t(vlevel[size_t(I)])]++)] = I;
  }
  std::vector<int32_t>&panel_row = H->panel_row, &panel_pool = H->panel_pool;
  struct Src { int32_t level, tgt, row, qa, qb; };
  std::vector<Src> srcs;
  for (int l = 0; l < L; ++l) {
    lpb[size_t(l)] = int32_t(panel_row.size());
    for (int32_t k = lrb[size_t(l)]; k < lrb[size_t(l) + 1]; ++k) {
      const int I = level_rows[size_t(k)];
      const int32_t q0 = row_start[size_t(I)], q1 = row_start[size_t(I) + 1];
      for (int32_t q = q0 + 1; q < q1; ++q) { panel_row.push_back(I); panel_pool.push_back(q); }
      for (int32_t qa = q0 + 1; qa < q1 - 1; ++qa) {  // Ja a real tile row; Jb up to the right-hand-side tile
        const int Ja = row_tiles[size_t(qa)];
        const auto first = row_tiles.begin() + row_start[size_t(Ja)], last = row_tiles.begin() + row_start[size_t(Ja) + 1];
        auto it = first;
        for (int32_t qb = qa; qb < q1; ++qb) {
          it = std::lower_bound(it, last, row_tiles[size_t(qb)]);  // present by construction of the symbolic fill
          srcs.push_back(Src{l, int32_t(it - row_tiles.begin()), I, qa, qb});
        }
      }
      // the plan's source and target indices are int32 (and the list itself 20 bytes per update on the host): a structure
      // with more than 2^30 tile-pair updates is refused like one whose pool does not fit
      if (srcs.size() > (size_t(1) << 30)) { H->fits = false; return; }
    }
  }
  lpb[size_t(L)] = int32_t(panel_row.size());
  // WHEN a product runs.  The contribution of row I to the tile (Ja, Jb) may be subtracted after any level from I's own up
  // to the one before Ja's.  Right after I's level (a right-looking schedule) nearly every target of a launch has ONE source
  // (1.06 on the Final shape), and such a workgroup is a dispatch, an index chain, the target's read-modify-write and a
  // drain around 1.7 us of products (see k_sp_update).  So a target's contributions are collected over a window of
  // `window` levels and subtracted by one workgroup as one chain of sources (the operands of the next source are requested
  // behind the current products, the target is read and written once per chain): the window opens at the target's first
  // pending source level l0 and closes at min(l0 + window - 1, level(Ja) - 1) -- per target, so the closing levels are spread
  // evenly instead of spiking every `window` levels.  window = 1 is the right-looking schedule, a large window the
  // left-looking one (everything just in time, which leaves the narrow top levels with long serial chains).  A distributed
  // plan closes every window of own levels before the exchange of the replicated tiles.
  // Measured (tools/sparse_window_ab.sh, reduced solve, Final shape / second scene): window 1: 38.5 / 111.2 ms, 4: 32.2 / 81.8,
  // 8: 34.4 / 82.7, 16: 39.9 / 92.0, 64: 73.3 / 157.2.  Per level (tools/sparse_levels.sh): the wide levels go from 18.5 to
  // 12-15 ns per product (the sustained fp64 MFMA rate is 10.9), the target's 64 KB of read-modify-write being shared by a
  // chain; longer windows leave the same rate on the wide levels and serial chains of 16-32 sources on the narrow ones.
  {
    int window = 4;
    if (const char* e = std::getenv("CX_SPARSE_WINDOW")) window = std::max(1, std::atoi(e));
    if (H->window > 0) window = H->window;
    if (window > 1) {
      std::vector<int32_t> row_of_slot(static_cast<size_t>(num_tiles));
      for (int J = 0; J < T; ++J)
        for (int32_t q = row_start[size_t(J)]; q < row_start[size_t(J) + 1]; ++q) row_of_slot[size_t(q)] = J;
      std::sort(srcs.begin(), srcs.end(), [](const Src& x, const Src& y) {
        return x.tgt != y.tgt ? x.tgt < y.tgt : (x.level != y.level ? x.level < y.level : x.row < y.row);
      });
      const int32_t split = distributed ? tree_levels : 0;
      // ... where the level a window opens at is WIDE: a level whose rows produce fewer products than fill the chip twice
      // (2 x 256 CUs x 5 workgroups) keeps the right-looking schedule -- there a chain of four only lengthens the level
      // (Final shape, levels 170-186: 12-13 us at window 1 against 23-26 at 4; a dense 51-row S: 2.7 against 3.1 ms)
      int64_t wide = 2560;
      if (const char* e = std::getenv("CX_SPARSE_WINDOW_MIN_PRODUCTS")) wide = std::atoll(e);
      std::vector<int64_t> level_products(size_t(L), 0);
      for (const Src& x : srcs) level_products[size_t(x.level)]++;
      size_t i = 0;
      while (i < srcs.size()) {
        const int32_t tq = srcs[i].tgt, l0 = srcs[i].level;
        const int w = level_products[size_t(l0)] >= wide ? window : 1;
        int32_t close = std::min<int32_t>(l0 + w - 1, vlevel[size_t(row_of_slot[size_t(tq)])] - 1);
        if (l0 < split) close = std::min<int32_t>(close, split - 1);
        size_t j = i;
        while (j < srcs.size() && srcs[j].tgt == tq && srcs[j].level <= close) srcs[j++].level = close;
        i = j;
      }
    }
  }
  std::sort(srcs.begin(), srcs.end(), [](const Src& x, const Src& y) {
    return x.level != y.level ? x.level < y.level : (x.tgt != y.tgt ? x.tgt < y.tgt : x.row < y.row);
  });
  std::vector<int32_t>&tgt_pool = H->tgt_pool, &tgt_flags = H->tgt_flags, &src_begin = H->src_begin, &src_a = H->src_a, &src_b = H->src_b;
  src_a.assign(srcs.size(), 0);
  src_b.assign(srcs.size(), 0);
  {
    int level = 0;
    for (size_t i = 0; i < srcs.size(); ++i) {
      while (level < srcs[i].level) ltb[size_t(++level)] = int32_t(tgt_pool.size());
      if (i == 0 || srcs[i].level != srcs[i - 1].level || srcs[i].tgt != srcs[i - 1].tgt) {
        const int32_t tq = srcs[i].tgt;
        // which row owns pool slot tq: the row whose range contains it
        const int Ja = int(std::upper_bound(row_start.begin(), row_start.end(), tq) - row_start.begin()) - 1;
        const int Jb = row_tiles[size_t(tq)];
        tgt_pool.push_back(tq);
        tgt_flags.push_back((Jb == Ja ? 1 : 0) | (Jb == T ? 2 : 0));
        src_begin.push_back(int32_t(i));
      }
      src_a[i] = srcs[i].qa;
      src_b[i] = srcs[i].qb;
    }
    if (H->keep_schedule) {
      H->src_row.resize(srcs.size());
      for (size_t i = 0; i < srcs.size(); ++i) H->src_row[i] = srcs[i].row;
      H->row_level = vlevel;
    }
    while (level < L) ltb[size_t(++level)] = int32_t(tgt_pool.size());
    src_begin.push_back(int32_t(srcs.size()));
  }
  if (std::getenv("CX_SPARSE_PLAN_STATS") != nullptr) {  // per level: rows, panel tiles, update targets, products, longest chain
    for (int l = 0; l < L; ++l) {
      int32_t longest = 0;
      for (int32_t t = ltb[size_t(l)]; t < ltb[size_t(l) + 1]; ++t) longest = std::max(longest, src_begin[size_t(t) + 1] - src_begin[size_t(t)]);
      const int32_t t0 = ltb[size_t(l)], t1 = ltb[size_t(l) + 1];
      std::fprintf(stderr, "cxsp level %d rows %d panels %d targets %d products %d longest %d\n", l, lrb[size_t(l) + 1] - lrb[size_t(l)],
                   lpb[size_t(l) + 1] - lpb[size_t(l)], t1 - t0, t1 > t0 ? src_begin[size_t(t1)] - src_begin[size_t(t0)] : 0, longest);
    }
  }
  // transposed index for forward solves with the stored factor: the tiles (K < I, I) of tile column I, ascending K
  std::vector<int32_t>&col_start = H->col_start, &col_pool = H->col_pool;
  col_start.assign(size_t(T) + 1, 0);
  {
    for (int K = 0; K < T; ++K)
      for (int32_t q = row_start[size_t(K)] + 1; q < row_start[size_t(K) + 1] - 1; ++q) col_start[size_t(row_tiles[size_t(q)]) + 1]++;
    for (int I = 0; I < T; ++I) col_start[size_t(I) + 1] += col_start[size_t(I)];
    col_pool.resize(size_t(col_start[size_t(T)]));
    std::vector<int32_t> cur(col_start.begin(), col_start.end() - 1);
    for (int K = 0; K < T; ++K)
      for (int32_t q = row_start[size_t(K)] + 1; q < row_start[size_t(K) + 1] - 1; ++q) col_pool[size_t(cur[size_t(row_tiles[size_t(q)])]++)] = q;
  }
}
}  // namespace

int cxsp_plan_from_cells(cx_context* ctx, int C, const int32_t* cell_c1, const int32_t* cell_c2, int64_t num_cells, cx_sp_plan* P,
                         bool distribute) {
  if (P->state != 0) return CX_OK;
  P->C = C;
  const int n = 9 * C;
  HostPlan H;
  static const bool distribution_off = std::getenv("CX_SPARSE_DISTRIBUTE") && std::atoi(std::getenv("CX_SPARSE_DISTRIBUTE")) == 0;  // A/B switch
  const bool dist = distribute && ctx->nranks > 1 && !distribution_off;
  BuildHostPlan(C, cell_c1, cell_c2, num_cells, &H, dist ? ctx->rank : -1, dist ? ctx->nranks : 1);
  if (!H.fits) { P->state = 2; return CX_OK; }
  hipStream_t st = ctx->stream;
  const int T = H.T, L = H.L;
  const int64_t num_tiles = H.num_tiles;
  const PaddedLayout& layout = H.layout;
  const std::vector<int32_t>&row_start = H.row_start, &row_tiles = H.row_tiles, &level_rows = H.level_rows, &panel_row = H.panel_row,
      &panel_pool = H.panel_pool, &tgt_pool = H.tgt_pool, &tgt_flags = H.tgt_flags, &src_begin = H.src_begin, &src_a = H.src_a,
      &src_b = H.src_b, &col_start = H.col_start, &col_pool = H.col_pool, &lrb = H.lrb, &lpb = H.lpb, &ltb = H.ltb;
  CX_TRY(P->d_cam_pos.upload(layout.cam_row, st));
  CX_TRY(P->d_valid.upload(layout.valid, st));
  CX_TRY(P->d_row_start.upload(row_start, st));
  CX_TRY(P->d_row_tiles.upload(row_tiles, st));
  CX_TRY(P->d_level_rows.upload(level_rows, st));
  CX_TRY(P->d_panel_row.upload(panel_row, st));
  CX_TRY(P->d_panel_pool.upload(panel_pool, st));
  CX_TRY(P->d_tgt_pool.upload(tgt_pool, st));
  CX_TRY(P->d_tgt_flags.upload(tgt_flags, st));
  CX_TRY(P->d_src_begin.upload(src_begin, st));
  CX_TRY(P->d_src_a.upload(src_a, st));
  CX_TRY(P->d_src_b.upload(src_b, st));
  CX_TRY(P->d_col_start.upload(col_start, st));
  {  // the device keeps the inverse of the transposed index: for every tile its slot in its column's list
    std::vector<int32_t> col_slot(static_cast<size_t>(num_tiles), -1);
    for (size_t p = 0; p < col_pool.size(); ++p) col_slot[size_t(col_pool[p])] = int32_t(p);
    CX_TRY(P->d_col_pool.upload(col_slot, st));
  }
  P->h_level_row_begin = lrb;
  P->h_level_panel_begin = lpb;
  P->h_level_tgt_begin = ltb;
  P->num_levels = L;
  P->num_tiles = num_tiles;
  P->T = T;
  P->level_split = H.L_split;
  if (H.L_split >= 0) {
    std::vector<int32_t> shared_tiles, keep(static_cast<size_t>(T), 0);
    for (int I = 0; I < T; ++I) {
      const int32_t o = H.owner[size_t(I)];
      keep[size_t(I)] = (o == ctx->rank || (o < 0 && ctx->rank == 0)) ? 1 : 0;
      if (o < 0)
        for (int32_t q = row_start[size_t(I)]; q < row_start[size_t(I) + 1]; ++q) shared_tiles.push_back(q);
    }
    P->num_shared_tiles = int64_t(shared_tiles.size());
    CX_TRY(P->d_shared_tiles.upload(shared_tiles, st));
    CX_TRY(P->d_row_keep.upload(keep, st));
    // Round 4: who needs which cell of S.  A cell's entries land in the tile rows of its camera that comes first in the
    // elimination order (one row, or two when the camera's nine rows straddle a 64-row boundary); the rank that factors
    // such a row needs the cell summed over the ranks, a replicated row's assembled values count on rank 0 only
    // (cxsp_factor_and_solve_sharded) -- nobody else needs the cell at all.  The ranks' lists, padded to the longest, are
    // the ranges of ONE reduce-scatter that replaces the all-reduce of every cell to every rank.
    {
      const int nr = ctx->nranks;
      std::vector<std::vector<int32_t>> need(static_cast<size_t>(nr));
      // cells of rows with an owner first; then the cells that only replicated rows need, each to the rank with the shortest
      // list so far (any ONE rank may bring such a cell into the sum at the split).  A cell that straddles an owned and a
      // replicated row goes to that owner alone -- it would assemble the replicated row's part anyway.
      std::vector<int64_t> only_shared;
      for (int64_t k = 0; k < num_cells; ++k) {
        const int32_t p0 = std::min(layout.cam_row[size_t(cell_c1[k])], layout.cam_row[size_t(cell_c2[k])]);
        const int32_t o0 = H.owner[size_t(p0 >> 6)], o1 = H.owner[size_t((p0 + 8) >> 6)];
        if (o0 < 0 && o1 < 0) { only_shared.push_back(k); continue; }
        if (o0 >= 0) need[size_t(o0)].push_back(int32_t(k));
        if (o1 >= 0 && o1 != o0) need[size_t(o1)].push_back(int32_t(k));
      }
      for (int64_t k : only_shared) {
        int best = 0;
        for (int r = 1; r < nr; ++r)
          if (need[size_t(r)].size() < need[size_t(best)].size()) best = r;
        need[size_t(best)].push_back(int32_t(k));
      }
      // Round 4, stored-factor sweeps on a distributed factor (cxsp_solve): the forward sweep of a replicated row J sums
      // the partial products F(K,J)' y_K of ALL rows K < J -- those of rows with an owner exist on their owner only.  Per
      // replicated row the slots of such tiles, in ascending K: their sum is formed per rank, summed over the ranks and put
      // back into the first of them (k_sp_fwd_gather_shared / k_sp_fwd_scatter_shared).
      {
        std::vector<int32_t> row_of_tile(static_cast<size_t>(num_tiles), -1);
        for (int K = 0; K < T; ++K)
          for (int32_t q = row_start[size_t(K)]; q < row_start[size_t(K) + 1]; ++q) row_of_tile[size_t(q)] = K;
        std::vector<int32_t> xs_rows, xs_begin(1, 0), xs_slots;
        for (int J = 0; J < T; ++J) {
          if (H.owner[size_t(J)] >= 0) continue;
          const size_t before = xs_slots.size();
          for (int32_t pslot = col_start[size_t(J)]; pslot < col_start[size_t(J) + 1]; ++pslot)
            if (H.owner[size_t(row_of_tile[size_t(col_pool[size_t(pslot)])])] >= 0) xs_slots.push_back(pslot);
          if (xs_slots.size() > before) {
            xs_rows.push_back(J);
            xs_begin.push_back(int32_t(xs_slots.size()));
          }
        }
        P->xs_count = int32_t(xs_rows.size());
        if (xs_slots.empty()) xs_slots.push_back(0);
        if (xs_rows.empty()) xs_rows.push_back(0);
        CX_TRY(P->d_xs_begin.upload(xs_begin, st));
        CX_TRY(P->d_xs_slots.upload(xs_slots, st));
        CX_TRY(P->d_xs_t.alloc(size_t(std::max(P->xs_count, 1)) * kTile));
      }
      std::vector<int32_t> mine(static_cast<size_t>(std::max<int64_t>(num_cells, 1)), 0);
      for (int32_t k : need[size_t(ctx->rank)]) mine[size_t(k)] = 1;
      CX_TRY(P->d_cell_mine.upload(mine, st));
      size_t chunk = 1;
      for (const auto& v : need) chunk = std::max(chunk, v.size());
      std::vector<int32_t> send_cells(size_t(nr) * chunk, -1);
      for (int r = 0; r < nr; ++r) std::copy(need[size_t(r)].begin(), need[size_t(r)].end(), send_cells.begin() + int64_t(size_t(r) * chunk));
      P->rs_chunk_cells = int64_t(chunk);
      CX_TRY(P->d_rs_cells.upload(send_cells, st));
      if (std::getenv("CX_SPARSE_CHOLESKY_VERBOSE")) {
        std::fprintf(stderr, "[cxschur] cell values by owner, rank %d: %lld cells in the union, ranges of %lld cells (", ctx->rank, (long long)num_cells, (long long)chunk);
        for (const auto& v : need) std::fprintf(stderr, " %zu", v.size());
        std::fprintf(stderr, " ): %.2f GB into the reduce-scatter instead of %.2f GB all-reduced\n", double(nr) * double(chunk) * 648e-9, double(num_cells) * 648e-9);
      }
    }
    P->h_work_per_rank = H.work_per_rank;
    P->h_work_shared = H.work_shared;
    if (std::getenv("CX_SPARSE_CHOLESKY_VERBOSE")) {
      std::fprintf(stderr, "[cxschur] distributed factorisation, rank %d of %d: %lld tiles of replicated rows (%.2f GB exchanged per solve), "
                   "tile-pair updates replicated %lld, per rank", ctx->rank, ctx->nranks, (long long)P->num_shared_tiles,
                   double(P->num_shared_tiles) * kTileDoubles * 8e-9, (long long)H.work_shared);
      for (int64_t w : H.work_per_rank) std::fprintf(stderr, " %lld", (long long)w);
      std::fprintf(stderr, "\n");
    }
  }
  P->state = 1;
  if (std::getenv("CX_SPARSE_CHOLESKY_VERBOSE"))
    std::fprintf(stderr, "[cxschur] tile-sparse Cholesky plan: %d cameras, %d tile rows (%d rows of padding) in %d levels, %lld tiles (%.2f GB, "
                 "dense would be %.2f GB), %zu tile-pair updates on %zu (level, target) pairs\n", C, T, kTile * T - n, L, (long long)num_tiles,
                 double(num_tiles) * kTileDoubles * 8e-9, double(n) * n * 8e-9, src_a.size(), tgt_pool.size());
  return CX_OK;
}

int cxsp_build_plan(cx_matrix* A) {
  if (A->sp.state != 0) return CX_OK;
  const auto t0 = std::chrono::steady_clock::now();
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state != 1) { A->sp.state = 2; return CX_OK; }
  const auto t1 = std::chrono::steady_clock::now();
  const int rc = cxsp_plan_from_cells(A->ctx, A->C, A->h_cell_c1.data(), A->h_cell_c2.data(), A->num_cells, &A->sp);
  if (std::getenv("CX_SPARSE_CHOLESKY_VERBOSE"))
    std::fprintf(stderr, "[cxschur] one-time structure analysis: pair lists %.3f s, ordering + symbolic factorisation + plan %.3f s\n",
                 std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
  return rc;
}

namespace {
// layout of plan->d_x: [T * 64] vector in the padded elimination order | two inverted 32 x 32 diagonal blocks per tile row |
// [tiles][64] partial products of the sweeps
struct Scratch { double *xp, *uinv, *partial; };
// f(pool pointer) with the pool in its scalar type (see cx_sp_plan::f32)
template <typename F>
int WithPool(cx_sp_plan* P, F&& f) {
  return P->f32 ? f(P->d_W32.p) : f(P->d_W.p);
}
int AllocPool(cx_sp_plan* P, hipStream_t st) {
  const size_t pool = size_t(P->num_tiles) * kTileDoubles;
  // (f32 is a per-solve option: a matrix solved now with one pool, now with the other keeps only the one in use -- the
  // other's gigabytes go back to the allocator after the stream has drained, which hipFree waits for)
  if (P->f32) {
    P->d_W.release();
    CX_TRY(P->d_W32.alloc(pool));
    CX_HIP(hipMemsetAsync(P->d_W32.p, 0, pool * sizeof(float), st));
  } else {
    P->d_W32.release();
    CX_TRY(P->d_W.alloc(pool));
    CX_HIP(hipMemsetAsync(P->d_W.p, 0, pool * sizeof(double), st));
  }
  return CX_OK;
}
int GetScratch(cx_sp_plan* P, Scratch* s) {
  const size_t npad = size_t(P->T) * kTile;
  CX_TRY(P->d_x.alloc(npad + 2 * size_t(P->T) * NB * NB + size_t(P->num_tiles) * kTile));
  s->xp = P->d_x.p;
  s->uinv = s->xp + npad;
  s->partial = s->uinv + 2 * size_t(P->T) * NB * NB;
  return CX_OK;
}
}  // namespace

// zero the pool and scatter (the selected) S cells of the matrix into it; A's gather assembly must have run
// (cxs_assemble_pair_items)
int cxsp_assemble(cx_matrix* A, cx_sp_plan* P, const double* Df, const int32_t* sel_cells, const int32_t* sel_offdiag, int64_t num_sel,
                  double offdiag_scale) {
  hipStream_t st = A->ctx->stream;
  CX_TRY(AllocPool(P, st));
  const int64_t count = sel_cells ? num_sel : A->num_cells;
  if (count > 0)
    CX_TRY(WithPool(P, [&](auto* W) -> int {
      using TW = std::remove_pointer_t<decltype(W)>;
      hipLaunchKernelGGL(k_sp_assemble<TW>, dim3(unsigned((count + 2) / 3)), dim3(3 * 81), 0, st, (const int32_t*)A->d_cell_c1.p,
                         (const int32_t*)A->d_cell_c2.p, (const int32_t*)A->d_cell_item_start.p, (const double*)A->d_item_partial.p,
                         (const double*)A->d_elim_diag.p, Df, (const int32_t*)P->d_cam_pos.p, (const int32_t*)P->d_row_start.p,
                         (const int32_t*)P->d_row_tiles.p, W, count, sel_cells, sel_offdiag, offdiag_scale);
      return CX_OK;
    }));
  CX_HIP(hipGetLastError());
  return CX_OK;
}

namespace {
// ---- distributed factorisation: the exchange at the split and its helpers
template <typename TW>
__global__ void k_sp_zero_tiles(TW* __restrict__ W, const int32_t* __restrict__ tiles, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * kTileDoubles) return;
  W[int64_t(tiles[i / kTileDoubles]) * kTileDoubles + (i % kTileDoubles)] = TW(0);
}
// dir 0: pool -> packed (and the flag into the last slot), 1: packed -> pool (and the summed flag back)
template <typename TW>
__global__ void k_sp_pack_tiles(TW* __restrict__ W, const int32_t* __restrict__ tiles, int64_t n, double* __restrict__ packed,
                                int* __restrict__ flag, int dir) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i == n * kTileDoubles) {
    if (dir == 0) packed[i] = double(*flag != 0);
    else if (packed[i] > 0.0) *flag = 1;
    return;
  }
  if (i > n * kTileDoubles) return;
  const int64_t w = int64_t(tiles[i / kTileDoubles]) * kTileDoubles + (i % kTileDoubles);
  if (dir == 0) packed[i] = double(W[w]);
  else W[w] = TW(packed[i]);
}
__global__ void k_sp_mask_rows(double* __restrict__ xp, const int32_t* __restrict__ keep, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && !keep[i / kTile]) xp[i] = 0.0;
}
// The tiles of the replicated rows hold, on every rank, what the rank's own subtrees subtracted from them (rank 0: on top of
// the assembled values, the others: on top of zeros, cxsp_prepare_distributed): their sum is the matrix the top of the tree
// is factored from.  A rank whose subtree met a non-positive pivot says so in the same message, so that every rank ends the
// solve the same way.
int ExchangeSharedTiles(cx_context* ctx, cx_sp_plan* P, int* d_flag) {
  hipStream_t st = ctx->stream;
  const int64_t n = P->num_shared_tiles, count = n * kTileDoubles + 1;
  CX_TRY(P->d_exchange.alloc(size_t(count)));
  auto pack = [&](int dir) {
    return WithPool(P, [&](auto* W) -> int {
      using TW = std::remove_pointer_t<decltype(W)>;
      hipLaunchKernelGGL(k_sp_pack_tiles<TW>, dim3(unsigned((count + 255) / 256)), dim3(256), 0, st, W, (const int32_t*)P->d_shared_tiles.p, n,
                         P->d_exchange.p, d_flag, dir);
      CX_HIP(hipGetLastError());
      return CX_OK;
    });
  };
  CX_TRY(pack(0));
  CX_TRY(cx_allreduce_device(ctx, P->d_exchange.p, count));  // (in double for either pool: the ranks' sums round once)
  CX_TRY(pack(1));
  return CX_OK;
}
}  // namespace

// numeric factorisation of the assembled pool, in place, level by level; a right-hand side placed in the rows' last
// tiles (k_sp_rhs) is forward-substituted along
namespace {
// (Round 3 A/B, not kept: TWO STREAMS.  Only the chains whose targets lie in rows of the next level have to precede the next
// diagonal kernel; the others -- window closings, with the rule that no window closes at the level before its target's last --
// were launched on a second stream beside the diagonal and panel kernels of the next two levels, with one event each way per
// level.  Reduced solve 32.1 -> 31.1 ms (Final), 81.8 -> 78.6 (second scene), against the ~5 ms the 190 x 45 us of diagonal +
// panel kernels suggested.  The kernel trace says why: (1) the two streams' kernels are dispatched one grid at a time -- a
// 27-wavefront diagonal kernel issued beside a 15 k-workgroup update STARTS at once but ends 116-250 us later, about when the
// update's last workgroup has been placed, at equal and at highest-against-lowest stream priority alike; (2) every cross-stream
// event costs ~11 us between the panel kernel's end and the next launch's start, 2 ms per factorisation; (3) the just-in-time part
// is 40-50 % of a wide level's products.  What it hid did not pay for a second stream, five events and a schedule rule.)
template <typename TW>
int FactorLevels(cx_context* ctx, cx_sp_plan* P, TW* W, int* d_flag) {
  hipStream_t st = ctx->stream;
  Scratch sc;
  CX_TRY(GetScratch(P, &sc));
  const int32_t* valid = P->d_valid.p;
  const int32_t* rows = P->d_level_rows.p;
  for (int l = 0; l < P->num_levels; ++l) {
    if (l == P->level_split) CX_TRY(ExchangeSharedTiles(ctx, P, d_flag));  // the ranks' own subtrees are done: sum what they sent up
    const int r0 = P->h_level_row_begin[size_t(l)], nr = P->h_level_row_begin[size_t(l) + 1] - r0;
    const int p0 = P->h_level_panel_begin[size_t(l)], np = P->h_level_panel_begin[size_t(l) + 1] - p0;
    const int t0 = P->h_level_tgt_begin[size_t(l)], nt = P->h_level_tgt_begin[size_t(l) + 1] - t0;
    if (nr > 0)
      hipLaunchKernelGGL(k_sp_diag<TW>, dim3(unsigned(nr)), dim3(64), 0, st, W, (const int32_t*)P->d_row_start.p, rows + r0, valid, sc.uinv, d_flag);
    if (np > 0)
      hipLaunchKernelGGL(k_sp_panel<TW>, dim3(unsigned(np)), dim3(128), 0, st, W, (const int32_t*)P->d_row_start.p,
                         (const int32_t*)P->d_row_tiles.p, (const int32_t*)P->d_panel_row.p + p0,
                         (const int32_t*)P->d_panel_pool.p + p0, valid, P->T, (const double*)sc.uinv);
    if (nt > 0) {
      if constexpr (std::is_same_v<TW, float>) {
        static const int f32_lds = std::getenv("CX_SPARSE_F32_LDS") ? atoi(std::getenv("CX_SPARSE_F32_LDS")) : 1;  // A/B switch
#define CX_SP_F32_LDS(R, K)                                                                                                            \
  hipLaunchKernelGGL((k_sp_update_f32_lds<R, K>), dim3(unsigned(nt)), dim3(256), 0, st, W, (const int32_t*)P->d_tgt_pool.p + t0,         \
                     (const int32_t*)P->d_tgt_flags.p + t0, (const int32_t*)P->d_src_begin.p + t0, (const int32_t*)P->d_src_a.p,        \
                     (const int32_t*)P->d_src_b.p)
        if (f32_lds == 644) CX_SP_F32_LDS(64, 4);  // (A/B: whole tiles, four workgroups per CU)
        else if (f32_lds) CX_SP_F32_LDS(16, 8);
        else
          hipLaunchKernelGGL((k_sp_update_f32<4, 8>), dim3(unsigned(nt)), dim3(256), 0, st, W, (const int32_t*)P->d_tgt_pool.p + t0,
                             (const int32_t*)P->d_tgt_flags.p + t0, (const int32_t*)P->d_src_begin.p + t0, (const int32_t*)P->d_src_a.p,
                             (const int32_t*)P->d_src_b.p);
      } else {
      static const int f64_lds = std::getenv("CX_SPARSE_F64_LDS") ? atoi(std::getenv("CX_SPARSE_F64_LDS")) : 1;  // A/B switch
      if (f64_lds) {
#define CX_SP_F64_LDS(R, K)                                                                                                            \
  hipLaunchKernelGGL((k_sp_update_f64_lds<R, K>), dim3(unsigned(nt)), dim3(256), 0, st, W, (const int32_t*)P->d_tgt_pool.p + t0,         \
                     (const int32_t*)P->d_tgt_flags.p + t0, (const int32_t*)P->d_src_begin.p + t0, (const int32_t*)P->d_src_a.p,        \
                     (const int32_t*)P->d_src_b.p)
        if (f64_lds == 324) CX_SP_F64_LDS(32, 4);  // (A/B: 32-row parts, four workgroups per CU)
        else CX_SP_F64_LDS(16, 6);
#undef CX_SP_F64_LDS
        continue;
      }
      static const int occ = [] { const char* v = std::getenv("CX_SPARSE_UPDATE_OCCUPANCY"); return v ? atoi(v) : 3; }();
#define CX_SP_UPDATE(K)                                                                                                          \
  hipLaunchKernelGGL(k_sp_update<K>, dim3(unsigned(nt)), dim3(256), 0, st, W, (const int32_t*)P->d_tgt_pool.p + t0,              \
                     (const int32_t*)P->d_tgt_flags.p + t0, (const int32_t*)P->d_src_begin.p + t0,                               \
                     (const int32_t*)P->d_src_a.p, (const int32_t*)P->d_src_b.p)
      static const int slices = std::getenv("CX_SPARSE_UPDATE_SLICES") ? atoi(std::getenv("CX_SPARSE_UPDATE_SLICES")) : 8;  // rows; 0: k_sp_update
#define CX_SP_SLICES(S, K)                                                                                                      \
  hipLaunchKernelGGL((k_sp_update_slices<S, K>), dim3(unsigned(nt)), dim3(256), 0, st, W, (const int32_t*)P->d_tgt_pool.p + t0,  \
                     (const int32_t*)P->d_tgt_flags.p + t0, (const int32_t*)P->d_src_begin.p + t0,                               \
                     (const int32_t*)P->d_src_a.p, (const int32_t*)P->d_src_b.p)
      if (slices == 16) CX_SP_SLICES(4, 4);
      else if (slices == 8) CX_SP_SLICES(2, 5);
      else if (slices == 4) CX_SP_SLICES(1, 6);
      else if (occ >= 4) CX_SP_UPDATE(4);
      else if (occ == 3) CX_SP_UPDATE(3);
      else CX_SP_UPDATE(2);
#undef CX_SP_UPDATE
#undef CX_SP_SLICES
#undef CX_SP_F32_LDS
      }
    }
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
}  // namespace
int cxsp_factor(cx_context* ctx, cx_sp_plan* P, int* d_flag) {
  return WithPool(P, [&](auto* W) -> int { return FactorLevels(ctx, P, W, d_flag); });
}

namespace {
// backward sweep U x = y by levels, top down; y_in_x: y is sc.xp itself, otherwise the rows' right-hand-side tiles
int BackwardSweep(cx_context* ctx, cx_sp_plan* P, const Scratch& sc, int y_in_x) {
  hipStream_t st = ctx->stream;
  const int32_t* rows = P->d_level_rows.p;
  for (int l = P->num_levels - 1; l >= 0; --l) {
    const int r0 = P->h_level_row_begin[size_t(l)], nr = P->h_level_row_begin[size_t(l) + 1] - r0;
    const int p0 = P->h_level_panel_begin[size_t(l)], np = P->h_level_panel_begin[size_t(l) + 1] - p0;
    // (A/B, round 2: both parts of a level in one launch -- the workgroup that finishes a row's last tile, told by an arrival
    // counter, going on with the row's part 2 -- took the same time: 3.474 against 3.472 ms per CG iteration with
    // CLUSTER_TRIDIAGONAL on the Final shape; the ticket costs what the kernel boundary cost.  Not kept.)
    CX_TRY(WithPool(P, [&](auto* Wp) -> int {
      using TW = std::remove_pointer_t<decltype(Wp)>;
      const TW* W = Wp;
      if (np > 0)
        hipLaunchKernelGGL(k_sp_bwd_partial<TW>, dim3(unsigned(np)), dim3(256), 0, st, W, (const int32_t*)P->d_row_tiles.p,
                           (const int32_t*)P->d_panel_pool.p + p0, (const int32_t*)P->d_valid.p, P->T, (const double*)sc.xp, sc.partial);
      if (nr > 0)
        hipLaunchKernelGGL(k_sp_bwd_level<TW>, dim3(unsigned(nr)), dim3(256), 0, st, W, (const int32_t*)P->d_row_start.p,
                           rows + r0, (const int32_t*)P->d_valid.p, (const double*)sc.uinv, (const double*)sc.partial, sc.xp, y_in_x);
      return CX_OK;
    }));
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
}  // namespace

namespace {
// t[j] = sum of the listed partial-product slots of replicated row j (list order: ascending source row)
__global__ void k_sp_fwd_gather_shared(const int32_t* __restrict__ begin, const int32_t* __restrict__ slots, const double* __restrict__ partial,
                                       double* __restrict__ t) {
  const int j = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int p = begin[j]; p < begin[j + 1]; ++p) s += partial[size_t(slots[p]) * kTile + lane];
  t[size_t(j) * kTile + lane] = s;
}
// ... and back: the first listed slot takes the sum over the ranks, the others zero
__global__ void k_sp_fwd_scatter_shared(const int32_t* __restrict__ begin, const int32_t* __restrict__ slots, const double* __restrict__ t,
                                        double* __restrict__ partial) {
  const int j = blockIdx.x, lane = threadIdx.x;
  for (int p = begin[j]; p < begin[j + 1]; ++p) partial[size_t(slots[p]) * kTile + lane] = (p == begin[j]) ? t[size_t(j) * kTile + lane] : 0.0;
}
}  // namespace

// z = M^-1 r with the stored factor M = U'U (r, z in camera order): U' y = r bottom up, U x = y top down.
// On a distributed factor (round 4): every rank sweeps its own rows and the replicated top; at the split the replicated rows'
// incoming partial sums from owned rows are summed over the ranks (one collective of 64 doubles per such row), the top is
// swept on every rank, the backward sweep needs nothing from other ranks, and the solution is the sum of the rows each rank
// keeps (one collective of 64 T doubles, as in cxsp_factor_and_solve_sharded).
int cxsp_solve(cx_context* ctx, cx_sp_plan* P, const double* r, double* z) {
  hipStream_t st = ctx->stream;
  const int C = P->C, n = 9 * C;
  if (n == 0) return CX_OK;
  Scratch sc;
  CX_TRY(GetScratch(P, &sc));
  const bool distributed = P->level_split >= 0 && ctx->nranks > 1;
  CX_HIP(hipMemsetAsync(sc.xp, 0, size_t(P->T) * kTile * sizeof(double), st));
  if (distributed) CX_HIP(hipMemsetAsync(sc.partial, 0, size_t(P->num_tiles) * kTile * sizeof(double), st));  // (other ranks' rows: no products here)
  hipLaunchKernelGGL(k_sp_permute, dim3((n + 255) / 256), dim3(256), 0, st, r, (const int32_t*)P->d_cam_pos.p, sc.xp, C);
  const int32_t* rows = P->d_level_rows.p;
  for (int l = 0; l < P->num_levels; ++l) {
    if (distributed && l == P->level_split && P->xs_count > 0) {
      hipLaunchKernelGGL(k_sp_fwd_gather_shared, dim3(unsigned(P->xs_count)), dim3(kTile), 0, st, (const int32_t*)P->d_xs_begin.p,
                         (const int32_t*)P->d_xs_slots.p, (const double*)sc.partial, P->d_xs_t.p);
      CX_HIP(hipGetLastError());
      CX_TRY(cx_allreduce_device(ctx, P->d_xs_t.p, int64_t(P->xs_count) * kTile));
      hipLaunchKernelGGL(k_sp_fwd_scatter_shared, dim3(unsigned(P->xs_count)), dim3(kTile), 0, st, (const int32_t*)P->d_xs_begin.p,
                         (const int32_t*)P->d_xs_slots.p, (const double*)P->d_xs_t.p, sc.partial);
    }
    const int r0 = P->h_level_row_begin[size_t(l)], nr = P->h_level_row_begin[size_t(l) + 1] - r0;
    const int p0 = P->h_level_panel_begin[size_t(l)], np = P->h_level_panel_begin[size_t(l) + 1] - p0;
    CX_TRY(WithPool(P, [&](auto* Wp) -> int {
      using TW = std::remove_pointer_t<decltype(Wp)>;
      const TW* W = Wp;
      if (nr > 0)
        hipLaunchKernelGGL(k_sp_fwd_level<TW>, dim3(unsigned(nr)), dim3(256), 0, st, W, (const int32_t*)P->d_row_start.p,
                           rows + r0, (const int32_t*)P->d_valid.p, (const double*)sc.uinv, (const int32_t*)P->d_col_start.p,
                           (const double*)sc.partial, sc.xp);
      if (np > 0)
        hipLaunchKernelGGL(k_sp_fwd_partial<TW>, dim3(unsigned(np)), dim3(256), 0, st, W, (const int32_t*)P->d_row_tiles.p,
                           (const int32_t*)P->d_panel_row.p + p0, (const int32_t*)P->d_panel_pool.p + p0, (const int32_t*)P->d_col_pool.p, P->T,
                           (const double*)sc.xp, sc.partial);
      return CX_OK;
    }));
  }
  CX_TRY(BackwardSweep(ctx, P, sc, 1));
  if (distributed) {
    const int64_t npad = int64_t(P->T) * kTile;
    hipLaunchKernelGGL(k_sp_mask_rows, dim3(unsigned((npad + 255) / 256)), dim3(256), 0, st, sc.xp, (const int32_t*)P->d_row_keep.p, npad);
    CX_TRY(cx_allreduce_device(ctx, sc.xp, npad));
  }
  hipLaunchKernelGGL(k_sp_unpermute, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)sc.xp, (const int32_t*)P->d_cam_pos.p, z, C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// S z = rhs for the explicit S of the matrix, assembled straight into the tile pool (A's gather assembly must
// have run: cxs_assemble_pair_items); z and rhs in camera order.
int cxsp_factor_and_solve(cx_matrix* A, const double* Df, const double* rhs, double* z, int* d_flag) {
  cx_context* ctx = A->ctx;
  hipStream_t st = ctx->stream;
  cx_sp_plan* P = &A->sp;
  const int C = A->C, n = 9 * C;
  if (n == 0) return CX_OK;
  CX_TRY(cxsp_assemble(A, P, Df, nullptr, nullptr, 0, 1.0));
  CX_TRY(WithPool(P, [&](auto* W) -> int {
    using TW = std::remove_pointer_t<decltype(W)>;
    hipLaunchKernelGGL(k_sp_rhs<TW>, dim3((n + 255) / 256), dim3(256), 0, st, rhs, (const int32_t*)P->d_cam_pos.p,
                       (const int32_t*)P->d_row_start.p, W, C);
    return CX_OK;
  }));
  CX_TRY(cxsp_factor(ctx, P, d_flag));
  Scratch sc;
  CX_TRY(GetScratch(P, &sc));
  CX_TRY(BackwardSweep(ctx, P, sc, 0));
  hipLaunchKernelGGL(k_sp_unpermute, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)sc.xp, (const int32_t*)P->d_cam_pos.p, z, C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Sharded matrix: points over ranks.  Every rank sees the cells its own points create; the plan
// must be the same everywhere, so it is built from the union of the ranks' cells.
namespace {
__global__ void k_sp_mark_cells(const int32_t* __restrict__ c1, const int32_t* __restrict__ c2, int64_t num_cells, int C,
                                double* __restrict__ present) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < num_cells) present[int64_t(c1[k]) * C + c2[k]] = 1.0;
}
// this rank's cell values into their slots of the union's cell-major array (81 doubles per cell)
// the reduce-scatter of the cell values by owner: pack the ranks' ranges out of the (local) union array, unpack the own
// range of the sum back into it
__global__ void k_sp_pack_cells(const double* __restrict__ values, const int32_t* __restrict__ cells, int64_t n, double* __restrict__ packed) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * 81) return;
  const int64_t k = i / 81;
  const int32_t c = cells[k];
  packed[i] = c >= 0 ? values[int64_t(c) * 81 + (i - k * 81)] : 0.0;
}
__global__ void k_sp_unpack_cells(const double* __restrict__ packed, const int32_t* __restrict__ cells, int64_t n, double* __restrict__ values) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * 81) return;
  const int64_t k = i / 81;
  const int32_t c = cells[k];
  if (c >= 0) values[int64_t(c) * 81 + (i - k * 81)] = packed[i];
}
__global__ void k_sp_scatter_cells(const double* __restrict__ local, const int32_t* __restrict__ local_to_union, int64_t num_cells,
                                   double* __restrict__ values) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= num_cells * 81) return;
  const int64_t cell = i / 81;
  values[int64_t(local_to_union[cell]) * 81 + (i - cell * 81)] = local[i];
}
// k_sp_assemble for cell values that are already summed (over items and over ranks): D_f^2 is added here, once
template <typename TW>
__global__ __launch_bounds__(3 * 81) void k_sp_assemble_values(const int32_t* __restrict__ cell_c1, const int32_t* __restrict__ cell_c2,
                                                               const double* __restrict__ values, const double* __restrict__ Df,
                                                               const int32_t* __restrict__ cam_pos, const int32_t* __restrict__ row_start,
                                                               const int32_t* __restrict__ row_tiles, TW* __restrict__ W,
                                                               int64_t num_cells, const int32_t* __restrict__ mine = nullptr) {
  const int64_t cell = int64_t(blockIdx.x) * 3 + threadIdx.x / 81;
  if (cell >= num_cells) return;
  if (mine && !mine[cell]) return;  // (cell values by owner: another rank assembles this cell, D_f^2 included)
  const int el = threadIdx.x % 81;
  const int c1 = cell_c1[cell], c2 = cell_c2[cell];
  const int a = el / 9, c = el - a * 9;
  double v = values[cell * 81 + el];
  if (c1 == c2 && Df && a == c) {
    const double d = Df[9 * int64_t(c1) + a];
    v += d * d;
  }
  int row = cam_pos[c1] + a, col = cam_pos[c2] + c;
  if (c1 != c2 && row > col) { const int t = row; row = col; col = t; }
  if (row > col) return;
  const int I = row >> 6, J = col >> 6;
  const int idx = tile_find(row_tiles, row_start[I], row_start[I + 1], J);
  W[size_t(idx) * kTileDoubles + (row & 63) * kTile + (col & 63)] = TW(v);
}
}  // namespace

int cxsp_build_plan_sharded(cx_matrix* A) {
  cx_sp_plan* P = &A->sp;
  // (the solves with a stored factor that iterative refinement needs, cxsp_solve, exist for a whole factor only: a plan that
  // was distributed is rebuilt when the caller now wants the factor replicated -- the same decision on every rank)
  // Round 4: and back -- an unrefined solve after a refined one gets its distributed factorisation again.  The
  // flag is a per-solve option kept on the matrix' plan, so the plan remembers how it was built; re-planning uses the
  // union cell list kept on the host (no second presence exchange), the same on every rank.
  cx_context* ctx = A->ctx;
  if (P->state == 1 && P->built_replicate != P->replicate && !P->h_union_c1.empty()) {
    P->built_replicate = P->replicate;
    P->state = 0;
    return cxsp_plan_from_cells(ctx, A->C, P->h_union_c1.data(), P->h_union_c2.data(), P->num_union_cells, P, /*distribute=*/!P->replicate);
  }
  if (P->state != 0) return CX_OK;
  hipStream_t st = ctx->stream;
  const int C = A->C;
  if (int64_t(C) > 16384) {  // (the same on every rank: all of them return here)
    cx_set_error("SPARSE_SCHUR on a sharded matrix is limited to 16384 cameras (dense exchange of the cell structure)");
    return CX_ERR_UNSUPPORTED;
  }
  // Everything that can fail on ONE rank only (its pair lists, its 8 C^2 bytes of exchange buffer) happens before a
  // one-number agreement: a rank that failed says so, and then NO rank enters the large exchange -- instead of the
  // failed rank returning while the others wait in the all-reduce for good.
  DevBuf<double> present, agree;
  int local_rc = cxs_build_pair_lists(A);
  if (local_rc == CX_OK) local_rc = present.alloc(size_t(C) * C + 1);
  const std::string local_error = local_rc == CX_OK ? std::string() : std::string(cx_last_error());
  CX_TRY(agree.alloc(1));
  const double failed_here = local_rc == CX_OK ? 0.0 : 1.0;
  CX_HIP(hipMemcpyAsync(agree.p, &failed_here, sizeof(double), hipMemcpyHostToDevice, st));
  CX_TRY(cx_allreduce_device(ctx, agree.p, 1));
  double failed_anywhere = 0.0;
  CX_TRY(cx_read_back(ctx, &failed_anywhere, agree.p, sizeof(double), st));
  CX_TRY(cx_stream_sync(ctx, st));
  if (local_rc != CX_OK) {
    cx_set_error("%s", local_error.c_str());
    return local_rc;
  }
  if (failed_anywhere > 0.0) {
    cx_set_error("another rank could not build its part of the S structure");
    return CX_ERR_COMM;
  }
  CX_HIP(hipMemsetAsync(present.p, 0, (size_t(C) * C + 1) * sizeof(double), st));
  if (A->pairs_state == 1 && A->num_cells > 0)
    hipLaunchKernelGGL(k_sp_mark_cells, dim3(unsigned((A->num_cells + 255) / 256)), dim3(256), 0, st, (const int32_t*)A->d_cell_c1.p,
                       (const int32_t*)A->d_cell_c2.p, A->num_cells, C, present.p);
  if (A->pairs_state != 1) {  // a rank whose pair list does not fit tells the others: nobody builds a plan
    const double one = 1.0;
    CX_HIP(hipMemcpyAsync(present.p + size_t(C) * C, &one, sizeof(double), hipMemcpyHostToDevice, st));
  }
  CX_HIP(hipGetLastError());
  CX_TRY(cx_allreduce_device(ctx, present.p, int64_t(C) * C + 1));
  std::vector<double> h(size_t(C) * C + 1);
  CX_TRY(cx_read_back(ctx, h.data(), present.p, h.size() * sizeof(double), st));
  CX_TRY(cx_stream_sync(ctx, st));
  if (h[size_t(C) * C] > 0.0) { P->state = 2; return CX_OK; }
  std::vector<int32_t> u1, u2, local_to_union(size_t(A->num_cells), 0);
  {
    size_t k = 0;  // this rank's cells are sorted the same way (block row, then block column)
    for (int c1 = 0; c1 < C; ++c1)
      for (int c2 = c1; c2 < C; ++c2)
        if (c2 == c1 || h[size_t(c1) * C + c2] > 0.0) {
          if (k < size_t(A->num_cells) && A->h_cell_c1[k] == c1 && A->h_cell_c2[k] == c2) local_to_union[k++] = int32_t(u1.size());
          u1.push_back(c1);
          u2.push_back(c2);
        }
    if (k != size_t(A->num_cells)) {
      cx_set_error("internal: this rank's S cells are not a subset of the union");
      return CX_ERR_INVALID_ARGUMENT;
    }
  }
  std::vector<double>().swap(h);
  P->num_union_cells = int64_t(u1.size());
  CX_TRY(P->d_union_c1.upload(u1, st));
  CX_TRY(P->d_union_c2.upload(u2, st));
  CX_TRY(P->d_local_to_union.upload(local_to_union, st));
  P->built_replicate = P->replicate;
  const int rc = cxsp_plan_from_cells(ctx, C, u1.data(), u2.data(), P->num_union_cells, P, /*distribute=*/!P->replicate);
  P->h_union_c1.swap(u1);
  P->h_union_c2.swap(u2);
  return rc;
}

int cxsp_factor_and_solve_sharded(cx_matrix* A, const double* Df, const double* rhs, double* z, int* d_flag) {
  cx_context* ctx = A->ctx;
  hipStream_t st = ctx->stream;
  cx_sp_plan* P = &A->sp;
  const int C = A->C, n = 9 * C;
  if (n == 0) return CX_OK;
  const size_t count = size_t(P->num_union_cells) * 81;
  CX_TRY(P->d_union_values.alloc(count));
  CX_HIP(hipMemsetAsync(P->d_union_values.p, 0, count * sizeof(double), st));
  if (A->num_cells > 0)
    hipLaunchKernelGGL(k_sp_scatter_cells, dim3(unsigned((A->num_cells * 81 + 255) / 256)), dim3(256), 0, st, (const double*)A->d_S.p,
                       (const int32_t*)P->d_local_to_union.p, A->num_cells, P->d_union_values.p);
  CX_HIP(hipGetLastError());
  const bool distributed = P->level_split >= 0;
  static const bool by_owner = !(std::getenv("CX_SPARSE_CELLS_BY_OWNER") && std::atoi(std::getenv("CX_SPARSE_CELLS_BY_OWNER")) == 0);  // A/B switch
  if (distributed && by_owner && P->rs_chunk_cells > 0) {
    // every tile row has ONE owner: a rank receives the sum of the cells its own rows (rank 0: and the replicated rows)
    // are assembled from, nothing else -- a reduce-scatter over the ranks' cell lists instead of the all-reduce of all cells
    const int64_t chunk = P->rs_chunk_cells, total = chunk * ctx->nranks;
    CX_TRY(P->d_rs_send.alloc(size_t(total) * 81));
    CX_TRY(P->d_rs_recv.alloc(size_t(chunk) * 81));
    hipLaunchKernelGGL(k_sp_pack_cells, dim3(unsigned((total * 81 + 255) / 256)), dim3(256), 0, st, (const double*)P->d_union_values.p,
                       (const int32_t*)P->d_rs_cells.p, total, P->d_rs_send.p);
    CX_HIP(hipGetLastError());
    CX_TRY(cx_reduce_scatter_device(ctx, P->d_rs_send.p, P->d_rs_recv.p, chunk * 81));
    hipLaunchKernelGGL(k_sp_unpack_cells, dim3(unsigned((chunk * 81 + 255) / 256)), dim3(256), 0, st, (const double*)P->d_rs_recv.p,
                       (const int32_t*)P->d_rs_cells.p + int64_t(ctx->rank) * chunk, chunk, P->d_union_values.p);
    CX_HIP(hipGetLastError());
  } else {
    CX_TRY(cx_allreduce_device(ctx, P->d_union_values.p, int64_t(count)));
  }
  CX_TRY(AllocPool(P, st));
  const bool cells_by_owner = distributed && by_owner && P->rs_chunk_cells > 0;
  CX_TRY(WithPool(P, [&](auto* W) -> int {
    using TW = std::remove_pointer_t<decltype(W)>;
    if (cells_by_owner) {
      // every rank assembles the cells of its own range only (the pool starts from zeros) and the right-hand side of the rows
      // it keeps: what the replicated rows' tiles hold over the ranks then sums to the assembled values exactly once
      hipLaunchKernelGGL(k_sp_assemble_values<TW>, dim3(unsigned((P->num_union_cells + 2) / 3)), dim3(3 * 81), 0, st,
                         (const int32_t*)P->d_union_c1.p, (const int32_t*)P->d_union_c2.p, (const double*)P->d_union_values.p, Df,
                         (const int32_t*)P->d_cam_pos.p, (const int32_t*)P->d_row_start.p, (const int32_t*)P->d_row_tiles.p, W,
                         P->num_union_cells, (const int32_t*)P->d_cell_mine.p);
      hipLaunchKernelGGL(k_sp_rhs<TW>, dim3((n + 255) / 256), dim3(256), 0, st, rhs, (const int32_t*)P->d_cam_pos.p,
                         (const int32_t*)P->d_row_start.p, W, C, (const int32_t*)P->d_row_keep.p);
      return CX_OK;
    }
    hipLaunchKernelGGL(k_sp_assemble_values<TW>, dim3(unsigned((P->num_union_cells + 2) / 3)), dim3(3 * 81), 0, st,
                       (const int32_t*)P->d_union_c1.p, (const int32_t*)P->d_union_c2.p, (const double*)P->d_union_values.p, Df,
                       (const int32_t*)P->d_cam_pos.p, (const int32_t*)P->d_row_start.p, (const int32_t*)P->d_row_tiles.p, W,
                       P->num_union_cells);
    hipLaunchKernelGGL(k_sp_rhs<TW>, dim3((n + 255) / 256), dim3(256), 0, st, rhs, (const int32_t*)P->d_cam_pos.p,
                       (const int32_t*)P->d_row_start.p, W, C);
    if (distributed && ctx->rank != 0 && P->num_shared_tiles > 0) {
      // every rank assembled the whole matrix; in the sum over the ranks at the split the assembled values (and right-hand
      // side) of the replicated rows must count once: rank 0 keeps them, the others start those tiles from zero
      const int64_t cnt = P->num_shared_tiles * kTileDoubles;
      hipLaunchKernelGGL(k_sp_zero_tiles<TW>, dim3(unsigned((cnt + 255) / 256)), dim3(256), 0, st, W, (const int32_t*)P->d_shared_tiles.p,
                         P->num_shared_tiles);
    }
    return CX_OK;
  }));
  CX_TRY(cxsp_factor(ctx, P, d_flag));
  Scratch sc;
  CX_TRY(GetScratch(P, &sc));
  if (distributed) CX_HIP(hipMemsetAsync(sc.xp, 0, size_t(P->T) * kTile * sizeof(double), st));  // (rows of other ranks stay zero)
  CX_TRY(BackwardSweep(ctx, P, sc, 0));
  if (distributed) {
    // every rank holds the solution of its own rows (and of the replicated ones): the whole vector is their sum
    const int64_t npad = int64_t(P->T) * kTile;
    hipLaunchKernelGGL(k_sp_mask_rows, dim3(unsigned((npad + 255) / 256)), dim3(256), 0, st, sc.xp, (const int32_t*)P->d_row_keep.p, npad);
    CX_TRY(cx_allreduce_device(ctx, sc.xp, npad));
  }
  hipLaunchKernelGGL(k_sp_unpermute, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)sc.xp, (const int32_t*)P->d_cam_pos.p, z, C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// Host half of the plan on its own (no device): the CPU test suite checks the layout, the symbolic fill and the level
// schedule through this entry point.
extern "C" int cx_sparse_cholesky_plan_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                            int32_t* camera_first_row, int32_t* num_tile_rows, int32_t* num_levels,
                                            int64_t* num_tiles, int64_t* num_tile_pair_updates, int32_t* tile_row_level,
                                            int32_t* tile_row_start, int32_t capacity_rows, int32_t* tile_cols,
                                            int64_t capacity_tiles) {
  CX_CHECK_ARG(num_cameras > 0 && cell_row && cell_col && num_cells >= num_cameras && num_tile_rows && num_levels && num_tiles &&
               num_tile_pair_updates);
  for (int64_t k = 0; k < num_cells; ++k)
    CX_CHECK_ARG(cell_row[k] >= 0 && cell_row[k] <= cell_col[k] && cell_col[k] < num_cameras);
  HostPlan H;
  BuildHostPlan(num_cameras, cell_row, cell_col, num_cells, &H);
  *num_tile_rows = H.T;
  *num_levels = H.L;
  *num_tiles = H.num_tiles;
  *num_tile_pair_updates = int64_t(H.src_a.size());
  if (!H.fits) {
    cx_set_error("the tile-sparse Cholesky of this structure would need more than 160 GB");
    return CX_ERR_UNSUPPORTED;
  }
  if (camera_first_row) std::copy(H.layout.cam_row.begin(), H.layout.cam_row.end(), camera_first_row);
  if (tile_row_level && capacity_rows >= H.T) std::copy(H.height.begin(), H.height.end(), tile_row_level);
  if (tile_row_start && capacity_rows >= H.T) std::copy(H.row_start.begin(), H.row_start.end(), tile_row_start);
  if (tile_cols && capacity_tiles >= H.num_tiles) std::copy(H.row_tiles.begin(), H.row_tiles.end(), tile_cols);
  return CX_OK;
}

// How the distributed factorisation of a sharded SPARSE_SCHUR divides the tile rows of this structure over nranks ranks
// (no device): see cxschur.h
extern "C" int cx_sparse_cholesky_distribution_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                                    int32_t nranks, int64_t* updates_per_rank, int64_t* updates_replicated,
                                                    int64_t* tiles_replicated, int32_t* tile_row_owner, int32_t capacity_rows) {
  CX_CHECK_ARG(num_cameras > 0 && cell_row && cell_col && num_cells >= num_cameras && nranks >= 1 && nranks <= 64 && updates_per_rank &&
               updates_replicated && tiles_replicated);
  for (int64_t k = 0; k < num_cells; ++k)
    CX_CHECK_ARG(cell_row[k] >= 0 && cell_row[k] <= cell_col[k] && cell_col[k] < num_cameras);
  HostPlan H;
  BuildHostPlan(num_cameras, cell_row, cell_col, num_cells, &H, 0, nranks);
  if (!H.fits) {
    cx_set_error("the tile-sparse Cholesky of this structure would need more than 160 GB");
    return CX_ERR_UNSUPPORTED;
  }
  if (nranks == 1) {  // nothing is distributed: everything is rank 0's
    updates_per_rank[0] = int64_t(H.src_a.size());
    *updates_replicated = 0;
    *tiles_replicated = 0;
    if (tile_row_owner && capacity_rows >= H.T) std::fill(tile_row_owner, tile_row_owner + H.T, 0);
    return H.T;
  }
  std::copy(H.work_per_rank.begin(), H.work_per_rank.end(), updates_per_rank);
  *updates_replicated = H.work_shared;
  int64_t tiles = 0;
  for (int I = 0; I < H.T; ++I)
    if (H.owner[size_t(I)] < 0) tiles += H.row_start[size_t(I) + 1] - H.row_start[size_t(I)];
  *tiles_replicated = tiles;
  if (tile_row_owner && capacity_rows >= H.T) std::copy(H.owner.begin(), H.owner.end(), tile_row_owner);
  return H.T;
}


// The schedule of the tile-pair updates under a window (see BuildHostPlan), checked from its definition -- no device: every
// product F(I, Ja)' F(I, Jb) of the symbolic factor appears exactly once, runs after its source row I is factored and before
// its target's row Ja is, and (distributed plan of `rank` of `nranks`) products of the rank's own rows all run before the
// exchange of the replicated tiles.  Returns the number of tile rows, or a negative error.
extern "C" int cx_sparse_cholesky_schedule_host(int32_t num_cameras, const int32_t* cell_row, const int32_t* cell_col, int64_t num_cells,
                                                int32_t nranks, int32_t rank, int32_t window, int64_t* num_products, int64_t* num_chains,
                                                int32_t* longest_chain, int64_t* violations) {
  CX_CHECK_ARG(num_cameras > 0 && cell_row && cell_col && num_cells >= num_cameras && nranks >= 1 && nranks <= 64 && rank >= 0 &&
               rank < nranks && window >= 1 && num_products && num_chains && longest_chain && violations);
  for (int64_t k = 0; k < num_cells; ++k)
    CX_CHECK_ARG(cell_row[k] >= 0 && cell_row[k] <= cell_col[k] && cell_col[k] < num_cameras);
  HostPlan H;
  H.window = window;
  H.keep_schedule = true;
  BuildHostPlan(num_cameras, cell_row, cell_col, num_cells, &H, nranks > 1 ? rank : -1, nranks);
  if (!H.fits) {
    cx_set_error("the tile-sparse Cholesky of this structure would need more than 160 GB");
    return CX_ERR_UNSUPPORTED;
  }
  const int T = H.T;
  int64_t bad = 0, expected = 0;
  for (int I = 0; I < T; ++I)
    if (H.row_level[size_t(I)] >= 0) {
      const int64_t m = H.row_start[size_t(I) + 1] - H.row_start[size_t(I)] - 1;  // tiles right of the diagonal, right-hand side included
      expected += m * (m + 1) / 2 - 1 + (m == 0 ? 1 : 0);                           // pairs (Ja <= Jb), Ja a real tile row: all but (rhs, rhs)
    }
  std::vector<int32_t> row_of_slot(static_cast<size_t>(H.num_tiles));
  for (int J = 0; J < T; ++J)
    for (int32_t q = H.row_start[size_t(J)]; q < H.row_start[size_t(J) + 1]; ++q) row_of_slot[size_t(q)] = J;
  std::vector<std::pair<int32_t, int32_t>> seen;  // (qa, qb) of every product: each once
  seen.reserve(H.src_a.size());
  int32_t longest = 0;
  for (int l = 0; l < H.L; ++l)
    for (int32_t t = H.ltb[size_t(l)]; t < H.ltb[size_t(l) + 1]; ++t) {
      const int32_t tq = H.tgt_pool[size_t(t)], Ja = row_of_slot[size_t(tq)], Jb = H.row_tiles[size_t(tq)];
      longest = std::max(longest, H.src_begin[size_t(t) + 1] - H.src_begin[size_t(t)]);
      for (int32_t i = H.src_begin[size_t(t)]; i < H.src_begin[size_t(t) + 1]; ++i) {
        const int32_t I = H.src_row[size_t(i)], qa = H.src_a[size_t(i)], qb = H.src_b[size_t(i)];
        const int32_t lI = H.row_level[size_t(I)], lJ = H.row_level[size_t(Ja)];
        bool ok = lI >= 0 && lJ >= 0 && lI <= l && l < lJ;                                      // after I is factored, before Ja is
        ok = ok && row_of_slot[size_t(qa)] == I && row_of_slot[size_t(qb)] == I;                // both operands are tiles of row I
        ok = ok && H.row_tiles[size_t(qa)] == Ja && H.row_tiles[size_t(qb)] == Jb && qa <= qb;   // ... the ones this target needs
        ok = ok && (H.L_split < 0 || lI >= H.L_split || l < H.L_split);                          // own rows: before the exchange
        ok = ok && (i == H.src_begin[size_t(t)] || H.src_row[size_t(i) - 1] < I);               // ascending source rows in a chain
        if (!ok) ++bad;
        seen.emplace_back(qa, qb);
      }
    }
  std::sort(seen.begin(), seen.end());
  for (size_t i = 1; i < seen.size(); ++i)
    if (seen[i] == seen[i - 1]) ++bad;
  if (int64_t(seen.size()) != expected) ++bad;
  *num_products = int64_t(seen.size());
  *num_chains = int64_t(H.tgt_pool.size());
  *longest_chain = longest;
  *violations = bad;
  return T;
}
