// Tile-sparse Cholesky of the reduced camera matrix: the reduced solve of SPARSE_SCHUR when 9C is too
// large for a dense S (SparseSchurComplementSolver::SolveReducedLinearSystem, schur_complement_solver.cc:
// 292-335, which hands a CRS copy of S to CHOLMOD, suitesparse.cc:397-469, after a fill-reducing ordering
// of the cameras, reorder_program.cc:341-382).
//
// CHOLMOD is third-party code that is not part of the reference tree: the factorisation here is this
// library's own and parity is stated on the solution (it must equal the dense solve), not on the ordering
// ("parity unpinned" for CAMD / AMD).
//
//   ordering   reverse Cuthill-McKee on the camera graph of S, optionally followed by minimum degree on groups
//              of 64 cameras -- whichever leaves fewer tiles after fill (host, once per structure)
//   structure  S in 64x64 tiles (upper triangle), symbolic fill at tile level; every tile row ends with one
//              extra tile that carries the right-hand side as its column 0, so the forward substitution
//              is part of the factorisation (as in the dense solver)
//   assembly   the gather assembly of cx_schur.hip (k_pair_items) scattered into the tile pool at the
//              permuted positions
//   numeric    the dense solver's step kernel with tile indirection: one launch per 32-column block step,
//              a workgroup per pair of non-zero tiles of the step's tile row, panel solve and trailing
//              update on fp64 MFMA, look-ahead factorisation of the next diagonal block
//   solve      one launch per tile row (64 rows) for the backward substitution, with the kept inverses of the
//              diagonal blocks, then the inverse permutation
// The factorisation is bound by the ~25 us a dependent launch + diagonal block cost on this part
// (n / 32 steps), not by flops or bytes.
#include <algorithm>
#include <cstdlib>
#include <queue>

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

// (a, b), a <= b < m, number t in row-major order of the upper triangle
__device__ __forceinline__ void pair_decode(int t, int m, int& a, int& b) {
  int row = int((2.0 * m + 1.0 - sqrt((2.0 * m + 1.0) * (2.0 * m + 1.0) - 8.0 * t)) * 0.5);
  row = max(0, min(m - 1, row));
  while (row > 0 && row * m - row * (row - 1) / 2 > t) --row;
  while ((row + 1) * m - (row + 1) * row / 2 <= t) ++row;
  a = row;
  b = row + (t - (row * m - row * (row - 1) / 2));
}

// S cells (gather assembly of cx_schur.hip) -> tile pool at the permuted positions; 81 threads per cell
__global__ __launch_bounds__(3 * 81) void k_sp_assemble(const int32_t* __restrict__ cell_c1, const int32_t* __restrict__ cell_c2,
                                                        const int32_t* __restrict__ cell_item_start,
                                                        const double* __restrict__ item_partial, const double* __restrict__ diag,
                                                        const double* __restrict__ Df, const int32_t* __restrict__ cam_pos,
                                                        const int32_t* __restrict__ row_start, const int32_t* __restrict__ row_tiles,
                                                        double* __restrict__ W, int64_t num_cells) {
  const int64_t cell = int64_t(blockIdx.x) * 3 + threadIdx.x / 81;
  if (cell >= num_cells) return;
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
  int row = 9 * cam_pos[c1] + a, col = 9 * cam_pos[c2] + c;
  if (c1 != c2 && row > col) { const int t = row; row = col; col = t; }  // the cell lands transposed
  if (row > col) return;                                                 // lower half of a diagonal cell
  const int I = row >> 6, J = col >> 6;
  const int idx = tile_find(row_tiles, row_start[I], row_start[I + 1], J);
  W[size_t(idx) * kTileDoubles + (row & 63) * kTile + (col & 63)] = v;
}

// right-hand side (camera order) -> column 0 of every tile row's last tile
__global__ void k_sp_rhs(const double* __restrict__ rhs, const int32_t* __restrict__ cam_pos, const int32_t* __restrict__ row_start,
                         double* __restrict__ W, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, a = i - 9 * c;
  const int row = 9 * cam_pos[c] + a;
  const int I = row >> 6;
  W[size_t(row_start[I + 1] - 1) * kTileDoubles + (row & 63) * kTile] = rhs[i];
}

__global__ __launch_bounds__(64) void k_sp_first(const double* __restrict__ W, double* __restrict__ F, int n, double* __restrict__ uinv,
                                                 int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  cxchol::potrf_inverse_block(W, kTile, F, kTile, min(NB, n), uinv, not_pd, lds);
}

// One block step k0 (see the file header).  Tile row I = k0 / 64, half = upper / lower 32 rows of it.  The
// step's column tiles are the row's list (without the diagonal tile in a lower-half step); block t owns the
// pair (a <= b) number t of that list: target tile (L[a], L[b]), panel pieces from tiles (I, L[a]), (I, L[b]).
__global__ __launch_bounds__(256) void k_sp_step(double* __restrict__ W, double* __restrict__ F, const int32_t* __restrict__ row_start,
                                                 const int32_t* __restrict__ row_tiles, int n, int T, double* __restrict__ uinv,
                                                 int k0, int* __restrict__ not_pd) {
  __shared__ double lds[cxchol::kPotrfLds];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int kb = min(NB, n - k0), rest = k0 + kb;
  const int I = k0 >> 6, half = (k0 >> 5) & 1;
  const double* ui = uinv + size_t(k0 / NB) * NB * NB;  // all block inverses are kept (backward substitution)
  const int rs = row_start[I] + half;  // a lower-half step has left the diagonal tile behind
  const int m = row_start[I + 1] - rs;
  int a, b;
  pair_decode(blockIdx.x, m, a, b);
  const int Ji = row_tiles[rs + a], Jj = row_tiles[rs + b];
  const int qi = wave >> 1, qj = wave & 1;
  const size_t prow = size_t(32 * half) * kTile;  // the panel's rows inside the tiles of row I
  // X_j = U_kk^-T W(k, columns of quadrant qj of tile Jj); the rhs tile T has one column
  const int ncols_j = (Jj < T) ? min(32, n - (kTile * Jj + 32 * qj)) : (qj == 0 ? 1 : 0);
  double4_t Xj[2][2];
  cxchol::panel_x(W + size_t(rs + b) * kTileDoubles + prow + 32 * qj, kTile, ui, kb, ncols_j, Xj);
  if (a == 0 && qi == 0) {
    // rows k0.. of the factor; columns left of `rest` belong to U_kk itself
    double* Frow = F + size_t(rs + b) * kTileDoubles + prow + 32 * qj;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int mr = 16 * mt + lk + 4 * g;
          const int cl = 16 * nt + li;
          const bool ok = (Jj < T) ? (kTile * Jj + 32 * qj + cl >= rest && cl < ncols_j) : (cl < ncols_j);
          if (mr < kb && ok) Frow[size_t(mr) * kTile + cl] = Xj[mt][nt][g];
        }
  }
  if (Ji < T) {
    double4_t Xi[2][2];
    if (Ji == Jj && qi == qj) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) Xi[mt][nt] = Xj[mt][nt];
    } else {
      const int ncols_i = min(32, n - (kTile * Ji + 32 * qi));
      cxchol::panel_x(W + size_t(rs + a) * kTileDoubles + prow + 32 * qi, kTile, ui, kb, ncols_i, Xi);
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
    // target tile (Ji, Jj): present by construction of the symbolic fill
    const int idx = tile_find(row_tiles, row_start[Ji], row_start[Ji + 1], Jj);
    double* Wt = W + size_t(idx) * kTileDoubles;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int il = 32 * qi + x * 16 + lk + 4 * g;  // local row / column inside the tile
          const int jl = 32 * qj + y * 16 + li;
          const int i = kTile * Ji + il;
          const bool col_ok = (Jj < T) ? (kTile * Jj + jl < n && kTile * Jj + jl >= i) : (jl == 0);
          if (i >= rest && i < n && col_ok) Wt[il * kTile + jl] -= acc[x][y][g];
        }
  }
  if (blockIdx.x == 0 && rest < n) {
    // look-ahead: the next diagonal block lies in the tile this workgroup has just updated, or in a
    // tile this step does not touch
    // owner of the next diagonal block: an upper-half step updates it as quadrant (1, 1) of the diagonal tile
    // (wavefront 3), a lower-half step as quadrant (0, 0) of the next diagonal tile (wavefront 0) or not at all --
    // either way one wavefront's own program order suffices (see k_chol_step)
    if (wave == (half ? 0 : 3)) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const int In = rest >> 6, hn = (rest >> 5) & 1;
      const size_t off = size_t(row_start[In]) * kTileDoubles + size_t(32 * hn) * kTile + 32 * hn;
      cxchol::potrf_inverse_block(W + off, kTile, F + off, kTile, min(NB, n - rest), uinv + size_t(rest / NB) * NB * NB, not_pd, lds);
    }
  }
}

// y (dense, permuted order) = column 0 of every tile row's last tile of the factor (U^-T rhs)
__global__ void k_sp_gather_y(const double* __restrict__ F, const int32_t* __restrict__ row_start, double* __restrict__ y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = F[size_t(row_start[(i >> 6) + 1] - 1) * kTileDoubles + size_t(i & 63) * kTile];
}

// out[m] = sum_c M[m][c] v[c] for a 32 x 32 block, 8 threads per row
__device__ __forceinline__ void sp_gemv32(const double* __restrict__ M, int ldm, const double* __restrict__ v, double* __restrict__ out,
                                          int rows_valid, int cols_valid) {
  const int t = threadIdx.x, m = t >> 3, part = t & 7;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * part + q;
    s += (m < rows_valid && c < cols_valid) ? M[size_t(m) * ldm + c] * v[c] : 0.0;
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (part == 0) out[m] = s;
}

// Backward substitution, one tile row (64 rows) per launch, as the dense solver's k_trsv_bwd64: every workgroup solves
// the tile row's 64 x 64 triangular system with the stored inverses of its two diagonal blocks (x2 = U22^-1 y2,
// x1 = U11^-1 (y1 - U12 x2)), then workgroup w subtracts (tile w of tile column I) x from that tile's 64 rows.
__global__ __launch_bounds__(256) void k_sp_bwd64(const double* __restrict__ F, const int32_t* __restrict__ row_start,
                                                  const int32_t* __restrict__ col_start, const int32_t* __restrict__ col_pool,
                                                  const int32_t* __restrict__ col_row, int n, int I, const double* __restrict__ uinv,
                                                  double* __restrict__ y, double* __restrict__ x) {
  __shared__ double y1[NB], y2[NB], x1[NB], x2[NB], tmp[NB];
  const int t = threadIdx.x;
  const int k0 = kTile * I;
  const int kb1 = min(NB, n - k0), kb2 = max(0, min(NB, n - k0 - NB));
  if (t < NB) {
    y1[t] = t < kb1 ? y[k0 + t] : 0.0;
    y2[t] = t < kb2 ? y[k0 + NB + t] : 0.0;
  }
  __syncthreads();
  const double* __restrict__ D = F + size_t(row_start[I]) * kTileDoubles;  // the diagonal tile
  const double* __restrict__ ui1 = uinv + size_t(2 * I) * NB * NB;
  if (kb2 > 0) {
    sp_gemv32(ui1 + NB * NB, NB, y2, x2, NB, NB);
    __syncthreads();
    sp_gemv32(D + NB, kTile, x2, tmp, kb1, kb2);  // U12 x2
    __syncthreads();
    if (t < NB) y1[t] -= tmp[t];
  } else if (t < NB) {
    x2[t] = 0.0;
  }
  __syncthreads();
  sp_gemv32(ui1, NB, y1, x1, NB, NB);
  __syncthreads();
  if (blockIdx.x == 0 && t < 64) {
    const double v = t < NB ? x1[t] : x2[t - NB];
    if (t < kb1 + kb2) x[k0 + t] = v;  // not into y: other workgroups still read the tile row's y
  }
  const int p = col_start[I] + blockIdx.x;
  if (p >= col_start[I + 1]) return;
  const int Ii = col_row[p];
  if (Ii >= I) return;  // the diagonal tile's part (U12) is inside the group solve
  const int il = t >> 2, part = t & 3;
  const double* __restrict__ row = F + size_t(col_pool[p]) * kTileDoubles + size_t(il) * kTile + 16 * part;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = 16 * part + q;
    const double xv = c < NB ? x1[c] : x2[c - NB];
    s += c < kb1 + kb2 ? row[q] * xv : 0.0;
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (part == 0) y[kTile * Ii + il] -= s;
}

__global__ void k_sp_unpermute(const double* __restrict__ xp, const int32_t* __restrict__ cam_pos, double* __restrict__ x, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int c = i / 9, a = i - 9 * c;
  x[i] = xp[9 * cam_pos[c] + a];
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

// tile-level structure of the permuted S (upper) and its symbolic fill (eliminating tile row k connects every pair
// of its later column tiles); rows as sorted lists, each closed by the right-hand-side tile T
void TileStructure(const cx_matrix* A, const std::vector<int32_t>& pos, int T, std::vector<int32_t>* row_start,
                   std::vector<int32_t>* row_tiles) {
  std::vector<std::vector<char>> nz(static_cast<size_t>(T), std::vector<char>(static_cast<size_t>(T), 0));
  auto mark = [&](int p1, int p2) {  // camera positions p1 <= p2
    for (int rt = (9 * p1) >> 6; rt <= (9 * p1 + 8) >> 6; ++rt)
      for (int ct = (9 * p2) >> 6; ct <= (9 * p2 + 8) >> 6; ++ct) nz[size_t(std::min(rt, ct))][size_t(std::max(rt, ct))] = 1;
  };
  for (int64_t k = 0; k < A->num_cells; ++k) {
    const int p1 = pos[size_t(A->h_cell_c1[size_t(k)])], p2 = pos[size_t(A->h_cell_c2[size_t(k)])];
    mark(std::min(p1, p2), std::max(p1, p2));
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

int cxsp_build_plan(cx_matrix* A) {
  if (A->sp_state != 0) return CX_OK;
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state != 1) { A->sp_state = 2; return CX_OK; }
  const int C = A->C;
  const int n = 9 * C;
  const int T = (n + kTile - 1) / kTile;
  std::vector<std::vector<int32_t>> adj(static_cast<size_t>(C));
  for (int64_t k = 0; k < A->num_cells; ++k) {
    const int c1 = A->h_cell_c1[size_t(k)], c2 = A->h_cell_c2[size_t(k)];
    if (c1 != c2) { adj[size_t(c1)].push_back(c2); adj[size_t(c2)].push_back(c1); }
  }
  // two candidate orderings, the one with fewer tiles after fill wins
  std::vector<int32_t> pos = ReverseCuthillMcKee(C, adj);
  std::vector<int32_t> row_start, row_tiles;
  TileStructure(A, pos, T, &row_start, &row_tiles);
  {
    std::vector<int32_t> pos2 = GroupMinimumDegree(C, adj, pos), rs2, rt2;
    TileStructure(A, pos2, T, &rs2, &rt2);
    if (rt2.size() < row_tiles.size()) {
      pos.swap(pos2);
      row_start.swap(rs2);
      row_tiles.swap(rt2);
    }
  }
  const int64_t num_tiles = int64_t(row_tiles.size());
  // transposed index: the tiles (Ii <= I, I) of tile column I, ascending Ii, with their pool positions
  std::vector<int32_t> col_start(size_t(T) + 1, 0), col_pool, col_row;
  {
    std::vector<std::vector<std::pair<int32_t, int32_t>>> cols(static_cast<size_t>(T));
    for (int I = 0; I < T; ++I)
      for (int32_t q = row_start[size_t(I)]; q < (I + 1 < T ? row_start[size_t(I) + 1] : int32_t(row_tiles.size())); ++q)
        if (row_tiles[size_t(q)] < T) cols[size_t(row_tiles[size_t(q)])].push_back({I, q});
    for (int J = 0; J < T; ++J) {
      col_start[size_t(J)] = int32_t(col_pool.size());
      for (auto& e : cols[size_t(J)]) { col_row.push_back(e.first); col_pool.push_back(e.second); }
    }
    col_start[size_t(T)] = int32_t(col_pool.size());
  }
  // two pools (working copy and factor): refuse structures that would not fit comfortably
  if (double(num_tiles) * kTileDoubles * 8.0 * 2.0 > 160e9) { A->sp_state = 2; return CX_OK; }
  hipStream_t st = A->ctx->stream;
  CX_TRY(A->d_sp_cam_pos.upload(pos, st));
  CX_TRY(A->d_sp_row_start.upload(row_start, st));
  CX_TRY(A->d_sp_row_tiles.upload(row_tiles, st));
  CX_TRY(A->d_sp_col_start.upload(col_start, st));
  CX_TRY(A->d_sp_col_pool.upload(col_pool, st));
  CX_TRY(A->d_sp_col_row.upload(col_row, st));
  A->h_sp_col_start = col_start;
  A->h_sp_row_start = row_start;
  A->sp_num_tiles = num_tiles;
  A->sp_T = T;
  A->sp_state = 1;
  if (std::getenv("CX_SPARSE_CHOLESKY_VERBOSE"))
    std::fprintf(stderr, "[cxschur] tile-sparse Cholesky plan: %d cameras, %d tile rows, %lld tiles (%.2f GB per pool, dense would be %.2f GB)\n", C, T,
                 (long long)num_tiles, double(num_tiles) * kTileDoubles * 8e-9, double(n) * n * 8e-9);
  return CX_OK;
}

// S z = rhs for the explicit S of the matrix, assembled straight into the tile pool (A's gather assembly must
// have run: cxs_assemble_pair_items); z and rhs in camera order.
int cxsp_factor_and_solve(cx_matrix* A, const double* Df, const double* rhs, double* z, int* d_flag) {
  cx_context* ctx = A->ctx;
  hipStream_t st = ctx->stream;
  const int C = A->C, n = 9 * C, T = A->sp_T;
  if (n == 0) return CX_OK;
  const size_t pool = size_t(A->sp_num_tiles) * kTileDoubles;
  CX_TRY(A->d_sp_W.alloc(pool));
  CX_TRY(A->d_sp_F.alloc(pool));
  CX_TRY(A->d_sp_x.alloc(2 * size_t(n) + size_t((n + NB - 1) / NB) * NB * NB));
  double* W = A->d_sp_W.p;
  double* F = A->d_sp_F.p;
  double* xp = A->d_sp_x.p;
  double* yv = xp + n;
  double* uinv = yv + n;
  CX_HIP(hipMemsetAsync(W, 0, pool * sizeof(double), st));
  CX_HIP(hipMemsetAsync(F, 0, pool * sizeof(double), st));
  if (A->num_cells > 0)
    hipLaunchKernelGGL(k_sp_assemble, dim3(unsigned((A->num_cells + 2) / 3)), dim3(3 * 81), 0, st, (const int32_t*)A->d_cell_c1.p,
                       (const int32_t*)A->d_cell_c2.p, (const int32_t*)A->d_cell_item_start.p, (const double*)A->d_item_partial.p,
                       (const double*)A->d_elim_diag.p, Df, (const int32_t*)A->d_sp_cam_pos.p, (const int32_t*)A->d_sp_row_start.p,
                       (const int32_t*)A->d_sp_row_tiles.p, W, A->num_cells);
  hipLaunchKernelGGL(k_sp_rhs, dim3((n + 255) / 256), dim3(256), 0, st, rhs, (const int32_t*)A->d_sp_cam_pos.p,
                     (const int32_t*)A->d_sp_row_start.p, W, C);
  hipLaunchKernelGGL(k_sp_first, dim3(1), dim3(64), 0, st, (const double*)W, F, n, uinv, d_flag);
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int I = k0 >> 6, half = (k0 >> 5) & 1;
    const int m = A->h_sp_row_start[size_t(I) + 1] - A->h_sp_row_start[size_t(I)] - half;
    hipLaunchKernelGGL(k_sp_step, dim3(unsigned(m * (m + 1) / 2)), dim3(256), 0, st, W, F, (const int32_t*)A->d_sp_row_start.p,
                       (const int32_t*)A->d_sp_row_tiles.p, n, T, uinv, k0, d_flag);
  }
  CX_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_sp_gather_y, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)F, (const int32_t*)A->d_sp_row_start.p, yv, n);
  for (int I = T - 1; I >= 0; --I) {
    const int tiles = A->h_sp_col_start[size_t(I) + 1] - A->h_sp_col_start[size_t(I)];
    hipLaunchKernelGGL(k_sp_bwd64, dim3(unsigned(std::max(1, tiles))), dim3(256), 0, st, (const double*)F,
                       (const int32_t*)A->d_sp_row_start.p, (const int32_t*)A->d_sp_col_start.p, (const int32_t*)A->d_sp_col_pool.p,
                       (const int32_t*)A->d_sp_col_row.p, n, I, (const double*)uinv, yv, xp);
  }
  hipLaunchKernelGGL(k_sp_unpermute, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)xp, (const int32_t*)A->d_sp_cam_pos.p, z, C);
  CX_HIP(hipGetLastError());
  return CX_OK;
}
