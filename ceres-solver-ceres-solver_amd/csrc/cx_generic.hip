// Dynamic-block-size path: any CompressedRowBlockStructure with blocks of up to
// kMaxBlock rows/columns.  This is what the Eigen::Dynamic instantiations are in the
// reference (schur_eliminator.cc:55-143, partitioned_matrix_view.cc) -- a correctness
// path so that every matrix the reference accepts, in particular its own unit-test
// fixtures (linear_least_squares_problems.cc), runs through the same C ABI on the GPU.
// One thread per row block or per chunk; transposed products and the S scatter use fp64
// atomics.  The hot bundle-adjustment structure never comes here (cx_matrix.hip Detect239).
#include <algorithm>
#include <numeric>

#include "cx_kernels.h"
#include "cx_schur.h"
#include "cx_solver_internal.h"

namespace {

constexpr int kMaxBlock = 16;
enum Sel : int { SEL_ALL = 0, SEL_E = 1, SEL_F = 2 };

static int grid_for(int64_t n, int block) { return int((n + block - 1) / block); }

struct GStruct {  // device view of the flat structure
  const cx_block* rows;
  const cx_block* cols;
  const int32_t* rcb;
  const cx_cell* cells;
  int R, nrows_e, nelim;
};

__device__ __forceinline__ void cell_range(const GStruct& g, int r, int sel, int& b, int& e) {
  b = g.rcb[r];
  e = g.rcb[r + 1];
  const bool has_e = r < g.nrows_e;
  if (sel == SEL_E) e = has_e ? min(e, b + 1) : b;
  else if (sel == SEL_F && has_e) b = min(e, b + 1);
}

// y += A_sel x  (PartitionedMatrixView::RightMultiplyAndAccumulate{E,F}, partitioned_matrix_view_impl.h:92-170)
__global__ void kg_mult(GStruct g, const double* __restrict__ values, const double* __restrict__ x,
                        double* __restrict__ y, int sel, int col_off) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= g.R) return;
  int cb, ce;
  cell_range(g, r, sel, cb, ce);
  const int rs = g.rows[r].size, rp = g.rows[r].position;
  for (int c = cb; c < ce; ++c) {
    const cx_cell cell = g.cells[c];
    const int cs = g.cols[cell.block_id].size, cp = g.cols[cell.block_id].position - col_off;
    const double* m = values + cell.position;
    for (int i = 0; i < rs; ++i) {
      double s = 0.0;
      for (int j = 0; j < cs; ++j) s += m[i * cs + j] * x[cp + j];
      y[rp + i] += s;
    }
  }
}

// y += A_sel' x  (LeftMultiplyAndAccumulate{E,F}, :172-320), atomics on y
__global__ void kg_mult_t(GStruct g, const double* __restrict__ values, const double* __restrict__ x,
                          double* __restrict__ y, int sel, int col_off) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= g.R) return;
  int cb, ce;
  cell_range(g, r, sel, cb, ce);
  const int rs = g.rows[r].size, rp = g.rows[r].position;
  for (int c = cb; c < ce; ++c) {
    const cx_cell cell = g.cells[c];
    const int cs = g.cols[cell.block_id].size, cp = g.cols[cell.block_id].position - col_off;
    const double* m = values + cell.position;
    for (int j = 0; j < cs; ++j) {
      double s = 0.0;
      for (int i = 0; i < rs; ++i) s += m[i * cs + j] * x[rp + i];
      atomicAdd(&y[cp + j], s);
    }
  }
}

// blocks[off[c] - off0 ..] += cell' cell   (UpdateBlockDiagonal{EtE,FtF} :420-658,
// BlockSparseJacobiPreconditioner::UpdateImpl block_jacobi_preconditioner.cc:59-115)
__global__ void kg_block_diag(GStruct g, const double* __restrict__ values, const int64_t* __restrict__ blk_off,
                              int64_t off0, double* __restrict__ blocks, int sel) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= g.R) return;
  int cb, ce;
  cell_range(g, r, sel, cb, ce);
  const int rs = g.rows[r].size;
  for (int c = cb; c < ce; ++c) {
    const cx_cell cell = g.cells[c];
    const int cs = g.cols[cell.block_id].size;
    const double* m = values + cell.position;
    double* out = blocks + (blk_off[cell.block_id] - off0);
    for (int a = 0; a < cs; ++a)
      for (int bcol = 0; bcol < cs; ++bcol) {
        double s = 0.0;
        for (int i = 0; i < rs; ++i) s += m[i * cs + a] * m[i * cs + bcol];
        atomicAdd(&out[a * cs + bcol], s);
      }
  }
}

// blocks[i] <- (blocks[i] + diag(D^2))^-1 via LLT on the upper triangle
// (AddDiagonalAndInvert implicit_schur_complement.cc:179-204; BlockRandomAccessDiagonalMatrix::Invert)
__global__ void kg_blockdiag_invert(const cx_block* __restrict__ cols, const int64_t* __restrict__ blk_off,
                                    int64_t off0, int first, int count, const double* __restrict__ D,
                                    double* __restrict__ blocks, int* __restrict__ not_pd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int s = cols[first + i].size, pos = cols[first + i].position;
  double* A = blocks + (blk_off[first + i] - off0);
  double U[kMaxBlock * kMaxBlock];
  bool ok = true;
  for (int j = 0; j < s; ++j) {
    for (int r = 0; r <= j; ++r) {
      double v = A[r * s + j];
      if (r == j && D) v += D[pos + j] * D[pos + j];
      for (int k = 0; k < r; ++k) v -= U[k * s + r] * U[k * s + j];
      if (r == j) {
        if (!(v > 0.0)) ok = false;
        U[r * s + r] = sqrt(v);
      } else {
        U[r * s + j] = v / U[r * s + r];
      }
    }
  }
  if (!ok) *not_pd = 1;
  double y[kMaxBlock];
  for (int col = 0; col < s; ++col) {
    for (int r = 0; r < s; ++r) {
      double v = (r == col) ? 1.0 : 0.0;
      for (int k = 0; k < r; ++k) v -= U[k * s + r] * y[k];
      y[r] = v / U[r * s + r];
    }
    for (int r = s - 1; r >= 0; --r) {
      double v = y[r];
      for (int k = r + 1; k < s; ++k) v -= U[r * s + k] * A[k * s + col];
      A[r * s + col] = v / U[r * s + r];
    }
  }
}

// y[block] = blocks[block] x[block]
__global__ void kg_blockdiag_mult(const cx_block* __restrict__ cols, const int64_t* __restrict__ blk_off, int64_t off0,
                                  int first, int count, int pos0, const double* __restrict__ blocks,
                                  const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int s = cols[first + i].size, p = cols[first + i].position - pos0;
  const double* B = blocks + (blk_off[first + i] - off0);
  for (int r = 0; r < s; ++r) {
    double v = 0.0;
    for (int k = 0; k < s; ++k) v += B[r * s + k] * x[p + k];
    y[p + r] = v;
  }
}

// copy the diagonal blocks of a dense n x n matrix into packed block storage
__global__ void kg_extract_diag_blocks(const cx_block* __restrict__ cols, const int64_t* __restrict__ blk_off,
                                       int64_t off0, int first, int count, int pos0, const double* __restrict__ lhs,
                                       int64_t n, double* __restrict__ blocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int s = cols[first + i].size, p = cols[first + i].position - pos0;
  double* B = blocks + (blk_off[first + i] - off0);
  for (int r = 0; r < s; ++r)
    for (int c = 0; c < s; ++c) B[r * s + c] = lhs[(p + r) * n + p + c];
}

// S += F_i' F_j over the f cells of each row, rhs += F'b for rows without an e block
// (EBlockRowOuterProduct :665-714, NoEBlockRowsUpdate :567-659)
__global__ void kg_row_outer(GStruct g, const double* __restrict__ values, const double* __restrict__ b,
                             double* __restrict__ lhs, int64_t n, double* __restrict__ rhs, int f_pos0) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= g.R) return;
  int cb, ce;
  cell_range(g, r, SEL_F, cb, ce);
  const int rs = g.rows[r].size, rp = g.rows[r].position;
  for (int i = cb; i < ce; ++i) {
    const cx_cell ci = g.cells[i];
    const int s1 = g.cols[ci.block_id].size, p1 = g.cols[ci.block_id].position - f_pos0;
    const double* m1 = values + ci.position;
    for (int j = i; j < ce; ++j) {
      const cx_cell cj = g.cells[j];
      const int s2 = g.cols[cj.block_id].size, p2 = g.cols[cj.block_id].position - f_pos0;
      const double* m2 = values + cj.position;
      for (int a = 0; a < s1; ++a)
        for (int c = 0; c < s2; ++c) {
          double s = 0.0;
          for (int k = 0; k < rs; ++k) s += m1[k * s1 + a] * m2[k * s2 + c];
          atomicAdd(&lhs[(p1 + a) * n + p2 + c], s);
        }
    }
    if (rhs && r >= g.nrows_e) {
      for (int a = 0; a < s1; ++a) {
        double s = 0.0;
        for (int k = 0; k < rs; ++k) s += m1[k * s1 + a] * b[rp + k];
        atomicAdd(&rhs[p1 + a], s);
      }
    }
  }
}

// One thread per chunk: UpdateRhs (:379-420) and ChunkOuterProduct (:512-561).
// ete_inv holds (E'E + D_e^2)^-1 per e block.
__global__ void kg_chunk_eliminate(GStruct g, const int32_t* __restrict__ chunk_start, int num_chunks,
                                   const double* __restrict__ values, const double* __restrict__ b,
                                   const int64_t* __restrict__ blk_off, const double* __restrict__ ete_inv,
                                   double* __restrict__ lhs, int64_t n, double* __restrict__ rhs, int f_pos0) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= num_chunks) return;
  const int r0 = chunk_start[ch], r1 = chunk_start[ch + 1];
  const int eb = g.cells[g.rcb[r0]].block_id;
  const int es = g.cols[eb].size;
  const double* inv = ete_inv + blk_off[eb];
  if (rhs) {
    double gsum[kMaxBlock], invg[kMaxBlock];
    for (int k = 0; k < es; ++k) gsum[k] = 0.0;
    for (int r = r0; r < r1; ++r) {
      const double* E = values + g.cells[g.rcb[r]].position;
      const int rs = g.rows[r].size, rp = g.rows[r].position;
      for (int k = 0; k < es; ++k)
        for (int i = 0; i < rs; ++i) gsum[k] += E[i * es + k] * b[rp + i];
    }
    for (int k = 0; k < es; ++k) {
      double v = 0.0;
      for (int q = 0; q < es; ++q) v += inv[k * es + q] * gsum[q];
      invg[k] = v;
    }
    for (int r = r0; r < r1; ++r) {
      const double* E = values + g.cells[g.rcb[r]].position;
      const int rs = g.rows[r].size, rp = g.rows[r].position;
      double sj[kMaxBlock];
      for (int i = 0; i < rs; ++i) {
        double v = b[rp + i];
        for (int k = 0; k < es; ++k) v -= E[i * es + k] * invg[k];
        sj[i] = v;
      }
      for (int c = g.rcb[r] + 1; c < g.rcb[r + 1]; ++c) {
        const cx_cell cell = g.cells[c];
        const int fs = g.cols[cell.block_id].size, fp = g.cols[cell.block_id].position - f_pos0;
        const double* F = values + cell.position;
        for (int a = 0; a < fs; ++a) {
          double v = 0.0;
          for (int i = 0; i < rs; ++i) v += F[i * fs + a] * sj[i];
          atomicAdd(&rhs[fp + a], v);
        }
      }
    }
  }
  // S(b1,b2) -= sum_{i,j} F_{i,b1}' E_i inv E_j' F_{j,b2},  b1 <= b2
  for (int ri = r0; ri < r1; ++ri) {
    const double* Ei = values + g.cells[g.rcb[ri]].position;
    const int rsi = g.rows[ri].size;
    for (int ci = g.rcb[ri] + 1; ci < g.rcb[ri + 1]; ++ci) {
      const cx_cell celli = g.cells[ci];
      const int s1 = g.cols[celli.block_id].size, p1 = g.cols[celli.block_id].position - f_pos0;
      const double* Fi = values + celli.position;
      for (int a = 0; a < s1; ++a) {
        double u[kMaxBlock], w[kMaxBlock];  // u = E_i' F_i(:,a) ; w = u' inv
        for (int k = 0; k < es; ++k) {
          double v = 0.0;
          for (int i = 0; i < rsi; ++i) v += Ei[i * es + k] * Fi[i * s1 + a];
          u[k] = v;
        }
        for (int q = 0; q < es; ++q) {
          double v = 0.0;
          for (int k = 0; k < es; ++k) v += u[k] * inv[k * es + q];
          w[q] = v;
        }
        for (int rj = r0; rj < r1; ++rj) {
          const double* Ej = values + g.cells[g.rcb[rj]].position;
          const int rsj = g.rows[rj].size;
          for (int cj = g.rcb[rj] + 1; cj < g.rcb[rj + 1]; ++cj) {
            const cx_cell cellj = g.cells[cj];
            if (celli.block_id > cellj.block_id) continue;
            const int s2 = g.cols[cellj.block_id].size, p2 = g.cols[cellj.block_id].position - f_pos0;
            const double* Fj = values + cellj.position;
            for (int c = 0; c < s2; ++c) {
              double val = 0.0;
              for (int q = 0; q < es; ++q) {
                double v = 0.0;
                for (int i = 0; i < rsj; ++i) v += Ej[i * es + q] * Fj[i * s2 + c];
                val += w[q] * v;
              }
              atomicAdd(&lhs[(p1 + a) * n + p2 + c], -val);
            }
          }
        }
      }
    }
  }
}

// ---- the same elimination as a GATHER (round 4): no atomics, every cell of S summed in a fixed order.
// Per f cell q of an e-row: B_q = E_r' F_q and G_q = (E'E + D_e^2)^-1 B_q (kg_fcell_bg); then one wavefront per cell
// (b1 <= b2) of S walks the cell's products in list order -- + F_i' F_j for two f cells of one row (EBlockRowOuterProduct /
// NoEBlockRowsUpdate), - B_i' G_j for two f cells of one chunk (ChunkOuterProduct) -- and writes the cell once.
__global__ void kg_fcell_bg(GStruct g, const int32_t* __restrict__ fcell, int num_fcells, const double* __restrict__ values,
                            const int64_t* __restrict__ blk_off, const double* __restrict__ ete_inv, double* __restrict__ bg) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= num_fcells) return;
  const int r = fcell[4 * q], c = fcell[4 * q + 1];
  const cx_cell ecell = g.cells[g.rcb[r]], fc = g.cells[c];
  const int rs = g.rows[r].size, es = g.cols[ecell.block_id].size, fs = g.cols[fc.block_id].size;
  const double* E = values + ecell.position;
  const double* F = values + fc.position;
  const double* inv = ete_inv + blk_off[ecell.block_id];
  double* B = bg + fcell[4 * q + 2];
  double* G = bg + fcell[4 * q + 3];
  for (int a = 0; a < fs; ++a) {
    double u[kMaxBlock];
    for (int k = 0; k < es; ++k) {
      double v = 0.0;
      for (int i = 0; i < rs; ++i) v += E[i * es + k] * F[i * fs + a];
      u[k] = v;
      B[k * fs + a] = v;
    }
    for (int k = 0; k < es; ++k) {
      double v = 0.0;
      for (int t = 0; t < es; ++t) v += inv[k * es + t] * u[t];
      G[k * fs + a] = v;
    }
  }
}

__global__ __launch_bounds__(256) void kg_gather_cells(const int32_t* __restrict__ target, const int32_t* __restrict__ tuple_begin,
                                                       const int32_t* __restrict__ tuples, int num_targets,
                                                       const double* __restrict__ values, const double* __restrict__ bg,
                                                       const double* __restrict__ Df, double* __restrict__ lhs, int64_t n) {
  const int t = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
  if (t >= num_targets) return;
  const int lane = threadIdx.x & 63;
  const int p1 = target[4 * t], p2 = target[4 * t + 1], s1 = target[4 * t + 2], s2 = target[4 * t + 3];
  const int entries = s1 * s2;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // blocks of up to 16 x 16: four entries per lane
  for (int k = tuple_begin[t]; k < tuple_begin[t + 1]; ++k) {
    const int kind = tuples[4 * k + 3], inner = tuples[4 * k + 2];
    const double* X = (kind ? bg : values) + tuples[4 * k];
    const double* Y = (kind ? bg : values) + tuples[4 * k + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      if (e < entries) {
        const int a = e / s2, c = e - a * s2;
        double sum = 0.0;
        for (int i = 0; i < inner; ++i) sum += X[i * s1 + a] * Y[i * s2 + c];
        acc[q] += kind ? -sum : sum;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = lane + 64 * q;
    if (e < entries) {
      const int a = e / s2, c = e - a * s2;
      double v = acc[q];
      if (Df && p1 + a == p2 + c) v += Df[p1 + a] * Df[p1 + a];
      lhs[int64_t(p1 + a) * n + p2 + c] = v;
    }
  }
}

// s = b - E (E'E + D^2)^-1 E'b for the rows of a chunk (the vector UpdateRhs multiplies by F', :379-420); rows without an
// e block keep b.  One thread per chunk, sums in row order.
__global__ void kg_chunk_rhs_rows(GStruct g, const int32_t* __restrict__ chunk_start, int num_chunks, const double* __restrict__ values,
                                  const double* __restrict__ b, const int64_t* __restrict__ blk_off, const double* __restrict__ ete_inv,
                                  double* __restrict__ s) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= num_chunks) return;
  const int r0 = chunk_start[ch], r1 = chunk_start[ch + 1];
  const int eb = g.cells[g.rcb[r0]].block_id;
  const int es = g.cols[eb].size;
  const double* inv = ete_inv + blk_off[eb];
  double gsum[kMaxBlock], invg[kMaxBlock];
  for (int k = 0; k < es; ++k) gsum[k] = 0.0;
  for (int r = r0; r < r1; ++r) {
    const double* E = values + g.cells[g.rcb[r]].position;
    const int rs = g.rows[r].size, rp = g.rows[r].position;
    for (int k = 0; k < es; ++k)
      for (int i = 0; i < rs; ++i) gsum[k] += E[i * es + k] * b[rp + i];
  }
  for (int k = 0; k < es; ++k) {
    double v = 0.0;
    for (int q = 0; q < es; ++q) v += inv[k * es + q] * gsum[q];
    invg[k] = v;
  }
  for (int r = r0; r < r1; ++r) {
    const double* E = values + g.cells[g.rcb[r]].position;
    const int rs = g.rows[r].size, rp = g.rows[r].position;
    for (int i = 0; i < rs; ++i) {
      double v = b[rp + i];
      for (int k = 0; k < es; ++k) v -= E[i * es + k] * invg[k];
      s[rp + i] = v;
    }
  }
}

// SchurEliminator::BackSubstitute (:307-373): y_e = (E'E + D^2)^-1 sum E_i'(b_i - F_i z)
__global__ void kg_chunk_backsub(GStruct g, const int32_t* __restrict__ chunk_start, int num_chunks,
                                 const double* __restrict__ values, const double* __restrict__ b,
                                 const int64_t* __restrict__ blk_off, const double* __restrict__ ete_inv,
                                 const double* __restrict__ z, double* __restrict__ y, int f_pos0) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= num_chunks) return;
  const int r0 = chunk_start[ch], r1 = chunk_start[ch + 1];
  const int eb = g.cells[g.rcb[r0]].block_id;
  const int es = g.cols[eb].size, ep = g.cols[eb].position;
  const double* inv = ete_inv + blk_off[eb];
  double acc[kMaxBlock];
  for (int k = 0; k < es; ++k) acc[k] = 0.0;
  for (int r = r0; r < r1; ++r) {
    const int rs = g.rows[r].size, rp = g.rows[r].position;
    double sj[kMaxBlock];
    for (int i = 0; i < rs; ++i) sj[i] = b[rp + i];
    for (int c = g.rcb[r] + 1; c < g.rcb[r + 1]; ++c) {
      const cx_cell cell = g.cells[c];
      const int fs = g.cols[cell.block_id].size, fp = g.cols[cell.block_id].position - f_pos0;
      const double* F = values + cell.position;
      for (int i = 0; i < rs; ++i)
        for (int a = 0; a < fs; ++a) sj[i] -= F[i * fs + a] * z[fp + a];
    }
    const double* E = values + g.cells[g.rcb[r]].position;
    for (int k = 0; k < es; ++k)
      for (int i = 0; i < rs; ++i) acc[k] += E[i * es + k] * sj[i];
  }
  for (int k = 0; k < es; ++k) {
    double v = 0.0;
    for (int q = 0; q < es; ++q) v += inv[k * es + q] * acc[q];
    y[ep + k] = v;
  }
}

__global__ void kg_add_diag_sq(double* __restrict__ lhs, int64_t n, const double* __restrict__ Df) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) lhs[i * n + i] += Df[i] * Df[i];
}
__global__ void kg_d2x(double* __restrict__ y, const double* __restrict__ d, const double* __restrict__ x, int64_t n, int assign) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (assign ? 0.0 : y[i]) + (d ? d[i] * d[i] * x[i] : 0.0);
}
__global__ void kg_sub(const double* __restrict__ a, const double* __restrict__ bvec, double* __restrict__ out, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - bvec[i];
}
// out = rhs - S z for a dense symmetric S of which the upper triangle is valid (row-major, n x n); one thread per row
__global__ void kg_sym_residual(const double* __restrict__ S, int64_t n, const double* __restrict__ rhs, const double* __restrict__ z,
                                double* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int64_t j = 0; j < i; ++j) s += S[j * n + i] * z[j];
  for (int64_t j = i; j < n; ++j) s += S[i * n + j] * z[j];
  out[i] = rhs[i] - s;
}
__global__ void kg_axpy(double* __restrict__ y, const double* __restrict__ x, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
__global__ void kg_negate(double* __restrict__ a, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] = -a[i];
}

// ------------------------------------------------------------------- host side
int PrepareGenericGather(cx_matrix* A);

struct Generic {
  cx_solver* S;
  cx_matrix* A;
  cx_context* ctx;
  hipStream_t st;
  GStruct g;
  int64_t ne, nf, off_f;  // columns in e / f, packed-block offset of the first f block
  int nfb;                // number of f blocks

  int zero(double* p, int64_t n) { CX_HIP(hipMemsetAsync(p, 0, size_t(std::max<int64_t>(n, 1)) * sizeof(double), st)); return CX_OK; }
  int mult(int sel, bool transpose, const double* x, double* y) {
    const int off = (sel == SEL_F) ? int(ne) : 0;
    if (g.R == 0) return CX_OK;
    static const bool atomics = std::getenv("CX_GENERIC_ATOMICS") != nullptr;  // A/B switch: the scatter form with fp64 atomics
    if (transpose && !atomics) return cxk_generic_left_multiply(A, sel, off, x, y);  // gather over the transposed index, fixed order
    if (transpose) hipLaunchKernelGGL(kg_mult_t, dim3(grid_for(g.R, 128)), dim3(128), 0, st, g, (const double*)A->d_values.p, x, y, sel, off);
    else hipLaunchKernelGGL(kg_mult, dim3(grid_for(g.R, 128)), dim3(128), 0, st, g, (const double*)A->d_values.p, x, y, sel, off);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  // packed inverse block diagonal of A_sel' A_sel + D^2 over column blocks [first, first+count)
  int block_diag_inverse(int sel, int first, int count, const double* D, double* blocks, bool allreduce) {
    const int64_t off0 = A->blk_off[first], total = A->blk_off[first + count] - off0;
    CX_TRY(zero(blocks, total));
    static const bool atomics = std::getenv("CX_GENERIC_ATOMICS") != nullptr;  // A/B switch
    if (g.R && !atomics) CX_TRY(cxk_generic_block_diagonal(A, sel, first, count, off0, blocks));  // gather over the transposed index
    else if (g.R) hipLaunchKernelGGL(kg_block_diag, dim3(grid_for(g.R, 128)), dim3(128), 0, st, g, (const double*)A->d_values.p,
                                (const int64_t*)A->d_blk_off.p, off0, blocks, sel);
    if (allreduce && ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, blocks, total));
    if (count) hipLaunchKernelGGL(kg_blockdiag_invert, dim3(grid_for(count, 64)), dim3(64), 0, st, (const cx_block*)A->d_cols.p,
                                  (const int64_t*)A->d_blk_off.p, off0, first, count, D, blocks, S->flag.p);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int blockdiag_mult(int first, int count, int pos0, const double* blocks, const double* x, double* y) {
    if (count) hipLaunchKernelGGL(kg_blockdiag_mult, dim3(grid_for(count, 64)), dim3(64), 0, st, (const cx_block*)A->d_cols.p,
                                  (const int64_t*)A->d_blk_off.p, A->blk_off[first], first, count, pos0, blocks, x, y);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  // SchurEliminator::Eliminate into dense lhs (n = nf); ete_inv must hold (E'E + D_e^2)^-1
  int eliminate(const double* b, const double* D, const double* ete_inv, bool add_df, double* lhs, double* rhs) {
    static const bool atomics = std::getenv("CX_GENERIC_ATOMICS") != nullptr;  // A/B switch: the scatter kernels
    if (!atomics) {
      CX_TRY(PrepareGenericGather(A));
      if (A->gather_state == 1) return eliminate_gather(b, D, ete_inv, add_df, lhs, rhs);
    }
    CX_TRY(zero(lhs, nf * nf));
    if (rhs) CX_TRY(zero(rhs, nf));
    if (D && add_df && nf) hipLaunchKernelGGL(kg_add_diag_sq, dim3(grid_for(nf, 256)), dim3(256), 0, st, lhs, nf, D + ne);
    if (g.R) hipLaunchKernelGGL(kg_row_outer, dim3(grid_for(g.R, 128)), dim3(128), 0, st, g, (const double*)A->d_values.p, b, lhs, nf, rhs, int(ne));
    if (A->num_chunks) hipLaunchKernelGGL(kg_chunk_eliminate, dim3(grid_for(A->num_chunks, 64)), dim3(64), 0, st, g,
                                          (const int32_t*)A->d_chunk_start.p, A->num_chunks, (const double*)A->d_values.p, b,
                                          (const int64_t*)A->d_blk_off.p, ete_inv, lhs, nf, rhs, int(ne));
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  // the same without atomics: every cell of S gathered in a fixed order (kg_gather_cells), the right-hand side through the
  // transposed index (F's with s = b - E (E'E)^-1 E'b): the same bits every time
  int eliminate_gather(const double* b, const double* D, const double* ete_inv, bool add_df, double* lhs, double* rhs) {
    CX_TRY(zero(lhs, nf * nf));
    if (D && add_df && nf) hipLaunchKernelGGL(kg_add_diag_sq, dim3(grid_for(nf, 256)), dim3(256), 0, st, lhs, nf, D + ne);  // blocks no row touches
    if (A->g_num_fcells)
      hipLaunchKernelGGL(kg_fcell_bg, dim3(grid_for(A->g_num_fcells, 64)), dim3(64), 0, st, g, (const int32_t*)A->d_g_fcell.p, A->g_num_fcells,
                         (const double*)A->d_values.p, (const int64_t*)A->d_blk_off.p, ete_inv, A->d_g_bg.p);
    if (A->g_num_targets)
      hipLaunchKernelGGL(kg_gather_cells, dim3(grid_for(A->g_num_targets, 4)), dim3(256), 0, st, (const int32_t*)A->d_g_target.p,
                         (const int32_t*)A->d_g_tuple_begin.p, (const int32_t*)A->d_g_tuples.p, A->g_num_targets, (const double*)A->d_values.p,
                         (const double*)A->d_g_bg.p, (D && add_df) ? D + ne : (const double*)nullptr, lhs, nf);
    if (rhs) {
      CX_TRY(S->v_rows.alloc(size_t(std::max<int64_t>(A->num_rows, 1))));
      double* s = S->v_rows.p;
      CX_HIP(hipMemcpyAsync(s, b, size_t(A->num_rows) * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (A->num_chunks)
        hipLaunchKernelGGL(kg_chunk_rhs_rows, dim3(grid_for(A->num_chunks, 64)), dim3(64), 0, st, g, (const int32_t*)A->d_chunk_start.p,
                           A->num_chunks, (const double*)A->d_values.p, b, (const int64_t*)A->d_blk_off.p, ete_inv, s);
      CX_TRY(zero(rhs, nf));
      CX_TRY(mult(SEL_F, true, s, rhs));
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  int backsub_chunks(const double* b, const double* ete_inv, const double* z, double* y) {
    if (A->num_chunks) hipLaunchKernelGGL(kg_chunk_backsub, dim3(grid_for(A->num_chunks, 64)), dim3(64), 0, st, g,
                                          (const int32_t*)A->d_chunk_start.p, A->num_chunks, (const double*)A->d_values.p, b,
                                          (const int64_t*)A->d_blk_off.p, ete_inv, z, y, int(ne));
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

// S x composed exactly as ImplicitSchurComplement::RightMultiplyAndAccumulate (implicit_schur_complement.cc:106-144)
struct GenericSchurOp : LinOp {
  Generic* G;
  const double* D;
  double *tmp_rows, *tmp_e, *tmp_e2;
  const double* ete_inv;
  int64_t size() const override { return G->nf; }
  int apply(const double* x, double* y) override {
    Generic& q = *G;
    CX_TRY(q.zero(tmp_rows, q.A->num_rows));
    CX_TRY(q.mult(SEL_F, false, x, tmp_rows));
    CX_TRY(q.zero(tmp_e, q.ne));
    CX_TRY(q.mult(SEL_E, true, tmp_rows, tmp_e));
    CX_TRY(q.blockdiag_mult(0, q.g.nelim, 0, ete_inv, tmp_e, tmp_e2));
    if (q.ne) hipLaunchKernelGGL(kg_negate, dim3(grid_for(q.ne, 256)), dim3(256), 0, q.st, tmp_e2, q.ne);
    CX_TRY(q.mult(SEL_E, false, tmp_e2, tmp_rows));
    CX_TRY(q.zero(y, q.nf));
    CX_TRY(q.mult(SEL_F, true, tmp_rows, y));
    if (q.ctx->nranks > 1) CX_TRY(cx_allreduce_device(q.ctx, y, q.nf));
    if (D && q.nf) hipLaunchKernelGGL(kg_d2x, dim3(grid_for(q.nf, 256)), dim3(256), 0, q.st, y, D + q.ne, x, q.nf, 0);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

struct GenericBlockDiagOp : LinOp {
  Generic* G;
  int first, count, pos0;
  const double* blocks;
  int64_t n;
  int64_t size() const override { return n; }
  int apply(const double* x, double* y) override { return G->blockdiag_mult(first, count, pos0, blocks, x, y); }
};

struct GenericIdentityOp : LinOp {
  Generic* G;
  int64_t n;
  int64_t size() const override { return n; }
  int apply(const double* x, double* y) override {
    CX_HIP(hipMemcpyAsync(y, x, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, G->st));
    return CX_OK;
  }
};

// PowerSeriesExpansionPreconditioner (power_series_expansion_preconditioner.cc:57-84) on the dynamic-size operators:
// y = sum_{k = 0 .. max_iterations} Z^k (F'F)^-1 x, Z = (F'F)^-1 F'E (E'E)^-1 E'F (InversePowerSeriesOperatorRightMultiply-
// Accumulate, implicit_schur_complement.cc:146-177), a fixed number of terms (the preconditioner's tolerance is 0,
// iterative_schur_complement_solver.cc:178-186).  blocks: the inverted diagonal blocks of F'F (+ D_f^2).
struct GenericSpseOp : LinOp {
  Generic* G;
  const double* ftf_inv;
  const double* ete_inv;
  int max_iterations;
  double *tmp_rows, *tmp_e, *tmp_e2, *series, *previous, *tmp_f;
  int64_t size() const override { return G->nf; }
  int apply(const double* x, double* y) override {
    Generic& q = *G;
    const int64_t n = q.nf;
    const int first = q.g.nelim, count = q.nfb, pos0 = int(q.ne);
    CX_TRY(q.blockdiag_mult(first, count, pos0, ftf_inv, x, y));
    CX_HIP(hipMemcpyAsync(previous, y, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, q.st));
    double* prev = previous;
    double* ser = series;
    for (int i = 1; i <= max_iterations; ++i) {
      CX_TRY(q.zero(tmp_rows, q.A->num_rows));
      CX_TRY(q.mult(SEL_F, false, prev, tmp_rows));
      CX_TRY(q.zero(tmp_e, q.ne));
      CX_TRY(q.mult(SEL_E, true, tmp_rows, tmp_e));
      CX_TRY(q.blockdiag_mult(0, q.g.nelim, 0, ete_inv, tmp_e, tmp_e2));
      CX_TRY(q.zero(tmp_rows, q.A->num_rows));
      CX_TRY(q.mult(SEL_E, false, tmp_e2, tmp_rows));
      CX_TRY(q.zero(tmp_f, n));
      CX_TRY(q.mult(SEL_F, true, tmp_rows, tmp_f));
      CX_TRY(q.blockdiag_mult(first, count, pos0, ftf_inv, tmp_f, ser));
      if (n) hipLaunchKernelGGL(kg_axpy, dim3(grid_for(n, 256)), dim3(256), 0, q.st, y, (const double*)ser, n);
      std::swap(prev, ser);
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

struct GenericCgnrOp : LinOp {
  Generic* G;
  const double* D;
  double* tmp_rows;
  int64_t size() const override { return G->A->num_cols; }
  int apply(const double* x, double* y) override {
    Generic& q = *G;
    CX_TRY(q.zero(tmp_rows, q.A->num_rows));
    CX_TRY(q.mult(SEL_ALL, false, x, tmp_rows));
    CX_TRY(q.zero(y, size()));
    CX_TRY(q.mult(SEL_ALL, true, tmp_rows, y));
    if (D) hipLaunchKernelGGL(kg_d2x, dim3(grid_for(size(), 256)), dim3(256), 0, q.st, y, D, x, size(), 0);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
};

// The product lists of the gather eliminator, from the structure alone (once per matrix).
int PrepareGenericGather(cx_matrix* A) {
  if (A->gather_state != 0) return CX_OK;
  const int nelim = A->nelim, nre = A->num_row_blocks_e;
  const int64_t f_pos0 = A->num_cols_e;
  // f cells of e-rows and their B / G slots
  std::vector<int32_t> fcell;
  std::vector<int32_t> q_of_cell(A->cells.size(), -1);
  int64_t bg = 0;
  for (int r = 0; r < nre; ++r) {
    const int es = A->cols[size_t(A->cells[size_t(A->rcb[size_t(r)])].block_id)].size;
    for (int c = A->rcb[size_t(r)] + 1; c < A->rcb[size_t(r) + 1]; ++c) {
      const int fs = A->cols[size_t(A->cells[size_t(c)].block_id)].size;
      q_of_cell[size_t(c)] = int32_t(fcell.size() / 4);
      fcell.insert(fcell.end(), {int32_t(r), int32_t(c), int32_t(bg), int32_t(bg + int64_t(es) * fs)});
      bg += 2 * int64_t(es) * fs;
      if (bg >= (int64_t(1) << 31)) { A->gather_state = 2; return CX_OK; }
    }
  }
  struct Product { int64_t key; int32_t x, y, inner, kind; };
  std::vector<Product> products;
  const int64_t kMaxProducts = int64_t(1) << 26;
  auto key_of = [&](int b1, int b2) { return int64_t(b1) * (int64_t(A->Cb) + 1) + b2; };
  // + F_i' F_j over the f cells of every row (cells i <= j in row order)
  for (int r = 0; r < A->R; ++r) {
    int cb = A->rcb[size_t(r)], ce = A->rcb[size_t(r) + 1];
    if (r < nre) cb = std::min(ce, cb + 1);
    for (int i = cb; i < ce; ++i)
      for (int j = i; j < ce; ++j) {
        const cx_cell& ci = A->cells[size_t(i)];
        const cx_cell& cj = A->cells[size_t(j)];
        products.push_back({key_of(ci.block_id, cj.block_id), ci.position, cj.position, A->rows[size_t(r)].size, 0});
      }
    if (int64_t(products.size()) > kMaxProducts) { A->gather_state = 2; return CX_OK; }
  }
  // - B_i' G_j over the f cells of every chunk, blocks b1 <= b2 (both orders when they are equal, as ChunkOuterProduct)
  {
    int r = 0;
    while (r < nre) {
      const int id = A->cells[size_t(A->rcb[size_t(r)])].block_id;
      const int r0 = r;
      while (r < nre && A->cells[size_t(A->rcb[size_t(r)])].block_id == id) ++r;
      const int es = A->cols[size_t(id)].size;
      for (int ri = r0; ri < r; ++ri)
        for (int ci = A->rcb[size_t(ri)] + 1; ci < A->rcb[size_t(ri) + 1]; ++ci)
          for (int rj = r0; rj < r; ++rj)
            for (int cj = A->rcb[size_t(rj)] + 1; cj < A->rcb[size_t(rj) + 1]; ++cj) {
              const int b1 = A->cells[size_t(ci)].block_id, b2 = A->cells[size_t(cj)].block_id;
              if (b1 > b2) continue;
              products.push_back({key_of(b1, b2), fcell[size_t(4 * q_of_cell[size_t(ci)] + 2)], fcell[size_t(4 * q_of_cell[size_t(cj)] + 3)], es, 1});
            }
      if (int64_t(products.size()) > kMaxProducts) { A->gather_state = 2; return CX_OK; }
    }
  }
  std::stable_sort(products.begin(), products.end(), [](const Product& a, const Product& b) { return a.key < b.key; });
  std::vector<int32_t> target, tuple_begin, tuples;
  tuples.reserve(products.size() * 4);
  for (size_t k = 0; k < products.size(); ++k) {
    if (k == 0 || products[k].key != products[k - 1].key) {
      const int b1 = int(products[k].key / (int64_t(A->Cb) + 1)), b2 = int(products[k].key % (int64_t(A->Cb) + 1));
      if (b1 < nelim || b2 < nelim) {
        cx_set_error("an f cell of row products lies in an e-block column");
        return CX_ERR_INVALID_ARGUMENT;
      }
      target.insert(target.end(), {int32_t(A->cols[size_t(b1)].position - f_pos0), int32_t(A->cols[size_t(b2)].position - f_pos0),
                                   A->cols[size_t(b1)].size, A->cols[size_t(b2)].size});
      tuple_begin.push_back(int32_t(k));
    }
    tuples.insert(tuples.end(), {products[k].x, products[k].y, products[k].inner, products[k].kind});
  }
  tuple_begin.push_back(int32_t(products.size()));
  hipStream_t st = A->ctx->stream;
  A->g_num_targets = int32_t(target.size() / 4);
  A->g_num_fcells = int32_t(fcell.size() / 4);
  if (target.empty()) target.assign(4, 0);
  if (tuples.empty()) tuples.assign(4, 0);
  if (fcell.empty()) fcell.assign(4, 0);
  CX_TRY(A->d_g_target.upload(target, st));
  CX_TRY(A->d_g_tuple_begin.upload(tuple_begin, st));
  CX_TRY(A->d_g_tuples.upload(tuples, st));
  CX_TRY(A->d_g_fcell.upload(fcell, st));
  CX_TRY(A->d_g_bg.alloc(size_t(std::max<int64_t>(bg, 1))));
  A->gather_state = 1;
  return CX_OK;
}

int PrepareGeneric(cx_matrix* A) {
  if (A->generic_ready) return CX_OK;
  int mx = 0;
  for (auto& r : A->rows) mx = std::max(mx, int(r.size));
  for (auto& c : A->cols) mx = std::max(mx, int(c.size));
  if (mx > kMaxBlock) {
    cx_set_error("dynamic path supports blocks of up to %d rows/columns, matrix has %d", kMaxBlock, mx);
    return CX_ERR_UNSUPPORTED;
  }
  A->blk_off.assign(size_t(A->Cb) + 1, 0);
  for (int c = 0; c < A->Cb; ++c) A->blk_off[c + 1] = A->blk_off[c] + int64_t(A->cols[c].size) * A->cols[c].size;
  // chunks (SchurEliminator::Init, schur_eliminator_impl.h:115-156)
  std::vector<int32_t> chunk_start;
  int r = 0;
  while (r < A->R) {
    if (A->rcb[r + 1] == A->rcb[r]) break;
    const int id = A->cells[A->rcb[r]].block_id;
    if (id >= A->nelim) break;
    chunk_start.push_back(r);
    while (r < A->R && A->rcb[r + 1] > A->rcb[r] && A->cells[A->rcb[r]].block_id == id) ++r;
  }
  A->num_chunks = int(chunk_start.size());
  chunk_start.push_back(r);
  if (r != A->num_row_blocks_e) {
    cx_set_error("rows of one e block are not contiguous (row %d); the matrix is not ordered for Schur elimination", r);
    return CX_ERR_INVALID_ARGUMENT;
  }
  CX_TRY(A->d_chunk_start.upload(chunk_start, A->ctx->stream));
  CX_TRY(A->d_blk_off.upload(A->blk_off, A->ctx->stream));
  A->generic_ready = true;
  return CX_OK;
}

}  // namespace

int cxg_solve(cx_solver* S, cx_matrix* A, const double* b, const double* D, double r_tol, double q_tol, double* x,
              cx_summary* summary) {
  CX_TRY(PrepareGeneric(A));
  cx_context* ctx = S->ctx;
  hipStream_t st = ctx->stream;
  const cx_solver_options& o = S->opt;
  Generic G;
  G.S = S; G.A = A; G.ctx = ctx; G.st = st;
  G.g = GStruct{A->d_rows.p, A->d_cols.p, A->d_rcb.p, A->d_cells.p, A->R, A->num_row_blocks_e, A->nelim};
  G.ne = A->num_cols_e; G.nf = A->num_cols_f; G.off_f = A->blk_off[A->nelim]; G.nfb = A->Cb - A->nelim;
  const int64_t n = A->num_cols;
  CX_TRY(S->flag.alloc(1));
  CX_HIP(hipMemsetAsync(S->flag.p, 0, sizeof(int), st));
  CX_TRY(S->v_rows.alloc(size_t(A->num_rows)));
  bool failed = false;

  if (o.type == CX_CGNR) {
    if (o.preconditioner_type != CX_JACOBI && o.preconditioner_type != CX_IDENTITY) {
      cx_set_error("CGNR supports JACOBI and IDENTITY preconditioners (cgnr_solver.cc:125-133)");
      return CX_ERR_UNSUPPORTED;
    }
    CX_TRY(S->v_rhs.alloc(n));
    CX_TRY(S->cam_blocks.alloc(size_t(A->blk_off[A->Cb])));
    if (o.preconditioner_type == CX_JACOBI) CX_TRY(G.block_diag_inverse(SEL_ALL, 0, A->Cb, D, S->cam_blocks.p, false));
    CX_TRY(G.zero(S->v_rhs.p, n));
    CX_TRY(G.mult(SEL_ALL, true, b, S->v_rhs.p));
    CX_TRY(G.zero(x, n));
    GenericCgnrOp lhs; lhs.G = &G; lhs.D = D; lhs.tmp_rows = S->v_rows.p;
    GenericBlockDiagOp jac; jac.G = &G; jac.first = 0; jac.count = A->Cb; jac.pos0 = 0; jac.blocks = S->cam_blocks.p; jac.n = n;
    GenericIdentityOp id; id.G = &G; id.n = n;
    LinOp& pre = (o.preconditioner_type == CX_IDENTITY) ? static_cast<LinOp&>(id) : static_cast<LinOp&>(jac);
    CX_TRY(cx_cg_run(S, n, n, lhs, pre, S->v_rhs.p, x, true, r_tol, q_tol, summary));
    return CX_OK;
  }

  // ---- Schur family: (E'E + D_e^2)^-1 first
  CX_TRY(S->ete_inv.alloc(size_t(std::max<int64_t>(G.off_f, 1))));
  CX_TRY(G.block_diag_inverse(SEL_E, 0, A->nelim, D, S->ete_inv.p, false));
  double* z = x + G.ne;

  if (o.type == CX_DENSE_SCHUR || o.type == CX_SPARSE_SCHUR) {
    CX_TRY(S->lhs.alloc(size_t(std::max<int64_t>(G.nf * G.nf, 1))));
    CX_TRY(S->v_rhs.alloc(size_t(std::max<int64_t>(G.nf, 1))));
    CX_TRY(G.zero(x, n));
    const bool sharded = ctx->nranks > 1;
    CX_TRY(G.eliminate(b, D, S->ete_inv.p, !sharded || ctx->rank == 0, S->lhs.p, S->v_rhs.p));
    if (sharded) {
      CX_TRY(cx_allreduce_device(ctx, S->lhs.p, G.nf * G.nf));
      CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p, G.nf));
    }
    summary->termination_type = CX_SUCCESS;
    summary->num_iterations = 0;
    std::snprintf(summary->message, sizeof(summary->message), "Success.");
    if (G.nf > 0) {
      if (S->opt.max_num_refinement_iterations > 0) {  // refinement needs S itself beside its factor
        CX_TRY(S->lhs_copy.alloc(size_t(G.nf * G.nf)));
        CX_HIP(hipMemcpyAsync(S->lhs_copy.p, S->lhs.p, size_t(G.nf * G.nf) * sizeof(double), hipMemcpyDeviceToDevice, st));
      }
      CX_TRY(cxd_cholesky_solve(ctx, int(G.nf), S->lhs.p, S->v_rhs.p, z, S->flag.p));
      summary->num_iterations = 1;
      CX_TRY(cx_check_flag(S, "Dense Cholesky factorization failed: the reduced matrix is not positive definite.", summary, &failed));
    }
    // max_num_refinement_iterations (RefinedDenseCholesky -> DenseIterativeRefiner::Refine, dense_cholesky.cc:122-128,
    // iterative_refiner.cc:80-103): residual = rhs - S z in double, z += S^-1 residual, a fixed number of times.  The dense
    // kernel factors in place and solves inside the factorisation, so a step factors a fresh copy of S -- the same factor
    // every time (this is the correctness path of the dynamic-size structures, not a fast one).
    const int refinements = std::max(0, S->opt.max_num_refinement_iterations);
    if (summary->termination_type == CX_SUCCESS && refinements > 0 && G.nf > 0) {
      CX_TRY(S->v_p.alloc(size_t(G.nf)));
      CX_TRY(S->v_tmp.alloc(size_t(G.nf)));
      for (int it = 0; it < refinements; ++it) {
        hipLaunchKernelGGL(kg_sym_residual, dim3(grid_for(G.nf, 64)), dim3(64), 0, st, (const double*)S->lhs_copy.p, G.nf,
                           (const double*)S->v_rhs.p, (const double*)z, S->v_p.p);
        CX_HIP(hipMemcpyAsync(S->lhs.p, S->lhs_copy.p, size_t(G.nf * G.nf) * sizeof(double), hipMemcpyDeviceToDevice, st));
        CX_TRY(cxd_cholesky_solve(ctx, int(G.nf), S->lhs.p, S->v_p.p, S->v_tmp.p, S->flag.p));
        hipLaunchKernelGGL(kg_axpy, dim3(grid_for(G.nf, 256)), dim3(256), 0, st, z, (const double*)S->v_tmp.p, G.nf);
      }
      CX_HIP(hipGetLastError());
    }
    if (summary->termination_type == CX_SUCCESS) CX_TRY(G.backsub_chunks(b, S->ete_inv.p, z, x));
    // use_mixed_precision_solves: the single precision factor lives in the tile code of the static <2,3,9> layout
    // (cx_solver.hip: SolveDenseSchur239); here the factor is double -- said in the message and in summary->notes
    if (summary->termination_type == CX_SUCCESS && (S->opt.use_mixed_precision_solves || refinements > 0)) {
      if (S->opt.use_mixed_precision_solves) summary->notes |= CX_NOTE_DOUBLE_PRECISION_FACTOR;
      std::snprintf(summary->message, sizeof(summary->message), "Success. (dynamic-size structure: double precision dense factorisation, %d refinement step%s)",
                    refinements, refinements == 1 ? "" : "s");
    }
    return CX_OK;
  }

  // ---- ITERATIVE_SCHUR (iterative_schur_complement_solver.cc:64-157)
  CX_TRY(S->v_rhs.alloc(size_t(std::max<int64_t>(G.nf, 1))));
  CX_TRY(S->v_x.alloc(size_t(std::max<int64_t>(G.nf, 1))));
  CX_TRY(S->v_cols.alloc(size_t(std::max<int64_t>(2 * G.ne, 1))));
  double* tmp_e = S->v_cols.p;
  double* tmp_e2 = S->v_cols.p + G.ne;
  double* tmp_rows = S->v_rows.p;
  // UpdateRhs (implicit_schur_complement.cc:251-276)
  CX_TRY(G.zero(tmp_e, G.ne));
  CX_TRY(G.mult(SEL_E, true, b, tmp_e));
  CX_TRY(G.blockdiag_mult(0, A->nelim, 0, S->ete_inv.p, tmp_e, tmp_e2));
  CX_TRY(G.zero(tmp_rows, A->num_rows));
  CX_TRY(G.mult(SEL_E, false, tmp_e2, tmp_rows));
  if (A->num_rows) hipLaunchKernelGGL(kg_sub, dim3(grid_for(A->num_rows, 256)), dim3(256), 0, st, b, (const double*)tmp_rows, tmp_rows, A->num_rows);
  CX_TRY(G.zero(S->v_rhs.p, G.nf));
  CX_TRY(G.mult(SEL_F, true, tmp_rows, S->v_rhs.p));
  if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, S->v_rhs.p, G.nf));
  if (G.nfb == 0) {  // nothing left in the Schur complement (:84-91)
    summary->num_iterations = 0;
    summary->termination_type = CX_SUCCESS;
    CX_TRY(G.zero(x, n));
    CX_TRY(G.backsub_chunks(b, S->ete_inv.p, z, x));
    return CX_OK;
  }
  // preconditioner
  CX_TRY(S->cam_blocks.alloc(size_t(std::max<int64_t>(A->blk_off[A->Cb] - G.off_f, 1))));
  if (o.preconditioner_type == CX_JACOBI) {
    CX_TRY(G.block_diag_inverse(SEL_F, A->nelim, G.nfb, D, S->cam_blocks.p, true));
  } else if (o.preconditioner_type == CX_SCHUR_JACOBI) {
    // block diagonal of S through the eliminator (schur_jacobi_preconditioner.cc:88-98)
    CX_TRY(S->lhs.alloc(size_t(G.nf * G.nf)));
    const bool sharded = ctx->nranks > 1;
    CX_TRY(G.eliminate(nullptr, D, S->ete_inv.p, !sharded, S->lhs.p, nullptr));
    hipLaunchKernelGGL(kg_extract_diag_blocks, dim3(grid_for(G.nfb, 64)), dim3(64), 0, st, (const cx_block*)A->d_cols.p,
                       (const int64_t*)A->d_blk_off.p, G.off_f, A->nelim, G.nfb, int(G.ne), (const double*)S->lhs.p, G.nf, S->cam_blocks.p);
    if (sharded) CX_TRY(cx_allreduce_device(ctx, S->cam_blocks.p, A->blk_off[A->Cb] - G.off_f));
    hipLaunchKernelGGL(kg_blockdiag_invert, dim3(grid_for(G.nfb, 64)), dim3(64), 0, st, (const cx_block*)A->d_cols.p,
                       (const int64_t*)A->d_blk_off.p, G.off_f, A->nelim, G.nfb, sharded ? D : (const double*)nullptr, S->cam_blocks.p, S->flag.p);
    CX_HIP(hipGetLastError());
  } else if (o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION) {
    CX_TRY(G.block_diag_inverse(SEL_F, A->nelim, G.nfb, D, S->cam_blocks.p, true));  // (F'F + D_f^2)^-1, as for JACOBI
    CX_TRY(S->v_spse.alloc(size_t(std::max<int64_t>(3 * G.nf, 1))));
    CX_TRY(S->v_spse_rows.alloc(size_t(std::max<int64_t>(A->num_rows + 2 * G.ne, 1))));
  } else if (o.preconditioner_type != CX_IDENTITY) {
    cx_set_error("preconditioner %d on a dynamic-size structure: the visibility based preconditioners exist for the static <2,3,9> "
                 "layout (and the structures embedded in it) only", o.preconditioner_type);
    return CX_ERR_UNSUPPORTED;
  }
  CX_TRY(cx_check_flag(S, "Preconditioner update failed.", summary, &failed));
  if (failed) return CX_OK;
  GenericSchurOp lhs;
  lhs.G = &G; lhs.D = D; lhs.tmp_rows = tmp_rows; lhs.tmp_e = tmp_e; lhs.tmp_e2 = tmp_e2; lhs.ete_inv = S->ete_inv.p;
  GenericBlockDiagOp bd; bd.G = &G; bd.first = A->nelim; bd.count = G.nfb; bd.pos0 = int(G.ne); bd.blocks = S->cam_blocks.p; bd.n = G.nf;
  GenericIdentityOp id; id.G = &G; id.n = G.nf;
  GenericSpseOp spse;
  spse.G = &G; spse.ftf_inv = S->cam_blocks.p; spse.ete_inv = S->ete_inv.p; spse.max_iterations = o.max_num_spse_iterations;
  if (o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION) {  // its own temporaries: the operator's are live inside CG
    spse.tmp_rows = S->v_spse_rows.p; spse.tmp_e = S->v_spse_rows.p + A->num_rows; spse.tmp_e2 = spse.tmp_e + G.ne;
    spse.series = S->v_spse.p; spse.previous = S->v_spse.p + G.nf; spse.tmp_f = S->v_spse.p + 2 * G.nf;
  }
  LinOp& pre = (o.preconditioner_type == CX_IDENTITY) ? static_cast<LinOp&>(id)
               : (o.preconditioner_type == CX_SCHUR_POWER_SERIES_EXPANSION) ? static_cast<LinOp&>(spse) : static_cast<LinOp&>(bd);
  if (o.use_spse_initialization) summary->notes |= CX_NOTE_SPSE_INITIALIZATION_SKIPPED;  // CG starts from zero here
  CX_TRY(G.zero(S->v_x.p, G.nf));
  CX_TRY(cx_cg_run(S, G.nf, G.nf, lhs, pre, S->v_rhs.p, S->v_x.p, true, r_tol, q_tol, summary));
  if (summary->termination_type != CX_FAILURE && summary->termination_type != CX_FATAL_ERROR) {
    // ImplicitSchurComplement::BackSubstitute (:208-243); rows without an e block do not enter
    CX_TRY(G.zero(x, n));
    CX_TRY(G.backsub_chunks(b, S->ete_inv.p, S->v_x.p, x));
    CX_HIP(hipMemcpyAsync(z, S->v_x.p, size_t(G.nf) * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  return CX_OK;
}

// entry points used by cx_solver.hip for matrices outside the static layout
int cxg_eliminate_dense(cx_solver* S, cx_matrix* A, const double* b, const double* D, double* lhs, double* rhs) {
  CX_TRY(PrepareGeneric(A));
  Generic G;
  G.S = S; G.A = A; G.ctx = A->ctx; G.st = A->ctx->stream;
  G.g = GStruct{A->d_rows.p, A->d_cols.p, A->d_rcb.p, A->d_cells.p, A->R, A->num_row_blocks_e, A->nelim};
  G.ne = A->num_cols_e; G.nf = A->num_cols_f; G.off_f = A->blk_off[A->nelim]; G.nfb = A->Cb - A->nelim;
  CX_TRY(S->flag.alloc(1));
  CX_TRY(S->ete_inv.alloc(size_t(std::max<int64_t>(G.off_f, 1))));
  CX_TRY(G.block_diag_inverse(SEL_E, 0, A->nelim, D, S->ete_inv.p, false));
  return G.eliminate(b, D, S->ete_inv.p, true, lhs, rhs);
}

int cxg_back_substitute(cx_solver* S, cx_matrix* A, const double* b, const double* D, const double* z, double* x) {
  CX_TRY(PrepareGeneric(A));
  Generic G;
  G.S = S; G.A = A; G.ctx = A->ctx; G.st = A->ctx->stream;
  G.g = GStruct{A->d_rows.p, A->d_cols.p, A->d_rcb.p, A->d_cells.p, A->R, A->num_row_blocks_e, A->nelim};
  G.ne = A->num_cols_e; G.nf = A->num_cols_f; G.off_f = A->blk_off[A->nelim]; G.nfb = A->Cb - A->nelim;
  CX_TRY(S->flag.alloc(1));
  CX_TRY(S->ete_inv.alloc(size_t(std::max<int64_t>(G.off_f, 1))));
  CX_TRY(G.block_diag_inverse(SEL_E, 0, A->nelim, D, S->ete_inv.p, false));
  return G.backsub_chunks(b, S->ete_inv.p, z, x);
}

int cxg_implicit_schur_multiply(cx_solver* S, cx_matrix* A, const double* D, const double* b, const double* x,
                                double* y, double* rhs) {
  CX_TRY(PrepareGeneric(A));
  Generic G;
  G.S = S; G.A = A; G.ctx = A->ctx; G.st = A->ctx->stream;
  G.g = GStruct{A->d_rows.p, A->d_cols.p, A->d_rcb.p, A->d_cells.p, A->R, A->num_row_blocks_e, A->nelim};
  G.ne = A->num_cols_e; G.nf = A->num_cols_f; G.off_f = A->blk_off[A->nelim]; G.nfb = A->Cb - A->nelim;
  hipStream_t st = G.st;
  CX_TRY(S->flag.alloc(1));
  CX_TRY(S->ete_inv.alloc(size_t(std::max<int64_t>(G.off_f, 1))));
  CX_TRY(S->v_rows.alloc(size_t(std::max<int64_t>(A->num_rows, 1))));
  CX_TRY(S->v_cols.alloc(size_t(std::max<int64_t>(2 * G.ne, 1))));
  CX_TRY(G.block_diag_inverse(SEL_E, 0, A->nelim, D, S->ete_inv.p, false));
  double* tmp_e = S->v_cols.p;
  double* tmp_e2 = S->v_cols.p + G.ne;
  if (x && y) {
    GenericSchurOp op;
    op.G = &G; op.D = D; op.tmp_rows = S->v_rows.p; op.tmp_e = tmp_e; op.tmp_e2 = tmp_e2; op.ete_inv = S->ete_inv.p;
    CX_TRY(op.apply(x, y));
  }
  if (rhs && b) {
    CX_TRY(G.zero(tmp_e, G.ne));
    CX_TRY(G.mult(SEL_E, true, b, tmp_e));
    CX_TRY(G.blockdiag_mult(0, A->nelim, 0, S->ete_inv.p, tmp_e, tmp_e2));
    CX_TRY(G.zero(S->v_rows.p, A->num_rows));
    CX_TRY(G.mult(SEL_E, false, tmp_e2, S->v_rows.p));
    if (A->num_rows) hipLaunchKernelGGL(kg_sub, dim3(grid_for(A->num_rows, 256)), dim3(256), 0, st, b, (const double*)S->v_rows.p, S->v_rows.p, A->num_rows);
    CX_TRY(G.zero(rhs, G.nf));
    CX_TRY(G.mult(SEL_F, true, S->v_rows.p, rhs));
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
