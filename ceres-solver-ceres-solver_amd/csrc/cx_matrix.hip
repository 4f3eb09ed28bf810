// Device-resident BlockSparseMatrix: structure analysis (host) and the J product
// kernels (gfx950).  Reference: block_sparse_matrix.cc:220-450,
// partitioned_matrix_view_impl.h, detect_structure.cc.
//
// Kernels and the roofline that bounds them (all HBM-bound, fp64, 2 flop / 8 B):
//   k_right_239      y += J x        240 B/row block read+write (24 values, 2 ids, y rmw)
//   k_left_e_239     y_e += E' x     64 B/row block
//   k_cam_ft         y_f  = F' x     164 B/row block from the camera-major copy (+ 16 B gather)
//   k_permute_ft     Ft <- F         288 B/row block, once per value update
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "cx_internal.h"
#include "cx_kernels.h"

// ============================================================ static 239 kernels

// y[2r..2r+1] += E_r x_pt + F_r x_cam        (block_sparse_matrix.cc:239-274)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_right_239(const T* __restrict__ E,
                                                      const T* __restrict__ F,
                                                      const int32_t* __restrict__ row_pt,
                                                      const int32_t* __restrict__ row_cam,
                                                      const double* __restrict__ xe,
                                                      const double* __restrict__ xf,
                                                      double* __restrict__ y, int64_t O, int use_e,
                                                      int use_f, int accumulate, const int* __restrict__ stop) {
  __shared__ double lds[FStage<T>::kLdsDoubles];  // 18 KB for fp64: 8 workgroups per CU
  if (stop && *stop) return;
  const int64_t r0 = int64_t(blockIdx.x) * kBlock;
  const int nvalid = int(min(int64_t(kBlock), O - r0));
  const int tid = threadIdx.x;
  const int64_t r = r0 + tid;
  // F is staged and consumed before E is requested (measured 17 % faster than requesting
  // everything up front: 6.4 vs 5.5 TB/s on Final-13682, DESIGN.md "A/B notes")
  const bool live = tid < nvalid;
  double acc0 = 0.0, acc1 = 0.0;
  if (use_f) {
    double f[18];
    FStage<T>::run(F + 18 * r0, nvalid, lds, f);
    if (live) {
      const double* xc = xf + 9 * int64_t(row_cam[r]);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const double xv = xc[k];
        acc0 += f[k] * xv;
        acc1 += f[9 + k] * xv;
      }
    }
  }
  if (use_e) {
    double e[6];
    stage_cells<6>(E + 6 * r0, nvalid, lds, e);
    if (live) {
      const double* xp = xe + 3 * int64_t(row_pt[r]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double xv = xp[k];
        acc0 += e[k] * xv;
        acc1 += e[3 + k] * xv;
      }
    }
  }
  if (live) {
    double2* yp = reinterpret_cast<double2*>(y) + r;
    double2 v = accumulate ? *yp : make_double2(0.0, 0.0);
    v.x += acc0;
    v.y += acc1;
    *yp = v;
  }
}

// y_pt (+)= sum over the rows of the point of E_r' x_r   (chunk-aligned tiles)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_left_e_239(const T* __restrict__ E,
                                                       const int32_t* __restrict__ tile_row,
                                                       const int32_t* __restrict__ tile_pt,
                                                       const int32_t* __restrict__ pt_start,
                                                       const double* __restrict__ x,
                                                       double* __restrict__ ye, int accumulate,
                                                       const int* __restrict__ stop,
                                                       const double* __restrict__ d = nullptr,
                                                       const double* __restrict__ dx = nullptr) {
  // d != nullptr: + diag(d)^2 dx on top (the LM diagonal of CGNR's (J'J + D'D) x, point part)
  __shared__ double lds[kBlock * 6];
  if (stop && *stop) return;
  __shared__ double w[kBlock * 3];
  __shared__ double red[3 * 4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  const int p0 = tile_pt[t], p1 = tile_pt[t + 1];
  if (r1 - r0 <= kBlock) {
    const int nvalid = r1 - r0;
    double e[6];
    stage_cells<6>(E + 6 * int64_t(r0), nvalid, lds, e);
    if (tid < nvalid) {
      const double2 xv = reinterpret_cast<const double2*>(x)[r0 + tid];
      w[tid * 3 + 0] = e[0] * xv.x + e[3] * xv.y;
      w[tid * 3 + 1] = e[1] * xv.x + e[4] * xv.y;
      w[tid * 3 + 2] = e[2] * xv.x + e[5] * xv.y;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      const int p = p0 + tid;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
      for (int j = pt_start[p] - r0; j < pt_start[p + 1] - r0; ++j) {
        s0 += w[j * 3];
        s1 += w[j * 3 + 1];
        s2 += w[j * 3 + 2];
      }
      double* yp = ye + 3 * int64_t(p);
      if (d) {
        const double* dp = d + 3 * int64_t(p);
        const double* xp = dx + 3 * int64_t(p);
        s0 += dp[0] * dp[0] * xp[0];
        s1 += dp[1] * dp[1] * xp[1];
        s2 += dp[2] * dp[2] * xp[2];
      }
      if (accumulate) { yp[0] += s0; yp[1] += s1; yp[2] += s2; }
      else { yp[0] = s0; yp[1] = s1; yp[2] = s2; }
    }
  } else {
    // one point whose chunk is longer than a tile: strided loop + block reduction
    double s[3] = {0.0, 0.0, 0.0};
    for (int r = r0 + tid; r < r1; r += kBlock) {
      const T* e = E + 6 * int64_t(r);
      const double2 xv = reinterpret_cast<const double2*>(x)[r];
      s[0] += double(e[0]) * xv.x + double(e[3]) * xv.y;
      s[1] += double(e[1]) * xv.x + double(e[4]) * xv.y;
      s[2] += double(e[2]) * xv.x + double(e[5]) * xv.y;
    }
    block_sum<3>(s, red);
    if (tid == 0) {
      double* yp = ye + 3 * int64_t(p0);
      if (d) {
        const double* dp = d + 3 * int64_t(p0);
        const double* xp = dx + 3 * int64_t(p0);
        s[0] += dp[0] * dp[0] * xp[0];
        s[1] += dp[1] * dp[1] * xp[1];
        s[2] += dp[2] * dp[2] * xp[2];
      }
      if (accumulate) { yp[0] += s[0]; yp[1] += s[1]; yp[2] += s[2]; }
      else { yp[0] = s[0]; yp[1] = s[1]; yp[2] = s[2]; }
    }
  }
}

// Ft[k] = F[cam_rows[k]]: camera-major copy of the F cells.  One wavefront moves
// 64 cells: gathered 144-byte reads, contiguous writes.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_permute_ft(const T* __restrict__ F,
                                                       const int32_t* __restrict__ cam_rows,
                                                       T* __restrict__ Ft, int64_t O) {
  using P2 = typename PieceOf<T>::type;
  // 9 lanes per cell (16 B each): 256 threads move 28 cells per pass; simpler: thread per 16-byte piece
  const int64_t piece = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (piece >= O * 9) return;
  const int64_t k = piece / 9;
  const int part = int(piece - k * 9);
  const int64_t r = cam_rows[k];
  reinterpret_cast<P2*>(Ft)[piece] = reinterpret_cast<const P2*>(F)[r * 9 + part];
}

__global__ void k_to_float(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    const double2 v = *reinterpret_cast<const double2*>(src + i);
    *reinterpret_cast<float2*>(dst + i) = make_float2(float(v.x), float(v.y));
  } else if (i < n) {
    dst[i] = float(src[i]);
  }
}

// Camera-major pass: one workgroup per segment (<= kSegRows rows of ONE camera),
// partial[seg][0..8] = sum over the segment of Ft_r' t_row(r).
#ifndef CX_CAM_FT_OCCUPANCY
#define CX_CAM_FT_OCCUPANCY 4
#endif
template <typename T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? CX_CAM_FT_OCCUPANCY + 1 : CX_CAM_FT_OCCUPANCY) void k_cam_ft(const T* __restrict__ Ft,
                                                   const int32_t* __restrict__ cam_rows,
                                                   const int32_t* __restrict__ seg_begin,
                                                   const double* __restrict__ t,
                                                   double* __restrict__ partial, const int* __restrict__ stop,
                                                   int num_segs) {
  __shared__ double lds[FStage<T>::kLdsDoubles];
  __shared__ double red[9 * 4];
  if (stop && *stop) return;
  const int s = xcd_segment(num_segs < 0 ? -num_segs : num_segs), tid = threadIdx.x;  // (num_segs < 0: A/B, the unpipelined loop)
  if (s < 0) return;
  const int b = seg_begin[s], e = seg_begin[s + 1];
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  // Software pipeline over the segment's passes (round 2, as in k_cam_init): the cells of pass k + 1, its row ids and the
  // t' values they point to (a two-hop chain) are requested before the arithmetic of pass k.
  typename FStage<T>::Pieces v;
  double2 tv_next = make_double2(0.0, 0.0);
  const bool pipe = num_segs > 0;
  if (pipe) {
    FStage<T>::load(Ft + 18 * int64_t(b), min(kBlock, e - b), v);
    if (b + tid < e) tv_next = reinterpret_cast<const double2*>(t)[cam_rows[b + tid]];
  }
  for (int k0 = b; k0 < e; k0 += kBlock) {
    const int nvalid = min(kBlock, e - k0);
    double f[18];
    if (!pipe) {
      FStage<T>::load(Ft + 18 * int64_t(k0), nvalid, v);
    }
    FStage<T>::exchange(v, lds, f);
    if (!pipe && tid < nvalid) tv_next = reinterpret_cast<const double2*>(t)[cam_rows[k0 + tid]];
    const double2 tv = tv_next;
    if (pipe && k0 + kBlock < e) {
      FStage<T>::load(Ft + 18 * int64_t(k0 + kBlock), min(kBlock, e - k0 - kBlock), v);
      if (k0 + kBlock + tid < e) tv_next = reinterpret_cast<const double2*>(t)[cam_rows[k0 + kBlock + tid]];
    }
    if (tid < nvalid) {
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[k] += f[k] * tv.x + f[9 + k] * tv.y;
    }
  }
  // (13 shuffles per wavefront instead of 54: a segment of a mid-size problem is one pass, and then the reduction was half
  // of the kernel's instructions)
  block_sum_store_multi<9>(acc, red, partial + int64_t(s) * 9);
}

// y_f[9c + k] (+)= sum of the camera's segment partials, in segment order (deterministic),
// optionally + d[9c+k]^2 * x[9c+k]
__global__ void k_cam_reduce9(const double* __restrict__ partial, const int32_t* __restrict__ cam_seg_start,
                              double* __restrict__ yf, int C, int accumulate,
                              const double* __restrict__ d, const double* __restrict__ x,
                              const int* __restrict__ stop) {
  if (stop && *stop) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 9) return;
  const int c = i / 9, k = i - c * 9;
  double s = 0.0;
  for (int sg = cam_seg_start[c]; sg < cam_seg_start[c + 1]; ++sg) s += partial[int64_t(sg) * 9 + k];
  if (d) s += d[i] * d[i] * x[i];
  yf[i] = accumulate ? yf[i] + s : s;
}

// x_e[3p+k] = sum over the chunk of E(:,k)^2 ; x_f via the camera-major copy
__global__ __launch_bounds__(kBlock) void k_sqnorm_e_239(const double* __restrict__ E,
                                                         const int32_t* __restrict__ pt_start,
                                                         double* __restrict__ xe, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int r = pt_start[p]; r < pt_start[p + 1]; ++r) {
    const double* e = E + 6 * int64_t(r);
    s0 += e[0] * e[0] + e[3] * e[3];
    s1 += e[1] * e[1] + e[4] * e[4];
    s2 += e[2] * e[2] + e[5] * e[5];
  }
  xe[3 * int64_t(p)] = s0;
  xe[3 * int64_t(p) + 1] = s1;
  xe[3 * int64_t(p) + 2] = s2;
}

__global__ __launch_bounds__(kBlock) void k_cam_sqnorm(const double* __restrict__ Ft,
                                                       const int32_t* __restrict__ seg_begin,
                                                       double* __restrict__ partial) {
  __shared__ double lds[kBlock * 18];
  __shared__ double red[9 * 4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int b = seg_begin[s], e = seg_begin[s + 1];
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int k0 = b; k0 < e; k0 += kBlock) {
    const int nvalid = min(kBlock, e - k0);
    double f[18];
    stage_cells<18>(Ft + 18 * int64_t(k0), nvalid, lds, f);
    if (tid < nvalid) {
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[k] += f[k] * f[k] + f[9 + k] * f[9 + k];
    }
  }
  block_sum_store<9>(acc, red, partial + int64_t(s) * 9);
}

// J <- J diag(scale)   (block_sparse_matrix.cc:403-450): cells in through LDS, scaled in
// registers, out through LDS -- coalesced both ways (416 B per residual block incl. ids).  The scaled F cells
// also go to their places in the camera-major copy Ft (+148 B per residual block), which is therefore current
// when the kernel ends: the LM iteration needs no k_permute_ft pass (292 B per residual block) afterwards.
// (Round 2 A/B: issuing every load of the tile before the first store, the cooperative gather of the camera block's
// scales (gather_by_row), half-size staging and cell numbers passed through LDS -- what sped up k_bal_evaluate -- left
// this kernel at the same 3.45 ms on the Final shape: 15.6 GB at 4.5 TB/s, against 4.7 TB/s for a plain copy.)
__global__ __launch_bounds__(kBlock) void k_scale_239(double* __restrict__ E, double* __restrict__ F,
                                                      const int32_t* __restrict__ row_pt,
                                                      const int32_t* __restrict__ row_cam,
                                                      const double* __restrict__ scale, int64_t O,
                                                      int64_t xf_off, double* __restrict__ Ft,
                                                      const int32_t* __restrict__ cam_pos, int num_tiles) {
  __shared__ double lds[kBlock * 18];
  // num_tiles > 0: XCD-aware tile map (xcd_segment) -- consecutive tiles run on ONE XCD, so the partially written
  // cache lines of Ft (a 144-byte cell straddles two) are completed in that XCD's L2 by the neighbouring tiles
  const int tile = num_tiles > 0 ? xcd_segment(num_tiles) : int(blockIdx.x);
  if (tile < 0) return;
  const int64_t r0 = int64_t(tile) * kBlock;
  const int nvalid = int(min(int64_t(kBlock), O - r0));
  const int tid = threadIdx.x;
  const bool live = tid < nvalid;
  {
    double f[18];
    stage_cells<18>(F + 18 * r0, nvalid, lds, f);
    if (live) {
      const double* sc = scale + xf_off + 9 * int64_t(row_cam[r0 + tid]);
#pragma unroll
      for (int k = 0; k < 9; ++k) { const double sv = sc[k]; f[k] *= sv; f[9 + k] *= sv; }
    }
    unstage_f_cells_two(F + 18 * r0, Ft, cam_pos + r0, nvalid, lds, f);
  }
  {
    double e[6];
    stage_cells<6>(E + 6 * r0, nvalid, lds, e);
    if (live) {
      const double* sp = scale + 3 * int64_t(row_pt[r0 + tid]);
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double sv = sp[k]; e[k] *= sv; e[3 + k] *= sv; }
    }
    unstage_cells<6>(E + 6 * r0, nvalid, lds, e);
  }
}

// ============================================================== generic kernels
// Dynamic block sizes: one thread per row block, atomics for transposed products.
// This is the fallback the reference's Eigen::Dynamic template instantiations are.

// sel: 0 all cells, 1 the e cell (first cell of rows r < nrows_e), 2 the f cells; col_off is
// subtracted from column positions (num_cols_e for products with the F half alone)
__device__ __forceinline__ void kg_cell_range(const int32_t* __restrict__ rcb, int r, int nrows_e, int sel, int& b, int& e) {
  b = rcb[r];
  e = rcb[r + 1];
  const bool has_e = r < nrows_e;
  if (sel == 1) e = has_e ? min(e, b + 1) : b;
  else if (sel == 2 && has_e) b = min(e, b + 1);
}

__global__ void kg_right(const cx_block* __restrict__ rows, const cx_block* __restrict__ cols,
                         const int32_t* __restrict__ rcb, const cx_cell* __restrict__ cells,
                         const double* __restrict__ values, const double* __restrict__ x,
                         double* __restrict__ y, int R, int nrows_e, int sel, int col_off) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int rs = rows[r].size, rp = rows[r].position;
  int cb, ce;
  kg_cell_range(rcb, r, nrows_e, sel, cb, ce);
  for (int c = cb; c < ce; ++c) {
    const cx_cell cell = cells[c];
    const int cs = cols[cell.block_id].size, cp = cols[cell.block_id].position - col_off;
    const double* m = values + cell.position;
    for (int i = 0; i < rs; ++i) {
      double s = 0.0;
      for (int j = 0; j < cs; ++j) s += m[i * cs + j] * x[cp + j];
      y[rp + i] += s;
    }
  }
}

__global__ void kg_left(const cx_block* __restrict__ rows, const cx_block* __restrict__ cols,
                        const int32_t* __restrict__ rcb, const cx_cell* __restrict__ cells,
                        const double* __restrict__ values, const double* __restrict__ x,
                        double* __restrict__ y, int R, int nrows_e, int sel, int col_off) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int rs = rows[r].size, rp = rows[r].position;
  int cb, ce;
  kg_cell_range(rcb, r, nrows_e, sel, cb, ce);
  for (int c = cb; c < ce; ++c) {
    const cx_cell cell = cells[c];
    const int cs = cols[cell.block_id].size, cp = cols[cell.block_id].position - col_off;
    const double* m = values + cell.position;
    for (int j = 0; j < cs; ++j) {
      double s = 0.0;
      for (int i = 0; i < rs; ++i) s += m[i * cs + j] * x[rp + i];
      atomicAdd(&y[cp + j], s);
    }
  }
}

// y += A_sel' x as a GATHER over the transposed index (round 3; the atomics form kg_left took 10 x the time of y += A x).
// The entries of a column block (its cells in ascending row order) are cut into segments of at most kTransposeSegment; one
// wavefront per segment: the 64 lanes form 64 / cs_pad groups of cs_pad lanes (cs_pad = the block's size rounded up to a
// power of two), lane j of group g sums column j over the entries g, g + groups, ... of the segment, the group sums are
// added by a fixed butterfly.  A column with one segment is finished there; the others leave a partial sum per segment and
// kg_left_finish adds them in segment order.  No atomics, the same bits every time.
// Block sizes: a column block of up to 64 scalars fits the lane groups (cs_pad <= 64: at least one group); the per-segment
// partial sums are `stride` apart, stride = the largest padded column block of the matrix (16 for everything the reference
// instantiates).  Matrices with a wider column block take the scatter kernel kg_left (cxk_generic_left_multiply).  A
// transposed entry packs its row block's size into the low 30 bits and the e-cell flag above them.
constexpr int kTransposeSegment = 128;
constexpr int kMetaEBit = 30;
constexpr int kMetaSizeMask = (1 << kMetaEBit) - 1;
constexpr int kGatherMaxColSize = 64;
__global__ __launch_bounds__(256) void kg_left_gather(const cx_block* __restrict__ cols, const int32_t* __restrict__ seg_begin,
                                                      const int32_t* __restrict__ seg_col, const int32_t* __restrict__ col_seg,
                                                      const int32_t* __restrict__ t_pos, const int32_t* __restrict__ t_rp,
                                                      const int32_t* __restrict__ t_meta, const double* __restrict__ values,
                                                      const double* __restrict__ x, double* __restrict__ y, double* __restrict__ partial,
                                                      int num_segments, int first_col, int end_col, int sel, int col_off, int stride) {
  const int s = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
  if (s >= num_segments) return;
  const int cb = seg_col[s];
  if (cb < first_col || cb >= end_col) return;
  const int lane = threadIdx.x & 63;
  const int cs = cols[cb].size;  // <= kGatherMaxColSize (checked on the host)
  int cs_pad = 1;
  while (cs_pad < cs) cs_pad <<= 1;
  const int groups = 64 / cs_pad, g = lane / cs_pad, j = lane - g * cs_pad;
  double acc = 0.0;
  if (j < cs) {
    for (int k = seg_begin[s] + g; k < seg_begin[s + 1]; k += groups) {
      const int meta = t_meta[k];
      const bool is_e_cell = (meta >> kMetaEBit) != 0;
      if ((sel == 1 && !is_e_cell) || (sel == 2 && is_e_cell)) continue;
      const int rs = meta & kMetaSizeMask;
      const double* m = values + t_pos[k] + j;
      const double* xr = x + t_rp[k];
      double sum = 0.0;
      for (int i = 0; i < rs; ++i) sum += m[i * cs] * xr[i];
      acc += sum;
    }
  }
  for (int off = cs_pad; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
  if (g == 0 && j < cs) {
    if (col_seg[cb + 1] - col_seg[cb] == 1) y[cols[cb].position - col_off + j] += acc;
    else partial[int64_t(s) * stride + j] = acc;
  }
}
__global__ void kg_left_finish(const cx_block* __restrict__ cols, const int32_t* __restrict__ col_seg, const double* __restrict__ partial,
                               double* __restrict__ y, int first_col, int end_col, int col_off, int stride) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t c = t / stride;
  const int j = int(t - c * stride);
  if (c >= end_col - first_col) return;
  const int cb = first_col + int(c);
  if (j >= cols[cb].size) return;
  const int s0 = col_seg[cb], s1 = col_seg[cb + 1];
  if (s1 - s0 <= 1) return;
  double v = 0.0;
  for (int s = s0; s < s1; ++s) v += partial[int64_t(s) * stride + j];
  y[cols[cb].position - col_off + j] += v;
}

__global__ void kg_sqnorm(const cx_block* __restrict__ rows, const cx_block* __restrict__ cols,
                          const int32_t* __restrict__ rcb, const cx_cell* __restrict__ cells,
                          const double* __restrict__ values, double* __restrict__ x, int R) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int rs = rows[r].size;
  for (int c = rcb[r]; c < rcb[r + 1]; ++c) {
    const cx_cell cell = cells[c];
    const int cs = cols[cell.block_id].size, cp = cols[cell.block_id].position;
    const double* m = values + cell.position;
    for (int j = 0; j < cs; ++j) {
      double s = 0.0;
      for (int i = 0; i < rs; ++i) s += m[i * cs + j] * m[i * cs + j];
      atomicAdd(&x[cp + j], s);
    }
  }
}

__global__ void kg_scale(const cx_block* __restrict__ rows, const cx_block* __restrict__ cols,
                         const int32_t* __restrict__ rcb, const cx_cell* __restrict__ cells,
                         double* __restrict__ values, const double* __restrict__ scale, int R) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int rs = rows[r].size;
  for (int c = rcb[r]; c < rcb[r + 1]; ++c) {
    const cx_cell cell = cells[c];
    const int cs = cols[cell.block_id].size, cp = cols[cell.block_id].position;
    double* m = values + cell.position;
    for (int i = 0; i < rs; ++i)
      for (int j = 0; j < cs; ++j) m[i * cs + j] *= scale[cp + j];
  }
}

// ============================================================= host: structure

static int grid_for(int64_t n, int block) { return int((n + block - 1) / block); }

// DetectStructure (detect_structure.cc:39-120); -1 == Eigen::Dynamic
static void DetectStructureHost(const cx_block_structure* bs, int nelim, int32_t* rsz, int32_t* esz, int32_t* fsz) {
  *rsz = *esz = *fsz = 0;
  for (int r = 0; r < bs->num_row_blocks; ++r) {
    const int nc = bs->row_cell_begin[r + 1] - bs->row_cell_begin[r];
    if (nc == 0) continue;
    const cx_cell* cells = bs->cells + bs->row_cell_begin[r];
    if (cells[0].block_id >= nelim) break;
    if (*rsz == 0) *rsz = bs->row_blocks[r].size;
    else if (*rsz != -1 && *rsz != bs->row_blocks[r].size) *rsz = -1;
    const int e = bs->col_blocks[cells[0].block_id].size;
    if (*esz == 0) *esz = e;
    else if (*esz != -1 && *esz != e) *esz = -1;
    if (nc > 1) {
      if (*fsz == 0) *fsz = bs->col_blocks[cells[1].block_id].size;
      for (int c = 1; c < nc && *fsz != -1; ++c)
        if (*fsz != bs->col_blocks[cells[c].block_id].size) *fsz = -1;
    }
    if (*rsz == -1 && *esz == -1 && *fsz == -1) break;
  }
}

// Is this exactly the bundle-adjustment layout the <2,3,9> kernels assume?
static bool Detect239(const cx_matrix* A, int32_t* P_out, int32_t* C_out) {
  const int R = A->R, Cb = A->Cb;
  if (R == 0 || Cb < 2) return false;
  int P = 0;
  while (P < Cb && A->cols[P].size == 3) ++P;
  const int C = Cb - P;
  if (P == 0 || C == 0) return false;
  if (A->nelim != 0 && A->nelim != P) return false;
  for (int j = 0; j < P; ++j) if (A->cols[j].position != 3 * j) return false;
  for (int i = 0; i < C; ++i) if (A->cols[P + i].size != 9 || int64_t(A->cols[P + i].position) != 3ll * P + 9ll * i) return false;
  const int64_t O = R;
  if (24 * O >= (int64_t(1) << 31)) return false;
  std::vector<int32_t> last_pt(C, -1);
  int prev_pt = 0;
  for (int r = 0; r < R; ++r) {
    if (A->rows[r].size != 2 || A->rows[r].position != 2 * r) return false;
    if (A->rcb[r] != 2 * r || A->rcb[r + 1] != 2 * r + 2) return false;
    const cx_cell& ce = A->cells[2 * r];
    const cx_cell& cf = A->cells[2 * r + 1];
    if (ce.block_id < 0 || ce.block_id >= P || cf.block_id < P || cf.block_id >= Cb) return false;
    if (int64_t(ce.position) != 6ll * r || int64_t(cf.position) != 6 * O + 18ll * r) return false;
    if (ce.block_id < prev_pt) return false;  // chunks must be contiguous and ascending
    prev_pt = ce.block_id;
    const int cam = cf.block_id - P;
    if (last_pt[cam] == ce.block_id) return false;  // a camera seeing a point twice: generic path
    last_pt[cam] = ce.block_id;
  }
  *P_out = P;
  *C_out = C;
  return true;
}

static int Build239(cx_matrix* A) {
  const int64_t O = A->R;
  const int P = A->P, C = A->C;
  hipStream_t st = A->ctx->stream;
  std::vector<int32_t> row_pt(O), row_cam(O), pt_start(P + 1, 0);
  for (int64_t r = 0; r < O; ++r) {
    row_pt[r] = A->cells[2 * r].block_id;
    row_cam[r] = A->cells[2 * r + 1].block_id - P;
    pt_start[row_pt[r] + 1]++;
  }
  std::partial_sum(pt_start.begin(), pt_start.end(), pt_start.begin());
  // chunk-aligned tiles
  std::vector<int32_t> tile_row{0}, tile_pt{0};
  {
    int p = 0;
    while (p < P) {
      const int p_begin = p;
      const int r_begin = pt_start[p];
      int rows = pt_start[p + 1] - pt_start[p];
      ++p;
      if (rows <= kTileRows) {
        while (p < P && p - p_begin < kTileRows && (pt_start[p + 1] - r_begin) <= kTileRows) ++p;
      }
      tile_row.push_back(pt_start[p]);
      tile_pt.push_back(p);
      if (rows > kTileRows) A->has_big_tiles = true;
    }
  }
  A->num_tiles = int(tile_row.size()) - 1;
  // camera-major order: stable counting sort by camera (ascending row inside a camera,
  // the order of the reference's transpose block structure, block_sparse_matrix.cc:784-808)
  std::vector<int32_t> cam_start(C + 1, 0), cam_rows(O);
  for (int64_t r = 0; r < O; ++r) cam_start[row_cam[r] + 1]++;
  std::partial_sum(cam_start.begin(), cam_start.end(), cam_start.begin());
  {
    std::vector<int32_t> cur(cam_start.begin(), cam_start.end() - 1);
    for (int64_t r = 0; r < O; ++r) cam_rows[cur[row_cam[r]]++] = int32_t(r);
  }
  // segment length: long enough that the per-segment block reduction is amortised over several
  // rows per thread, short enough that the launch still has far more workgroups than CUs
  static const int seg_env = [] { const char* v = getenv("CX_SEG_ROWS"); return v ? atoi(v) : 0; }();
  int seg_rows = seg_env > 0 ? seg_env : kSegRows;
  if (seg_env <= 0) {
    while (seg_rows < 4096 && O / (2 * seg_rows) > int64_t(A->ctx->num_cus) * 16) seg_rows *= 2;
  }
  std::vector<int32_t> seg_begin, seg_cam, cam_seg_start(C + 1, 0);
  for (int c = 0; c < C; ++c) {
    cam_seg_start[c] = int32_t(seg_cam.size());
    // equal parts (no short tail segment): nseg = ceil(len / seg_rows)
    const int len = cam_start[c + 1] - cam_start[c];
    const int nseg = (len + seg_rows - 1) / seg_rows;
    int b = cam_start[c];
    for (int k = 0; k < nseg; ++k) {
      seg_begin.push_back(b);
      seg_cam.push_back(c);
      b += len / nseg + (k < len % nseg ? 1 : 0);
    }
  }
  cam_seg_start[C] = int32_t(seg_cam.size());
  seg_begin.push_back(int32_t(O));
  A->num_segs = int(seg_cam.size());
  CX_TRY(A->d_row_pt.upload(row_pt, st));
  CX_TRY(A->d_row_cam.upload(row_cam, st));
  CX_TRY(A->d_pt_start.upload(pt_start, st));
  CX_TRY(A->d_tile_row.upload(tile_row, st));
  CX_TRY(A->d_tile_pt.upload(tile_pt, st));
  CX_TRY(A->d_cam_rows.upload(cam_rows, st));
  {
    std::vector<int32_t> cam_pos(static_cast<size_t>(O));
    for (int64_t k = 0; k < O; ++k) cam_pos[size_t(cam_rows[size_t(k)])] = int32_t(k);
    CX_TRY(A->d_cam_pos.upload(cam_pos, st));
  }
  CX_TRY(A->d_seg_begin.upload(seg_begin, st));
  CX_TRY(A->d_seg_cam.upload(seg_cam, st));
  CX_TRY(A->d_cam_seg_start.upload(cam_seg_start, st));
  CX_TRY(A->d_partials.alloc(size_t(std::max(1, A->num_segs)) * 81));
  return CX_OK;
}

int cx_matrix_ensure_ft(cx_matrix* A) {
  if (!A->is239 || A->ft_valid) return CX_OK;
  CX_TRY(A->d_Ft.alloc(size_t(A->O) * 18));
  const int64_t pieces = A->O * 9;
  hipLaunchKernelGGL(k_permute_ft<double>, dim3(grid_for(pieces, kBlock)), dim3(kBlock), 0, A->ctx->stream,
                     (const double*)(A->d_values.p + 6 * A->O), A->d_cam_rows.p, A->d_Ft.p, A->O);
  CX_HIP(hipGetLastError());
  A->ft_valid = true;
  return CX_OK;
}

// fp32 copies of the values (row-major E, F and the camera-major Ft) for mixed-precision CGNR
int cx_matrix_ensure_f32(cx_matrix* A) {
  if (!A->is239 || A->f32_valid) return CX_OK;
  hipStream_t st = A->ctx->stream;
  CX_TRY(A->d_vals32.alloc(size_t(A->nnz)));
  CX_TRY(A->d_Ft32.alloc(size_t(A->O) * 18));
  hipLaunchKernelGGL(k_to_float, dim3(grid_for((A->nnz + 1) / 2, 256)), dim3(256), 0, st, (const double*)A->d_values.p,
                     A->d_vals32.p, A->nnz);
  hipLaunchKernelGGL(k_permute_ft<float>, dim3(grid_for(A->O * 9, kBlock)), dim3(kBlock), 0, st,
                     (const float*)(A->d_vals32.p + 6 * A->O), A->d_cam_rows.p, A->d_Ft32.p, A->O);
  CX_HIP(hipGetLastError());
  A->f32_valid = true;
  return CX_OK;
}

// ------------------------------------------------------------ product drivers
// per-segment partial sums of F' t into A->d_partials (9 per segment); cxk_ft_multiply or a fused
// consumer (cx_solver.hip: k_cam_reduce9_dot) adds them up per camera in segment order
int cxk_ft_partials(cx_matrix* A, const double* t) {
  CX_TRY(cx_matrix_ensure_ft(A));
  hipStream_t st = A->ctx->stream;
  if (A->num_segs > 0) {
    static const bool plain = std::getenv("CX_CAM_FT_PLAIN") != nullptr;  // A/B switch: the unpipelined loop
    const int segs = plain ? -A->num_segs : A->num_segs;
    if (A->use_f32)
      hipLaunchKernelGGL(k_cam_ft<float>, dim3(xcd_grid(A->num_segs)), dim3(kBlock), 0, st, (const float*)A->d_Ft32.p,
                         A->d_cam_rows.p, A->d_seg_begin.p, t, A->d_partials.p, A->stop, segs);
    else
      hipLaunchKernelGGL(k_cam_ft<double>, dim3(xcd_grid(A->num_segs)), dim3(kBlock), 0, st, (const double*)A->d_Ft.p,
                         A->d_cam_rows.p, A->d_seg_begin.p, t, A->d_partials.p, A->stop, segs);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxk_ft_multiply(cx_matrix* A, const double* t, double* yf, bool accumulate, const double* d_f, const double* x_f) {
  CX_TRY(cxk_ft_partials(A, t));
  hipLaunchKernelGGL(k_cam_reduce9, dim3(grid_for(int64_t(A->C) * 9, 256)), dim3(256), 0, A->ctx->stream,
                     A->d_partials.p, A->d_cam_seg_start.p, yf, A->C, accumulate ? 1 : 0, d_f, x_f, A->stop);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxk_right_multiply(cx_matrix* A, const double* x, double* y, bool accumulate) {
  hipStream_t st = A->ctx->stream;
  const int acc = accumulate ? 1 : 0;
  if (!accumulate && !A->is239) CX_HIP(hipMemsetAsync(y, 0, size_t(A->num_rows) * sizeof(double), st));
  if (A->is239) {
    if (A->use_f32)
      hipLaunchKernelGGL(k_right_239<float>, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, (const float*)A->d_vals32.p,
                         (const float*)(A->d_vals32.p + 6 * A->O), A->d_row_pt.p, A->d_row_cam.p, x, x + 3 * int64_t(A->P), y,
                         A->O, 1, 1, acc, A->stop);
    else
      hipLaunchKernelGGL(k_right_239<double>, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, (const double*)A->d_values.p,
                         (const double*)(A->d_values.p + 6 * A->O), A->d_row_pt.p, A->d_row_cam.p, x, x + 3 * int64_t(A->P), y,
                         A->O, 1, 1, acc, A->stop);
  } else if (A->R > 0) {
    hipLaunchKernelGGL(kg_right, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p,
                       A->d_rcb.p, A->d_cells.p, A->d_values.p, x, y, A->R, A->num_row_blocks_e, 0, 0);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// accumulate = false: y = A'x (the static kernels write every entry themselves: their tiles cover all points, empty ones
// included, and the camera reduction writes every camera; the dynamic-size path zeroes y first).
// d != nullptr: y (+)= A'x + diag(d)^2 dx in the same launches (static layout; *folded says whether it happened).
int cxk_left_multiply(cx_matrix* A, const double* x, double* y, bool accumulate, const double* d, const double* dx, bool* folded) {
  hipStream_t st = A->ctx->stream;
  if (folded) *folded = false;
  if (!accumulate && !A->is239) CX_HIP(hipMemsetAsync(y, 0, size_t(A->num_cols) * sizeof(double), st));
  if (A->is239) {
    const int acc = accumulate ? 1 : 0;
    const int64_t ne = 3 * int64_t(A->P);
    if (A->use_f32)
      hipLaunchKernelGGL(k_left_e_239<float>, dim3(A->num_tiles), dim3(kBlock), 0, st, (const float*)A->d_vals32.p,
                         A->d_tile_row.p, A->d_tile_pt.p, A->d_pt_start.p, x, y, acc, A->stop, d, dx);
    else
      hipLaunchKernelGGL(k_left_e_239<double>, dim3(A->num_tiles), dim3(kBlock), 0, st, (const double*)A->d_values.p,
                         A->d_tile_row.p, A->d_tile_pt.p, A->d_pt_start.p, x, y, acc, A->stop, d, dx);
    CX_TRY(cxk_ft_multiply(A, x, y + ne, accumulate, d ? d + ne : nullptr, d ? dx + ne : nullptr));
    if (folded && d) *folded = true;
  } else if (A->R > 0) {
    static const bool atomics = std::getenv("CX_GENERIC_ATOMICS") != nullptr;  // A/B switch: round 1's scatter form
    if (atomics)
      hipLaunchKernelGGL(kg_left, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p,
                         A->d_rcb.p, A->d_cells.p, A->d_values.p, x, y, A->R, A->num_row_blocks_e, 0, 0);
    else
      CX_TRY(cxk_generic_left_multiply(A, 0, 0, x, y));
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxk_squared_column_norm(cx_matrix* A, double* x) {
  hipStream_t st = A->ctx->stream;
  if (A->is239) {
    CX_TRY(cx_matrix_ensure_ft(A));
    hipLaunchKernelGGL(k_sqnorm_e_239, dim3(grid_for(A->P, kBlock)), dim3(kBlock), 0, st, A->d_values.p,
                       A->d_pt_start.p, x, A->P);
    if (A->num_segs > 0)
      hipLaunchKernelGGL(k_cam_sqnorm, dim3(A->num_segs), dim3(kBlock), 0, st, A->d_Ft.p, A->d_seg_begin.p,
                         A->d_partials.p);
    hipLaunchKernelGGL(k_cam_reduce9, dim3(grid_for(int64_t(A->C) * 9, 256)), dim3(256), 0, st,
                       A->d_partials.p, A->d_cam_seg_start.p, x + 3 * int64_t(A->P), A->C, 0,
                       (const double*)nullptr, (const double*)nullptr, (const int*)nullptr);
  } else {
    CX_HIP(hipMemsetAsync(x, 0, size_t(A->num_cols) * sizeof(double), st));
    if (A->R > 0)
      hipLaunchKernelGGL(kg_sqnorm, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p,
                         A->d_rcb.p, A->d_cells.p, A->d_values.p, x, A->R);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxk_scale_columns(cx_matrix* A, const double* scale) {
  hipStream_t st = A->ctx->stream;
  if (A->is239) {
    static const bool emit_ft = std::getenv("CX_NO_FT_EMIT") == nullptr;  // A/B switch: separate k_permute_ft pass
    if (emit_ft) CX_TRY(A->d_Ft.alloc(size_t(A->O) * 18));
    static const bool xcd_tiles = std::getenv("CX_NO_XCD_TILES") == nullptr;
    const int ntiles = grid_for(A->O, kBlock);
    hipLaunchKernelGGL(k_scale_239, dim3(xcd_tiles ? xcd_grid(ntiles) : ntiles), dim3(kBlock), 0, st, A->d_values.p,
                       A->d_values.p + 6 * A->O, A->d_row_pt.p, A->d_row_cam.p, scale, A->O,
                       3 * int64_t(A->P), emit_ft ? A->d_Ft.p : (double*)nullptr, (const int32_t*)A->d_cam_pos.p,
                       xcd_tiles ? ntiles : 0);
    A->ft_valid = emit_ft;
  } else if (A->R > 0) {
    hipLaunchKernelGGL(kg_scale, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p,
                       A->d_rcb.p, A->d_cells.p, A->d_values.p, scale, A->R);
    A->ft_valid = false;
  }
  A->f32_valid = false;
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// The block diagonal of A_sel' A_sel over the transposed index: one wavefront per segment, lane l owns the entries l, l + 64,
// ... of the cs x cs block and walks the segment's cells in row order; single-segment columns are written directly, the
// others through per-segment partial blocks summed in segment order (kg_blockdiag_finish).
__global__ __launch_bounds__(256) void kg_blockdiag_gather(const cx_block* __restrict__ cols, const int32_t* __restrict__ seg_begin,
                                                           const int32_t* __restrict__ seg_col, const int32_t* __restrict__ col_seg,
                                                           const int32_t* __restrict__ t_pos, const int32_t* __restrict__ t_meta,
                                                           const double* __restrict__ values, const int64_t* __restrict__ blk_off,
                                                           int64_t off0, double* __restrict__ blocks, double* __restrict__ partial,
                                                           int num_segments, int first_col, int end_col, int sel) {
  const int s = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
  if (s >= num_segments) return;
  const int cb = seg_col[s];
  if (cb < first_col || cb >= end_col) return;
  const int lane = threadIdx.x & 63;
  const int cs = cols[cb].size, n = cs * cs;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // cs <= 16: at most four entries per lane
  for (int k = seg_begin[s]; k < seg_begin[s + 1]; ++k) {
    const int meta = t_meta[k];
    const bool is_e_cell = (meta >> kMetaEBit) != 0;
    if ((sel == 1 && !is_e_cell) || (sel == 2 && is_e_cell)) continue;
    const int rs = meta & kMetaSizeMask;
    const double* m = values + t_pos[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      if (e < n) {
        const int a = e / cs, b = e - a * cs;
        double sum = 0.0;
        for (int i = 0; i < rs; ++i) sum += m[i * cs + a] * m[i * cs + b];
        acc[q] += sum;
      }
    }
  }
  const bool single = col_seg[cb + 1] - col_seg[cb] == 1;
  double* out = single ? blocks + (blk_off[cb] - off0) : partial + int64_t(s) * 256;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = lane + 64 * q;
    if (e < n) out[e] = acc[q];
  }
}
__global__ void kg_blockdiag_finish(const cx_block* __restrict__ cols, const int32_t* __restrict__ col_seg, const double* __restrict__ partial,
                                    const int64_t* __restrict__ blk_off, int64_t off0, double* __restrict__ blocks, int first_col, int end_col) {
  const int cb = first_col + int(blockIdx.x);
  if (cb >= end_col) return;
  const int s0 = col_seg[cb], s1 = col_seg[cb + 1];
  const int cs = cols[cb].size, n = cs * cs;
  if (s1 - s0 == 1) return;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    double v = 0.0;
    for (int s = s0; s < s1; ++s) v += partial[int64_t(s) * 256 + e];
    blocks[blk_off[cb] - off0 + e] = v;  // (a column without cells: zero)
  }
}

int cxk_build_transpose(cx_matrix* A) {
  if (A->transpose_ready) return CX_OK;
  const size_t ncells = A->cells.size();
  std::vector<int32_t> start(size_t(A->Cb) + 1, 0), pos(ncells), rp(ncells), meta(ncells);
  for (const cx_cell& c : A->cells) start[size_t(c.block_id) + 1]++;
  for (int c = 0; c < A->Cb; ++c) start[size_t(c) + 1] += start[size_t(c)];
  std::vector<int32_t> fill(start.begin(), start.end() - 1);
  for (int r = 0; r < A->R; ++r)  // ascending rows: every column's cells come out in row order
    for (int c = A->rcb[size_t(r)]; c < A->rcb[size_t(r) + 1]; ++c) {
      const int32_t slot = fill[size_t(A->cells[size_t(c)].block_id)]++;
      pos[size_t(slot)] = A->cells[size_t(c)].position;
      rp[size_t(slot)] = A->rows[size_t(r)].position;
      const bool is_e_cell = r < A->num_row_blocks_e && c == A->rcb[size_t(r)];
      meta[size_t(slot)] = A->rows[size_t(r)].size | (is_e_cell ? 1 << kMetaEBit : 0);
    }
  int32_t max_cs = 1;
  for (const cx_block& c : A->cols) max_cs = std::max(max_cs, c.size);
  A->t_max_col_size = max_cs;
  A->t_stride = 16;  // partial sums of a segment: padded size of the widest column block the gather kernels take
  while (A->t_stride < std::min(max_cs, kGatherMaxColSize)) A->t_stride <<= 1;
  std::vector<int32_t> seg_begin, seg_col, col_seg(size_t(A->Cb) + 1, 0);
  for (int c = 0; c < A->Cb; ++c) {
    col_seg[size_t(c)] = int32_t(seg_col.size());
    for (int32_t k = start[size_t(c)]; k < start[size_t(c) + 1]; k += kTransposeSegment) {
      seg_begin.push_back(k);
      seg_col.push_back(c);
    }
  }
  col_seg[size_t(A->Cb)] = int32_t(seg_col.size());
  seg_begin.push_back(int32_t(ncells));
  hipStream_t st = A->ctx->stream;
  CX_TRY(A->d_t_pos.upload(pos, st));
  CX_TRY(A->d_t_rp.upload(rp, st));
  CX_TRY(A->d_t_meta.upload(meta, st));
  CX_TRY(A->d_t_seg_begin.upload(seg_begin, st));
  CX_TRY(A->d_t_seg_col.upload(seg_col, st));
  CX_TRY(A->d_t_col_seg.upload(col_seg, st));
  A->num_t_segments = int32_t(seg_col.size());
  CX_TRY(A->d_t_partial.alloc(size_t(std::max(A->num_t_segments, 1)) * size_t(A->t_stride)));
  A->transpose_ready = true;
  return CX_OK;
}

int cxk_generic_left_multiply(cx_matrix* A, int sel, int col_off, const double* x, double* y) {
  if (A->R == 0 || A->Cb == 0) return CX_OK;
  CX_TRY(cxk_build_transpose(A));
  // sel 1: only the e-blocks' columns can receive anything, sel 2: only the f-blocks' (and every column of a matrix
  // without e-blocks)
  const int first = sel == 2 ? A->nelim : 0;
  const int end = sel == 1 ? A->nelim : A->Cb;
  if (end <= first || A->num_t_segments == 0) return CX_OK;
  hipStream_t st = A->ctx->stream;
  if (A->t_max_col_size > kGatherMaxColSize) {
    // a column block wider than a wavefront's lane groups: the scatter form (atomic adds; not bitwise repeatable) takes any size
    hipLaunchKernelGGL(kg_left, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p, A->d_rcb.p, A->d_cells.p,
                       A->d_values.p, x, y, A->R, A->num_row_blocks_e, sel, col_off);
    CX_HIP(hipGetLastError());
    return CX_OK;
  }
  hipLaunchKernelGGL(kg_left_gather, dim3(unsigned((A->num_t_segments + 3) / 4)), dim3(256), 0, st, (const cx_block*)A->d_cols.p,
                     (const int32_t*)A->d_t_seg_begin.p, (const int32_t*)A->d_t_seg_col.p, (const int32_t*)A->d_t_col_seg.p,
                     (const int32_t*)A->d_t_pos.p, (const int32_t*)A->d_t_rp.p, (const int32_t*)A->d_t_meta.p, (const double*)A->d_values.p,
                     x, y, A->d_t_partial.p, A->num_t_segments, first, end, sel, col_off, A->t_stride);
  hipLaunchKernelGGL(kg_left_finish, dim3(unsigned((int64_t(end - first) * A->t_stride + 255) / 256)), dim3(256), 0, st, (const cx_block*)A->d_cols.p,
                     (const int32_t*)A->d_t_col_seg.p, (const double*)A->d_t_partial.p, y, first, end, col_off, A->t_stride);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

int cxk_generic_block_diagonal(cx_matrix* A, int sel, int first, int count, int64_t off0, double* blocks) {
  if (count <= 0) return CX_OK;
  CX_TRY(cxk_build_transpose(A));
  for (int c = first; c < first + count; ++c)
    if (A->cols[size_t(c)].size > 16) {  // four entries per lane: blocks of up to 16 x 16
      cx_set_error("block diagonal of a %d-wide column block: the dynamic-size kernels take blocks of up to 16", A->cols[size_t(c)].size);
      return CX_ERR_UNSUPPORTED;
    }
  CX_TRY(A->d_t_partial_blocks.alloc(size_t(std::max(A->num_t_segments, 1)) * 256));
  hipStream_t st = A->ctx->stream;
  if (A->num_t_segments > 0)
    hipLaunchKernelGGL(kg_blockdiag_gather, dim3(unsigned((A->num_t_segments + 3) / 4)), dim3(256), 0, st, (const cx_block*)A->d_cols.p,
                       (const int32_t*)A->d_t_seg_begin.p, (const int32_t*)A->d_t_seg_col.p, (const int32_t*)A->d_t_col_seg.p,
                       (const int32_t*)A->d_t_pos.p, (const int32_t*)A->d_t_meta.p, (const double*)A->d_values.p,
                       (const int64_t*)A->d_blk_off.p, off0, blocks, A->d_t_partial_blocks.p, A->num_t_segments, first, first + count, sel);
  hipLaunchKernelGGL(kg_blockdiag_finish, dim3(unsigned(count)), dim3(64), 0, st, (const cx_block*)A->d_cols.p,
                     (const int32_t*)A->d_t_col_seg.p, (const double*)A->d_t_partial_blocks.p, (const int64_t*)A->d_blk_off.p, off0, blocks,
                     first, first + count);
  CX_HIP(hipGetLastError());
  return CX_OK;
}

// run `body` between two events and keep the device time
template <typename Fn>
static int Timed(cx_matrix* A, Fn body) {
  cx_context* ctx = A->ctx;
  CX_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
  CX_TRY(body());
  CX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
  CX_TRY(cx_event_sync(ctx, ctx->ev[1]));
  CX_HIP(hipEventElapsedTime(&A->last_ms, ctx->ev[0], ctx->ev[1]));
  return CX_OK;
}

// =================================================================== C ABI
extern "C" {

int cx_detect_structure(const cx_block_structure* bs, int32_t nelim, int32_t* r, int32_t* e, int32_t* f) {
  CX_CHECK_ARG(bs && r && e && f);
  DetectStructureHost(bs, nelim, r, e, f);
  return CX_OK;
}

int cx_partition_points(const cx_block_structure* bs, int32_t nelim, int32_t nranks, int32_t* bounds) {
  CX_CHECK_ARG(bs && bounds && nranks >= 1 && nelim >= 0 && nelim <= bs->num_col_blocks);
  // non-zeros of each eliminated column block
  std::vector<int64_t> nnz(nelim + 1, 0);
  for (int r = 0; r < bs->num_row_blocks; ++r) {
    const int b = bs->row_cell_begin[r], e = bs->row_cell_begin[r + 1];
    if (b == e) continue;
    const int first = bs->cells[b].block_id;
    if (first >= nelim) continue;
    int64_t n = 0;
    for (int c = b; c < e; ++c) n += int64_t(bs->row_blocks[r].size) * bs->col_blocks[bs->cells[c].block_id].size;
    nnz[first + 1] += n;
  }
  std::partial_sum(nnz.begin(), nnz.end(), nnz.begin());
  const int64_t total = nnz[nelim];
  bounds[0] = 0;
  for (int k = 1; k < nranks; ++k) {
    const int64_t target = total * k / nranks;
    int32_t j = int32_t(std::lower_bound(nnz.begin(), nnz.end(), target) - nnz.begin());
    bounds[k] = std::max(bounds[k - 1], std::min(j, nelim));
  }
  bounds[nranks] = nelim;
  return CX_OK;
}

int cx_matrix_create(cx_context* ctx, const cx_block_structure* bs, int32_t nelim, cx_matrix** out) {
  CX_CHECK_ARG(ctx && bs && out);
  CX_CHECK_ARG(bs->num_row_blocks >= 0 && bs->num_col_blocks >= 0);
  CX_CHECK_ARG(nelim >= 0 && nelim <= bs->num_col_blocks);
  if (cxm_is_front(ctx)) return cxm_matrix_create(ctx, bs, nelim, out);
  CX_HIP(hipSetDevice(ctx->device));
  auto A = new cx_matrix;
  A->ctx = ctx;
  A->R = bs->num_row_blocks;
  A->Cb = bs->num_col_blocks;
  A->nelim = nelim;
  A->rows.assign(bs->row_blocks, bs->row_blocks + A->R);
  A->cols.assign(bs->col_blocks, bs->col_blocks + A->Cb);
  A->rcb.assign(bs->row_cell_begin, bs->row_cell_begin + A->R + 1);
  A->cells.assign(bs->cells, bs->cells + A->rcb[A->R]);
  A->num_rows = A->R ? int64_t(A->rows.back().position) + A->rows.back().size : 0;
  A->num_cols = A->Cb ? int64_t(A->cols.back().position) + A->cols.back().size : 0;
  // validate + count
  int64_t nnz = 0, max_end = 0;
  for (int r = 0; r < A->R; ++r) {
    for (int c = A->rcb[r]; c < A->rcb[r + 1]; ++c) {
      const cx_cell& cell = A->cells[c];
      if (cell.block_id < 0 || cell.block_id >= A->Cb || cell.position < 0) {
        delete A;
        cx_set_error("cell %d of row block %d is out of range", c, r);
        return CX_ERR_INVALID_ARGUMENT;
      }
      const int64_t sz = int64_t(A->rows[r].size) * A->cols[cell.block_id].size;
      nnz += sz;
      max_end = std::max(max_end, int64_t(cell.position) + sz);
    }
  }
  A->nnz = nnz;
  if (max_end > nnz) {  // the reference allocates exactly num_nonzeros values
    delete A;
    cx_set_error("cell positions reach %lld but the matrix has only %lld non-zeros", (long long)max_end, (long long)nnz);
    return CX_ERR_INVALID_ARGUMENT;
  }
  for (int c = 0; c < A->Cb; ++c) (c < nelim ? A->num_cols_e : A->num_cols_f) += A->cols[c].size;
  for (int r = 0; r < A->R; ++r)
    if (A->rcb[r + 1] > A->rcb[r] && A->cells[A->rcb[r]].block_id < nelim) A->num_row_blocks_e = r + 1;
  DetectStructureHost(bs, nelim, &A->row_size, &A->e_size, &A->f_size);

  int rc = A->d_values.alloc(size_t(std::max<int64_t>(nnz, 1)));
  if (rc == CX_OK) rc = (hipMemsetAsync(A->d_values.p, 0, size_t(std::max<int64_t>(nnz, 1)) * sizeof(double), ctx->stream) == hipSuccess) ? CX_OK : CX_ERR_HIP;
  int32_t P = 0, C = 0;
  if (rc == CX_OK && Detect239(A, &P, &C)) {
    A->is239 = true;
    A->O = A->R;
    A->P = P;
    A->C = C;
    rc = Build239(A);
  }
  if (rc == CX_OK && !A->is239) {
    hipStream_t st = ctx->stream;
    rc = A->d_rows.upload(A->rows, st);
    if (rc == CX_OK) rc = A->d_cols.upload(A->cols, st);
    if (rc == CX_OK) rc = A->d_rcb.upload(A->rcb, st);
    if (rc == CX_OK) rc = A->d_cells.upload(A->cells, st);
    if (rc == CX_OK) rc = cxe_try_embed(A);  // <2, e <= 3, f <= 9> structures get a static image (cx_embed.hip)
  }
  if (rc != CX_OK) {
    cxe_destroy(A);
    delete A;
    return rc;
  }
  *out = A;
  return CX_OK;
}

void cx_matrix_destroy(cx_matrix* A) {
  if (!A) return;
  if (!A->parts.empty() || cxm_is_front(A->ctx)) return cxm_matrix_destroy(A);
  (void)hipSetDevice(A->ctx->device);
  (void)hipStreamSynchronize(A->ctx->stream);
  cxe_destroy(A);
  delete A;
}

int64_t cx_matrix_num_rows(const cx_matrix* A) { return A ? A->num_rows : 0; }
int64_t cx_matrix_num_cols(const cx_matrix* A) { return A ? A->num_cols : 0; }
int64_t cx_matrix_num_nonzeros(const cx_matrix* A) { return A ? A->nnz : 0; }
int cx_matrix_is_static_239(const cx_matrix* A) { return A && A->is239 ? 1 : 0; }
int cx_matrix_static_path(const cx_matrix* A) { return !A ? 0 : (A->is239 ? 1 : (A->embed ? 2 : 0)); }
double* cx_matrix_device_values(cx_matrix* A) { return A ? A->d_values.p : nullptr; }  // NULL for a multi-shard front

int cx_matrix_values_changed(cx_matrix* A) {
  CX_CHECK_ARG(A);
  if (!A->parts.empty()) return cxm_matrix_values_changed(A);
  A->ft_valid = false;
  A->f32_valid = false;
  if (A->embed) A->embed->dirty = true;
  return CX_OK;
}

int cx_matrix_set_values(cx_matrix* A, const double* src, int32_t memspace) {
  CX_CHECK_ARG(A && (src || A->nnz == 0));
  if (!A->parts.empty()) return cxm_matrix_set_values(A, src, memspace);
  if (A->nnz) CX_TRY(cx_vector_in(A->ctx, A->d_values.p, src, size_t(A->nnz), memspace));
  CX_TRY(cx_stream_sync(A->ctx, A->ctx->stream));
  A->ft_valid = false;
  A->f32_valid = false;
  if (A->embed) A->embed->dirty = true;
  return CX_OK;
}

int cx_matrix_get_values(const cx_matrix* A, double* dst) {
  CX_CHECK_ARG(A && (dst || A->nnz == 0));
  if (!A->parts.empty()) return cxm_matrix_get_values(A, dst);
  if (A->nnz) CX_TRY(cx_copy_d2h(A->ctx, dst, A->d_values.p, size_t(A->nnz) * sizeof(double)));
  CX_TRY(cx_stream_sync(A->ctx, A->ctx->stream));
  return CX_OK;
}

int cx_matrix_set_zero(cx_matrix* A) {
  CX_CHECK_ARG(A);
  if (!A->parts.empty()) return cxm_matrix_set_zero(A);
  if (A->nnz) CX_HIP(hipMemsetAsync(A->d_values.p, 0, size_t(A->nnz) * sizeof(double), A->ctx->stream));
  A->ft_valid = false;
  A->f32_valid = false;
  if (A->embed) A->embed->dirty = true;
  return CX_OK;
}

int cx_matrix_right_multiply(cx_matrix* A, const double* x, double* y, int32_t memspace) {
  CX_CHECK_ARG(A && x && y);
  if (!A->parts.empty()) return cxm_matrix_op(A, 0, x, y, memspace);
  HostOrDevice hx(A->ctx), hy(A->ctx);
  CX_TRY(hx.in(x, size_t(A->num_cols), memspace));
  CX_TRY(hy.inout(y, size_t(A->num_rows), memspace, true));
  CX_TRY(Timed(A, [&] { return A->embed ? cxe_matrix_op(A, 0, hx.dptr, hy.dptr) : cxk_right_multiply(A, hx.dptr, hy.dptr); }));
  return hy.out();
}

int cx_matrix_right_multiply_overwrite(cx_matrix* A, const double* x, double* y, int32_t memspace) {
  CX_CHECK_ARG(A && x && y);
  if (!A->parts.empty()) return cxm_matrix_op(A, 4, x, y, memspace);
  HostOrDevice hx(A->ctx), hy(A->ctx);
  CX_TRY(hx.in(x, size_t(A->num_cols), memspace));
  CX_TRY(hy.inout(y, size_t(A->num_rows), memspace, false));  // nothing of y crosses PCIe on the way in
  // the static kernel writes every row itself (no zeroing, no read of the old value); an embedding accumulates into zeros
  if (A->embed) CX_HIP(hipMemsetAsync(hy.dptr, 0, size_t(A->num_rows) * sizeof(double), A->ctx->stream));
  CX_TRY(Timed(A, [&] { return A->embed ? cxe_matrix_op(A, 0, hx.dptr, hy.dptr) : cxk_right_multiply(A, hx.dptr, hy.dptr, false); }));
  return hy.out();
}

int cx_matrix_left_multiply(cx_matrix* A, const double* x, double* y, int32_t memspace) {
  CX_CHECK_ARG(A && x && y);
  if (!A->parts.empty()) return cxm_matrix_op(A, 1, x, y, memspace);
  HostOrDevice hx(A->ctx), hy(A->ctx);
  CX_TRY(hx.in(x, size_t(A->num_rows), memspace));
  CX_TRY(hy.inout(y, size_t(A->num_cols), memspace, true));
  CX_TRY(cx_matrix_ensure_ft(A));  // the camera-major copy is part of the matrix, not of the product
  CX_TRY(Timed(A, [&] { return A->embed ? cxe_matrix_op(A, 1, hx.dptr, hy.dptr) : cxk_left_multiply(A, hx.dptr, hy.dptr); }));
  return hy.out();
}

int cx_matrix_squared_column_norm(cx_matrix* A, double* x, int32_t memspace) {
  CX_CHECK_ARG(A && x);
  if (!A->parts.empty()) return cxm_matrix_op(A, 2, nullptr, x, memspace);
  HostOrDevice hx(A->ctx);
  CX_TRY(hx.inout(x, size_t(A->num_cols), memspace, false));
  CX_TRY(cx_matrix_ensure_ft(A));
  CX_TRY(Timed(A, [&] { return A->embed ? cxe_matrix_op(A, 2, nullptr, hx.dptr) : cxk_squared_column_norm(A, hx.dptr); }));
  return hx.out();
}

int cx_matrix_scale_columns(cx_matrix* A, const double* scale, int32_t memspace) {
  CX_CHECK_ARG(A && scale);
  if (!A->parts.empty()) return cxm_matrix_op(A, 3, scale, nullptr, memspace);
  HostOrDevice hs(A->ctx);
  CX_TRY(hs.in(scale, size_t(A->num_cols), memspace));
  CX_TRY(Timed(A, [&] { return cxk_scale_columns(A, hs.dptr); }));  // (an embedded matrix scales the caller's values: dynamic-size kernel)
  if (A->embed) A->embed->dirty = true;
  return CX_OK;
}

// PartitionedMatrixView products: part 1 = E, 2 = F
static int PartitionedMultiply(cx_matrix* A, int part, bool transpose, const double* x, double* y, int32_t memspace) {
  CX_CHECK_ARG(A && x && y && A->nelim > 0);
  if (!A->parts.empty()) {
    cx_set_error("the PartitionedMatrixView products are not offered on a multi-shard front");
    return CX_ERR_UNSUPPORTED;
  }
  const size_t ncol = size_t(part == 1 ? A->num_cols_e : A->num_cols_f);
  HostOrDevice hx(A->ctx), hy(A->ctx);
  CX_TRY(hx.in(x, transpose ? size_t(A->num_rows) : ncol, memspace));
  CX_TRY(hy.inout(y, transpose ? ncol : size_t(A->num_rows), memspace, true));
  if (A->is239 && part == 2 && transpose) CX_TRY(cx_matrix_ensure_ft(A));
  CX_TRY(Timed(A, [&]() -> int {
    hipStream_t st = A->ctx->stream;
    if (A->is239) {
      const double* E = A->d_values.p;
      const double* F = A->d_values.p + 6 * A->O;
      if (!transpose) {
        hipLaunchKernelGGL(k_right_239<double>, dim3(grid_for(A->O, kBlock)), dim3(kBlock), 0, st, E, F, A->d_row_pt.p,
                           A->d_row_cam.p, (const double*)hx.dptr, (const double*)hx.dptr, hy.dptr, A->O, part == 1 ? 1 : 0,
                           part == 2 ? 1 : 0, 1, (const int*)nullptr);
      } else if (part == 1) {
        hipLaunchKernelGGL(k_left_e_239<double>, dim3(A->num_tiles), dim3(kBlock), 0, st, E, A->d_tile_row.p, A->d_tile_pt.p,
                           A->d_pt_start.p, (const double*)hx.dptr, hy.dptr, 1, (const int*)nullptr);
      } else {
        CX_TRY(cxk_ft_multiply(A, hx.dptr, hy.dptr, true));
      }
    } else if (A->R > 0) {
      const int off = part == 2 ? int(A->num_cols_e) : 0;
      if (!transpose)
        hipLaunchKernelGGL(kg_right, dim3(grid_for(A->R, 128)), dim3(128), 0, st, A->d_rows.p, A->d_cols.p, A->d_rcb.p,
                           A->d_cells.p, A->d_values.p, (const double*)hx.dptr, hy.dptr, A->R, A->num_row_blocks_e, part, off);
      else
        CX_TRY(cxk_generic_left_multiply(A, part, off, hx.dptr, hy.dptr));
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
  }));
  return hy.out();
}

int cx_matrix_right_multiply_e(cx_matrix* A, const double* x, double* y, int32_t m) { return PartitionedMultiply(A, 1, false, x, y, m); }
int cx_matrix_right_multiply_f(cx_matrix* A, const double* x, double* y, int32_t m) { return PartitionedMultiply(A, 2, false, x, y, m); }
int cx_matrix_left_multiply_e(cx_matrix* A, const double* x, double* y, int32_t m) { return PartitionedMultiply(A, 1, true, x, y, m); }
int cx_matrix_left_multiply_f(cx_matrix* A, const double* x, double* y, int32_t m) { return PartitionedMultiply(A, 2, true, x, y, m); }

double cx_matrix_last_kernel_ms(const cx_matrix* A) { return A ? double(A->last_ms) : 0.0; }

}  // extern "C"
