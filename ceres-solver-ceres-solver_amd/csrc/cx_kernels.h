// Device-side helpers shared by the gfx950 kernels of libcxschur.
//
// Data layout (static <2,3,9> bundle-adjustment path).  The value array is the
// reference's BlockSparseMatrix::values_ verbatim: E cells (2x3 row-major, 48 B)
// of all row blocks first, then the F cells (2x9 row-major, 144 B).  A wavefront
// reads 64 consecutive cells as one contiguous run with 16-byte loads per lane
// (coalesced), parks them in LDS and every lane then picks up its own row's cell
// with ds_read_b128 (row stride 144 B / 48 B: conflict free).
#ifndef CX_KERNELS_H_
#define CX_KERNELS_H_

#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int kBlock = 256;  // threads per workgroup = rows per tile (4 wavefronts of 64)

// Stage `nvalid` (<= kBlock) consecutive cells of DPR doubles each, starting at `base`
// (16-byte aligned), so that the calling thread ends up with its own cell.  Two steps, so a
// kernel can put the global loads of several arrays in flight before the first barrier:
//   load_cells      issues the coalesced 16-byte loads (piece i*kBlock + tid of the run)
//   exchange_cells  parks the pieces in `lds` (kBlock*DPR doubles) and reads the own cell back
//                   (two block barriers)
template <int DPR>
__device__ __forceinline__ void load_cells(const double* __restrict__ base, int nvalid, double2 (&v)[DPR / 2]) {
  static_assert(DPR % 2 == 0, "cells are moved as 16-byte pieces");
  constexpr int kPieces = DPR / 2;
  const int tid = threadIdx.x;
  const double2* __restrict__ src = reinterpret_cast<const double2*>(base);
  const int total = nvalid * kPieces;
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = i * kBlock + tid;
    v[i] = (idx < total) ? src[idx] : make_double2(0.0, 0.0);
  }
}

template <int DPR>
__device__ __forceinline__ void exchange_cells(const double2 (&v)[DPR / 2], double* __restrict__ lds, double (&out)[DPR]) {
  constexpr int kPieces = DPR / 2;
  const int tid = threadIdx.x;
  double2* l2 = reinterpret_cast<double2*>(lds);
#pragma unroll
  for (int i = 0; i < kPieces; ++i) l2[i * kBlock + tid] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const double2 t = l2[tid * kPieces + i];
    out[2 * i] = t.x;
    out[2 * i + 1] = t.y;
  }
  __syncthreads();
}

// fp32-storage variants (mixed-precision CGNR): the run of nvalid cells of DPR floats moves as 16-byte
// pieces too (half as many load instructions as the fp64 run -- with 8-byte pieces the kernels were no
// faster than fp64, they are bound by memory instructions in flight, not bytes); pieces may straddle
// cells and start on 4- or 8-byte boundaries (camera segments), hence the under-aligned vector type.
// The lane widens its own cell to fp64 when it picks it up; all arithmetic stays fp64.
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

template <int DPR>
__device__ __forceinline__ void load_cells(const float* __restrict__ base, int nvalid, float4_u (&v)[(DPR + 3) / 4]) {
  constexpr int kPieces = (DPR + 3) / 4;
  const int tid = threadIdx.x;
  const int total = nvalid * DPR;  // floats in the run
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int f0 = 4 * (i * kBlock + tid);
    if (f0 + 3 < total) {
      v[i] = *reinterpret_cast<const float4_u*>(base + f0);
    } else {
      v[i] = float4_u{0.f, 0.f, 0.f, 0.f};
      if (f0 < total) v[i].x = base[f0];
      if (f0 + 1 < total) v[i].y = base[f0 + 1];
      if (f0 + 2 < total) v[i].z = base[f0 + 2];
    }
  }
}

template <int DPR>
__device__ __forceinline__ void exchange_cells(const float4_u (&v)[(DPR + 3) / 4], double* __restrict__ lds, double (&out)[DPR]) {
  static_assert(DPR % 2 == 0, "the own cell is read back as 8-byte pairs");
  constexpr int kPieces = (DPR + 3) / 4;
  const int tid = threadIdx.x;
  typedef float float4_a __attribute__((ext_vector_type(4)));
  float4_a* l4 = reinterpret_cast<float4_a*>(lds);  // kPieces * kBlock * 16 B <= kBlock * DPR * 8 B
#pragma unroll
  for (int i = 0; i < kPieces; ++i) l4[i * kBlock + tid] = float4_a{v[i].x, v[i].y, v[i].z, v[i].w};
  __syncthreads();
  const float2* l2 = reinterpret_cast<const float2*>(lds);
#pragma unroll
  for (int i = 0; i < DPR / 2; ++i) {
    const float2 t = l2[tid * (DPR / 2) + i];
    out[2 * i] = double(t.x);
    out[2 * i + 1] = double(t.y);
  }
  __syncthreads();
}

template <int DPR>
__device__ __forceinline__ void stage_cells(const float* __restrict__ base, int nvalid,
                                            double* __restrict__ lds, double (&out)[DPR]) {
  float4_u v[(DPR + 3) / 4];
  load_cells<DPR>(base, nvalid, v);
  exchange_cells<DPR>(v, lds, out);
}

template <typename T> struct PieceOf;
template <> struct PieceOf<double> { using type = double2; };
template <> struct PieceOf<float> { using type = float2; };

template <int DPR>
__device__ __forceinline__ void stage_cells(const double* __restrict__ base, int nvalid,
                                            double* __restrict__ lds, double (&out)[DPR]) {
  double2 v[DPR / 2];
  load_cells<DPR>(base, nvalid, v);
  exchange_cells<DPR>(v, lds, out);
}

// stage_cells with half the LDS (kBlock * DPR / 2 doubles): all loads are issued first, then the run goes
// through the buffer in two rounds -- the first half of the bytes (the cells of threads 0..kBlock/2-1),
// then the second.  Two more barriers, but an 18 KB instead of a 36 KB workgroup lets twice as many
// wavefronts share a CU (the streaming kernels are bound by loads in flight).
template <int DPR>
__device__ __forceinline__ void exchange_cells_halves(const double2 (&v)[DPR / 2], double* __restrict__ lds, double (&out)[DPR]);
template <int DPR>
__device__ __forceinline__ void stage_cells_halves(const double* __restrict__ base, int nvalid,
                                                   double* __restrict__ lds, double (&out)[DPR]) {
  constexpr int kPieces = DPR / 2;
  static_assert((kBlock * kPieces) % 2 == 0, "the run splits into two equal halves");
  double2 v[kPieces];
  load_cells<DPR>(base, nvalid, v);
  exchange_cells_halves<DPR>(v, lds, out);
}
template <int DPR>
__device__ __forceinline__ void exchange_cells_halves(const double2 (&v)[DPR / 2], double* __restrict__ lds, double (&out)[DPR]) {
  constexpr int kPieces = DPR / 2;
  constexpr int kHalf = kBlock * kPieces / 2;
  const int tid = threadIdx.x;
  double2* l2 = reinterpret_cast<double2*>(lds);
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = i * kBlock + tid;
    if (idx < kHalf) l2[idx] = v[i];
  }
  __syncthreads();
  if (tid < kBlock / 2) {
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const double2 t = l2[tid * kPieces + i];
      out[2 * i] = t.x;
      out[2 * i + 1] = t.y;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = i * kBlock + tid;
    if (idx >= kHalf) l2[idx - kHalf] = v[i];
  }
  __syncthreads();
  if (tid >= kBlock / 2) {
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const double2 t = l2[(tid - kBlock / 2) * kPieces + i];
      out[2 * i] = t.x;
      out[2 * i + 1] = t.y;
    }
  }
  __syncthreads();
}

// The F cells (18 values) of a tile: fp64 through the half-size buffer, fp32 through the plain one.
template <typename T> struct FStage;
// (load + exchange = run, for kernels that put other requests in flight between the two)
template <> struct FStage<double> {
  static constexpr int kLdsDoubles = kBlock * 9;
  struct Pieces { double2 v[9]; };
  static __device__ __forceinline__ void run(const double* __restrict__ base, int nvalid, double* __restrict__ lds, double (&out)[18]) {
    stage_cells_halves<18>(base, nvalid, lds, out);
  }
  static __device__ __forceinline__ void load(const double* __restrict__ base, int nvalid, Pieces& p) { load_cells<18>(base, nvalid, p.v); }
  static __device__ __forceinline__ void exchange(const Pieces& p, double* __restrict__ lds, double (&out)[18]) {
    exchange_cells_halves<18>(p.v, lds, out);
  }
};
template <> struct FStage<float> {
  // 5 sixteen-byte pieces per thread: 20 KB (round 1 declared the fp64 size, 36 KB, and left the fp32 kernels at four
  // workgroups per CU)
  static constexpr int kLdsDoubles = kBlock * 10;
  struct Pieces { float4_u v[5]; };
  static __device__ __forceinline__ void run(const float* __restrict__ base, int nvalid, double* __restrict__ lds, double (&out)[18]) {
    stage_cells<18>(base, nvalid, lds, out);
  }
  static __device__ __forceinline__ void load(const float* __restrict__ base, int nvalid, Pieces& p) { load_cells<18>(base, nvalid, p.v); }
  static __device__ __forceinline__ void exchange(const Pieces& p, double* __restrict__ lds, double (&out)[18]) {
    exchange_cells<18>(p.v, lds, out);
  }
};

// reverse of stage_cells: every thread hands in its own cell, the workgroup stores
// the nvalid cells as one contiguous run of 16-byte pieces
// (tid is a parameter for callers inside a loop that hide the thread index from the optimiser per iteration, so that
// what derives from it -- addresses, index arithmetic -- is not hoisted out of the loop into registers: k_bal_evaluate)
template <int DPR>
__device__ __forceinline__ void unstage_cells(double* __restrict__ base, int nvalid, double* __restrict__ lds,
                                              const double (&in)[DPR], int tid) {
  constexpr int kPieces = DPR / 2;
  double2* l2 = reinterpret_cast<double2*>(lds);
#pragma unroll
  for (int i = 0; i < kPieces; ++i) l2[tid * kPieces + i] = make_double2(in[2 * i], in[2 * i + 1]);
  __syncthreads();
  double2* dst = reinterpret_cast<double2*>(base);
  const int total = nvalid * kPieces;
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = i * kBlock + tid;
    if (idx < total) dst[idx] = l2[idx];
  }
  __syncthreads();
}

template <int DPR>
__device__ __forceinline__ void unstage_cells(double* __restrict__ base, int nvalid, double* __restrict__ lds,
                                              const double (&in)[DPR]) {
  unstage_cells<DPR>(base, nvalid, lds, in, int(threadIdx.x));
}

// unstage_cells for the F cells with a second destination: the camera-major copy Ft (cx_matrix.hip).  The
// row-major run leaves as one contiguous stream; the same 16-byte pieces then go to cell cam_pos[row] of Ft,
// nine consecutive lanes writing the 144 contiguous bytes of one cell.  This is what lets the kernels that
// produce F values (evaluation, column scaling) keep Ft current, so that no separate permutation pass over
// F is needed afterwards.  ft == nullptr: plain unstage_cells.
__device__ __forceinline__ void unstage_f_cells_two(double* __restrict__ base, double* __restrict__ ft,
                                                    const int32_t* __restrict__ cam_pos_of_row, int nvalid,
                                                    double* __restrict__ lds, const double (&in)[18]) {
  constexpr int kPieces = 9;
  const int tid = threadIdx.x;
  double2* l2 = reinterpret_cast<double2*>(lds);
#pragma unroll
  for (int i = 0; i < kPieces; ++i) l2[tid * kPieces + i] = make_double2(in[2 * i], in[2 * i + 1]);
  __syncthreads();
  double2* dst = reinterpret_cast<double2*>(base);
  const int total = nvalid * kPieces;
#pragma unroll
  for (int i = 0; i < kPieces; ++i) {
    const int idx = i * kBlock + tid;
    if (idx < total) dst[idx] = l2[idx];
  }
  if (ft != nullptr) {
    double2* dft = reinterpret_cast<double2*>(ft);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
      const int idx = i * kBlock + tid;
      if (idx < total) {
        const int cell = idx / kPieces;
        const int part = idx - cell * kPieces;
        dft[int64_t(cam_pos_of_row[cell]) * kPieces + part] = l2[idx];
      }
    }
  }
  __syncthreads();
}

// unstage_cells / unstage_f_cells_two with half the LDS (kBlock * DPR / 2 doubles): the cells of threads
// 0 .. kBlock/2-1 go out in a first round, those of the other half in a second.  Two more barriers, but a producer
// kernel holds its workgroup slot while its stores drain, so that the number of workgroups a CU can hold -- not the
// arithmetic and not the write rate -- sets its speed (k_bal_evaluate: compute alone 1.07 ms, stores alone 1.02 ms,
// together 1.91 ms with 36 KB workgroups).  ft == nullptr: no second destination.
// The second destination's cell numbers come from the callers' registers (own_cam_pos = cam_pos of the thread's own row,
// loaded early) through cpos (kBlock ints of LDS): a global load between the stores would wait for them (vmcnt).
template <int DPR>
__device__ __forceinline__ void unstage_cells_halves(double* __restrict__ base, double* __restrict__ ft,
                                                     int own_cam_pos, int* __restrict__ cpos, int nvalid,
                                                     double* __restrict__ lds, const double (&in)[DPR], int tid) {
  constexpr int kPieces = DPR / 2;
  constexpr int kHalfRows = kBlock / 2;
  constexpr int kHalfPieces = kHalfRows * kPieces;
  constexpr int kIter = (kHalfPieces + kBlock - 1) / kBlock;
  double2* l2 = reinterpret_cast<double2*>(lds);
  double2* dst = reinterpret_cast<double2*>(base);
  double2* dft = reinterpret_cast<double2*>(ft);
  if (ft != nullptr) cpos[tid] = own_cam_pos;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((tid >= kHalfRows) == (h == 1)) {
      const int lt = tid - h * kHalfRows;
#pragma unroll
      for (int i = 0; i < kPieces; ++i) l2[lt * kPieces + i] = make_double2(in[2 * i], in[2 * i + 1]);
    }
    __syncthreads();
    const int total = max(0, min(kHalfRows, nvalid - h * kHalfRows)) * kPieces;
    double2 v[kIter];  // (read unconditionally from a clamped index: a conditionally filled array went to scratch memory)
#pragma unroll
    for (int i = 0; i < kIter; ++i) v[i] = l2[min(i * kBlock + tid, kHalfPieces - 1)];
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int idx = i * kBlock + tid;
      if (idx < total) dst[h * kHalfPieces + idx] = v[i];
    }
    if (ft != nullptr) {
#pragma unroll
      for (int i = 0; i < kIter; ++i) {
        const int idx = i * kBlock + tid;
        if (idx < total) {
          const int cell = idx / kPieces;
          const int part = idx - cell * kPieces;
          dft[int64_t(cpos[h * kHalfRows + cell]) * kPieces + part] = v[i];
        }
      }
    }
    __syncthreads();
  }
}

// N consecutive doubles per row, gathered by a per-row index (camera parameters, column scales of a camera block) for
// the 64 rows of a wavefront.  Plain form: lane l reads the N values of its own row's block -- 64 different cache
// lines per load instruction.  Cooperative form (kCoop): value number f = l + 64 j of the wavefront's [row][N] array is
// read by lane l, so that consecutive lanes read consecutive words of one block (64 / N blocks, about as many lines,
// per instruction: a third of the L2 requests of the whole evaluation kernel); transpose_gathered then hands every lane
// its own row through the wavefront's slice of LDS (64 * N doubles).
template <int N, bool kCoop>
__device__ __forceinline__ void gather_by_row(const double* __restrict__ blocks, int32_t block_of_row, int lane, double (&out)[N]) {
  if constexpr (kCoop) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int f = lane + 64 * j;
      const int row = f / N;
      const int c = __shfl(block_of_row, row, 64);
      out[j] = blocks[N * int64_t(c) + (f - row * N)];
    }
  } else {
    const double* cp = blocks + N * int64_t(block_of_row);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = cp[i];
  }
}
template <int N>
__device__ __forceinline__ void transpose_gathered(double* __restrict__ wave_slice, int lane, double (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) wave_slice[lane + 64 * j] = v[j];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_slice[lane * N + k];
}

// XCD-aware block -> work item map for camera-major kernels.  Workgroups are dealt round-robin
// over the 8 XCDs (blockIdx % 8 picks the XCD), so giving XCD x the contiguous range
// [x*per, (x+1)*per) of segments makes workgroups that run at the same time on one XCD work on
// neighbouring cameras, whose rows share cache lines of the row-sized vector they gather from
// (each XCD has its own L2).  Launch 8*per blocks; returns -1 for the padding blocks.
// Placement only changes speed, never results.
__device__ __forceinline__ int xcd_segment(int num_items) {
  const int per = (num_items + 7) >> 3;
  const int s = int(blockIdx.x & 7) * per + int(blockIdx.x >> 3);
  return s < num_items ? s : -1;
}
__host__ __device__ inline int xcd_grid(int num_items) { return ((num_items + 7) >> 3) << 3; }

// Sum of v over the 64 lanes of a wavefront, returned in every lane.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Deterministic block sum of N values per thread: wave butterflies, then the 4
// wave results are added in wave order.  scratch: N*4 doubles of LDS.  Result in
// every thread.  Contains two block barriers.
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* scratch, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) scratch[i * 4 + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = ((scratch[i * 4] + scratch[i * 4 + 1]) + scratch[i * 4 + 2]) + scratch[i * 4 + 3];
  __syncthreads();
}

template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* scratch) {
  block_sum<N>(v, scratch, int(threadIdx.x));
}

// Same reduction, but the N sums are stored to out[0..N) by threads 0..N-1 (no
// runtime-indexed register array, which would spill to scratch).  N <= kBlock.
template <int N>
__device__ __forceinline__ void block_sum_store(const double (&v)[N], double* scratch, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) scratch[i * 4 + wave] = s;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < N) out[t] = ((scratch[t * 4] + scratch[t * 4 + 1]) + scratch[t * 4 + 2]) + scratch[t * 4 + 3];
  __syncthreads();
}

// ---- many sums at once.  The butterfly above costs 6 shuffles per value; with N values per lane a wavefront can do with
// about 2: at the step with partner lane ^ MASK a lane keeps the lower or the upper half of its values (by its bit MASK),
// hands the other half over and adds what it receives -- 54 -> 27 -> 14 -> 7 -> 4 -> 2 -> 1.  After the six steps every lane holds
// ONE finished 64-lane total, that of value wave_multi_index<N, 32>(lane) (-1: none).  A fixed tree, the same in every call:
// deterministic (and a different association than wave_sum's, so a kernel uses one or the other for a given quantity).
// k_cam_init / k_cam_diag: 45 (+ 9) sums per segment were 324 shuffles of 64-bit values per wavefront, the LDS crossbar's
// busiest customer on mid-size problems (a Dubrovnik-356 segment is ONE pass of 256 rows); now 55.
template <int N, int MASK>
struct WaveMulti {
  static_assert(N >= 1 && N <= 2 * MASK, "N values need log2 steps down to one per lane");
  static constexpr int H = (N + 1) / 2;
  __device__ __forceinline__ static double run(const double (&v)[N], int lane) {
    double w[H];
    const bool upper = (lane & MASK) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const double a = v[j];
      const double b = (j + H < N) ? v[j + H] : 0.0;
      const double keep = upper ? b : a, give = upper ? a : b;
      w[j] = keep + __shfl_xor(give, MASK, 64);
    }
    if constexpr (MASK == 1) return w[0];
    else return WaveMulti<H, MASK / 2>::run(w, lane);
  }
  // which of the N values this lane's result is the total of (-1: of none)
  __device__ __forceinline__ static int index(int lane) {
    int pos = 0;
    if constexpr (MASK > 1) {
      pos = WaveMulti<H, MASK / 2>::index(lane);
      if (pos < 0) return -1;
    }
    const int src = pos + ((lane & MASK) ? H : 0);
    return src < N ? src : -1;
  }
};

// block_sum_store with the wavefront stage above: out[0..N) = the N sums over the workgroup (4 wavefronts, added in wave
// order).  scratch: N * 4 doubles of LDS.  N <= 64.
template <int N>
__device__ __forceinline__ void block_sum_store_multi(const double (&v)[N], double* scratch, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double total = WaveMulti<N, 32>::run(v, lane);
  const int idx = WaveMulti<N, 32>::index(lane);
  if (idx >= 0) scratch[idx * 4 + wave] = total;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < N) out[t] = ((scratch[t * 4] + scratch[t * 4 + 1]) + scratch[t * 4 + 2]) + scratch[t * 4 + 3];
  __syncthreads();
}

// Inverse of a symmetric positive definite 3x3 (row-major, upper triangle read)
// through LLT, the arithmetic of selfadjointView<Upper>().llt().solve(I)
// (implicit_schur_complement.cc:201-202).  ok = false if not PD.
__device__ __forceinline__ void inv3_llt(const double (&m)[9], double (&inv)[9], bool& ok) {
  // A = U'U, U upper
  const double a00 = m[0], a01 = m[1], a02 = m[2], a11 = m[4], a12 = m[5], a22 = m[8];
  ok = a00 > 0.0;
  const double u00 = sqrt(a00);
  const double u01 = a01 / u00, u02 = a02 / u00;
  const double d1 = a11 - u01 * u01;
  ok = ok && d1 > 0.0;
  const double u11 = sqrt(d1);
  const double u12 = (a12 - u01 * u02) / u11;
  const double d2 = a22 - u02 * u02 - u12 * u12;
  ok = ok && d2 > 0.0;
  const double u22 = sqrt(d2);
  // X = U^-1 (upper): solve U X = I
  const double x00 = 1.0 / u00, x11 = 1.0 / u11, x22 = 1.0 / u22;
  const double x01 = -u01 * x11 / u00;
  const double x12 = -u12 * x22 / u11;
  const double x02 = -(u01 * x12 + u02 * x22) / u00;
  // A^-1 = X X'
  inv[0] = x00 * x00 + x01 * x01 + x02 * x02;
  inv[1] = x01 * x11 + x02 * x12;
  inv[2] = x02 * x22;
  inv[4] = x11 * x11 + x12 * x12;
  inv[5] = x12 * x22;
  inv[8] = x22 * x22;
  inv[3] = inv[1];
  inv[6] = inv[2];
  inv[7] = inv[5];
}

// Closed-form (cofactor) inverse of a general 3x3 -- Eigen's fixed-size
// inverse() that InvertPSDMatrix<3> uses (invert_psd_matrix.h:60-63).
__device__ __forceinline__ void inv3_cofactor(const double (&m)[9], double (&inv)[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double invdet = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  inv[0] = c00 * invdet; inv[1] = (m[2] * m[7] - m[1] * m[8]) * invdet; inv[2] = (m[1] * m[5] - m[2] * m[4]) * invdet;
  inv[3] = c01 * invdet; inv[4] = (m[0] * m[8] - m[2] * m[6]) * invdet; inv[5] = (m[2] * m[3] - m[0] * m[5]) * invdet;
  inv[6] = c02 * invdet; inv[7] = (m[1] * m[6] - m[0] * m[7]) * invdet; inv[8] = (m[0] * m[4] - m[1] * m[3]) * invdet;
}

#endif
