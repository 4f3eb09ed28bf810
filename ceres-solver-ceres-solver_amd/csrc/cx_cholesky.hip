// Dense Cholesky of the reduced camera matrix on gfx950 (DenseCholesky::
// FactorAndSolve, dense_cholesky.cc:139-207; the reference's device variant is
// cusolverDnDpotrf/Dpotrs, dense_cholesky.cc:360-444).
//
// A (n x n, row-major, upper triangle valid) = U'U, blocked right-looking with
// 32-wide panels:
//   k_potrf_diag   factor the 32x32 diagonal block, one wavefront, columns in registers
//   k_trsm_panel   U(k, j>k) = U_kk^-T A(k, j>k)       one thread per column
//   k_syrk_mfma    A(i,j) -= U(k,i)' U(k,j), i <= j     v_mfma_f64_16x16x4_f64, 64x64 tile per
//                  workgroup, 32x32 per wavefront -- the one genuinely dense, MFMA-bound piece
// followed by blocked forward / backward substitution.
#include "cx_internal.h"
#include "cx_schur.h"

namespace {

constexpr int NB = 32;  // panel width: 32 keeps the wave-level factor/solve kernels in registers (64 spills)

typedef double double4_t __attribute__((ext_vector_type(4)));

// v at lane `l` (compile-time constant), as a wave-uniform value: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane(unsigned(u), l);
  const unsigned hi = __builtin_amdgcn_readlane(unsigned(u >> 32), l);
  return __longlong_as_double((static_cast<unsigned long long>(hi) << 32) | lo);
}

// Diagonal block: ONE wavefront, lane c owns column c of the NB x NB block in NB registers.
// Right-looking, fully unrolled: the pivot and the scaled pivot row reach the other lanes
// through v_readlane (scalar broadcast), so the NB-step dependency chain contains no LDS
// round trip and no barrier.
__global__ __launch_bounds__(64) void k_potrf_diag(double* __restrict__ A, int n, int k0, int kb,
                                                   int* __restrict__ not_pd) {
  const int lane = threadIdx.x;
  double T[NB];  // T[r] = block(r, lane); identity padding beyond kb keeps the recurrence valid
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool in = r < kb && lane < kb && lane >= r;
    T[r] = (r == lane) ? 1.0 : 0.0;
    if (in) T[r] = A[size_t(k0 + r) * n + k0 + lane];
  }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double d = readlane_f64(T[j], j);
    ok = ok && (d > 0.0);
    const double sq = sqrt(d);
    const double rs = 1.0 / sq;
    T[j] = (lane == j) ? sq : T[j] * rs;
    const double uj = T[j];  // U(j, own column)
#pragma unroll
    for (int i = j + 1; i < NB; ++i) {
      const double uji = readlane_f64(T[j], i);  // U(j, i)
      T[i] = (lane >= i) ? T[i] - uji * uj : T[i];
    }
  }
  if (!ok && lane == 0) *not_pd = 1;
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (r < kb && lane < kb && lane >= r) A[size_t(k0 + r) * n + k0 + lane] = T[r];
}

// Panel: columns c >= k0 + kb, solve U_kk' x = a(:, c).  One lane per column with the NB
// unknowns in registers; U_kk is wave-uniform and arrives through scalar loads (SGPR operands).
// (An LDS-broadcast variant of the same loop makes hipcc 7.2 spill 2.4 KB per lane.)
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int n, int k0, int kb) {
  const int c = k0 + kb + blockIdx.x * 256 + threadIdx.x;
  const bool live = c < n;
  const double* __restrict__ Ukk = A + size_t(k0) * n + k0;
  double x[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) x[i] = (live && i < kb) ? A[size_t(k0 + i) * n + c] : 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i < kb) {
      double sum = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) sum -= Ukk[size_t(k) * n + i] * x[k];
      x[i] = sum / Ukk[size_t(i) * n + i];
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i < kb) A[size_t(k0 + i) * n + c] = x[i];
  }
}

// trailing update with fp64 MFMA.  Workgroup = 64x64 tile of the trailing matrix,
// wavefront w = 32x32 quadrant, 2x2 MFMA 16x16 tiles, K = kb in steps of 4.
// Operand maps (cdna guide, f64 16x16x4): lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result reg g of lane l is C[(l >> 4) + 4 g][l & 15].
__global__ __launch_bounds__(256) void k_syrk_mfma(double* __restrict__ A, int n, int k0, int kb) {
  const int rest = k0 + kb;
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (ti > tj) return;  // upper block triangle only
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = rest + ti * 64 + (wave >> 1) * 32;
  const int j0 = rest + tj * 64 + (wave & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < kb; kk += 4) {
    const int k = kk + lk;
    const bool kv = k < kb;
    const double* Urow = A + size_t(k0 + (kv ? k : 0)) * n;
    double av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int i = i0 + a * 16 + li;
      av[a] = (kv && i < n) ? Urow[i] : 0.0;
      const int j = j0 + a * 16 + li;
      bv[a] = (kv && j < n) ? Urow[j] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = i0 + a * 16 + lk + 4 * g;
        const int j = j0 + b * 16 + li;
        if (i < n && j < n && j >= i) A[size_t(i) * n + j] -= acc[a][b][g];
      }
}

// Forward substitution, one launch per panel: every workgroup first solves the small system
// U_kk' y_blk = y_blk itself (redundantly, in its first wavefront: lane t owns column t of U_kk,
// 32 steps of one multiply + one lane broadcast; the reciprocals of the diagonal are formed
// beforehand so no division sits in the chain), then subtracts U(blk, j)' y_blk from its slice
// of the remaining right-hand side.  Workgroup 0 stores y_blk.
__global__ __launch_bounds__(256) void k_trsv_fwd(const double* __restrict__ A, int n, int k0, int kb,
                                                  double* __restrict__ y, double* __restrict__ sol) {
  __shared__ double ys[NB];
  const int t = threadIdx.x;
  if (t < 64) {
    double Ucol[NB];  // Ucol[i] = U(i, t)
#pragma unroll
    for (int i = 0; i < NB; ++i) Ucol[i] = (i < kb && t < kb && t >= i) ? A[size_t(k0 + i) * n + k0 + t] : ((i == t) ? 1.0 : 0.0);
    double diag = 1.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) if (i == t) diag = Ucol[i];
    const double rdiag = 1.0 / diag;
    double yt = (t < kb) ? y[k0 + t] : 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const double yi = readlane_f64(yt * rdiag, i);  // y_i is final once steps < i are subtracted
      if (t == i) yt = yi;
      if (t > i) yt -= Ucol[i] * yi;
    }
    if (t < NB) ys[t] = yt;
    if (blockIdx.x == 0 && t < kb) sol[k0 + t] = yt;  // not into y: other workgroups still read y_blk
  }
  __syncthreads();
  const int j = k0 + kb + blockIdx.x * 256 + t;
  if (j >= n) return;
  double s = 0.0;
#pragma unroll 8
  for (int k = 0; k < kb; ++k) s += A[size_t(k0 + k) * n + j] * ys[k];
  y[j] -= s;
}

// Backward substitution, same scheme: U_kk x_blk = x_blk (lane t owns row t of U_kk), then
// x(i) -= U(i, blk) x_blk for the rows above the block.
__global__ __launch_bounds__(256) void k_trsv_bwd(const double* __restrict__ A, int n, int k0, int kb,
                                                  double* __restrict__ x, double* __restrict__ sol) {
  __shared__ double Us[NB * (NB + 1)];
  __shared__ double xs[NB];
  const int t = threadIdx.x;
  if (t < NB)
    for (int i = 0; i < NB; ++i) Us[i * (NB + 1) + t] = (i < kb && t < kb && t >= i) ? A[size_t(k0 + i) * n + k0 + t] : ((i == t) ? 1.0 : 0.0);
  __syncthreads();
  if (t < 64) {
    double Urow[NB];  // Urow[c] = U(t, c)
#pragma unroll
    for (int c = 0; c < NB; ++c) Urow[c] = (t < NB) ? Us[t * (NB + 1) + c] : ((c == t) ? 1.0 : 0.0);
    double diag = 1.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) if (c == t) diag = Urow[c];
    const double rdiag = 1.0 / diag;
    double xt = (t < kb) ? x[k0 + t] : 0.0;
#pragma unroll
    for (int i = NB - 1; i >= 0; --i) {
      const double xi = readlane_f64(xt * rdiag, i);
      if (t == i) xt = xi;
      if (t < i) xt -= Urow[i] * xi;
    }
    if (t < NB) xs[t] = xt;
    if (blockIdx.x == 0 && t < kb) sol[k0 + t] = xt;
  }
  __syncthreads();
  // 4 threads per row above the block, 8 columns each
  const int gi = blockIdx.x * 64 + (t >> 2);
  const int part = t & 3;
  double s = 0.0;
  if (gi < k0) {
    const double* row = A + size_t(gi) * n + k0;
    for (int c = part * (NB / 4); c < min(kb, (part + 1) * (NB / 4)); ++c) s += row[c] * xs[c];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (gi < k0 && part == 0) x[gi] -= s;
}

}  // namespace

int cxd_cholesky_solve(cx_context* ctx, int n, double* a, const double* rhs, double* x, int* d_flag) {
  hipStream_t st = ctx->stream;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(64), 0, st, a, n, k0, kb, d_flag);
    const int rem = n - k0 - kb;
    if (rem > 0) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((rem + 255) / 256), dim3(256), 0, st, a, n, k0, kb);
      const int T = (rem + 63) / 64;
      hipLaunchKernelGGL(k_syrk_mfma, dim3(T, T), dim3(256), 0, st, a, n, k0, kb);
    }
  }
  CX_HIP(hipGetLastError());
  // work = rhs (updated in place), ysol = U'^-1 rhs (then updated in place), x = U^-1 ysol
  CX_TRY(ctx->chol_scratch.alloc(2 * size_t(n)));
  double* work = ctx->chol_scratch.p;
  double* ysol = ctx->chol_scratch.p + n;
  CX_HIP(hipMemcpyAsync(work, rhs, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, st));
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    const int rem = n - k0 - kb;
    hipLaunchKernelGGL(k_trsv_fwd, dim3(std::max(1, (rem + 255) / 256)), dim3(256), 0, st, (const double*)a, n, k0, kb, work, ysol);
  }
  const int last = ((n - 1) / NB) * NB;
  for (int k0 = last; k0 >= 0; k0 -= NB) {
    const int kb = std::min(NB, n - k0);
    hipLaunchKernelGGL(k_trsv_bwd, dim3(std::max(1, (k0 + 63) / 64)), dim3(256), 0, st, (const double*)a, n, k0, kb, ysol, x);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
