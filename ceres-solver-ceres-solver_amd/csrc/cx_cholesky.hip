// Dense Cholesky of the reduced camera matrix on gfx950 (DenseCholesky::
// FactorAndSolve, dense_cholesky.cc:139-207; the reference's device variant is
// cusolverDnDpotrf/Dpotrs, dense_cholesky.cc:360-444).
//
// A (n x n, row-major, upper triangle valid) = U'U, blocked right-looking with
// 64-wide panels:
//   k_potrf_diag   factor the 64x64 diagonal block in LDS
//   k_trsm_panel   U(k, j>k) = U_kk^-T A(k, j>k)       one thread per column
//   k_syrk_mfma    A(i,j) -= U(k,i)' U(k,j), i <= j     v_mfma_f64_16x16x4_f64, 64x64 tile per
//                  workgroup, 32x32 per wavefront -- the one genuinely dense, MFMA-bound piece
// followed by blocked forward / backward substitution.
#include "cx_internal.h"
#include "cx_schur.h"

namespace {

constexpr int NB = 64;

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ A, int n, int k0, int kb,
                                                    int* __restrict__ not_pd) {
  __shared__ double T[NB][NB + 1];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx - i * NB;
    T[i][j] = (i < kb && j < kb && j >= i) ? A[size_t(k0 + i) * n + k0 + j] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < kb; ++j) {
    const double d = T[j][j];
    if (!(d > 0.0)) {
      if (tid == 0) *not_pd = 1;
      return;  // uniform: every thread reads the same d
    }
    const double s = sqrt(d);
    __syncthreads();
    // scale row j
    for (int c = j + tid; c < kb; c += 256) T[j][c] = (c == j) ? s : T[j][c] / s;
    __syncthreads();
    // trailing update of the upper triangle
    const int m = kb - j - 1;
    for (int idx = tid; idx < m * m; idx += 256) {
      const int i = j + 1 + idx / m, c = j + 1 + idx % m;
      if (c >= i) T[i][c] -= T[j][i] * T[j][c];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx - i * NB;
    if (i < kb && j < kb && j >= i) A[size_t(k0 + i) * n + k0 + j] = T[i][j];
  }
}

// columns c >= k0 + kb: solve U_kk' x = a(:, c)
__global__ __launch_bounds__(128) void k_trsm_panel(double* __restrict__ A, int n, int k0, int kb) {
  __shared__ double U[NB][NB + 1];
  __shared__ double xs[NB][128];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 128) {
    const int i = idx / NB, j = idx - i * NB;
    U[i][j] = (i < kb && j < kb && j >= i) ? A[size_t(k0 + i) * n + k0 + j] : 0.0;
  }
  __syncthreads();
  const int c = k0 + kb + blockIdx.x * 128 + tid;
  if (c >= n) return;
  for (int i = 0; i < kb; ++i) {
    double s = A[size_t(k0 + i) * n + c];
    for (int k = 0; k < i; ++k) s -= U[k][i] * xs[k][tid];
    s /= U[i][i];
    xs[i][tid] = s;
    A[size_t(k0 + i) * n + c] = s;
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

// forward: solve U_kk' y_blk = y_blk in place (one workgroup), then y(j) -= U(k, j) y_k for j >= rest
__global__ __launch_bounds__(64) void k_trsv_diag_fwd(const double* __restrict__ A, int n, int k0, int kb,
                                                      double* __restrict__ y) {
  __shared__ double ys[NB];
  const int t = threadIdx.x;
  if (t < kb) ys[t] = y[k0 + t];
  __syncthreads();
  for (int i = 0; i < kb; ++i) {
    if (t == i) ys[i] = ys[i] / A[size_t(k0 + i) * n + k0 + i];
    __syncthreads();
    if (t > i && t < kb) ys[t] -= A[size_t(k0 + i) * n + k0 + t] * ys[i];
    __syncthreads();
  }
  if (t < kb) y[k0 + t] = ys[t];
}

__global__ __launch_bounds__(256) void k_gemv_fwd(const double* __restrict__ A, int n, int k0, int kb,
                                                  double* __restrict__ y) {
  __shared__ double ys[NB];
  if (threadIdx.x < kb) ys[threadIdx.x] = y[k0 + threadIdx.x];
  __syncthreads();
  const int j = k0 + kb + blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int k = 0; k < kb; ++k) s += A[size_t(k0 + k) * n + j] * ys[k];
  y[j] -= s;
}

// backward: solve U_kk x_blk = x_blk in place, then x(i) -= U(i, blk) x_blk for i < k0
__global__ __launch_bounds__(64) void k_trsv_diag_bwd(const double* __restrict__ A, int n, int k0, int kb,
                                                      double* __restrict__ x) {
  __shared__ double xs[NB];
  const int t = threadIdx.x;
  if (t < kb) xs[t] = x[k0 + t];
  __syncthreads();
  for (int i = kb - 1; i >= 0; --i) {
    if (t == i) xs[i] = xs[i] / A[size_t(k0 + i) * n + k0 + i];
    __syncthreads();
    if (t < i) xs[t] -= A[size_t(k0 + t) * n + k0 + i] * xs[i];
    __syncthreads();
  }
  if (t < kb) x[k0 + t] = xs[t];
}

__global__ __launch_bounds__(256) void k_gemv_bwd(const double* __restrict__ A, int n, int k0, int kb,
                                                  double* __restrict__ x) {
  __shared__ double xs[NB];
  if (threadIdx.x < kb) xs[threadIdx.x] = x[k0 + threadIdx.x];
  __syncthreads();
  // 4 threads per row i, 16 columns each, shuffled together
  const int gi = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  double s = 0.0;
  if (gi < k0) {
    const double* row = A + size_t(gi) * n + k0;
    for (int c = part * 16; c < min(kb, part * 16 + 16); ++c) s += row[c] * xs[c];
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
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, st, a, n, k0, kb, d_flag);
    const int rem = n - k0 - kb;
    if (rem > 0) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((rem + 127) / 128), dim3(128), 0, st, a, n, k0, kb);
      const int T = (rem + 63) / 64;
      hipLaunchKernelGGL(k_syrk_mfma, dim3(T, T), dim3(256), 0, st, a, n, k0, kb);
    }
  }
  CX_HIP(hipGetLastError());
  if (x != rhs) CX_HIP(hipMemcpyAsync(x, rhs, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, st));
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    hipLaunchKernelGGL(k_trsv_diag_fwd, dim3(1), dim3(64), 0, st, (const double*)a, n, k0, kb, x);
    const int rem = n - k0 - kb;
    if (rem > 0) hipLaunchKernelGGL(k_gemv_fwd, dim3((rem + 255) / 256), dim3(256), 0, st, (const double*)a, n, k0, kb, x);
  }
  const int last = ((n - 1) / NB) * NB;
  for (int k0 = last; k0 >= 0; k0 -= NB) {
    const int kb = std::min(NB, n - k0);
    hipLaunchKernelGGL(k_trsv_diag_bwd, dim3(1), dim3(64), 0, st, (const double*)a, n, k0, kb, x);
    if (k0 > 0) hipLaunchKernelGGL(k_gemv_bwd, dim3((k0 + 63) / 64), dim3(256), 0, st, (const double*)a, n, k0, kb, x);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}
