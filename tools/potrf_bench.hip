// Micro-benchmark of the 32x32 diagonal-block kernel of the blocked Cholesky factorisations
// (cxchol::potrf_inverse_block, ceres-solver-ceres-solver_amd/csrc/cx_chol_blocks.h): time per call and the error
// of U and U^-1 against a host computation.  Build: hipcc -O3 --offload-arch=gfx950 -I<csrc> tools/potrf_bench.hip
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "cx_chol_blocks.h"

void cx_set_error(const char*, ...) {}

using cxchol::NB;

template <int VARIANT>
__global__ __launch_bounds__(64) void k_variant(const double* __restrict__ W, double* __restrict__ F, double* __restrict__ uinv,
                                                int* __restrict__ not_pd, int reps, int kb) {
  __shared__ double lds[cxchol::kPotrfLds];
  for (int r = 0; r < reps; ++r) {
    if (VARIANT == 0) cxchol::potrf_inverse_block_scalar(W, NB, F, NB, kb, uinv, not_pd, lds);
    else cxchol::potrf_inverse_block_mfma(W, NB, F, NB, kb, uinv, not_pd, lds);
  }
}

template <int VARIANT>
static int check(int kb, bool timing) {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  std::vector<double> G(NB * NB), A(NB * NB, 0.0);
  for (auto& g : G) g = nd(rng);
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      double s = (i == j) ? 4.0 : 0.0;
      for (int k = 0; k < NB; ++k) s += G[k * NB + i] * G[k * NB + j];
      A[i * NB + j] = s;
    }
  // host U'U = A (upper) and U^-1
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j)
      if (i >= kb || j >= kb) A[i * NB + j] = (i == j) ? 1.0 : 0.0;  // what the kernel assumes outside the kb x kb block
  std::vector<double> U(A), Ui(NB * NB, 0.0);
  for (int j = 0; j < NB; ++j) {
    for (int k = 0; k < j; ++k)
      for (int c = j; c < NB; ++c) U[j * NB + c] -= U[k * NB + j] * U[k * NB + c];
    const double d = std::sqrt(U[j * NB + j]);
    for (int c = j; c < NB; ++c) U[j * NB + c] /= d;
  }
  for (int c = 0; c < NB; ++c)
    for (int r = NB - 1; r >= 0; --r) {
      double s = (r == c) ? 1.0 : 0.0;
      for (int k = r + 1; k < NB; ++k) s -= U[r * NB + k] * Ui[k * NB + c];
      Ui[r * NB + c] = (c >= r) ? s / U[r * NB + r] : 0.0;
    }
  double *dW, *dF, *dUi;
  int* dflag;
  hipMalloc(&dW, NB * NB * 8);
  hipMalloc(&dF, NB * NB * 8);
  hipMalloc(&dUi, NB * NB * 8);
  hipMalloc(&dflag, 4);
  hipMemcpy(dW, A.data(), NB * NB * 8, hipMemcpyHostToDevice);
  hipMemset(dF, 0, NB * NB * 8);
  hipMemset(dflag, 0, 4);
  hipLaunchKernelGGL(k_variant<VARIANT>, dim3(1), dim3(64), 0, 0, dW, dF, dUi, dflag, 1, kb);
  hipDeviceSynchronize();
  std::vector<double> F(NB * NB), V(NB * NB);
  hipMemcpy(F.data(), dF, NB * NB * 8, hipMemcpyDeviceToHost);
  hipMemcpy(V.data(), dUi, NB * NB * 8, hipMemcpyDeviceToHost);
  double eu = 0.0, ei = 0.0;
  for (int r = 0; r < NB; ++r)
    for (int c = 0; c < NB; ++c) {
      if (c >= r && r < kb && c < kb) eu = std::fmax(eu, std::fabs(F[r * NB + c] - U[r * NB + c]));
      if (c < r && F[r * NB + c] != 0.0) eu = 1.0;  // nothing may be written left of the diagonal (band storage aliases it)
      ei = std::fmax(ei, std::fabs(V[r * NB + c] - Ui[r * NB + c]));  // the whole 32 x 32 inverse, zeros included
    }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = timing ? 2000 : 1;
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_variant<VARIANT>, dim3(1), dim3(64), 0, 0, dW, dF, dUi, dflag, reps, kb);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  int flag = 0;
  hipMemcpy(&flag, dflag, 4, hipMemcpyDeviceToHost);
  std::printf("kb = %d: %s: %.2f us per call, max |U - ref| = %.2e, max |Uinv - ref| = %.2e, not_pd = %d\n", kb, VARIANT == 0 ? "potrf_inverse_block     " : "potrf_inverse_block_mfma", ms * 1e3 / reps, eu, ei, flag);
  return (eu < 1e-12 && ei < 1e-11 && flag == 0) ? 0 : 1;
}

int main() {
  int rc = 0;
  rc |= check<0>(32, true);
  rc |= check<1>(32, true);
  for (int kb : {20, 9, 16, 17, 1, 4, 5, 31}) {
    rc |= check<0>(kb, false);
    rc |= check<1>(kb, false);
  }
  // not positive definite: both variants must raise the flag (checked by eye in the output: not_pd = 1)
  return rc;
}
