// What v_mfma_f64_16x16x4_f64 sustains on the whole chip (all SIMDs busy, operands in registers): the yardstick for the
// "fraction of the fp64 matrix peak" figures in DESIGN.md.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  double4_t acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
  double a = double(threadIdx.x & 7) * 1e-3, b = double(threadIdx.x >> 3) * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
static void run(int wg_per_cu, int cus) {
  const int grid = wg_per_cu * cus, iters = 20000;
  double* out;
  hipMalloc(&out, size_t(grid) * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(grid), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * 4 /*waves*/ * iters * CHAINS * 2048.0;
  std::printf("%d independent accumulators per wave, %d workgroups per CU: %.1f TFLOP/s (%.2f ms), %.1f cycles per MFMA and SIMD at 2.4 GHz\n",
              CHAINS, wg_per_cu, flops / ms * 1e-9, ms,
              ms * 1e-3 * 2.4e9 / (double(wg_per_cu) * iters * CHAINS));
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<1>(1, cus);
  run<4>(1, cus);
  run<4>(2, cus);
  run<8>(1, cus);
  run<4>(4, cus);
  run<8>(2, cus);
  run<8>(4, cus);
  run<2>(8, cus);
  return 0;
}
