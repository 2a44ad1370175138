// Dev aid: what clock does the chip hold under fp32 MFMA load, and what rate does a bare
// v_mfma_f32_32x32x2_f32 loop reach with 1 or 2 waves per SIMD (1 or 2 accumulator chains)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void bare(float *out, long long *cyc, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int q = 0; q < CHAINS; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int q = 0; q < CHAINS; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int q = 0; q < CHAINS; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(int wg_per_cu, int iters) {
  int blocks = 256 * wg_per_cu;
  float *out; long long *cyc;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bare<CHAINS><<<blocks, 256>>>(out, cyc, iters / 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bare<CHAINS><<<blocks, 256>>>(out, cyc, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double flops = (double)blocks * 4 * iters * 16 * CHAINS * 2.0 * 32 * 32 * 2;
  printf("chains %d, %d WG/CU (waves/SIMD %d): %8.3f ms  %7.1f TF/s   clock64 ticks %lld -> %6.1f MHz   cyc/MFMA/SIMD %.1f\n",
         CHAINS, wg_per_cu, wg_per_cu, ms, flops / ms / 1e9, c, c / (ms * 1e3), (double)c / ((double)iters * 16 * CHAINS * wg_per_cu));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(1, 20000); run<2>(1, 10000); run<1>(2, 10000); run<2>(2, 5000); run<1>(3, 8000);
  return 0;
}
