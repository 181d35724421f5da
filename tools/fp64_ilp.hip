// fp64_ilp.hip -- how many independent fp64 FMA chains one wavefront needs to keep a CDNA4 SIMD's DP pipe busy.
// One wave per SIMD (1024 blocks of 64 threads), CHAINS independent accumulators, each a dependent FMA chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CHAINS>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
  double acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = fma(acc[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS>
void run(double* out, int waves_per_simd) {
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(1024 * waves_per_simd);
  hipLaunchKernelGGL(k<CHAINS>, grid, dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, grid, dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fmas = (double)iters * CHAINS;                 // per wave
  double cyc_per_fma = ms * 1e-3 * 2.4e9 / (fmas * waves_per_simd);   // per SIMD, at nominal 2.4 GHz
  printf("chains %2d waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-FMA per SIMD (@2.4GHz), %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms,
         cyc_per_fma, 2.0 * fmas * 64 * 1024 * waves_per_simd / (ms * 1e-3) / 1e12);
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 1024 * 8);
  for (int w : {1, 2, 4}) { run<1>(out, w); run<2>(out, w); run<4>(out, w); run<8>(out, w); }
  return 0;
}
