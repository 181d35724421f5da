// issue_probe.hip -- what ONE wavefront per SIMD can issue: fp64 vector FMAs with 8 ... 32 independent chains (tools/fp64_ilp.hip stops at 8:
// 10.8 cycles per FMA), and v_mfma_f64_16x16x4 with 1 ... 4 independent accumulator tiles.  The two config-4 kernels run one wavefront per
// SIMD; this says how far their instruction cadence is from what the hardware gives a lone wavefront, and what the
// matrix unit would give it.   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(64) void k_valu(double* out, int iters, double a, double b) {
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

template <int TILES>
__global__ __launch_bounds__(64) void k_mfma(double* out, int iters, double a, double b) {
  d4 acc[TILES];
#pragma unroll
  for (int c = 0; c < TILES; c++) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const double av = a + threadIdx.x * 1e-9, bv = b + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < TILES; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < TILES; c++) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int CHAINS>
void valu(double* out, int w) {
  const int iters = 4096;
  float ms = timed([&] { hipLaunchKernelGGL(k_valu<CHAINS>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  double n = (double)iters * CHAINS;
  printf("valu  chains %2d waves/SIMD %d: %8.3f ms  %6.2f cycles per FMA instruction per SIMD (2.4 GHz)  %6.1f TFLOP/s\n", CHAINS, w, ms,
         ms * 1e-3 * 2.4e9 / (n * w), 2.0 * n * 64 * 1024 * w / (ms * 1e-3) / 1e12);
}

template <int TILES>
void mfma(double* out, int w) {
  const int iters = 4096;
  float ms = timed([&] { hipLaunchKernelGGL(k_mfma<TILES>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0000001, 1e-3); });
  double n = (double)iters * TILES;
  printf("mfma  tiles  %2d waves/SIMD %d: %8.3f ms  %6.2f cycles per MFMA per SIMD (2.4 GHz)  %6.1f TFLOP/s\n", TILES, w, ms,
         ms * 1e-3 * 2.4e9 / (n * w), 2.0 * 1024 * n * 1024 * w / (ms * 1e-3) / 1e12);
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 1024 * 4);
  for (int w : {1, 2}) { valu<8>(out, w); valu<16>(out, w); valu<24>(out, w); valu<32>(out, w); }
  for (int w : {1, 2}) { mfma<1>(out, w); mfma<2>(out, w); mfma<4>(out, w); }
  return 0;
}
