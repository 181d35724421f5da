// exec_probe.hip -- does an fp64 vector instruction of a wavefront with few active lanes cost the SIMD fewer cycles?  A wave64 fp64 FMA runs as
// four passes of 16 lanes; this times a throughput-bound stream of FMAs (8 chains per wavefront, 4 / 8 wavefronts per SIMD) with all lanes
// active, with lanes 0-7 only, lanes 0-15 only, and with one lane in eight (0, 8, .. 56: the scalar phases of the lane-group kernels).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/exec_probe.hip -o tools/exec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(64) void k_fma(double* out, int iters, double a, double b) {
  const int l = threadIdx.x;
  const bool on = MODE == 0 ? true : MODE == 1 ? (l < 8) : MODE == 2 ? (l < 16) : MODE == 3 ? (l % 8 == 0) : (l < 32);
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = l * 1e-3 + c;
  if (on) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < 8; c++) acc[c] = fma(acc[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) s += acc[c];
  out[blockIdx.x * 64 + l] = s;
}

template <int MODE>
void run(double* out, int w, const char* what) {
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fma<MODE>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma<MODE>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s waves/SIMD %d: %8.3f ms  %5.2f cycles per FMA instruction per SIMD (2.4 GHz)\n", what, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w));
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 1024 * 8);
  for (int w : {1, 4, 8}) {
    run<0>(out, w, "all 64 lanes");
    run<4>(out, w, "lanes 0-31");
    run<2>(out, w, "lanes 0-15");
    run<1>(out, w, "lanes 0-7");
    run<3>(out, w, "lanes 0, 8, .. 56");
  }
  return 0;
}
