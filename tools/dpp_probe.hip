// dpp_probe.hip -- the one cross-lane form CDNA3/4 give fp64 arithmetic: `row_newbcast:L` (lane L of every 16-lane row feeds all 16
// lanes) on v_mov_b64_dpp / v_fmac_f64_dpp.  (1) semantics, (2) whether a DPP read of a register the previous vector instruction wrote
// needs software wait states inside inline asm, (3) issue cadence of v_fmac_f64_dpp against plain v_fma_f64 at 1 and 2 wavefronts per
// SIMD -- the smoother kernel k_rts4 takes every broadcast operand this way instead of through LDS.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/dpp_probe.hip -o tools/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

#define BC_MOV(dst, src, L) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
#define BC_MOV_NOP(dst, src, L) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
#define BC_FMAC(acc, src, coef, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef))

__global__ __launch_bounds__(64) void k_sem(double* out, double seed) {
  const int l = threadIdx.x;
  double x = 100.0 * l + seed, y, z = 1.0, w, v;
  BC_MOV_NOP(y, x, 5);
  double c = 2.0 + l;
  asm volatile("s_nop 1");
  BC_FMAC(z, x, c, 11);                 // z = 1 + x[row lane 11] * c[own]
  // hazard: the source is produced by the instruction right in front of the DPP read
  double t = fma(x, 3.0, seed);
  BC_MOV(w, t, 7);                      // no software wait states
  double t2 = fma(x, 5.0, seed);
  BC_MOV_NOP(v, t2, 7);
  out[l] = y; out[64 + l] = z; out[128 + l] = w; out[192 + l] = v;
}

template <int CHAINS, bool DPP>
__global__ __launch_bounds__(64) void k_rate(double* out, int iters, double a, double b) {
  double acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = threadIdx.x * 1e-3 + c;
  double src = a * 1e-9 + threadIdx.x * 1e-12, coef = b;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (DPP) { BC_FMAC(acc[c], src, coef, 3); }
      else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[c]) : "v"(src), "v"(coef));
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// dependent chain: broadcast of a value the previous FMA produced (the factorisation's pivot path)
template <bool DPP>
__global__ __launch_bounds__(64) void k_chain(double* out, int iters, double a) {
  double x = threadIdx.x * 1e-3 + 1.0, y = 0.5;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x = fma(x, a, y);
      if (DPP) { double t; BC_MOV_NOP(t, x, 2); y = t * 1e-3; } else y = x * 1e-3;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x + y;
}

template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int CHAINS, bool DPP>
void rate(double* out, int w) {
  const int iters = 4096;
  float ms = timed([&] { hipLaunchKernelGGL((k_rate<CHAINS, DPP>), dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  double n = (double)iters * CHAINS;
  printf("%s chains %2d waves/SIMD %d: %8.3f ms  %6.2f cycles per instruction per SIMD (2.4 GHz)\n", DPP ? "fmac_dpp" : "fmac    ", CHAINS, w, ms,
         ms * 1e-3 * 2.4e9 / (n * w));
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 1024 * 4);
  double h[256];
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, out, 0.25);
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; l++) {
    const int r = l & ~15;
    const double x5 = 100.0 * (r + 5) + 0.25, x11 = 100.0 * (r + 11) + 0.25, x7 = 100.0 * (r + 7) + 0.25;
    bad[0] += h[l] != x5;
    bad[1] += h[64 + l] != fma(x11, 2.0 + l, 1.0);
    bad[2] += h[128 + l] != fma(x7, 3.0, 0.25);
    bad[3] += h[192 + l] != fma(x7, 5.0, 0.25);
  }
  printf("semantics: mov_dpp row_newbcast wrong lanes %d, fmac_dpp wrong lanes %d, back-to-back read without s_nop wrong lanes %d, with s_nop 1 wrong lanes %d\n",
         bad[0], bad[1], bad[2], bad[3]);
  for (int w : {1, 2}) {
    rate<8, false>(out, w); rate<8, true>(out, w); rate<16, false>(out, w); rate<16, true>(out, w);
    rate<32, false>(out, w); rate<32, true>(out, w); rate<44, true>(out, w);
  }
  for (int w : {1, 2}) {
    const int iters = 2048;
    float m0 = timed([&] { hipLaunchKernelGGL(k_chain<false>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 0.999); });
    float m1 = timed([&] { hipLaunchKernelGGL(k_chain<true>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 0.999); });
    printf("dependent chain (fma -> mul) x 8 per iteration, waves/SIMD %d: plain %7.1f cycles per link, with a broadcast in the link %7.1f\n", w,
           m0 * 1e-3 * 2.4e9 / (iters * 8.0), m1 * 1e-3 * 2.4e9 / (iters * 8.0));
  }
  return 0;
}
