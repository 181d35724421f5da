// Accuracy of the v_rsq_f64 / v_rcp_f64 seeds on gfx950 and of one / two Newton steps on top (decides how many steps
// rn::fast_rsqrt / rn::fast_recip need: they sit on the critical path of every factorisation).
//   hipcc --offload-arch=gfx950 -O2 tools/rsq_probe.hip -o tools/rsq_probe && tools/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double* a, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i];
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  const double y1 = fma(y, fma(-h * y, y, 0.5), y);
  const double y2 = fma(y1, fma(-h * y1, y1, 0.5), y1);
  double r = __builtin_amdgcn_rcp(x);
  const double r1 = fma(fma(-x, r, 1.0), r, r);
  const double r2 = fma(fma(-x, r1, 1.0), r1, r1);
  out[6 * i + 0] = y; out[6 * i + 1] = y1; out[6 * i + 2] = y2;
  out[6 * i + 3] = r; out[6 * i + 4] = r1; out[6 * i + 5] = r2;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> a(n), o(6 * n);
  unsigned long long s = 12345;
  for (int i = 0; i < n; i++) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    a[i] = std::pow(10.0, -8.0 + 16.0 * u);           // 1e-8 .. 1e8
  }
  double *da, *dout;
  hipMalloc((void**)&da, n * 8); hipMalloc((void**)&dout, 6 * n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    const long double rs = 1.0L / sqrtl((long double)a[i]), rc = 1.0L / (long double)a[i];
    for (int k = 0; k < 3; k++) { const double d = (double)fabsl(((long double)o[6 * i + k] - rs) / rs); if (d > e[k]) e[k] = d; }
    for (int k = 3; k < 6; k++) { const double d = (double)fabsl(((long double)o[6 * i + k] - rc) / rc); if (d > e[k]) e[k] = d; }
  }
  std::printf("max relative error over %d values in [1e-8, 1e8] (2^-53 = 1.1e-16):\n", n);
  std::printf("  v_rsq_f64 seed %.3e   + 1 Newton %.3e   + 2 Newton %.3e\n", e[0], e[1], e[2]);
  std::printf("  v_rcp_f64 seed %.3e   + 1 Newton %.3e   + 2 Newton %.3e\n", e[3], e[4], e[5]);
  return 0;
}
