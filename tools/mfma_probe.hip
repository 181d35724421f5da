// Probe of v_mfma_f64_16x16x4_f64's operand / result layout on this GPU (the smoother's products use it; the layout is taken
// from here, not from memory): D = A B + C with A 16x4, B 4x16.  Assumed inputs: lane l supplies A[l % 16][l / 16] and
// B[l / 16][l % 16].  The program finds, for every (lane, result register), which entry of the reference product it holds.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  v4d c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}

int main() {
  double hA[64], hB[64], hD[256], ref[16][16];
  for (int i = 0; i < 64; i++) { hA[i] = 1.0 + 0.37 * i + 0.01 * i * i; hB[i] = 2.0 - 0.11 * i + 0.003 * i * i; }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i][j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int ok = 0, as_expected = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (std::fabs(hD[l * 4 + r] - ref[i][j]) < 1e-9 * std::fabs(ref[i][j])) { fi = i; fj = j; }
    if (fi >= 0) ok++;
    if (fi == 4 * (l / 16) + r && fj == l % 16) as_expected++;
    if (l < 20 && r < 4 && (l % 16 < 2 || l == 16 || l == 17)) std::printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, fi, fj);
  }
  std::printf("matched %d of 256; layout D[4*(l/16)+r][l%%16]: %d of 256\n", ok, as_expected);
  return 0;
}
