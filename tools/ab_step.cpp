// A/B harness for the step-granular fused entry point `{name}_batch_predict_update_{kind}`, no Python: loads several builds of
// one model's library, drives the same launches through each (W warm-up + K timed, back to back on the null stream, HIP events
// around the K), reports microseconds per launch and the largest difference of x, P, y and the flags against the first build.
//   hipcc -O2 -std=c++17 tools/ab_step.cpp -o tools/ab_step -ldl
//   tools/ab_step <name> <kind> <D> <E> <Z> <n> <W> <K> <dt> <inputs.bin|-> <lib.so> [<lib.so> ...]
// inputs.bin (tools/ab_inputs.py): doubles x0[D], P0[E*E], Q[E*E], R[Z*Z], z[Z] of one filter, replicated with small
// per-filter perturbations; "-" = synthetic (unit covariance, diagonal Q and R), fine for the linear kinematic models.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(e)                                                                  \
  do {                                                                         \
    hipError_t e_ = (e);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(3);                                                            \
    }                                                                          \
  } while (0)

typedef int (*step_fn)(double*, double*, const double*, const double*, double, double*, const double*, int, const double*, int64_t, int,
                       uint8_t*, void*);

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double urand() {
  g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17;
  return (double)(g_s >> 11) * (1.0 / 9007199254740992.0);
}

static double maxdiff(const std::vector<double>& a, const std::vector<double>& b, double* scale) {
  double m = 0.0, s = 0.0;
  for (size_t i = 0; i < a.size(); i++) {
    const double d = std::fabs(a[i] - b[i]);
    if (!(d <= m)) m = d;
    if (std::fabs(a[i]) > s) s = std::fabs(a[i]);
  }
  *scale = s;
  return m;
}

int main(int argc, char** argv) {
  if (argc < 12) {
    std::fprintf(stderr, "usage: %s name kind D E Z n W K dt inputs.bin|- lib.so [lib.so ...]\n", argv[0]);
    return 2;
  }
  const std::string name = argv[1], kind = argv[2];
  const int D = std::atoi(argv[3]), E = std::atoi(argv[4]), Z = std::atoi(argv[5]);
  const int64_t n = std::atoll(argv[6]);
  const int W = std::atoi(argv[7]), K = std::atoi(argv[8]);
  const double dt = std::atof(argv[9]);
  const std::string inputs = argv[10];
  const int EE = E * E, ZZ = Z * Z, POOL = 4;

  std::vector<double> x1(D, 0.0), P1(EE, 0.0), Q(EE, 0.0), R(ZZ, 0.0), z1(Z, 0.0);
  if (inputs == "-") {
    for (int i = 0; i < E; i++) { P1[i * E + i] = 1.0; Q[i * E + i] = 0.01 * (1 + i); }
    for (int i = 0; i < Z; i++) R[i * Z + i] = 0.01;
  } else {
    FILE* f = std::fopen(inputs.c_str(), "rb");
    if (!f) { std::perror(inputs.c_str()); return 2; }
    size_t ok = std::fread(x1.data(), 8, D, f) + std::fread(P1.data(), 8, EE, f) + std::fread(Q.data(), 8, EE, f) +
                std::fread(R.data(), 8, ZZ, f) + std::fread(z1.data(), 8, Z, f);
    std::fclose(f);
    if (ok != (size_t)(D + 2 * EE + ZZ + Z)) { std::fprintf(stderr, "%s: short file\n", inputs.c_str()); return 2; }
  }
  std::vector<double> x0(n * D), P0(n * EE), z0((size_t)POOL * n * Z);
  for (int64_t f = 0; f < n; f++) {
    for (int i = 0; i < D; i++) x0[f * D + i] = x1[i] + (inputs == "-" ? 0.1 * (urand() - 0.5) : 0.0);
    const double s = 1.0 + 0.2 * urand();
    for (int i = 0; i < EE; i++) P0[f * EE + i] = P1[i] * s;
  }
  for (size_t i = 0; i < z0.size(); i++) {
    const double c = z1[i % Z];
    z0[i] = c + (0.01 * std::fabs(c) + (inputs == "-" ? 0.3 : 0.01)) * (urand() - 0.5);
  }
  double *dx, *dP, *dz, *dQ, *dR;
  uint8_t* dfl;
  CK(hipMalloc((void**)&dx, sizeof(double) * n * D));
  CK(hipMalloc((void**)&dP, sizeof(double) * n * EE));
  CK(hipMalloc((void**)&dz, sizeof(double) * POOL * n * Z));
  CK(hipMalloc((void**)&dQ, sizeof(double) * EE));
  CK(hipMalloc((void**)&dR, sizeof(double) * ZZ));
  CK(hipMalloc((void**)&dfl, (size_t)n));
  CK(hipMemcpy(dQ, Q.data(), sizeof(double) * EE, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, R.data(), sizeof(double) * ZZ, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  std::vector<double> rx, rP, ry;
  std::vector<uint8_t> rf;
  for (int li = 11; li < argc; li++) {
    void* h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::printf("%s: dlopen failed: %s\n", argv[li], dlerror()); continue; }
    step_fn step = (step_fn)dlsym(h, (name + "_batch_predict_update_" + kind).c_str());
    if (!step) { std::printf("%s: entry point missing\n", argv[li]); continue; }
    float best = 1e30f;
    std::vector<double> hx(n * D), hP(n * EE), hy((size_t)POOL * n * Z);
    std::vector<uint8_t> hf(n);
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemcpy(dx, x0.data(), sizeof(double) * n * D, hipMemcpyHostToDevice));
      CK(hipMemcpy(dP, P0.data(), sizeof(double) * n * EE, hipMemcpyHostToDevice));
      CK(hipMemcpy(dz, z0.data(), sizeof(double) * POOL * n * Z, hipMemcpyHostToDevice));
      CK(hipMemset(dfl, 0xEE, (size_t)n));
      int rc = 0;
      for (int i = 0; i < W; i++) rc |= step(dx, dP, dQ, nullptr, dt, dz + (size_t)(i % POOL) * n * Z, dR, 0, nullptr, n, 1, dfl, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      for (int i = W; i < W + K; i++) rc |= step(dx, dP, dQ, nullptr, dt, dz + (size_t)(i % POOL) * n * Z, dR, 0, nullptr, n, 1, dfl, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      if (rc != 0) { std::printf("%s: entry point returned %d\n", argv[li], rc); break; }
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      if (rep == 0) {
        CK(hipMemcpy(hx.data(), dx, sizeof(double) * n * D, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hP.data(), dP, sizeof(double) * n * EE, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hy.data(), dz, sizeof(double) * POOL * n * Z, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), dfl, (size_t)n, hipMemcpyDeviceToHost));
      }
    }
    std::printf("%-40s %9.3f us per launch (best of 3 x %d)", argv[li], best * 1e3 / K, K);
    if (rx.empty()) {
      rx = hx; rP = hP; ry = hy; rf = hf;
      double s, sp;
      maxdiff(hx, hx, &s);
      maxdiff(hP, hP, &sp);
      size_t nz = 0;
      for (auto f : hf) nz += f != 0;
      std::printf("  (reference; |x|max %.3g |P|max %.3g nonzero flags %zu)\n", s, sp, nz);
    } else {
      double sx, sP, sy;
      const double dxm = maxdiff(rx, hx, &sx), dPm = maxdiff(rP, hP, &sP), dym = maxdiff(ry, hy, &sy);
      size_t fd = 0;
      for (size_t i = 0; i < hf.size(); i++) fd += hf[i] != rf[i];
      std::printf("  diff x %.3g (of %.3g) P %.3g (of %.3g) y %.3g (of %.3g) flags %zu\n", dxm, sx, dPm, sP, dym, sy, fd);
    }
    std::fflush(stdout);
  }
  return 0;
}
