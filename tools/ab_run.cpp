// A/B harness for the fused run (`{name}_batch_run`) of lane-per-filter models, no Python: loads several builds of one model's
// library, runs the same seeded schedule through each, reports time per launch and the largest difference of x, P, y and the
// flags against the first build.  Starts in well under a second, so a gpurun call can compare many variants.
//   hipcc -O2 -std=c++17 tools/ab_run.cpp -o tools/ab_run -ldl
//   tools/ab_run <name> <D> <E> <zmax> <n> <T> <reps> <trace 0|1> <lib.so> [<lib.so> ...]
// trace = 1 passes trace buffers (the traced kernel); keep n * T small then (the trace is T * n * (D + E * E) doubles).
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

typedef int (*run_fn)(double*, double*, const double*, const int32_t*, const double*, int64_t, double*, const double*, int64_t, int,
                      uint8_t*, double*, double*, const double*, const int32_t*, void*);

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double urand() {
  g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17;
  return (double)(g_s >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() {
  const double u = urand() + 1e-300, v = urand();
  return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
}

template <class T>
static T* dev(const std::vector<T>& h) {
  T* d = nullptr;
  CK(hipMalloc((void**)&d, sizeof(T) * (h.size() ? h.size() : 1)));
  CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return d;
}

static double maxdiff(const std::vector<double>& a, const std::vector<double>& b, double* scale) {
  double m = 0.0, s = 0.0;
  for (size_t i = 0; i < a.size(); i++) {
    const double d = std::fabs(a[i] - b[i]);
    if (!(d <= m)) m = d;                 // NaN-propagating
    if (std::fabs(a[i]) > s) s = std::fabs(a[i]);
  }
  *scale = s;
  return m;
}

int main(int argc, char** argv) {
  if (argc < 10) {
    std::fprintf(stderr, "usage: %s name D E zmax n T reps trace lib.so [lib.so ...]\n", argv[0]);
    return 2;
  }
  const std::string name = argv[1];
  const int D = std::atoi(argv[2]), E = std::atoi(argv[3]), Z = std::atoi(argv[4]);
  const int64_t n = std::atoll(argv[5]), T = std::atoll(argv[6]);
  const int reps = std::atoi(argv[7]), trace = std::atoi(argv[8]);
  const int EE = E * E, ZZ = Z * Z;

  // schedule and inputs (identical for every build): unit-ish covariance, small process noise, unit observations with 0.1 sigma
  std::vector<double> x0(n * D), P0(n * EE, 0.0), Q(EE, 0.0), z0(T * n * Z), R(T * ZZ, 0.0), dts(T);
  std::vector<int32_t> kinds(T, 1);
  if (const char* pat = std::getenv("AB_KINDS")) {        // e.g. AB_KINDS=4,10,4,10,12: the schedule cycles through these kinds
    std::vector<int32_t> cyc;
    for (const char* q = pat; *q;) {
      cyc.push_back((int32_t)std::strtol(q, const_cast<char**>(&q), 10));
      if (*q == ',') q++;
    }
    for (int64_t t = 0; t < T && !cyc.empty(); t++) kinds[t] = cyc[t % cyc.size()];
  }
  for (auto& v : x0) v = 0.1 * nrand();
  for (int64_t f = 0; f < n; f++)
    for (int i = 0; i < E; i++) P0[f * EE + i * E + i] = 1.0 + 0.5 * urand();
  for (int i = 0; i < E; i++) Q[i * E + i] = 0.01 * (1 + i);
  for (auto& v : z0) v = 3.4641016151377544 * (urand() - 0.5);      // unit variance, cheap (a schedule is up to 2^27 values)
  for (int64_t t = 0; t < T; t++) {
    dts[t] = 0.01 * (1 + (t % 3));
    for (int i = 0; i < Z; i++) R[t * ZZ + i * Z + i] = 0.01 * (1.0 + 0.1 * ((t + i) % 5));
  }
  double *dx, *dP, *dz, *dtx = nullptr, *dtP = nullptr;
  uint8_t* dfl;
  CK(hipMalloc((void**)&dx, sizeof(double) * n * D));
  CK(hipMalloc((void**)&dP, sizeof(double) * n * EE));
  CK(hipMalloc((void**)&dz, sizeof(double) * T * n * Z));
  CK(hipMalloc((void**)&dfl, (size_t)T * n));
  // AB_D2D=1: the observations are put in place by a device-to-device copy (what a Python caller's clone() does) instead of a
  // host-to-device copy -- the lines are then dirty in the L2 / Infinity Cache when the run starts
  double* dz0 = nullptr;
  if (std::getenv("AB_D2D")) {
    CK(hipMalloc((void**)&dz0, sizeof(double) * T * n * Z));
    CK(hipMemcpy(dz0, z0.data(), sizeof(double) * T * n * Z, hipMemcpyHostToDevice));
  }
  if (trace) {
    CK(hipMalloc((void**)&dtx, sizeof(double) * T * n * D));
    CK(hipMalloc((void**)&dtP, sizeof(double) * T * n * EE));
  }
  double* dQ = dev(Q);
  double* dR = dev(R);
  double* ddt = dev(dts);
  int32_t* dk = dev(kinds);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  std::vector<double> rx, rP, ry;
  std::vector<uint8_t> rf;
  for (int li = 9; li < argc; li++) {
    void* h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::printf("%s: dlopen failed: %s\n", argv[li], dlerror()); continue; }
    run_fn run = (run_fn)dlsym(h, (name + "_batch_run").c_str());
    auto unroll = (int (*)())dlsym(h, (name + "_run_unroll").c_str());
    if (!run) { std::printf("%s: no %s_batch_run\n", argv[li], name.c_str()); continue; }
    float best = 1e30f, sum = 0.f;
    std::vector<double> hx(n * D), hP(n * EE), hy(T * n * Z);
    std::vector<uint8_t> hf((size_t)T * n);
    for (int r = 0; r < reps + 1; r++) {            // first pass: warm-up and the results that are compared
      CK(hipMemcpy(dx, x0.data(), sizeof(double) * n * D, hipMemcpyHostToDevice));
      CK(hipMemcpy(dP, P0.data(), sizeof(double) * n * EE, hipMemcpyHostToDevice));
      if (dz0) CK(hipMemcpy(dz, dz0, sizeof(double) * T * n * Z, hipMemcpyDeviceToDevice));
      else CK(hipMemcpy(dz, z0.data(), sizeof(double) * T * n * Z, hipMemcpyHostToDevice));
      CK(hipMemset(dfl, 0xEE, (size_t)T * n));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      const int rc = run(dx, dP, dQ, dk, ddt, T, dz, dR, n, 1, r == 0 ? dfl : nullptr, dtx, dtP, nullptr, nullptr, nullptr);      // timed passes: no flags, like bench.py
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      if (rc != 0) { std::printf("%s: batch_run returned %d\n", argv[li], rc); break; }
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r == 0) {
        CK(hipMemcpy(hx.data(), dx, sizeof(double) * n * D, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hP.data(), dP, sizeof(double) * n * EE, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hy.data(), dz, sizeof(double) * T * n * Z, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), dfl, (size_t)T * n, hipMemcpyDeviceToHost));
      } else {
        if (ms < best) best = ms;
        sum += ms;
      }
    }
    std::printf("%-44s unroll %2d  %9.4f ms best %9.4f ms avg  %8.2f G steps/s", argv[li], unroll ? unroll() : -1, best, sum / (reps > 0 ? reps : 1),
                (double)n * T / (best * 1e-3) * 1e-9);
    if (rx.empty()) {
      rx = hx; rP = hP; ry = hy; rf = hf;
      double s;
      maxdiff(hx, hx, &s);
      size_t nz = 0;
      for (auto f : hf) nz += f != 0;
      std::printf("  (reference; |x|max %.3g, nonzero flags %zu)\n", s, nz);
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
