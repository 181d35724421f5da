// Per-call cost of the C++ orchestrator (include/rednose_amd/ekf_sym_batch.hpp) at step granularity: EKFSymBatch::predict_and_update_batch on the
// headline workload (kinematic6, 65 536 filters, one fused predict + update launch per call), host wall clock over K calls + one synchronize,
// beside the same launches issued straight through the C ABI.
//   hipcc -O2 -std=c++17 -Iinclude tools/cpp_orch_time.cpp -o tools/cpp_orch_time -ldl && tools/cpp_orch_time generated [K]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rednose_amd/ekf_sym_batch.hpp"

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "generated";
  const int K = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int64_t n = 65536;
  const int D = 6, E = 6, Z = 3;
  std::vector<double> Q(E * E, 0.0), x0(D, 0.0), P0(E * E, 0.0), R(Z * Z, 0.0);
  for (int i = 0; i < E; i++) { Q[i * E + i] = 0.01; P0[i * E + i] = 1.0; }
  for (int i = 0; i < Z; i++) R[i * Z + i] = 0.04;
  for (int ring : {0, 8}) {
    rednose_amd::EKFSymBatch f(dir, "kinematic6", Q, x0, P0, n, false, nullptr, ring);
    double* z = nullptr;
    if (hipMalloc((void**)&z, sizeof(double) * n * Z) != hipSuccess) return 3;
    (void)hipMemset(z, 0, sizeof(double) * n * Z);
    double t = 0.0;
    for (int i = 0; i < 500; i++) { t += 0.01; f.predict_and_update_batch(t, 1, z, R.data()); }
    f.synchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; i++) { t += 0.01; f.predict_and_update_batch(t, 1, z, R.data()); }
    f.synchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    std::printf("EKFSymBatch::predict_and_update_batch, kinematic6 x %lld, rewind ring %d: %.2f us per call (wall, %d calls)\n", (long long)n, ring, us, K);
    (void)hipFree(z);
  }
  // per-filter timelines: every filter on its own clock (times on the host, one observation each per call), without and with per-filter rings
  for (int ring : {0, 8}) {
    rednose_amd::EKFSymBatch f(dir, "kinematic6", Q, x0, P0, n, false, nullptr, ring);
    double* z = nullptr;
    if (hipMalloc((void**)&z, sizeof(double) * n * Z) != hipSuccess) return 3;
    (void)hipMemset(z, 0, sizeof(double) * n * Z);
    std::vector<double> tt(n), off(n), off_late(n);
    for (int64_t i = 0; i < n; i++) { off[i] = 0.005 * (double)i / (double)n; off_late[i] = off[i] - (i % 100 == 0 ? 0.015 : 0.0); }
    double t = 0.0, fill_us = 0.0;
    auto call = [&](bool late) {
      t += 0.01;
      const auto a = std::chrono::steady_clock::now();
      const double* o = late ? off_late.data() : off.data();
      for (int64_t i = 0; i < n; i++) tt[i] = t + o[i];
      fill_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();      // the caller's own work, not the orchestrator's
      f.predict_and_update_batch_per_filter(tt.data(), nullptr, 1, z, R.data());
    };
    for (int i = 0; i < 100; i++) call(false);
    f.synchronize();
    for (int late = 0; late <= (ring ? 1 : 0); late++) {
      const int Kp = 300;
      fill_us = 0.0;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < Kp; i++) call(late != 0);
      f.synchronize();
      const double us = (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - fill_us) / Kp;
      std::printf("EKFSymBatch::predict_and_update_batch_per_filter, kinematic6 x %lld, ring %d, %s: %.1f us per call (wall, %d calls)\n", (long long)n, ring,
                  late ? "1 % of the filters late by one call" : "in order", us, Kp);
    }
    (void)hipFree(z);
  }
  return 0;
}
