// C++ driver for rednose_amd::EKFSymBatch: replays a (t, z) stream read from a text file through the kinematic filter
// for a batch of identical filters and prints the final state / sqrt(diag P) of the first and last filter.
// tests/test_gpu_cpp.py feeds it the stream of /root/reference/examples/test_kinematic_kf.py and checks the literals.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "rednose_amd/ekf_sym_batch.hpp"

// Rewind mode: the stream arrives out of order (test_compare.py swaps two samples); with the checkpoint ring the filter must
// end where the reference's orchestrators end (tests/golden/compare_rewind.npz).  Prints the filter time and state after every
// observation of filter 0 and the last filter.
static int run_rewind(const char* dir, const char* stream, int64_t n) {
  rednose_amd::EKFSymBatch kf(dir, "kinematic", {0.1 * 0.1, 0.0, 0.0, 2.0 * 2.0}, {0.5, 0.0}, {1.0, 0.0, 0.0, 1.0}, n, false, nullptr, 512, 1.0);
  std::ifstream in(stream);
  std::vector<double> zs(n);
  double* z_dev = nullptr;
  if (hipMalloc((void**)&z_dev, sizeof(double) * n) != hipSuccess) return 3;
  const double R[1] = {0.1 * 0.1};
  double t, z;
  while (in >> t >> z) {
    std::fill(zs.begin(), zs.end(), z);
    if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
    const bool applied = kf.predict_and_update_batch(t, 1, z_dev, R);
    kf.synchronize();
    const std::vector<double> x = kf.state();
    std::printf("%d %.17g %.17g %.17g %.17g %.17g\n", applied ? 1 : 0, kf.get_filter_time(), x[0], x[1], x[(n - 1) * 2], x[(n - 1) * 2 + 1]);
  }
  // far older than max_rewind_age: dropped, state untouched
  const std::vector<double> before = kf.state();
  const bool dropped = !kf.predict_and_update_batch(kf.get_filter_time() - 5.0, 1, z_dev, R);
  std::printf("too_old_dropped %d untouched %d\n", dropped ? 1 : 0, (int)(kf.state() == before));
  (void)hipFree(z_dev);
  return 0;
}

// Timelines mode: per-filter logs (tests/golden/perfilter_timelines.npz part A: one line per arrival, "t0 z0 t1 z1 ..." for the n
// filters): every filter on its own clock with its own checkpoint ring.  Prints per arrival and filter: ignored, filter time, x.
static int run_timelines(const char* dir, const char* stream, int64_t n) {
  rednose_amd::EKFSymBatch kf(dir, "kinematic", {0.1 * 0.1, 0.0, 0.0, 2.0 * 2.0}, {0.5, 0.0}, {1.0, 0.0, 0.0, 1.0}, n, false, nullptr, 512, 1.0);
  std::ifstream in(stream);
  std::vector<double> ts(n), zs(n);
  double* z_dev = nullptr;
  if (hipMalloc((void**)&z_dev, sizeof(double) * n) != hipSuccess) return 3;
  const double R[1] = {0.1 * 0.1};
  while (true) {
    for (int64_t i = 0; i < n; i++) in >> ts[i] >> zs[i];
    if (!in) break;
    if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
    kf.predict_and_update_batch_per_filter(ts.data(), nullptr, 1, z_dev, R);
    kf.synchronize();
    const std::vector<double> x = kf.state();
    for (int64_t i = 0; i < n; i++)
      std::printf("%d %.17g %.17g %.17g%s", (int)kf.ignored()[i], kf.filter_times()[i], x[2 * i], x[2 * i + 1], i + 1 < n ? " " : "\n");
  }
  (void)hipFree(z_dev);
  return 0;
}

// Robustness mode (round-3 review of the per-filter path): (a) a time-varying R -- a new matrix on every call -- must not grow the
// host-side noise table without bound; (b) an observation that is too old for one filter's ring sets bits 4 and 5 of that filter's
// flag byte; (c) a call that is refused (a late observation without a ring) throws BEFORE anything is touched: filter times and
// states are what they were, and the next valid call works.
static int run_robustness(const char* dir) {
  const int64_t n = 5;
  const std::vector<double> Q = {0.1 * 0.1, 0.0, 0.0, 2.0 * 2.0}, x0 = {0.5, 0.0}, P0 = {1.0, 0.0, 0.0, 1.0};
  rednose_amd::EKFSymBatch kf(dir, "kinematic", Q, x0, P0, n, false, nullptr, 8, 1.0);
  double* z_dev = nullptr;
  uint8_t* fl_dev = nullptr;
  if (hipMalloc((void**)&z_dev, sizeof(double) * n) != hipSuccess || hipMalloc((void**)&fl_dev, n) != hipSuccess) return 3;
  std::vector<double> ts(n), zs(n, 0.1);
  size_t table_max = 0;
  for (int it = 0; it < 400; it++) {
    for (int64_t i = 0; i < n; i++) ts[i] = 0.01 * (it + 1) + 1e-4 * i;
    const double R[1] = {0.01 * (1.0 + 1e-3 * it)};                       // never the same matrix twice
    if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
    kf.predict_and_update_batch_per_filter(ts.data(), nullptr, 1, z_dev, R, fl_dev);
    table_max = std::max(table_max, kf.noise_table_size());
  }
  // (b) filter 2 gets an observation 3 s in its past (ring: 8 entries, max_rewind_age 1 s): ignored for it alone
  for (int64_t i = 0; i < n; i++) ts[i] = 0.01 * 401 + 1e-4 * i;
  ts[2] -= 3.0;
  const double R1[1] = {0.01};
  if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
  const int64_t ign = kf.predict_and_update_batch_per_filter(ts.data(), nullptr, 1, z_dev, R1, fl_dev);
  kf.synchronize();
  std::vector<uint8_t> fl(n);
  if (hipMemcpy(fl.data(), fl_dev, n, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  std::printf("table_max %zu ignored %lld flags", table_max, (long long)ign);
  for (int64_t i = 0; i < n; i++) std::printf(" %d", (int)fl[i]);
  std::printf("\n");
  // (c) no ring: a late observation for one filter is refused up front
  rednose_amd::EKFSymBatch nr(dir, "kinematic", Q, x0, P0, n, false, nullptr, 0, 1.0);
  for (int64_t i = 0; i < n; i++) ts[i] = 1.0 + 0.1 * i;
  if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
  nr.predict_and_update_batch_per_filter(ts.data(), nullptr, 1, z_dev, R1);
  nr.synchronize();
  const std::vector<double> ft0 = nr.filter_times(), xs0 = nr.state();
  std::vector<double> t2(ts);
  t2[0] += 0.5; t2[1] += 0.5; t2[3] -= 0.2;                                 // filters 0 and 1 would advance, filter 3 is late
  bool threw = false;
  try { nr.predict_and_update_batch_per_filter(t2.data(), nullptr, 1, z_dev, R1); } catch (const std::runtime_error&) { threw = true; }
  nr.synchronize();
  const bool untouched = nr.filter_times() == ft0 && nr.state() == xs0;
  for (int64_t i = 0; i < n; i++) t2[i] = ts[i] + 0.5;
  if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
  nr.predict_and_update_batch_per_filter(t2.data(), nullptr, 1, z_dev, R1);
  nr.synchronize();
  std::printf("refused %d untouched %d next_call_time %.17g\n", threw ? 1 : 0, untouched ? 1 : 0, nr.filter_times()[3]);
  (void)hipFree(z_dev); (void)hipFree(fl_dev);
  return 0;
}

// Multi mode: n observations per call (EKFSym::predict_and_update_batch with vectors of z / R, ekf_sym.cc:83-117,158-194) on the 9-state
// model with the shared-timeline ring.  The log is one filter's of tests/golden/multi_obs.npz part A -- calls of 1-3 observations with a
// different noise matrix each, one LATE multi-observation call that rewinds over multi-observation checkpoints -- fed to a batch of
// identical filters.  File: Q (81), x0 (9), P0 (81), R of kinds 1 / 2 / 3 (9 + 1 + 9), then per call "t kind nobs" and per
// observation Z values + the noise scale.  Prints per call: applied, filter time, x of filter 0 and of the last filter, and the
// residuals of filter 0 (nobs x Z).
static int run_multi(const char* dir, const char* stream, int64_t n) {
  std::ifstream in(stream);
  std::vector<double> Q(81), x0(9), P0(81), R1(9), R2(1), R3(9);
  for (auto* v : {&Q, &x0, &P0, &R1, &R2, &R3}) for (double& e : *v) in >> e;
  rednose_amd::EKFSymBatch kf(dir, "kinematic9", Q, x0, P0, n, false, nullptr, 64, 1.0);
  const int NM = 3;
  std::vector<double*> zbuf(NM, nullptr);
  uint8_t* fl_dev = nullptr;
  for (auto& p : zbuf) if (hipMalloc((void**)&p, sizeof(double) * n * 3 + 16) != hipSuccess) return 3;
  if (hipMalloc((void**)&fl_dev, NM * n) != hipSuccess) return 3;
  double t;
  int kind, nobs;
  while (in >> t >> kind >> nobs) {
    const int Z = kf.zdim(kind);
    const std::vector<double>& Rk = kind == 1 ? R1 : (kind == 2 ? R2 : R3);
    std::vector<std::vector<double>> Rs(nobs);
    std::vector<double*> zs;
    std::vector<const double*> Rp;
    for (int i = 0; i < nobs; i++) {
      std::vector<double> zi(Z), host((size_t)n * Z);
      for (double& e : zi) in >> e;
      double scale;
      in >> scale;
      for (int64_t f = 0; f < n; f++) std::copy(zi.begin(), zi.end(), host.begin() + f * Z);
      if (hipMemcpy(zbuf[i], host.data(), sizeof(double) * n * Z, hipMemcpyHostToDevice) != hipSuccess) return 3;
      Rs[i] = Rk;
      for (double& e : Rs[i]) e *= scale;
      zs.push_back(zbuf[i]);
      Rp.push_back(Rs[i].data());
    }
    const bool applied = kf.predict_and_update_batch(t, kind, zs, Rp, fl_dev);
    kf.synchronize();
    const std::vector<double> x = kf.state();
    std::printf("%d %.17g", applied ? 1 : 0, kf.get_filter_time());
    for (int64_t f : {(int64_t)0, n - 1}) for (int j = 0; j < 9; j++) std::printf(" %.17g", x[f * 9 + j]);
    for (int i = 0; i < nobs; i++) {
      std::vector<double> y(Z);
      if (hipMemcpy(y.data(), zbuf[i], sizeof(double) * Z, hipMemcpyDeviceToHost) != hipSuccess) return 3;
      for (double e : y) std::printf(" %.17g", e);
    }
    std::printf("\n");
  }
  bool threw = false;      // n = 0 observations / mismatched vectors are refused (the reference asserts, ekf_sym.cc:159-160)
  try { kf.predict_and_update_batch(1e3, 1, std::vector<double*>{zbuf[0], zbuf[1]}, std::vector<const double*>{R1.data()}); } catch (const std::runtime_error&) { threw = true; }
  std::printf("mismatch_threw %d\n", threw ? 1 : 0);
  for (auto p : zbuf) (void)hipFree(p);
  (void)hipFree(fl_dev);
  return 0;
}

// Timelines-multi mode: per-filter logs whose calls carry 1-3 observations (tests/golden/multi_obs.npz part C; the noise of observation m of a call is the
// kind's matrix x scale[m] for every filter).  File: Q (81), x0 (9), P0 (81), R of kinds 1 / 2 / 3 (9 + 1 + 9), scale (3), "N T", then per arrival N lines
// "t kind n z(9)".  One group of masked launches per (kind, n) present at an arrival.  Prints per arrival and filter: filter time, x (9), residuals (9).
static int run_timelines_multi(const char* dir, const char* stream) {
  std::ifstream in(stream);
  std::vector<double> Q(81), x0(9), P0(81), R1(9), R2(1), R3(9), scale(3);
  for (auto* v : {&Q, &x0, &P0, &R1, &R2, &R3, &scale}) for (double& e : *v) in >> e;
  int64_t n; int T;
  in >> n >> T;
  rednose_amd::EKFSymBatch kf(dir, "kinematic9", Q, x0, P0, n, false, nullptr, 64, 1.0);
  kf.set_max_observations_per_call(3);
  std::vector<double*> zbuf(3, nullptr);
  for (auto& p : zbuf) if (hipMalloc((void**)&p, sizeof(double) * n * 3 + 16) != hipSuccess) return 3;
  for (int a = 0; a < T; a++) {
    std::vector<double> ts(n), zz((size_t)n * 9), yy((size_t)n * 9, 0.0);
    std::vector<int> kd(n), no(n);
    for (int64_t i = 0; i < n; i++) { in >> ts[i] >> kd[i] >> no[i]; for (int e = 0; e < 9; e++) in >> zz[i * 9 + e]; }
    for (int k = 1; k <= 3; k++) {
      const int Z = kf.zdim(k);
      const std::vector<double>& Rk = k == 1 ? R1 : (k == 2 ? R2 : R3);
      for (int m = 1; m <= 3; m++) {
        std::vector<uint8_t> act(n, 0);
        bool any = false;
        for (int64_t i = 0; i < n; i++) { act[i] = (uint8_t)(kd[i] == k && no[i] == m); any = any || act[i]; }
        if (!any) continue;
        std::vector<std::vector<double>> Rs(m);
        std::vector<double*> zs;
        std::vector<const double*> Rp;
        for (int o = 0; o < m; o++) {
          std::vector<double> host((size_t)n * Z, 0.0);
          for (int64_t i = 0; i < n; i++) for (int e = 0; e < Z; e++) host[i * Z + e] = zz[i * 9 + o * 3 + e];
          if (hipMemcpy(zbuf[o], host.data(), sizeof(double) * n * Z, hipMemcpyHostToDevice) != hipSuccess) return 3;
          Rs[o] = Rk;
          for (double& e : Rs[o]) e *= scale[o];
          zs.push_back(zbuf[o]);
          Rp.push_back(Rs[o].data());
        }
        if (kf.predict_and_update_batch_per_filter(ts.data(), act.data(), k, zs, Rp) != 0) return 5;
        kf.synchronize();
        for (int o = 0; o < m; o++) {
          std::vector<double> host((size_t)n * Z);
          if (hipMemcpy(host.data(), zbuf[o], sizeof(double) * n * Z, hipMemcpyDeviceToHost) != hipSuccess) return 3;
          for (int64_t i = 0; i < n; i++) if (act[i]) for (int e = 0; e < Z; e++) yy[i * 9 + o * 3 + e] = host[i * Z + e];
        }
      }
    }
    kf.synchronize();
    const std::vector<double> x = kf.state();
    for (int64_t i = 0; i < n; i++) {
      std::printf("%.17g", kf.filter_times()[i]);
      for (int e = 0; e < 9; e++) std::printf(" %.17g", x[i * 9 + e]);
      for (int e = 0; e < 9; e++) std::printf(" %.17g", yy[i * 9 + e]);
      std::printf("\n");
    }
  }
  bool threw = false;      // more observations than the rings were sized for
  try {
    std::vector<double> ts(n, 1e3);
    kf.predict_and_update_batch_per_filter(ts.data(), nullptr, 1, std::vector<double*>{zbuf[0], zbuf[1], zbuf[2], zbuf[0]},
                                           std::vector<const double*>{R1.data(), R1.data(), R1.data(), R1.data()});
  } catch (const std::runtime_error&) { threw = true; }
  std::printf("too_many_threw %d\n", threw ? 1 : 0);
  for (auto p : zbuf) (void)hipFree(p);
  return 0;
}

// Globals mode: set_global / get_extra_routine on a model generated with global_vars (tests/test_global_vars.py's gv_runtime)
static int run_globals(const char* dir) {
  rednose_amd::EKFSymBatch kf(dir, "gv_runtime", {0.01, 0.0, 0.0, 4.0}, {0.5, 0.3}, {1.0, 0.0, 0.0, 1.0}, 3);
  kf.set_global("gain", 2.5);
  kf.predict(0.0);
  kf.predict(0.1);
  kf.synchronize();
  const std::vector<double> x = kf.state();
  bool threw = false;
  try { kf.set_global("nope", 1.0); } catch (const std::runtime_error&) { threw = true; }
  auto f = kf.get_extra_routine("f_fun");          // any exported host-pointer routine resolves by name
  std::printf("x %.17g %.17g unknown_global_threw %d routine %d\n", x[0], x[1], threw ? 1 : 0, f != nullptr);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <generated_dir> <stream.txt> <batch> [rewind] | %s <generated_dir> - - globals\n", argv[0], argv[0]);
    return 2;
  }
  const int64_t n = std::atoll(argv[3]);
  try {
    if (argc >= 5 && std::string(argv[4]) == "rewind") return run_rewind(argv[1], argv[2], n);
    if (argc >= 5 && std::string(argv[4]) == "globals") return run_globals(argv[1]);
    if (argc >= 5 && std::string(argv[4]) == "robustness") return run_robustness(argv[1]);
    if (argc >= 5 && std::string(argv[4]) == "timelines") return run_timelines(argv[1], argv[2], n);
    if (argc >= 5 && std::string(argv[4]) == "multi") return run_multi(argv[1], argv[2], n);
    if (argc >= 5 && std::string(argv[4]) == "timelines_multi") return run_timelines_multi(argv[1], argv[2]);
    rednose_amd::EKFSymBatch kf(argv[1], "kinematic", {0.1 * 0.1, 0.0, 0.0, 2.0 * 2.0}, {0.5, 0.0}, {1.0, 0.0, 0.0, 1.0}, n);
    std::ifstream in(argv[2]);
    std::vector<double> zs(n);
    double* z_dev = nullptr;
    if (hipMalloc((void**)&z_dev, sizeof(double) * n) != hipSuccess) return 3;
    const double R[1] = {0.1 * 0.1};
    double t, z;
    long steps = 0;
    while (in >> t >> z) {
      std::fill(zs.begin(), zs.end(), z);
      if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
      if (!kf.predict_and_update_batch(t, 1, z_dev, R)) return 4;
      steps++;
    }
    // a late observation must be rejected, an unknown kind must throw
    const bool late = kf.predict_and_update_batch(0.0, 1, z_dev, R);
    bool threw = false;
    try { kf.predict_and_update_batch(1e9, 7, z_dev, R); } catch (const std::out_of_range&) { threw = true; }
    kf.synchronize();
    const std::vector<double> x = kf.state(), P = kf.covs();
    // Mahalanobis distance of an observation 0.3 above every filter's position estimate (state must stay untouched)
    for (int64_t i = 0; i < n; i++) zs[i] = x[i * 2] + 0.3;
    double* d2_dev = nullptr;
    if (hipMalloc((void**)&d2_dev, sizeof(double) * n) != hipSuccess) return 3;
    if (hipMemcpy(z_dev, zs.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
    kf.maha_distance(1, z_dev, R, d2_dev);
    kf.synchronize();
    std::vector<double> d2(n);
    if (hipMemcpy(d2.data(), d2_dev, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    const bool untouched = kf.state() == x && kf.covs() == P;
    (void)hipFree(d2_dev);
    std::printf("steps %ld late_rejected %d unknown_kind_threw %d filter_time %.17g\n", steps, late ? 0 : 1, threw ? 1 : 0, kf.get_filter_time());
    for (int64_t i : {(int64_t)0, n - 1})
      std::printf("x %.17g %.17g std %.17g %.17g\n", x[i * 2], x[i * 2 + 1], std::sqrt(P[i * 4]), std::sqrt(P[i * 4 + 3]));
    std::printf("maha %.17g %.17g untouched %d\n", d2[0], 0.3 * 0.3 / (P[0] + R[0]), untouched ? 1 : 0);
    (void)hipFree(z_dev);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
