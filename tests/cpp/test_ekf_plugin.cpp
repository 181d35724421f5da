// Host side of the C++ plugin hook, written the way the reference's host code uses it (rednose/helpers/ekf_load.cc:4-39 keeps a
// registry filled by ekf_register and loads lib{name}.so + ekf_get(); rednose/helpers/ekf_sym.cc:196-219 then calls the filter
// through the descriptor): loads a rednose_amd library, checks the descriptor, and -- with a stream file -- replays
// (t, z) pairs through ekf->predict / ekf->updates.at(1), the two calls EKFSym::predict / ::update make.
//   test_ekf_plugin <generated_dir> <name>                       descriptor only (no GPU needed)
//   test_ekf_plugin <generated_dir> kinematic <stream.txt>       + known-answer stream (GPU)
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "rednose_amd/ekf_plugin.h"

static std::vector<const EKF*> registry;
void ekf_register(const EKF* e) { registry.push_back(e); }       // strong definition: libraries register themselves on load

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string path = std::string(argv[1]) + "/lib" + argv[2] + ".so";
  void* h = dlopen(path.c_str(), RTLD_NOW);
  if (!h) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  void* (*get)() = (void* (*)())dlsym(h, "ekf_get");
  if (!get) { std::fprintf(stderr, "no ekf_get\n"); return 4; }
  const EKF* ekf = (const EKF*)get();
  std::printf("name %s kinds", ekf->name.c_str());
  for (int k : ekf->kinds) std::printf(" %d", k);
  std::printf(" feature_kinds %zu registered %d same %d\n", ekf->feature_kinds.size(), (int)registry.size(),
              (int)(registry.size() == 1 && registry[0] == ekf));
  bool ok = ekf->f_fun && ekf->F_fun && ekf->err_fun && ekf->inv_err_fun && ekf->H_mod_fun && ekf->predict;
  for (int k : ekf->kinds) ok = ok && ekf->hs.count(k) && ekf->Hs.count(k) && ekf->updates.count(k);
  for (int k : ekf->feature_kinds) ok = ok && ekf->Hes.count(k);
  std::printf("complete %d sets %zu extra_routines %zu\n", (int)ok, ekf->sets.size(), ekf->extra_routines.size());
  if (argc >= 4) {
    double x[2] = {0.5, 0.0}, P[4] = {1.0, 0.0, 0.0, 1.0}, Q[4] = {0.1 * 0.1, 0.0, 0.0, 2.0 * 2.0}, R[1] = {0.1 * 0.1}, ea[1] = {0.0};
    std::ifstream in(argv[3]);
    double t, z, t_prev = NAN;
    long steps = 0;
    while (in >> t >> z) {
      const double dt = std::isnan(t_prev) ? 0.0 : t - t_prev;      // EKFSym::predict: first call has dt = 0 (ekf_sym.cc:198-204)
      t_prev = t;
      ekf->predict(x, P, Q, dt);
      double zz[1] = {z};
      ekf->updates.at(1)(x, P, zz, R, ea);
      steps++;
    }
    std::printf("steps %ld x %.17g %.17g std %.17g %.17g\n", steps, x[0], x[1], std::sqrt(P[0]), std::sqrt(P[3]));
  }
  return ok ? 0 : 1;
}
