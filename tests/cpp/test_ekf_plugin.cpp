// Host side of the C++ plugin hook, written the way the reference's host code uses it (rednose/helpers/ekf_load.cc:4-39 keeps a
// registry filled by ekf_register and loads lib{name}.so + ekf_get(); rednose/helpers/ekf_sym.cc:196-219 then calls the filter
// through the descriptor): loads a rednose_amd library, checks the descriptor, and -- with a stream file -- replays
// (t, z) pairs through ekf->predict / ekf->updates.at(1), the two calls EKFSym::predict / ::update make.
// Two builds: the look-alike host below (its own registry), and -DRN_REF_LOADER linked against the reference's own ekf_load.cc object.
//   test_ekf_plugin <generated_dir> <name>                       descriptor only (no GPU needed)
//   test_ekf_plugin <generated_dir> kinematic <stream.txt>       + known-answer stream (GPU)
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "rednose_amd/ekf_plugin.h"

#ifdef RN_REF_LOADER
// Host = the REFERENCE'S OWN loader: /root/reference/rednose/helpers/ekf_load.cc compiled unmodified into oracle/_ref/ref_ekf_load.o
// (__graft_entry__.build(), when /root/reference is present; the object travels to the GPU box like the oracle's libraries) and
// linked here.  These are its declarations (rednose/helpers/ekf_load.h:6-9); nothing of the registry is re-implemented.
std::vector<const EKF*>& ekf_get_all();
const EKF* ekf_lookup(const std::string& ekf_name);
void ekf_load_and_register(const std::string& ekf_directory, const std::string& ekf_name);
#else
static std::vector<const EKF*> registry;
void ekf_register(const EKF* e) { registry.push_back(e); }       // strong definition: libraries register themselves on load
#endif

int main(int argc, char** argv) {
  if (argc < 3) return 2;
#ifdef RN_REF_LOADER
  // ekf_load.cc:22-39: dlopen(dir/lib{name}.so, RTLD_NOW) -> dlsym("ekf_get") -> ekf_register.  The library's constructor has
  // already registered the descriptor through the loader's own ekf_register by then (ekf.h:35-42), so the registry holds the
  // same pointer twice -- exactly what the reference's libraries do to it; a second call is a no-op (:23-25).
  ekf_load_and_register(argv[1], argv[2]);
  ekf_load_and_register(argv[1], argv[2]);
  const EKF* ekf = ekf_lookup(argv[2]);
  if (!ekf) { std::fprintf(stderr, "ekf_lookup(%s) found nothing\n", argv[2]); return 4; }
  const std::vector<const EKF*>& registry = ekf_get_all();
  bool same = !registry.empty() && registry.size() <= 2;
  for (const EKF* e : registry) same = same && e == ekf;
  std::printf("name %s kinds", ekf->name.c_str());
  for (int k : ekf->kinds) std::printf(" %d", k);
  std::printf(" feature_kinds %zu registered %d same %d\n", ekf->feature_kinds.size(), (int)registry.size(), (int)same);
#else
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
#endif
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
