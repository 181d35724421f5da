// ekf_plugin.h -- the plugin descriptor a generated filter library hands to C++ hosts, and the hook that publishes it.
//
// This is the interface of /root/reference/rednose/helpers/ekf.h:14-42 (struct EKF, ekf_lib_init), restated so that a
// rednose_amd library can be loaded by the reference's own C++ host code -- `ekf_load_and_register(dir, name)`
// (rednose/helpers/ekf_load.cc:22-39) dlopens lib{name}.so, calls `ekf_get()` and reads the struct; `EKFSym`
// (rednose/helpers/ekf_sym.cc:196-219) then calls `ekf->predict(...)` / `ekf->updates.at(kind)(...)` through it.  The member
// list, order and types ARE the binary interface (a libstdc++-ABI struct, not a C one), so they are those of the reference;
// what is left out is its `#include <eigen3/Eigen/Dense>`, which the struct does not use.
// Every function pointer in the descriptor of a rednose_amd library is one of the scalar host-pointer entry points of
// {name}.h, i.e. a batch-of-one launch on the GPU.
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

typedef void (*extra_routine_t)(double *, double *);

struct EKF {
  std::string name;
  std::vector<int> kinds;
  std::vector<int> feature_kinds;

  void (*f_fun)(double *, double, double *);
  void (*F_fun)(double *, double, double *);
  void (*err_fun)(double *, double *, double *);
  void (*inv_err_fun)(double *, double *, double *);
  void (*H_mod_fun)(double *, double *);
  void (*predict)(double *, double *, double *, double);
  std::unordered_map<int, void (*)(double *, double *, double *)> hs = {};
  std::unordered_map<int, void (*)(double *, double *, double *)> Hs = {};
  std::unordered_map<int, void (*)(double *, double *, double *, double *, double *)> updates = {};
  std::unordered_map<int, void (*)(double *, double *, double *)> Hes = {};
  std::unordered_map<std::string, void (*)(double)> sets = {};
  std::unordered_map<std::string, extra_routine_t> extra_routines = {};
};

// host side defines this to collect descriptors at load time (rednose/helpers/ekf_load.cc:9-11); weak: absent is fine
extern void __attribute__((weak)) ekf_register(const EKF *ptr);
