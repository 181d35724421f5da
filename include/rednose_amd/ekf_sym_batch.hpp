// ekf_sym_batch.hpp -- C++ host-side orchestrator for N filters resident on one MI355X (header-only).
//
// The reference's C++ orchestrator is EKFS::EKFSym (/root/reference/rednose/helpers/ekf_sym.h:44-66,
// ekf_sym.cc:7-223): one filter, Eigen members, plugin lookup through ekf_load.cc.  This class is its batched
// counterpart above the C ABI of a generated rednose_amd library (include/rednose_amd_filter.h):
//   * library discovery as in ekf_load.cc:22-39 -- dlopen("<dir>/lib<name>.so"), symbols looked up by name;
//   * same vocabulary and semantics: init_state (:45-51), state/covs (:53-59), get/set_filter_time (:61-67),
//     predict (:196-209: first call adopts t, dt >= 0 required), predict_and_update_batch (:83-117 without the
//     rewind ring: late observations are rejected, false is returned), quaternion renormalisation after predict
//     and update when quaternion_idxs were given to gen_code (:207,213 -- done inside the kernels);
//   * no Eigen: state lives in HBM as x (N, D), P (N, E, E) row-major fp64; z is a DEVICE pointer (N, Z) that the
//     kernel overwrites with the residual y; R is a host Z x Z matrix shared by the batch.
// Errors throw std::runtime_error carrying {name}_last_error_string(); nothing aborts.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rednose_amd {

class EKFSymBatch {
 public:
  EKFSymBatch(const std::string& directory, const std::string& name, const std::vector<double>& Q,
              const std::vector<double>& x_initial, const std::vector<double>& P_initial, int64_t batch,
              bool normalize_quaternions = false, hipStream_t stream = nullptr)
      : name_(name), n_(batch), norm_quats_(normalize_quaternions ? 1 : 0), stream_(stream) {
    const std::string path = directory + "/lib" + name + ".so";
    handle_ = dlopen(path.c_str(), RTLD_NOW);
    if (!handle_) throw std::runtime_error("rednose_amd: cannot load " + path + ": " + dlerror());
    int dims[3];
    sym<void (*)(int*)>("dims")(dims);
    D_ = dims[0];
    E_ = dims[1];
    if ((int64_t)x_initial.size() != D_ || (int64_t)P_initial.size() != (int64_t)E_ * E_ || Q.size() != P_initial.size())
      throw std::runtime_error("rednose_amd: initial state / covariance / Q do not match the library's dimensions");
    const int nk = sym<int (*)()>("num_kinds")();
    std::vector<int> kinds(nk);
    sym<void (*)(int*)>("kinds")(kinds.data());
    for (int k : kinds) zdim_[k] = sym<int (*)(int)>("kind_zdim")(k);
    batch_predict_ = sym<predict_fn>("batch_predict");
    hip(hipMalloc((void**)&x_, sizeof(double) * n_ * D_), "hipMalloc x");
    hip(hipMalloc((void**)&P_, sizeof(double) * n_ * E_ * E_), "hipMalloc P");
    hip(hipMalloc((void**)&Q_, sizeof(double) * E_ * E_), "hipMalloc Q");
    hip(hipMalloc((void**)&R_, sizeof(double) * 64 * 64), "hipMalloc R");
    hip(hipMemcpy(Q_, Q.data(), sizeof(double) * E_ * E_, hipMemcpyHostToDevice), "copy Q");
    init_state(x_initial, P_initial, NAN);
  }

  ~EKFSymBatch() {
    (void)hipFree(x_);
    (void)hipFree(P_);
    (void)hipFree(Q_);
    (void)hipFree(R_);
    if (handle_) dlclose(handle_);
  }
  EKFSymBatch(const EKFSymBatch&) = delete;
  EKFSymBatch& operator=(const EKFSymBatch&) = delete;

  // one state / covariance broadcast to every filter (EKFSym::init_state)
  void init_state(const std::vector<double>& state, const std::vector<double>& covs, double filter_time) {
    std::vector<double> xs((size_t)n_ * D_), Ps((size_t)n_ * E_ * E_);
    for (int64_t i = 0; i < n_; i++) {
      std::copy(state.begin(), state.end(), xs.begin() + i * D_);
      std::copy(covs.begin(), covs.end(), Ps.begin() + i * E_ * E_);
    }
    init_state_batch(xs.data(), Ps.data(), filter_time);
  }

  // per-filter initial values: x (N, D), P (N, E, E) host arrays
  void init_state_batch(const double* xs, const double* Ps, double filter_time) {
    hip(hipMemcpy(x_, xs, sizeof(double) * n_ * D_, hipMemcpyHostToDevice), "copy x");
    hip(hipMemcpy(P_, Ps, sizeof(double) * n_ * E_ * E_, hipMemcpyHostToDevice), "copy P");
    filter_time_ = filter_time;
  }

  std::vector<double> state() const {
    std::vector<double> out((size_t)n_ * D_);
    hip(hipMemcpy(out.data(), x_, sizeof(double) * out.size(), hipMemcpyDeviceToHost), "read x");
    return out;
  }
  std::vector<double> covs() const {
    std::vector<double> out((size_t)n_ * E_ * E_);
    hip(hipMemcpy(out.data(), P_, sizeof(double) * out.size(), hipMemcpyDeviceToHost), "read P");
    return out;
  }
  double* state_device() { return x_; }
  double* covs_device() { return P_; }
  int dim_x() const { return D_; }
  int dim_err() const { return E_; }
  int64_t batch() const { return n_; }
  int zdim(int kind) const { return zdim_.at(kind); }       // std::out_of_range for an unknown kind, like updates.at(kind)
  void set_filter_time(double t) { filter_time_ = t; }
  double get_filter_time() const { return filter_time_; }

  void predict(double t) {
    const double dt = advance(t);
    check(batch_predict_(x_, P_, Q_, nullptr, dt, n_, norm_quats_, stream_), "batch_predict");
    filter_time_ = t;
  }

  // One fused predict(t - filter_time) + update(kind) launch over the batch.  z_dev: (N, Z) device, in: z, out: y.
  // R_host: Z x Z row-major, shared.  flags_dev: N bytes or nullptr.  Returns false (and does nothing) when the
  // observation is older than the filter time (the reference would rewind; this class does not keep a ring).
  // ea_dev: (N, kind_eadim) extra arguments of the observations (MSCKF feature tracks: the landmark), or nullptr for kinds
  // that take none; augment = true shifts the MSCKF window afterwards (EKFSym asserts !augment, ekf_sym.cc:186; the
  // Python class implements it, ekf_sym.py:365-391,527-528).
  bool predict_and_update_batch(double t, int kind, double* z_dev, const double* R_host, uint8_t* flags_dev = nullptr,
                                const double* ea_dev = nullptr, bool augment = false) {
    const int Z = zdim_.at(kind);
    if (!std::isnan(filter_time_) && t < filter_time_) return false;
    const double dt = advance(t);
    hip(hipMemcpyAsync(R_, R_host, sizeof(double) * Z * Z, hipMemcpyHostToDevice, stream_), "copy R");
    auto fn = sym<step_fn>("batch_predict_update_" + std::to_string(kind));
    check(fn(x_, P_, Q_, nullptr, dt, z_dev, R_, 0, ea_dev, n_, norm_quats_, flags_dev, stream_), "batch_predict_update");
    filter_time_ = t;
    if (augment) this->augment();
    return true;
  }

  // MSCKF window shift on every filter (libraries generated with msckf_params only)
  void augment() {
    check(sym<int (*)(double*, double*, int64_t, void*)>("batch_augment")(x_, P_, n_, stream_), "batch_augment");
  }

  // Mahalanobis distance of an observation per filter into d2_dev (N), state untouched (EKF_sym.maha_test, ekf_sym.py:626-649)
  void maha_distance(int kind, const double* z_dev, const double* R_host, double* d2_dev, const double* ea_dev = nullptr) {
    const int Z = zdim_.at(kind);
    hip(hipMemcpyAsync(R_, R_host, sizeof(double) * Z * Z, hipMemcpyHostToDevice, stream_), "copy R");
    auto fn = sym<int (*)(const double*, const double*, const double*, const double*, int, const double*, int64_t, double*, void*)>(
        "batch_maha_" + std::to_string(kind));
    check(fn(x_, P_, z_dev, R_, 0, ea_dev, n_, d2_dev, stream_), "batch_maha");
  }

  void synchronize() const { hip(hipStreamSynchronize(stream_), "synchronize"); }

 private:
  using predict_fn = int (*)(double*, double*, const double*, const double*, double, int64_t, int, void*);
  using step_fn = int (*)(double*, double*, const double*, const double*, double, double*, const double*, int, const double*,
                          int64_t, int, uint8_t*, void*);

  template <class F>
  F sym(const std::string& suffix) const {
    void* p = dlsym(handle_, (name_ + "_" + suffix).c_str());
    if (!p) throw std::runtime_error("rednose_amd: lib" + name_ + ".so does not export " + name_ + "_" + suffix);
    return reinterpret_cast<F>(p);
  }
  static void hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("rednose_amd: ") + what + ": " + hipGetErrorString(e));
  }
  void check(int rc, const char* what) const {
    if (rc == 0) return;
    const char* msg = sym<const char* (*)()>("last_error_string")();
    const std::string text = std::string("rednose_amd: ") + what + " -> " + std::to_string(rc) + ": " + msg;
    sym<void (*)()>("clear_error")();
    throw std::runtime_error(text);
  }
  double advance(double t) {
    if (std::isnan(filter_time_)) filter_time_ = t;       // first call adopts t, dt = 0 (ekf_sym.cc:198-200)
    const double dt = t - filter_time_;
    if (dt < 0.0) throw std::runtime_error("rednose_amd: dt < 0 in predict");
    return dt;
  }

  std::string name_;
  void* handle_ = nullptr;
  int D_ = 0, E_ = 0;
  int64_t n_;
  int norm_quats_;
  hipStream_t stream_;
  std::map<int, int> zdim_;
  predict_fn batch_predict_ = nullptr;
  double *x_ = nullptr, *P_ = nullptr, *Q_ = nullptr, *R_ = nullptr;
  double filter_time_ = NAN;
};

}  // namespace rednose_amd
