// Compiled Python binding of the C++ orchestrator rednose_amd::EKFSymBatch (include/rednose_amd/ekf_sym_batch.hpp) -- the analogue of the
// reference's Cython class EKF_sym_pyx over its C++ EKFSym (/root/reference/rednose/helpers/ekf_sym_pyx.pyx:85-195): same method names
// and argument meaning (init_state, state, covs, set_filter_time, get_filter_time, set_global, reset_rewind, predict,
// predict_and_update_batch returning the Estimate 9-tuple or None), for a BATCH of N filters resident on the GPU.
//   * observations: a list of n entries (the reference's z, one per observation of the call, ekf_sym_pyx.pyx:146-150); an entry is a host
//     array (Z,) -- the same observation for every filter -- or (N, Z), or a DEVICE pointer given as an int / an object with data_ptr()
//     (a torch tensor) to (N, Z) contiguous doubles, which the kernels overwrite with the residuals;
//   * R: a list of n (Z, Z) host arrays (:152-156); extra_args: a list of n lists (:158-165), the same for every filter;
//   * the Estimate holds the batch: xk1 / xk (N, D), Pk1 / Pk (N, E, E), y a list of n (N, Z) arrays -- N = 1 gives the reference's shapes
//     after a squeeze, which rednose_amd/helpers/ekf_sym_pyx.py applies;  estimate=False skips the split launches and the host copies.
// Built in-tree by __graft_entry__.build() (hipcc, host code only: the kernels live in the generated filter libraries this class dlopens).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <optional>

#include "rednose_amd/ekf_sym_batch.hpp"

namespace py = pybind11;
using rednose_amd::EKFSymBatch;
using arr = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

std::vector<double> flat(const arr& a) { return std::vector<double>(a.data(), a.data() + a.size()); }

struct DeviceBuf {
  double* p = nullptr;
  size_t cap = 0;
  void reserve(size_t doubles) {
    if (cap >= doubles) return;
    (void)hipFree(p);
    if (hipMalloc((void**)&p, sizeof(double) * doubles) != hipSuccess) throw std::runtime_error("rednose_amd: hipMalloc of an observation buffer failed");
    cap = doubles;
  }
  ~DeviceBuf() { (void)hipFree(p); }
};

class PyBatch {
 public:
  PyBatch(const std::string& gen_dir, const std::string& name, const arr& Q, const arr& x_initial, const arr& P_initial, int64_t batch,
          bool normalize_quaternions, int rewind_to_keep, double max_rewind_age)
      : kf_(gen_dir, name, flat(Q), flat(x_initial), flat(P_initial), batch, normalize_quaternions, nullptr, rewind_to_keep, max_rewind_age) {}

  void init_state(const arr& state, const arr& covs, py::object filter_time) {
    const double t = filter_time.is_none() ? NAN : filter_time.cast<double>();
    const int64_t n = kf_.batch(), D = kf_.dim_x(), E = kf_.dim_err();
    if (state.size() == D && covs.size() == E * E) {
      kf_.init_state(flat(state), flat(covs), t);
    } else if (state.size() == n * D && covs.size() == n * E * E) {
      kf_.init_state_batch(state.data(), covs.data(), t);
    } else {
      throw std::runtime_error("rednose_amd: init_state takes (D,), (E, E) or (N, D), (N, E, E)");
    }
  }
  arr state() {
    kf_.synchronize();
    const std::vector<double> v = kf_.state();
    arr out({(py::ssize_t)kf_.batch(), (py::ssize_t)kf_.dim_x()});
    std::copy(v.begin(), v.end(), out.mutable_data());
    return out;
  }
  arr covs() {
    kf_.synchronize();
    const std::vector<double> v = kf_.covs();
    arr out({(py::ssize_t)kf_.batch(), (py::ssize_t)kf_.dim_err(), (py::ssize_t)kf_.dim_err()});
    std::copy(v.begin(), v.end(), out.mutable_data());
    return out;
  }
  void set_filter_time(double t) { kf_.set_filter_time(t); }
  py::object get_filter_time() const { return std::isnan(kf_.get_filter_time()) ? py::object(py::none()) : py::object(py::float_(kf_.get_filter_time())); }
  void set_global(const std::string& var, double val) { kf_.set_global(var, val); }
  void reset_rewind() { kf_.reset_rewind(); }
  void predict(double t) { kf_.predict(t); }
  uintptr_t state_ptr() { return (uintptr_t)kf_.state_device(); }
  uintptr_t covs_ptr() { return (uintptr_t)kf_.covs_device(); }
  int64_t batch() const { return kf_.batch(); }
  int dim_x() const { return kf_.dim_x(); }
  int dim_err() const { return kf_.dim_err(); }

  py::object predict_and_update_batch(double t, int kind, py::list z, py::list R, py::object extra_args, bool augment, bool estimate) {
    const int Z = kf_.zdim(kind);
    const int64_t n = kf_.batch();
    const size_t nobs = py::len(z);
    if (py::len(R) != nobs) throw std::runtime_error("rednose_amd: one R per observation (ekf_sym.cc:159)");
    std::vector<arr> Rk;
    std::vector<const double*> Rp;
    for (py::handle r : R) {
      Rk.push_back(arr::ensure(r));
      if (!Rk.back() || Rk.back().size() != Z * Z) throw std::runtime_error("rednose_amd: R must be (Z, Z) per observation");
    }
    for (const arr& r : Rk) Rp.push_back(r.data());
    if (zbuf_.size() < nobs) zbuf_.resize(nobs);
    if (eabuf_.size() < nobs) eabuf_.resize(nobs);
    std::vector<double*> zp;
    std::vector<bool> own;
    for (size_t i = 0; i < nobs; i++) {
      py::handle zi = z[i];
      if (py::isinstance<py::int_>(zi)) {
        zp.push_back(reinterpret_cast<double*>(zi.cast<uintptr_t>()));
        own.push_back(false);
      } else if (py::hasattr(zi, "data_ptr")) {
        zp.push_back(reinterpret_cast<double*>(zi.attr("data_ptr")().cast<uintptr_t>()));
        own.push_back(false);
      } else {
        arr a = arr::ensure(zi);
        if (!a || (a.size() != Z && a.size() != n * Z)) throw std::runtime_error("rednose_amd: an observation must be (Z,) or (N, Z)");
        std::vector<double> host((size_t)n * Z);
        for (int64_t f = 0; f < n; f++) std::copy(a.data() + (a.size() == Z ? 0 : f * Z), a.data() + (a.size() == Z ? 0 : f * Z) + Z, host.begin() + f * Z);
        zbuf_[i].reserve(host.size() + 2);
        if (hipMemcpy(zbuf_[i].p, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("rednose_amd: observation upload failed");
        zp.push_back(zbuf_[i].p);
        own.push_back(true);
      }
    }
    // extra_args: [[]] / None (kinds without), or one list of kind_eadim values per observation -- broadcast to every filter
    std::vector<const double*> eap;
    if (!extra_args.is_none()) {
      py::list ea = extra_args.cast<py::list>();
      bool any = false;
      for (py::handle e : ea) any = any || py::len(e) > 0;
      if (any) {
        if (py::len(ea) != nobs) throw std::runtime_error("rednose_amd: one extra_args entry per observation (ekf_sym.cc:160)");
        for (size_t i = 0; i < nobs; i++) {
          arr e = arr::ensure(ea[i]);
          const size_t w = (size_t)e.size();
          std::vector<double> host((size_t)n * w);
          for (int64_t f = 0; f < n; f++) std::copy(e.data(), e.data() + w, host.begin() + f * w);
          eabuf_[i].reserve(host.size() + 2);
          if (hipMemcpy(eabuf_[i].p, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("rednose_amd: extra_args upload failed");
          eap.push_back(eabuf_[i].p);
        }
      }
    }
    EKFSymBatch::Estimate est;
    const bool applied = kf_.predict_and_update_batch(t, kind, zp, Rp, nullptr, eap, augment, estimate ? &est : nullptr);
    if (!applied) return py::none();              // too old: the reference returns None (ekf_sym_pyx.pyx:168-169)
    if (!estimate) return py::bool_(true);
    const py::ssize_t N = (py::ssize_t)n, D = kf_.dim_x(), E = kf_.dim_err();
    auto mk = [](const std::vector<double>& v, std::vector<py::ssize_t> shape) { arr a(shape); std::copy(v.begin(), v.end(), a.mutable_data()); return a; };
    py::list ys;
    kf_.synchronize();
    for (size_t i = 0; i < nobs; i++) {
      arr y({N, (py::ssize_t)Z});
      if (hipMemcpy(y.mutable_data(), zp[i], sizeof(double) * n * Z, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("rednose_amd: residual download failed");
      ys.append(y);
    }
    return py::make_tuple(mk(est.xk1, {N, D}), mk(est.xk, {N, D}), mk(est.Pk1, {N, E, E}), mk(est.Pk, {N, E, E}), est.t, est.kind, ys, z, extra_args);
  }

 private:
  EKFSymBatch kf_;
  std::vector<DeviceBuf> zbuf_, eabuf_;
};

}  // namespace

PYBIND11_MODULE(_ekf_sym_batch, m) {
  m.doc() = "rednose_amd::EKFSymBatch (include/rednose_amd/ekf_sym_batch.hpp) for Python: the compiled counterpart of the reference's EKF_sym_pyx";
  py::class_<PyBatch>(m, "EKFSymBatch")
      .def(py::init<const std::string&, const std::string&, const arr&, const arr&, const arr&, int64_t, bool, int, double>(), py::arg("gen_dir"), py::arg("name"),
           py::arg("Q"), py::arg("x_initial"), py::arg("P_initial"), py::arg("batch") = 1, py::arg("normalize_quaternions") = false,
           py::arg("rewind_to_keep") = 512, py::arg("max_rewind_age") = 1.0)
      .def("init_state", &PyBatch::init_state, py::arg("state"), py::arg("covs"), py::arg("filter_time"))
      .def("state", &PyBatch::state)
      .def("covs", &PyBatch::covs)
      .def("set_filter_time", &PyBatch::set_filter_time)
      .def("get_filter_time", &PyBatch::get_filter_time)
      .def("set_global", &PyBatch::set_global)
      .def("reset_rewind", &PyBatch::reset_rewind)
      .def("predict", &PyBatch::predict)
      .def("predict_and_update_batch", &PyBatch::predict_and_update_batch, py::arg("t"), py::arg("kind"), py::arg("z"), py::arg("R"),
           py::arg("extra_args") = py::none(), py::arg("augment") = false, py::arg("estimate") = true)
      .def("state_ptr", &PyBatch::state_ptr)
      .def("covs_ptr", &PyBatch::covs_ptr)
      .def_property_readonly("batch", &PyBatch::batch)
      .def_property_readonly("dim_x", &PyBatch::dim_x)
      .def_property_readonly("dim_err", &PyBatch::dim_err);
}
