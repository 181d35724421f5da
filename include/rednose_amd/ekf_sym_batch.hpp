// ekf_sym_batch.hpp -- C++ host-side orchestrator for N filters resident on one MI355X (header-only).
//
// The reference's C++ orchestrator is EKFS::EKFSym (/root/reference/rednose/helpers/ekf_sym.h:44-66,
// ekf_sym.cc:7-223): one filter, Eigen members, plugin lookup through ekf_load.cc.  This class is its batched
// counterpart above the C ABI of a generated rednose_amd library (include/rednose_amd_filter.h):
//   * library discovery as in ekf_load.cc:22-39 -- dlopen("<dir>/lib<name>.so"), symbols looked up by name;
//   * same vocabulary and semantics: init_state (:45-51), state/covs (:53-59), get/set_filter_time (:61-67),
//     predict (:196-209: first call adopts t, dt >= 0 required), predict_and_update_batch (:83-117) including the
//     late-observation path: with rewind_to_keep > 0 (the reference keeps REWIND_TO_KEEP = 512, ekf_sym.h:18) a ring of
//     checkpoints (filter time, x, P, the observation) lives in HBM; an observation older than the filter time rewinds
//     the whole batch to the last checkpoint at or before it (:119-140), is applied, and the overtaken observations are
//     replayed (:111-116); older than max_rewind_age or than the ring -> false (:87-94).  Quaternion renormalisation
//     after predict and update when quaternion_idxs were given to gen_code (:207,213 -- done inside the kernels);
//     set_global (:79-81) and get_extra_routine (:221-223);
//   * per-filter timelines (predict_and_update_batch_per_filter): N independent instances of that orchestrator in one batch --
//     every filter keeps its own filter time and its own ring of checkpoints in HBM; a call advances the filters named by a mask
//     to their own times through the library's `_masked` entry points, a late observation rewinds, applies and fast-forwards only
//     the filters it is late for, one that is too old is ignored for that filter alone;
//   * no Eigen: state lives in HBM as x (N, D), P (N, E, E) row-major fp64; z is a DEVICE pointer (N, Z) that the
//     kernel overwrites with the residual y; R is a host Z x Z matrix shared by the batch; a call carries one observation per
//     filter or, like the reference's (:83-85,172-180), n of them: vectors of z / R / extra_args pointers, one predict, n updates,
//     one checkpoint.
// Errors throw std::runtime_error carrying {name}_last_error_string(); nothing aborts.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rednose_amd {

class EKFSymBatch {
 public:
  EKFSymBatch(const std::string& directory, const std::string& name, const std::vector<double>& Q,
              const std::vector<double>& x_initial, const std::vector<double>& P_initial, int64_t batch,
              bool normalize_quaternions = false, hipStream_t stream = nullptr, int rewind_to_keep = 0, double max_rewind_age = 1.0)
      : name_(name), n_(batch), norm_quats_(normalize_quaternions ? 1 : 0), stream_(stream), rewind_to_keep_(rewind_to_keep),
        max_rewind_age_(max_rewind_age) {
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
    for (auto& c : ring_) release(c);
    for (auto& c : spare_) release(c);
    for (double* p : {x_, P_, Q_, R_, pf_.ring_x, pf_.ring_P, pf_.ring_z, pf_.ring_ea, pf_.dt, pf_.stage_z[0], pf_.stage_z[1], pf_.stage_ea[0],
                      pf_.stage_ea[1], pf_.Rn, pf_.zpack, pf_.eapack})
      (void)hipFree(p);
    (void)hipFree(pf_.act);
    for (Bounce& b : bounce_) { if (b.p) (void)hipHostFree(b.p); if (b.done) (void)hipEventDestroy(b.done); }
    (void)hipFree(pf_.slot);
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
    reset_rewind();
  }

  // forget every checkpoint (EKFSym::reset_rewind, ekf_sym.cc:69-73 there)
  void reset_rewind() {
    for (auto& c : ring_) spare_.push_back(c);
    ring_.clear();
  }

  // run-time model scalar of gen_code's global_vars (EKFSym::set_global, ekf_sym.cc:79-81): per LIBRARY, like the reference
  void set_global(const std::string& var, double value) {
    sym<void (*)(double)>("set_" + var)(value);
    if (sym<int (*)()>("last_error")() != 0) check(1, ("set_" + var).c_str());
  }

  // extra routine by name (EKFSym::get_extra_routine, ekf_sym.cc:221-223): the host-pointer entry point {name}_{routine}
  typedef void (*extra_routine_t)(double*, double*);
  extra_routine_t get_extra_routine(const std::string& routine) const { return sym<extra_routine_t>(routine); }

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
    return predict_and_update_batch(t, kind, std::vector<double*>{z_dev}, std::vector<const double*>{R_host}, flags_dev,
                                    ea_dev ? std::vector<const double*>{ea_dev} : std::vector<const double*>{}, augment);
  }

  // n observations of one kind in ONE call, in the shape the reference takes them (ekf_sym.cc:83-85: a vector of z, a vector of R, a
  // vector of extra_args): ONE predict to t, the n observations applied in order (ekf_sym.cc:172-180), ONE checkpoint (:191) --
  // a late call rewinds over calls, not over observations.  z_devs[i]: (N, Z) device, in: z_i, out: y_i (Estimate.y, :183);
  // R_hosts[i]: Z x Z row-major host, shared by the batch; ea_devs: empty, or one (N, kind_eadim) device pointer per observation;
  // flags_dev: n x N bytes (observation i's flags at flags_dev + i * N) or nullptr.  Launches: one fused predict + update, then
  // n - 1 batch_update_k -- the step-granular kernels, the reference's arithmetic on any P (BatchedEKF serves the same call with one
  // batch_run launch where the library allows it).
  // The reference's Estimate (ekf_sym.h:32-42) for the whole batch, as host copies: xk1 / Pk1 the predicted pair (after predict(t), before
  // the first update), xk / Pk the filtered pair right after THIS call's observations (before any fast-forward, ekf_sym.cc:106-116); the
  // residuals y stay where the observations were (z_devs).  Filled when a pointer to one is passed to predict_and_update_batch: the call
  // then predicts and updates in separate launches (the predicted pair has to exist in memory) and copies 2 N (D + E^2) doubles to the host.
  struct Estimate {
    std::vector<double> xk1, xk, Pk1, Pk;
    double t = NAN;
    int kind = 0;
  };

  bool predict_and_update_batch(double t, int kind, const std::vector<double*>& z_devs, const std::vector<const double*>& R_hosts,
                                uint8_t* flags_dev = nullptr, const std::vector<const double*>& ea_devs = {}, bool augment = false,
                                Estimate* estimate = nullptr) {
    const int Z = zdim_.at(kind);
    if (z_devs.empty() || z_devs.size() != R_hosts.size() || (!ea_devs.empty() && ea_devs.size() != z_devs.size()))
      throw std::runtime_error("rednose_amd: predict_and_update_batch needs n >= 1 observations with one R (and, if any, one extra_args) each");   // ekf_sym.cc:159-160
    if (z_devs.size() * (size_t)Z * Z > 64 * 64)
      throw std::runtime_error("rednose_amd: too many observations in one call for the noise staging buffer");
    std::vector<Checkpoint> replay;
    if (!std::isnan(filter_time_) && t < filter_time_) {
      // late observation (ekf_sym.cc:87-94): too old for the ring or for max_rewind_age -> ignored
      if (ring_.empty() || t < ring_.front().t || t < ring_.back().t - max_rewind_age_) return false;
      if (augment) throw std::runtime_error("rednose_amd: augment with a rewind is not supported (the reference asserts the same)");
      replay = rewind(t);
    }
    step(t, kind, Z, z_devs, R_hosts, flags_dev, ea_devs, estimate);
    if (estimate) {
      estimate->t = t;
      estimate->kind = kind;
      synchronize();
      estimate->xk = state();
      estimate->Pk = covs();
    }
    if (augment) this->augment();
    for (auto& c : replay) {             // fast-forward through the calls that were overtaken (ekf_sym.cc:111-116), each with all its observations
      const int Zc = zdim_.at(c.kind);
      std::vector<double*> zs;
      std::vector<const double*> Rs, eas;
      for (int i = 0; i < c.nobs; i++) {
        zs.push_back(c.z + (size_t)i * obs_stride(Zc));
        Rs.push_back(c.R.data() + (size_t)i * Zc * Zc);
        if (c.has_ea) eas.push_back(c.ea + (size_t)i * obs_stride(c.ead));
      }
      step(c.obs_t, c.kind, Zc, zs, Rs, nullptr, eas);
      spare_.push_back(c);
    }
    return true;
  }

  // ---- per-filter timelines --------------------------------------------------------------------------------------------
  // One observation kind, every filter on its own clock: t_host (N) observation times, active_host (N bytes; nullptr = every
  // filter has an observation).  z_dev (N, Z) in: z, out: y for the active filters, untouched for the others; flags_dev (N
  // bytes or nullptr): kernel flags, 16 for filters that were not active.  A filter whose observation is older than its own
  // filter time is rewound to its last checkpoint at or before it, gets the observation, and replays what it had overtaken
  // (needs rewind_to_keep > 0: otherwise std::runtime_error, like dt < 0 in the reference); an observation older than the
  // filter's ring or than max_rewind_age is ignored for that filter alone (ekf_sym.cc:87-94).  Returns the number of ignored
  // observations; ignored() tells which.  Do not mix with the shared-timeline calls on one object.
  int64_t predict_and_update_batch_per_filter(const double* t_host, const uint8_t* active_host, int kind, double* z_dev,
                                              const double* R_host, uint8_t* flags_dev = nullptr, const double* ea_dev = nullptr) {
    return predict_and_update_batch_per_filter(t_host, active_host, kind, std::vector<double*>{z_dev}, std::vector<const double*>{R_host}, flags_dev,
                                               ea_dev ? std::vector<const double*>{ea_dev} : std::vector<const double*>{});
  }

  // The same with n observations per active filter in ONE call, in the reference's argument shape (vectors of z / R / extra_args, ekf_sym.cc:83-85):
  // every active filter is predicted to its own time once, gets the n observations in order and writes ONE checkpoint into its ring; a late call
  // rewinds its filter over whole calls and the replay applies every overtaken call with all its observations.  z_devs[j]: (N, Z) device, in z_j /
  // out y_j; R_hosts[j]: Z x Z host, shared by the batch; flags_dev: n x N bytes (observation j at flags_dev + j * N; a filter that was not active
  // has 16 in every row) or nullptr.  n may not exceed set_max_observations_per_call() (default 1: the rings are sized by it).
  int64_t predict_and_update_batch_per_filter(const double* t_host, const uint8_t* active_host, int kind, const std::vector<double*>& z_devs,
                                              const std::vector<const double*>& R_hosts, uint8_t* flags_dev = nullptr,
                                              const std::vector<const double*>& ea_devs = {}) {
    const int Z = zdim_.at(kind);
    const int nobs = (int)z_devs.size();
    if (nobs < 1 || R_hosts.size() != z_devs.size() || (!ea_devs.empty() && ea_devs.size() != z_devs.size()))
      throw std::runtime_error("rednose_amd: per-filter call needs n >= 1 observations with one R (and, if any, one extra_args) each");
    if (nobs > pf_nmax_)
      throw std::runtime_error("rednose_amd: " + std::to_string(nobs) + " observations per call: call set_max_observations_per_call() before the first per-filter step");
    if ((size_t)nobs * Z * Z > 64 * 64) throw std::runtime_error("rednose_amd: too many observations in one call for the noise staging buffer");
    pf_init();
    PerFilter& s = pf_;
    const int K = rewind_to_keep_;
    // everything that can be refused is refused HERE, before the rings, the filter times or x / P are touched: a call that throws
    // leaves the object as it was.  (Kinds with different extra-argument counts are refused once, in pf_init; a pending observation
    // was accepted by these same checks when it first arrived.)
    if (s.ead_of.at(kind) > 0 && ea_devs.empty())
      throw std::runtime_error("rednose_amd: kind " + std::to_string(kind) + " takes extra arguments: ea_dev is null");
    for (int64_t i = 0; i < n_; i++) {
      const bool on = active_host ? active_host[i] != 0 : true;
      if (on && K <= 0 && !std::isnan(s.ft[i]) && t_host[i] < s.ft[i])
        throw std::runtime_error("rednose_amd: dt < 0 for a filter and no rewind ring (rewind_to_keep = 0)");
    }
    rtable_gc();
    std::vector<uint8_t> act(n_, 1), restore(n_, 0);
    if (active_host) act.assign(active_host, active_host + n_);
    std::fill(s.ignored.begin(), s.ignored.end(), 0);
    std::vector<int32_t> slot(n_, 0);
    std::vector<std::vector<Pending>> rep;           // rep[q]: overtaken call number q of every rewound filter
    int64_t n_ignored = 0;
    // ---- late observations: EKFSym::rewind (ekf_sym.cc:119-140) on every late filter's own ring ----
    for (int64_t i = 0; i < n_; i++) {
      if (!act[i] || std::isnan(s.ft[i]) || !(t_host[i] < s.ft[i])) continue;
      if (K <= 0) throw std::runtime_error("rednose_amd: dt < 0 for a filter and no rewind ring (rewind_to_keep = 0)");
      const int32_t L = s.len[i], H = s.head[i];
      auto at = [&](int32_t j) { return (size_t)((H + j) % K) * n_ + i; };
      if (L == 0 || t_host[i] < s.rt[at(0)] || t_host[i] < s.rt[at(L - 1)] - max_rewind_age_) {
        s.ignored[i] = 1; act[i] = 0; n_ignored++;
        continue;
      }
      int32_t ix = 0;
      while (ix < L && s.rt[at(ix)] <= t_host[i]) ix++;                // bisect_right
      slot[i] = (H + ix - 1) % K;
      restore[i] = 1;
      s.ft[i] = s.rt[at(ix - 1)];
      for (int32_t j = ix; j < L; j++) {
        if (rep.size() < (size_t)(j - ix + 1)) rep.emplace_back();
        Pending p{i, s.rt[at(j)], s.rkind[at(j)], s.rnobs[at(j)], {}, (int32_t)((H + j) % K)};
        p.ridx.assign(s.rridx.begin() + at(j) * pf_nmax_, s.rridx.begin() + at(j) * pf_nmax_ + p.nobs);
        rep[j - ix].push_back(std::move(p));
      }
      s.len[i] = ix;
    }
    if (!rep.empty() || std::count(restore.begin(), restore.end(), 1) > 0) {
      upload(s.slot, slot.data(), sizeof(int32_t) * n_);
      upload(s.act, restore.data(), n_);
      ring_copy(s.ring_x, D_, x_, D_, D_, false);
      ring_copy(s.ring_P, (int64_t)E_ * E_, P_, (int64_t)E_ * E_, (int64_t)E_ * E_, false);
    }
    // every checkpoint written below lands on the ring slot of the NEXT overtaken call: that one is staged out first
    int cur = 0;
    if (!rep.empty()) stage(rep[0], cur);
    std::vector<double> tt(t_host, t_host + n_);
    std::vector<std::vector<int>> ridx(nobs);
    std::vector<const double*> R_devs;
    for (int j = 0; j < nobs; j++) {
      ridx[j].assign(n_, rtable_index(R_hosts[j], Z));
      upload_R((size_t)j * Z * Z, R_hosts[j], (size_t)Z * Z);
      R_devs.push_back(R_ + (size_t)j * Z * Z);
    }
    masked_step(tt, act, kind, Z, z_devs, R_devs, 0, flags_dev, ea_devs, ridx, nullptr);
    // ---- fast-forward (ekf_sym.cc:111-116): position q of every rewound filter, one group of launches per kind present there ----
    for (size_t q = 0; q < rep.size(); q++) {
      if (q + 1 < rep.size()) stage(rep[q + 1], cur ^ 1);
      std::map<int, std::vector<const Pending*>> by_kind;
      for (const Pending& p : rep[q]) by_kind[p.kind].push_back(&p);
      for (auto& kv : by_kind) {
        const int k = kv.first, Zk = zdim_.at(k), ek = s.ead_of.at(k);
        std::vector<uint8_t> ra(n_, 0);
        std::vector<double> tq(s.ft);
        std::vector<int32_t> nq(n_, 0);
        int nmaxq = 0;
        for (const Pending* p : kv.second) { ra[p->f] = 1; tq[p->f] = p->t; nq[p->f] = p->nobs; nmaxq = std::max(nmaxq, p->nobs); }
        std::vector<std::vector<int>> rq(nmaxq, std::vector<int>(n_, 0));
        std::vector<double*> zs;
        std::vector<const double*> Rs, eas;
        for (int j = 0; j < nmaxq; j++) {
          std::vector<double> Rn((size_t)n_ * Zk * Zk, 0.0);
          for (int64_t f = 0; f < n_; f++)                                  // (filters masked out of launch j: any regular matrix)
            for (int d = 0; d < Zk; d++) Rn[(size_t)f * Zk * Zk + d * Zk + d] = 1.0;
          for (const Pending* p : kv.second) {
            if (p->nobs <= j) continue;
            rq[j][p->f] = p->ridx[j];
            std::copy(s.rtable[p->ridx[j]].begin(), s.rtable[p->ridx[j]].end(), Rn.begin() + (size_t)p->f * Zk * Zk);
          }
          // staged observations are (N, nmax, zmax) rows, the entry point takes (N, Zk) contiguous
          double* zj = s.zpack + (size_t)j * obs_stride(s.zmax);
          hip(hipMemcpy2DAsync(zj, sizeof(double) * Zk, s.stage_z[cur] + (size_t)j * s.zmax, sizeof(double) * pf_nmax_ * s.zmax, sizeof(double) * Zk, n_,
                               hipMemcpyDeviceToDevice, stream_), "pack z");
          double* Rj = s.Rn + (size_t)j * n_ * s.zmax * s.zmax;
          upload(Rj, Rn.data(), sizeof(double) * Rn.size());
          zs.push_back(zj);
          Rs.push_back(Rj);
          if (ek) {
            double* ej = s.eapack + (size_t)j * obs_stride(s.ead);
            hip(hipMemcpy2DAsync(ej, sizeof(double) * ek, s.stage_ea[cur] + (size_t)j * s.ead, sizeof(double) * pf_nmax_ * s.ead, sizeof(double) * ek, n_,
                                 hipMemcpyDeviceToDevice, stream_), "pack ea");
            eas.push_back(ej);
          }
        }
        masked_step(tq, ra, k, Zk, zs, Rs, 1, nullptr, eas, rq, &nq);
      }
      cur ^= 1;
    }
    // flag bit 5 (include/rednose_amd_filter.h): "observation too old for this filter's ring, ignored" -- on top of bit 4, which
    // the kernel set for every filter that was not active in the launch
    if (flags_dev != nullptr && n_ignored > 0) {
      // only the mask of the ignored filters goes up (N bytes; `upload` waits for that one copy because its source is pageable host memory), and a
      // small kernel sets their bytes on the stream: no download of the flags, the other filters' bytes are not rewritten
      upload(s.act, s.ignored.data(), (size_t)n_);
      for (int j = 0; j < nobs; j++)
        check(sym<int (*)(uint8_t*, const uint8_t*, int, int64_t, void*)>("batch_flags_set")(flags_dev + (size_t)j * n_, s.act, 16 | 32, n_, stream_), "batch_flags_set");
    }
    return n_ignored;
  }
  // capacity of a per-filter ring entry in observations (the rings, K x N x that many observation slots, are allocated by the first per-filter call)
  void set_max_observations_per_call(int nmax) {
    if (pf_.ready) throw std::runtime_error("rednose_amd: set_max_observations_per_call() after the per-filter rings were allocated");
    if (nmax < 1) throw std::runtime_error("rednose_amd: set_max_observations_per_call(n >= 1)");
    pf_nmax_ = nmax;
  }
  const std::vector<double>& filter_times() const { return pf_.ft; }          // NaN: the filter has not stepped yet
  const std::vector<uint8_t>& ignored() const { return pf_.ignored; }         // 1: the last per-filter call ignored this filter's observation
  size_t noise_table_size() const { return pf_.rtable.size(); }              // distinct R matrices kept for the per-filter rings (bounded: rtable_gc)

  // MSCKF window shift on every filter (libraries generated with msckf_params only)
  void augment() {
    check(sym<int (*)(double*, double*, int64_t, void*)>("batch_augment")(x_, P_, n_, stream_), "batch_augment");
  }

  // Mahalanobis distance of an observation per filter into d2_dev (N), state untouched (EKF_sym.maha_test, ekf_sym.py:626-649)
  void maha_distance(int kind, const double* z_dev, const double* R_host, double* d2_dev, const double* ea_dev = nullptr) {
    const int Z = zdim_.at(kind);
    upload_R(0, R_host, (size_t)Z * Z);
    auto fn = sym<int (*)(const double*, const double*, const double*, const double*, int, const double*, int64_t, double*, void*)>(
        "batch_maha_" + std::to_string(kind));
    check(fn(x_, P_, z_dev, R_, 0, ea_dev, n_, d2_dev, stream_), "batch_maha");
  }

  void synchronize() const { hip(hipStreamSynchronize(stream_), "synchronize"); }

 private:
  // one entry of the rewind ring: the batch state AFTER an observation was applied, and that observation
  struct Checkpoint {
    double t = NAN;                    // filter time after the step
    double *x = nullptr, *P = nullptr; // device copies of the batch state
    double obs_t = NAN;
    int kind = 0;
    int nobs = 1;                      // observations of the call this checkpoint closes (ekf_sym.cc:191: one checkpoint per call)
    double* z = nullptr;               // device copies of the observations, obs_stride(Z) doubles apart (the kernels overwrite the caller's with the residuals)
    double* ea = nullptr;              // device copies of the extra arguments, obs_stride(ead) apart
    size_t zcap = 0, eacap = 0;        // doubles allocated behind z / ea
    bool has_ea = false;               // ... which these observations had
    int ead = 0;
    std::vector<double> R;             // nobs x Z x Z, host
  };
  // distance between two observations' (N, w) blocks of a checkpoint: even, so that every block starts 16-byte aligned
  size_t obs_stride(int w) const { return ((size_t)n_ * w + 1) & ~(size_t)1; }

  void release(Checkpoint& c) {
    (void)hipFree(c.x); (void)hipFree(c.P); (void)hipFree(c.z); (void)hipFree(c.ea);
    c = Checkpoint();
  }

  Checkpoint slot() {
    if (!spare_.empty()) { Checkpoint c = spare_.back(); spare_.pop_back(); return c; }
    Checkpoint c;
    hip(hipMalloc((void**)&c.x, sizeof(double) * n_ * D_), "hipMalloc ring x");
    hip(hipMalloc((void**)&c.P, sizeof(double) * n_ * E_ * E_), "hipMalloc ring P");
    return c;
  }
  static void reserve(double** p, size_t* cap, size_t doubles) {
    if (*cap >= doubles) return;
    (void)hipFree(*p);
    *p = nullptr;
    hip(hipMalloc((void**)p, sizeof(double) * doubles), "hipMalloc ring observations");
    *cap = doubles;
  }

  // predict + n updates + ONE checkpoint (EKFSym::predict_and_update_batch's inner part, ekf_sym.cc:158-194)
  void step(double t, int kind, int Z, const std::vector<double*>& z_devs, const std::vector<const double*>& R_hosts, uint8_t* flags_dev,
            const std::vector<const double*>& ea_devs, Estimate* estimate = nullptr) {
    Checkpoint c;
    const bool keep = rewind_to_keep_ > 0;
    const int nobs = (int)z_devs.size();
    const int ead = sym<int (*)(int)>("kind_eadim")(kind);
    // one observation without extra arguments, no Estimate asked for: the fused launch writes the checkpoint itself (k_stepc_{kind} behind
    // {name}_batch_predict_update_{kind}_ckpt: the observation as it came, the filtered pair) -- no copies in front of or behind it
    ckpt_fn fused_ckpt = nullptr;
    if (keep && nobs == 1 && !estimate && (ea_devs.empty() || ead == 0)) fused_ckpt = sym_optional<ckpt_fn>("batch_predict_update_" + std::to_string(kind) + "_ckpt");
    if (keep) {
      c = slot();
      c.obs_t = t; c.kind = kind; c.nobs = nobs; c.ead = ead;
      c.R.clear();
      reserve(&c.z, &c.zcap, obs_stride(Z) * nobs);
      c.has_ea = !ea_devs.empty() && ead > 0;
      if (c.has_ea) reserve(&c.ea, &c.eacap, obs_stride(ead) * nobs);
      for (int i = 0; i < nobs; i++) {
        c.R.insert(c.R.end(), R_hosts[i], R_hosts[i] + Z * Z);
        if (!fused_ckpt) hip(hipMemcpyAsync(c.z + (size_t)i * obs_stride(Z), z_devs[i], sizeof(double) * n_ * Z, hipMemcpyDeviceToDevice, stream_), "ring z");
        if (c.has_ea) hip(hipMemcpyAsync(c.ea + (size_t)i * obs_stride(ead), ea_devs[i], sizeof(double) * n_ * ead, hipMemcpyDeviceToDevice, stream_), "ring ea");
      }
    }
    const double dt = advance(t);
    for (int i = 0; i < nobs; i++)
      upload_R((size_t)i * Z * Z, R_hosts[i], (size_t)Z * Z);
    int first_update = 1;
    if (estimate) {            // the predicted pair has to exist in memory: predict alone, then every observation as an update
      check(batch_predict_(x_, P_, Q_, nullptr, dt, n_, norm_quats_, stream_), "batch_predict");
      synchronize();
      estimate->xk1 = state();
      estimate->Pk1 = covs();
      first_update = 0;
    } else if (fused_ckpt) {
      check(fused_ckpt(x_, P_, Q_, nullptr, dt, z_devs[0], R_, 0, nullptr, n_, norm_quats_, flags_dev, c.x, c.P, c.z, stream_), "batch_predict_update_ckpt");
    } else {
      auto fused = sym<step_fn>("batch_predict_update_" + std::to_string(kind));
      check(fused(x_, P_, Q_, nullptr, dt, z_devs[0], R_, 0, ea_devs.empty() ? nullptr : ea_devs[0], n_, norm_quats_, flags_dev, stream_), "batch_predict_update");
    }
    if (nobs > first_update) {
      auto upd = sym<update_fn>("batch_update_" + std::to_string(kind));
      for (int i = first_update; i < nobs; i++)
        check(upd(x_, P_, z_devs[i], R_ + (size_t)i * Z * Z, 0, ea_devs.empty() ? nullptr : ea_devs[i], n_, norm_quats_,
                  flags_dev ? flags_dev + (size_t)i * n_ : nullptr, stream_), "batch_update");
    }
    filter_time_ = t;
    if (keep) {
      if (!fused_ckpt) {
        hip(hipMemcpyAsync(c.x, x_, sizeof(double) * n_ * D_, hipMemcpyDeviceToDevice, stream_), "ring x");
        hip(hipMemcpyAsync(c.P, P_, sizeof(double) * n_ * E_ * E_, hipMemcpyDeviceToDevice, stream_), "ring P");
      }
      c.t = t;
      ring_.push_back(c);
      while ((int)ring_.size() > rewind_to_keep_) { spare_.push_back(ring_.front()); ring_.pop_front(); }
    }
  }

  // EKFSym::rewind (ekf_sym.cc:119-140): back to the last checkpoint at or before t; returns the observations to replay
  std::vector<Checkpoint> rewind(double t) {
    size_t idx = 0;
    while (idx < ring_.size() && ring_[idx].t <= t) idx++;          // first checkpoint after t
    const Checkpoint& at = ring_[idx - 1];
    filter_time_ = at.t;
    hip(hipMemcpyAsync(x_, at.x, sizeof(double) * n_ * D_, hipMemcpyDeviceToDevice, stream_), "rewind x");
    hip(hipMemcpyAsync(P_, at.P, sizeof(double) * n_ * E_ * E_, hipMemcpyDeviceToDevice, stream_), "rewind P");
    std::vector<Checkpoint> replay(ring_.begin() + idx, ring_.end());
    ring_.erase(ring_.begin() + idx, ring_.end());
    return replay;
  }

  // ---- per-filter timelines ---------------------------------------------------------------------------------------------
  struct Pending { int64_t f; double t; int kind; int nobs; std::vector<int> ridx; int32_t slot; };   // an overtaken call (its observations' noise indices) still in filter f's ring
  struct PerFilter {
    bool ready = false;
    int zmax = 0, ead = 0;
    std::vector<double> ft;                  // filter time per filter (NaN: not started)
    std::vector<uint8_t> ignored;
    std::vector<int32_t> head, len;          // circular ring position per filter
    std::vector<double> rt;                  // (K, N) checkpoint times
    std::vector<int32_t> rkind, rnobs;       // (K, N) observation kind / observations of the call
    std::vector<int32_t> rridx;              // (K, N, nmax) index of every observation's noise matrix in rtable
    std::vector<std::vector<double>> rtable; // distinct noise matrices referenced by live ring slots (row-major Z x Z), see rtable_gc
    size_t rtable_gc_at = 64;
    std::map<int, int> ead_of;               // extra arguments per kind
    double *ring_x = nullptr, *ring_P = nullptr, *ring_z = nullptr, *ring_ea = nullptr;      // (K, N, rec) device
    double *dt = nullptr, *Rn = nullptr, *zpack = nullptr, *eapack = nullptr;
    double* stage_z[2] = {nullptr, nullptr};
    double* stage_ea[2] = {nullptr, nullptr};
    uint8_t* act = nullptr;
    int32_t* slot = nullptr;
  };
  using masked_fn = int (*)(double*, double*, const double*, const double*, double, double*, const double*, int, const double*,
                            int64_t, int, uint8_t*, const uint8_t*, void*);
  using masked_update_fn = int (*)(double*, double*, double*, const double*, int, const double*, int64_t, int, uint8_t*, const uint8_t*, void*);
  using ring_fn = int (*)(double*, int64_t, double*, int64_t, int64_t, const int32_t*, const uint8_t*, int64_t, int, void*);

  void pf_init() {
    PerFilter& s = pf_;
    if (s.ready) return;
    for (auto& kv : zdim_) {
      s.zmax = std::max(s.zmax, kv.second);
      s.ead_of[kv.first] = sym<int (*)(int)>("kind_eadim")(kv.first);
      s.ead = std::max(s.ead, s.ead_of[kv.first]);
    }
    for (auto& kv : s.ead_of)
      if (kv.second != 0 && kv.second != s.ead)
        throw std::runtime_error("rednose_amd: kinds with different extra-argument counts are not supported by the per-filter ring");
    s.ft.assign(n_, filter_time_);
    s.ignored.assign(n_, 0);
    s.head.assign(n_, 0);
    s.len.assign(n_, 0);
    const size_t K = (size_t)std::max(rewind_to_keep_, 0);
    s.rt.assign(K * n_, NAN);
    s.rkind.assign(K * n_, 0);
    s.rnobs.assign(K * n_, 1);
    s.rridx.assign(K * n_ * (size_t)pf_nmax_, 0);
    auto dmal = [&](double** p, size_t doubles) { hip(hipMalloc((void**)p, sizeof(double) * std::max<size_t>(doubles, 2)), "hipMalloc per-filter buffer"); };
    const size_t ea1 = (size_t)std::max(s.ead, 1);
    if (K > 0) {
      dmal(&s.ring_x, K * n_ * D_);
      dmal(&s.ring_P, K * n_ * E_ * E_);
      dmal(&s.ring_z, K * n_ * (size_t)pf_nmax_ * s.zmax);
      dmal(&s.ring_ea, K * n_ * (size_t)pf_nmax_ * ea1);
    }
    dmal(&s.dt, n_);
    dmal(&s.zpack, (size_t)pf_nmax_ * obs_stride(s.zmax));
    dmal(&s.eapack, (size_t)pf_nmax_ * obs_stride((int)ea1));
    dmal(&s.Rn, (size_t)pf_nmax_ * n_ * s.zmax * s.zmax);
    for (int b = 0; b < 2; b++) { dmal(&s.stage_z[b], (size_t)n_ * pf_nmax_ * s.zmax); dmal(&s.stage_ea[b], (size_t)n_ * pf_nmax_ * ea1); }
    hip(hipMalloc((void**)&s.act, n_ + 16), "hipMalloc mask");
    hip(hipMalloc((void**)&s.slot, sizeof(int32_t) * n_ + 16), "hipMalloc slots");
    ring_copy_ = sym<ring_fn>("batch_ring_copy");
    s.ready = true;
  }
  // Host array -> device, on the stream, without waiting for it: the bytes go through one of a few PINNED bounce buffers (a copy from pageable memory is staged by
  // the runtime anyway, synchronously, and the stream was synchronised on top of that so that the caller could reuse `src`: three such uploads were half of a
  // per-filter call's 220 us at 65 536 filters).  A bounce buffer is reused only after the copy that last read it has completed (its event).
  struct Bounce { void* p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; };
  void upload(void* dst, const void* src, size_t bytes) {
    Bounce& b = bounce_[bounce_next_++ % (sizeof(bounce_) / sizeof(bounce_[0]))];
    if (!b.done) hip(hipEventCreateWithFlags(&b.done, hipEventDisableTiming), "upload event");
    else hip(hipEventSynchronize(b.done), "upload wait");
    if (b.cap < bytes) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr; b.cap = 0;
      hip(hipHostMalloc(&b.p, bytes, hipHostMallocDefault), "hipHostMalloc bounce buffer");
      b.cap = bytes;
    }
    std::memcpy(b.p, src, bytes);
    hip(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyHostToDevice, stream_), "upload");
    hip(hipEventRecord(b.done, stream_), "upload record");
  }
  // one array of a checkpoint, every filter at its own ring position (slot vector and mask already on the device)
  void ring_copy(double* ring, int64_t ring_stride, double* flat, int64_t flat_stride, int64_t rec, bool to_ring) {
    check(ring_copy_(ring, ring_stride, flat, flat_stride, rec, pf_.slot, pf_.act, n_, to_ring ? 1 : 0, stream_), "batch_ring_copy");
  }
  // The table of distinct noise matrices would grow without bound under a time-varying R (one entry per call).  Entries that no
  // live ring slot references any more are dropped once the table has doubled since the last sweep: O(K N) then, amortised O(1).
  void rtable_gc() {
    PerFilter& s = pf_;
    if (s.rtable.size() < s.rtable_gc_at) return;
    const int K = rewind_to_keep_;
    std::vector<int> remap(s.rtable.size(), -1);
    std::vector<std::vector<double>> kept;
    for (int64_t i = 0; i < n_ && K > 0; i++) {
      for (int32_t j = 0; j < s.len[i]; j++) {
        const size_t at = (size_t)((s.head[i] + j) % K) * n_ + i;
        for (int32_t o = 0; o < s.rnobs[at]; o++) {
          int32_t& r = s.rridx[at * pf_nmax_ + o];
          if (remap[r] < 0) { remap[r] = (int)kept.size(); kept.push_back(s.rtable[r]); }
          r = remap[r];
        }
      }
    }
    s.rtable.swap(kept);
    s.rtable_gc_at = std::max<size_t>(64, 2 * s.rtable.size());
  }
  int rtable_index(const double* R, int Z) {
    for (size_t i = 0; i < pf_.rtable.size(); i++)
      if ((int)pf_.rtable[i].size() == Z * Z && std::equal(R, R + Z * Z, pf_.rtable[i].begin())) return (int)i;
    pf_.rtable.emplace_back(R, R + Z * Z);
    return (int)pf_.rtable.size() - 1;
  }
  // observations of one replay position -> flat staging buffers `b` (their ring slots are about to be overwritten)
  void stage(const std::vector<Pending>& list, int b) {
    PerFilter& s = pf_;
    std::vector<uint8_t> m(n_, 0);
    std::vector<int32_t> sl(n_, 0);
    for (const Pending& p : list) { m[p.f] = 1; sl[p.f] = p.slot; }
    upload(s.slot, sl.data(), sizeof(int32_t) * n_);
    upload(s.act, m.data(), n_);
    ring_copy(s.ring_z, (int64_t)pf_nmax_ * s.zmax, s.stage_z[b], (int64_t)pf_nmax_ * s.zmax, (int64_t)pf_nmax_ * s.zmax, false);
    if (s.ead > 0) ring_copy(s.ring_ea, (int64_t)pf_nmax_ * s.ead, s.stage_ea[b], (int64_t)pf_nmax_ * s.ead, (int64_t)pf_nmax_ * s.ead, false);
  }
  // masked predict + n updates of the filters in `act`, each to its own time, + ONE checkpoint each (EKFSym::predict_and_update_batch's
  // inner part, ekf_sym.cc:158-194, per filter).  z_devs / R_devs / ea_devs: one entry per observation of the call; nobs_of (replay): filter i
  // has only its first nobs_of[i] of them -- launch j is masked to the filters that have more than j.
  void masked_step(const std::vector<double>& t, const std::vector<uint8_t>& act, int kind, int Z, const std::vector<double*>& z_devs,
                   const std::vector<const double*>& R_devs, int r_per_filter, uint8_t* flags_dev, const std::vector<const double*>& ea_devs,
                   const std::vector<std::vector<int>>& ridx, const std::vector<int32_t>* nobs_of) {
    PerFilter& s = pf_;
    const int K = rewind_to_keep_;
    const int nobs = (int)z_devs.size();
    std::vector<double> dt(n_, 0.0);
    std::vector<int32_t> slot(n_, 0);
    for (int64_t i = 0; i < n_; i++) {
      if (!act[i]) continue;
      dt[i] = std::isnan(s.ft[i]) ? 0.0 : t[i] - s.ft[i];             // first call of a filter adopts t (ekf_sym.cc:198-200)
      if (dt[i] < 0.0) throw std::runtime_error("rednose_amd: dt < 0 in a per-filter step");
      s.ft[i] = t[i];
      if (K > 0) {
        if (s.len[i] == K) s.head[i] = (s.head[i] + 1) % K; else s.len[i]++;
        slot[i] = (s.head[i] + s.len[i] - 1) % K;
        const size_t at = (size_t)slot[i] * n_ + i;
        const int ni = nobs_of ? (*nobs_of)[i] : nobs;
        s.rt[at] = t[i]; s.rkind[at] = kind; s.rnobs[at] = ni;
        for (int j = 0; j < ni; j++) s.rridx[at * pf_nmax_ + j] = ridx[j][i];
      }
    }
    auto mask_of = [&](int j) {            // filters that take observation j of this call
      std::vector<uint8_t> m(act);
      if (nobs_of) for (int64_t i = 0; i < n_; i++) m[i] = (uint8_t)(act[i] && (*nobs_of)[i] > j);
      return m;
    };
    upload(s.dt, dt.data(), sizeof(double) * n_);
    const int ek = sym<int (*)(int)>("kind_eadim")(kind);
    if (K > 0) upload(s.slot, slot.data(), sizeof(int32_t) * n_);
    auto fused = sym<masked_fn>("batch_predict_update_" + std::to_string(kind) + "_masked");
    auto upd = nobs > 1 ? sym<masked_update_fn>("batch_update_" + std::to_string(kind) + "_masked") : nullptr;
    for (int j = 0; j < nobs; j++) {
      if (j == 0 || nobs_of) { const std::vector<uint8_t> m = mask_of(j); upload(s.act, m.data(), n_); }
      if (K > 0) {                          // the observation, before the kernel turns it into the residual
        ring_copy(s.ring_z + (size_t)j * s.zmax, (int64_t)pf_nmax_ * s.zmax, z_devs[j], Z, Z, true);
        if (ek > 0 && !ea_devs.empty()) ring_copy(s.ring_ea + (size_t)j * s.ead, (int64_t)pf_nmax_ * s.ead, const_cast<double*>(ea_devs[j]), ek, ek, true);
      }
      const double* ea = ea_devs.empty() ? nullptr : ea_devs[j];
      uint8_t* fl = flags_dev ? flags_dev + (size_t)j * n_ : nullptr;
      if (j == 0) check(fused(x_, P_, Q_, s.dt, 0.0, z_devs[0], R_devs[0], r_per_filter, ea, n_, norm_quats_, fl, s.act, stream_), "batch_predict_update_masked");
      else check(upd(x_, P_, z_devs[j], R_devs[j], r_per_filter, ea, n_, norm_quats_, fl, s.act, stream_), "batch_update_masked");
    }
    if (K > 0) {
      if (nobs_of && nobs > 1) upload(s.act, act.data(), n_);          // the state of EVERY filter of the call goes into its checkpoint
      ring_copy(s.ring_x, D_, x_, D_, D_, true);
      ring_copy(s.ring_P, (int64_t)E_ * E_, P_, (int64_t)E_ * E_, (int64_t)E_ * E_, true);
    }
  }

  using predict_fn = int (*)(double*, double*, const double*, const double*, double, int64_t, int, void*);
  using step_fn = int (*)(double*, double*, const double*, const double*, double, double*, const double*, int, const double*,
                          int64_t, int, uint8_t*, void*);
  using update_fn = int (*)(double*, double*, double*, const double*, int, const double*, int64_t, int, uint8_t*, void*);
  using ckpt_fn = int (*)(double*, double*, const double*, const double*, double, double*, const double*, int, const double*, int64_t, int, uint8_t*,
                          double*, double*, double*, void*);

  template <class F>
  F sym(const std::string& suffix) const {
    auto it = syms_.find(suffix);              // looked up once: a step-granular call is ~9 us of GPU time, dlsym + two string concatenations are not free beside it
    if (it == syms_.end()) it = syms_.emplace(suffix, dlsym(handle_, (name_ + "_" + suffix).c_str())).first;
    if (!it->second) throw std::runtime_error("rednose_amd: lib" + name_ + ".so does not export " + name_ + "_" + suffix);
    return reinterpret_cast<F>(it->second);
  }
  template <class F>
  F sym_optional(const std::string& suffix) const {       // nullptr for an entry point this library was built without
    auto it = syms_.find(suffix);
    if (it == syms_.end()) it = syms_.emplace(suffix, dlsym(handle_, (name_ + "_" + suffix).c_str())).first;
    return reinterpret_cast<F>(it->second);
  }
  // noise matrices go to the device staging buffer R_ only when they differ from what is there (the usual caller passes the same R for a kind on
  // every call; a pageable host-to-device copy per call costs more host time than the launch it feeds)
  void upload_R(size_t offset, const double* R_host, size_t count) {
    if (r_mirror_.size() < offset + count) r_mirror_.resize(64 * 64, NAN);
    if (std::equal(R_host, R_host + count, r_mirror_.begin() + offset)) return;       // (NaN never compares equal: the first call always uploads)
    hip(hipMemcpyAsync(R_ + offset, R_host, sizeof(double) * count, hipMemcpyHostToDevice, stream_), "copy R");
    std::copy(R_host, R_host + count, r_mirror_.begin() + offset);
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
  Bounce bounce_[8];                                   // pinned staging of upload()
  size_t bounce_next_ = 0;
  std::vector<double> r_mirror_;                       // host copy of what R_ holds (upload_R)
  mutable std::map<std::string, void*> syms_;          // resolved entry points (sym)
  double filter_time_ = NAN;
  int rewind_to_keep_ = 0;
  int pf_nmax_ = 1;                  // observations a per-filter ring entry can hold (set_max_observations_per_call)
  double max_rewind_age_ = 1.0;
  std::deque<Checkpoint> ring_;
  std::vector<Checkpoint> spare_;
  PerFilter pf_;
  ring_fn ring_copy_ = nullptr;
};

}  // namespace rednose_amd
