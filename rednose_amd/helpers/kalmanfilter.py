"""`KalmanFilter` base class: the thin user-facing surface named by the north star.

API parity with /root/reference/rednose/helpers/kalmanfilter.py:6-52 -- class-attribute model
configuration (name, initial_x, initial_P_diag, Q, obs_noise), `x/t/P` properties, `init_state`,
`get_R`, `predict_and_observe`.  `self.filter` may be an `EKF_sym` (one filter) or a `BatchedEKF`
(N filters on the GPU); for the latter `data` is (N, Z) -- or (N, n, Z): n observations per filter in one call -- and R may be one
shared (Z, Z) matrix.
"""
from typing import Any

import numpy as np


class KalmanFilter:
  name = "<name>"
  initial_x: np.ndarray = np.zeros(0)
  initial_P_diag: np.ndarray = np.zeros(0)
  Q: np.ndarray = np.zeros((0, 0))
  obs_noise: dict[int, Any] = {}

  # set by the concrete model's constructor
  filter: Any = None

  @property
  def x(self):
    return self.filter.state()

  @property
  def t(self):
    return self.filter.get_filter_time()

  @property
  def P(self):
    return self.filter.covs()

  def init_state(self, state, covs_diag=None, covs=None, filter_time=None):
    if covs_diag is not None:
      cov = np.diag(covs_diag)
    elif covs is not None:
      cov = covs
    else:
      cov = self.filter.covs()
    self.filter.init_state(state, cov, filter_time)

  def get_R(self, kind, n):
    """(n, Z, Z) stack of the configured observation noise for `kind`."""
    noise = np.atleast_2d(self.obs_noise[kind])
    return np.broadcast_to(noise, (n,) + noise.shape).copy()

  def predict_and_observe(self, t, kind, data, R=None):
    """One filter (EKF_sym): data (n, Z), the n observations of this call, R (n, Z, Z) -- the reference's shapes (kalmanfilter.py:45-52).
    Batched filter: data (N, Z) one observation per filter, or (N, n, Z) n observations per filter (then R defaults to the (n, Z, Z) stack)."""
    if len(data) > 0:
      data = np.atleast_2d(data)
    if R is None:
      R = self.get_R(kind, data.shape[1] if np.ndim(data) == 3 else len(data))
    return self.filter.predict_and_update_batch(t, kind, data, R)

  def predict_and_observe_stream(self, ts, kinds, data, Rs=None, extra_args=None, augment=None):
    """A whole schedule of observations: ts (T,), kinds (T,), data (T, N, zmax) -- one observation per filter and step
    (rows of kinds with Z < zmax padded).  With a batched filter whose library has the fused multi-step entry point this is ONE
    launch: x and P stay on chip for all T steps and only z / y cross HBM ({name}_batch_run).  That is the default path for
    streams because a launch per step cannot feed small models: the 2-state kinematic filter moves 112 B per filter-step,
    8 MB per launch at 65 536 filters -- 1.3 us of HBM time under a 4.5 us launch (14 G steps/s, 20 % of the HBM roofline),
    against 48 G steps/s for the fused run (round-2 bench, "kinematic_fused").
    extra_args (T, N, 3): per filter and step extra arguments of the kinds that take them (MSCKF feature tracks: the landmark);
    augment (T,) bool: MSCKF window shift after that step (EKF_sym.augment) -- both ride in the fused schedule.
    A single (host-pointer) filter, or a library without the entry point (more than 32 error states), falls back to one call per
    step.  Returns the residuals (T, N, zmax) for the fused path, a list of per-step results otherwise."""
    ts = np.asarray(ts, dtype=np.float64)
    kinds = np.asarray(kinds, dtype=np.int32)
    if Rs is None:
      Rs = {int(k): np.atleast_2d(self.obs_noise[int(k)]) for k in set(kinds.tolist())}
    f = self.filter
    if hasattr(f, "run"):
      from rednose_amd.helpers import KalmanError
      try:
        return f.run(ts, kinds, data, Rs, extra_args=extra_args, augment=augment)[0]
      except KalmanError as e:
        if "-> 4" not in str(e):      # status 4: this library has no fused run (more than 32 error states)
          raise
    out = []
    for i, (t, k, z) in enumerate(zip(ts, kinds, data)):
      Z = np.atleast_2d(Rs[int(k)]).shape[0]
      zk = z[..., :Z]
      if hasattr(f, "run"):
        ea = None if extra_args is None or not getattr(f, "eadims", {}).get(int(k), 0) else extra_args[i]
        out.append(f.predict_and_update_batch(float(t), int(k), zk, Rs[int(k)], extra_args=ea,
                                              augment=bool(augment[i]) if augment is not None else False))
      else:
        out.append(self.predict_and_observe(float(t), int(k), zk, None))
    return out
