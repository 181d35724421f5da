"""`KalmanFilter` base class: the thin user-facing surface named by the north star.

API parity with /root/reference/rednose/helpers/kalmanfilter.py:6-52 -- class-attribute model
configuration (name, initial_x, initial_P_diag, Q, obs_noise), `x/t/P` properties, `init_state`,
`get_R`, `predict_and_observe`.  `self.filter` may be an `EKF_sym` (one filter) or a `BatchedEKF`
(N filters on the GPU); for the latter `data` is (N, Z) and R may be one shared (Z, Z) matrix.
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
    if len(data) > 0:
      data = np.atleast_2d(data)
    if R is None:
      R = self.get_R(kind, len(data))
    return self.filter.predict_and_update_batch(t, kind, data, R)
