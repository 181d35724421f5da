#!/usr/bin/env python3
"""Small MSCKF: a moving camera observes point landmarks from a sliding window of its last N positions.

The reference ships the MSCKF machinery (gen_code's msckf_params, /root/reference/rednose/helpers/ekf_sym.py:57-73;
block-structured predict, /root/reference/rednose/templates/ekf_c.c:23-26; null-space projection through
He = dh/d(extra args), ekf_c.c:66-76 and ekf_sym.py:576-591; window shift augment(), ekf_sym.py:365-391) but no
model that uses it.  This is the smallest one that exercises all of it:

  main state   [pos(3), vel(3)]                  dim_main = dim_main_err = 6, f = x + dt*[vel; 0]
  window       N = 3 past positions               dim_augment = dim_augment_err = 3  ->  dim_x = dim_err = 15
  POSITION     h = pos                            ordinary 3-D kind
  FEATURE      h = normalised image coordinates of landmark l (the 3 extra args) seen from every window position,
               [(l - p_i).x / (l - p_i).z, (l - p_i).y / (l - p_i).z] for i < N  ->  Z = 6, He = dh/dl is 6x3, the
               residual is projected on the 3-dimensional left null space of He before the update
"""
import os
import sys

if __name__ == "__main__":  # allow running as a script from anywhere (generator CLI contract)
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import sympy as sp

from rednose_amd.helpers.kalmanfilter import KalmanFilter
from rednose_amd.helpers.ekf_sym import EKF_sym, BatchedEKF, gen_code


class ObservationKind:
  UNKNOWN = 0
  POSITION = 1
  FEATURE = 2


N_WINDOW = 3
DIM_MAIN, DIM_AUGMENT = 6, 3


class FeatureKalman(KalmanFilter):
  name = 'feature'
  n_window = N_WINDOW
  dim_augment = DIM_AUGMENT          # window entries are copies of the first dim_augment main states (ekf_sym.py:366-367)
  observed = tuple(range(N_WINDOW))  # window entries the camera model uses

  dim_state = DIM_MAIN + N_WINDOW * DIM_AUGMENT
  initial_x = np.concatenate([[0.0, 0.0, 0.0, 1.0, 0.5, 0.0], np.tile([0.0, 0.0, 0.0], N_WINDOW)])
  initial_P_diag = np.concatenate([[0.5**2] * 3 + [1.0**2] * 3, [0.5**2] * (3 * N_WINDOW)])
  Q = np.diag([0.05**2] * 3 + [0.5**2] * 3 + [0.0] * (3 * N_WINDOW))
  obs_noise = {ObservationKind.POSITION: np.eye(3) * 0.2**2,
               ObservationKind.FEATURE: np.eye(2 * N_WINDOW) * 0.01**2}

  @classmethod
  def model(cls):
    n = cls.dim_state
    state_sym = sp.MatrixSymbol('state', n, 1)
    state = sp.Matrix(state_sym)
    dt = sp.Symbol('dt')

    rate = sp.zeros(n, 1)
    rate[0:3, 0] = state[3:6, 0]
    f_sym = state + dt * rate

    landmark_sym = sp.MatrixSymbol('landmark', 3, 1)
    landmark = sp.Matrix(landmark_sym)
    rows = []
    for i in cls.observed:
      at = DIM_MAIN + cls.dim_augment * i
      ray = landmark - state[at:at + 3, 0]
      rows += [ray[0] / ray[2], ray[1] / ray[2]]
    obs_eqs = [[sp.Matrix(state[0:3, 0]), ObservationKind.POSITION, None],
               [sp.Matrix(rows), ObservationKind.FEATURE, landmark_sym]]
    msckf_params = [DIM_MAIN, cls.dim_augment, DIM_MAIN, cls.dim_augment, cls.n_window, [ObservationKind.FEATURE]]
    return dict(name=cls.name, f_sym=f_sym, dt_sym=dt, x_sym=state_sym, obs_eqs=obs_eqs, dim_x=n, dim_err=n,
                msckf_params=msckf_params)

  @classmethod
  def generate_code(cls, generated_dir, **gen_kwargs):
    gen_code(generated_dir, **cls.model(), **gen_kwargs)

  @classmethod
  def filter_kwargs(cls):
    return dict(N=cls.n_window, dim_augment=cls.dim_augment, dim_augment_err=cls.dim_augment)

  def __init__(self, generated_dir, batch=None, device=None):
    P0 = np.diag(self.initial_P_diag)
    if batch is None:
      self.filter = EKF_sym(generated_dir, self.name, self.Q, self.initial_x, P0, DIM_MAIN, DIM_MAIN, **self.filter_kwargs())
    else:
      self.filter = BatchedEKF(generated_dir, self.name, self.Q, self.initial_x, P0, DIM_MAIN, DIM_MAIN, batch=batch, device=device,
                               **self.filter_kwargs())


class WideFeatureKalman(FeatureKalman):
  """The same camera with a window of five [pos, vel] copies (36 error states: one filter per wavefront in the lane-group
  kernels), landmarks observed from window entries 0, 2 and 4."""
  name = 'feature36'
  n_window = 5
  dim_augment = 6
  observed = (0, 2, 4)

  dim_state = DIM_MAIN + 5 * 6
  initial_x = np.concatenate([[0.0, 0.0, 0.0, 1.0, 0.5, 0.0], np.tile([0.0, 0.0, 0.0, 1.0, 0.5, 0.0], 5)])
  initial_P_diag = np.concatenate([[0.5**2] * 3 + [1.0**2] * 3, np.tile([0.5**2] * 3 + [1.0**2] * 3, 5)])
  Q = np.diag([0.05**2] * 3 + [0.5**2] * 3 + [0.0] * 30)


if __name__ == "__main__":
  {"feature": FeatureKalman, "feature36": WideFeatureKalman}[sys.argv[1] if sys.argv[1] in ("feature", "feature36") else "feature"].generate_code(sys.argv[2])
