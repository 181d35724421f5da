#!/usr/bin/env python3
"""3-D constant-acceleration filter: state [pos(3), vel(3), acc(3)], three observation kinds.

A mid-size model (9 error states) between the reference's 2-state example and its 22-error-state live filter
(/root/reference/examples/kinematic_kf.py:31-67, /root/reference/examples/live_kf.py:54-237): the smallest
filter the lane-group kernel family serves, 7 filters per wavefront, odd-sized covariance records.  Kinds:
POSITION (linear, 3-D), RANGE (distance to a fixed anchor: nonlinear h, state-dependent H, 1-D) and VELOCITY
(linear, 3-D).  f = x + dt*[vel; acc; 0].
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
  RANGE = 2
  VELOCITY = 3


class States:
  POSITION = slice(0, 3)
  VELOCITY = slice(3, 6)
  ACCELERATION = slice(6, 9)


ANCHOR = (10.0, -5.0, 3.0)


class Kinematic9Kalman(KalmanFilter):
  name = 'kinematic9'

  initial_x = np.array([0.5, 0.5, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
  initial_P_diag = np.ones(9)
  Q = np.diag([0.1**2] * 3 + [0.5**2] * 3 + [2.0**2] * 3)
  obs_noise = {ObservationKind.POSITION: np.eye(3) * 0.1**2,
               ObservationKind.RANGE: np.eye(1) * 0.2**2,
               ObservationKind.VELOCITY: np.eye(3) * 0.3**2}

  @classmethod
  def model(cls):
    n = cls.initial_x.shape[0]
    state_sym = sp.MatrixSymbol('state', n, 1)
    state = sp.Matrix(state_sym)
    dt = sp.Symbol('dt')

    rate = sp.zeros(n, 1)
    rate[States.POSITION, 0] = state[States.VELOCITY, 0]
    rate[States.VELOCITY, 0] = state[States.ACCELERATION, 0]
    f_sym = state + dt * rate

    offset = sp.Matrix(state[States.POSITION, 0]) - sp.Matrix(ANCHOR)
    obs_eqs = [[sp.Matrix(state[States.POSITION, 0]), ObservationKind.POSITION, None],
               [sp.Matrix([sp.sqrt(offset.dot(offset))]), ObservationKind.RANGE, None],
               [sp.Matrix(state[States.VELOCITY, 0]), ObservationKind.VELOCITY, None]]
    return dict(name=cls.name, f_sym=f_sym, dt_sym=dt, x_sym=state_sym, obs_eqs=obs_eqs, dim_x=n, dim_err=n)

  @classmethod
  def generate_code(cls, generated_dir, **gen_kwargs):
    gen_code(generated_dir, **cls.model(), **gen_kwargs)

  def __init__(self, generated_dir, batch=None, device=None):
    n = self.initial_x.shape[0]
    P0 = np.diag(self.initial_P_diag)
    if batch is None:
      self.filter = EKF_sym(generated_dir, self.name, self.Q, self.initial_x, P0, n, n)
    else:
      self.filter = BatchedEKF(generated_dir, self.name, self.Q, self.initial_x, P0, n, n, batch=batch, device=device)


if __name__ == "__main__":
  Kinematic9Kalman.generate_code(sys.argv[2])
