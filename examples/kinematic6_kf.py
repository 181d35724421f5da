#!/usr/bin/env python3
"""3-D constant-velocity filter: state [pos(3), vel(3)], one 3-D POSITION observation kind.

BASELINE.json's headline config names a "6-state pos/vel" kinematic filter; the reference only ships
the 1-D (2-state) version (/root/reference/examples/kinematic_kf.py:31-67).  This is the obvious
3-axis generalisation written through the same gen_code API:
f = x + dt*[v; 0], h = pos, x0 = [0.5,0.5,0.5,0,0,0], P0 = I, Q = diag(0.1^2 x3, 2.0^2 x3),
R = 0.1^2 I3.  Each axis is an independent copy of the reference's 2-state filter, which
tests/test_oracle.py uses as a cross-check.
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


class States:
  POSITION = slice(0, 3)
  VELOCITY = slice(3, 6)


class Kinematic6Kalman(KalmanFilter):
  name = 'kinematic6'

  initial_x = np.array([0.5, 0.5, 0.5, 0.0, 0.0, 0.0])
  initial_P_diag = np.ones(6)
  Q = np.diag([0.1**2] * 3 + [2.0**2] * 3)
  obs_noise = {ObservationKind.POSITION: np.eye(3) * 0.1**2}

  @classmethod
  def model(cls):
    n = cls.initial_x.shape[0]
    state_sym = sp.MatrixSymbol('state', n, 1)
    state = sp.Matrix(state_sym)
    dt = sp.Symbol('dt')

    rate = sp.zeros(n, 1)
    rate[States.POSITION, 0] = state[States.VELOCITY, 0]
    f_sym = state + dt * rate

    obs_eqs = [[sp.Matrix(state[States.POSITION, 0]), ObservationKind.POSITION, None]]
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
  Kinematic6Kalman.generate_code(sys.argv[2])
