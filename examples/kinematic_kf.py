#!/usr/bin/env python3
"""1-D constant-velocity filter: state [position, velocity], one POSITION observation kind.

Model parity: /root/reference/examples/kinematic_kf.py:36-76 (x0, P0, Q, R, f = x + dt*[v, 0],
h = [position]); this is the DIM=2 model the reference's known-answer test pins
(/root/reference/examples/test_kinematic_kf.py:52-55).  Usage as a generator script follows the
reference CLI contract:  kinematic_kf.py <target> <out_dir>   (only argv[2] is read).
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
  NO_OBSERVATION = 1
  POSITION = 1

  names = ['Unknown', 'No observation', 'Position']

  @classmethod
  def to_string(cls, kind):
    return cls.names[kind]


class States:
  POSITION = slice(0, 1)
  VELOCITY = slice(1, 2)


class KinematicKalman(KalmanFilter):
  name = 'kinematic'

  initial_x = np.array([0.5, 0.0])
  initial_P_diag = np.array([1.0**2, 1.0**2])
  Q = np.diag([0.1**2, 2.0**2])
  obs_noise = {ObservationKind.POSITION: np.atleast_2d(0.1**2)}

  @classmethod
  def model(cls):
    """Symbolic definition -> keyword arguments of gen_code."""
    n = cls.initial_x.shape[0]
    state_sym = sp.MatrixSymbol('state', n, 1)
    state = sp.Matrix(state_sym)
    dt = sp.Symbol('dt')

    rate = sp.zeros(n, 1)
    rate[States.POSITION.start, 0] = state[States.VELOCITY.start, 0]
    f_sym = state + dt * rate

    obs_eqs = [[sp.Matrix([state[States.POSITION.start, 0]]), ObservationKind.POSITION, None]]
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
  KinematicKalman.generate_code(sys.argv[2])
