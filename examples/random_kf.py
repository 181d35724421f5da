#!/usr/bin/env python3
"""Seeded random filters: a family of small nonlinear models of assorted sizes, written through the gen_code API.

Not part of the reference; they exist to exercise the emitter where the hand-written examples do not reach -- every
lane layout of the kernels (lane per filter up to 8 error states; 7, 5, 4, 3, 2 and 1 filters per wavefront above),
odd and even covariance record sizes, random sparsity in F and H, state-dependent Jacobians, 1- to 3-dimensional
observations.  f = x + dt * (A x + bilinear + sine terms), h_k = H_k x + a product term, all coefficients drawn from
numpy's default_rng(seed) at model-construction time (so the reference's gen_code and ours see the same expressions).
"""
import os
import sys

if __name__ == "__main__":  # allow running as a script from anywhere (generator CLI contract)
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import sympy as sp

from rednose_amd.helpers.kalmanfilter import KalmanFilter
from rednose_amd.helpers.ekf_sym import EKF_sym, BatchedEKF, gen_code

SIZES = (3, 5, 8, 11, 13, 17, 24, 32, 40, 56)
AFFINE_SIZES = (5, 11)     # "randaff{n}": f = A0 x + dt (...) with A0 != I, i.e. predict(dt = 0) is NOT the identity (a legal
                           # model for the reference, which only asserts that F depends on dt, ekf_sym.py:82); one per kernel family


WIDE_OBS = ((10, 9),)      # "randz{n}": kind 1 is a zbig-dimensional observation (> 8: more than one observation entry per lane in
                           # the fused run, an 81-entry innovation covariance in registers; the reference puts no limit on ZDIM)


def make(n, seed=None, affine=False, zbig=None):
  """-> a KalmanFilter subclass for an n-state random model (seed defaults to 1000 + n)."""
  seed = (2000 if affine else 1000) + n if seed is None else seed
  if zbig:
    seed += 5000
  rng = np.random.default_rng(seed)
  A0 = np.eye(n)
  if affine:
    A0 = np.diag(np.round(rng.uniform(0.9, 0.99, size=n), 3))
    A0[0, n - 1] = 0.05
    A0[n // 2, 1] = -0.03
  A = np.where(rng.random((n, n)) < min(0.5, 3.0 / n), np.round(rng.normal(size=(n, n)), 3), 0.0)
  kinds = {}
  for k, z in ((1, zbig or 3), (2, 1), (3, 2)):
    H = np.where(rng.random((z, n)) < min(0.6, 4.0 / n), np.round(rng.normal(size=(z, n)), 3), 0.0)
    H[np.arange(z), rng.choice(n, size=z, replace=False)] = 1.0          # every row observes something
    kinds[k] = (H, int(rng.integers(0, n)), int(rng.integers(0, n)))
  bil = [(int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(0, n)), round(float(rng.normal()) * 0.3, 3)) for _ in range(3)]
  sines = [(int(rng.integers(0, n)), int(rng.integers(0, n)), round(float(rng.normal()) * 0.5, 3)) for _ in range(2)]

  class RandomKalman(KalmanFilter):
    name = f"randz{n}" if zbig else (f"randaff{n}" if affine else f"rand{n}")
    dim = n
    initial_x = np.round(rng.normal(size=n) * 0.5, 3)
    initial_P_diag = np.round(rng.uniform(0.5, 2.0, size=n), 3)
    Q = np.diag(np.round(rng.uniform(0.01, 0.5, size=n) ** 2, 6))
    obs_noise = {1: np.eye(zbig or 3) * 0.1**2, 2: np.eye(1) * 0.2**2, 3: np.diag([0.05**2, 0.3**2])}

    @classmethod
    def model(cls):
      state_sym = sp.MatrixSymbol('state', n, 1)
      state = sp.Matrix(state_sym)
      dt = sp.Symbol('dt')
      rate = sp.Matrix(A) * state
      for i, a, b, c in bil:
        rate[i] += c * state[a] * state[b]
      for i, a, c in sines:
        rate[i] += c * sp.sin(state[a])
      f_sym = (sp.Matrix(A0) * state if affine else state) + dt * rate
      obs_eqs = []
      for k, (H, a, b) in kinds.items():
        h = sp.Matrix(H) * state
        h[0] += 0.25 * state[a] * state[b]
        obs_eqs.append([h, k, None])
      return dict(name=cls.name, f_sym=f_sym, dt_sym=dt, x_sym=state_sym, obs_eqs=obs_eqs, dim_x=n, dim_err=n)

    @classmethod
    def generate_code(cls, generated_dir, **gen_kwargs):
      gen_code(generated_dir, **cls.model(), **gen_kwargs)

    def __init__(self, generated_dir, batch=None, device=None):
      P0 = np.diag(self.initial_P_diag)
      if batch is None:
        self.filter = EKF_sym(generated_dir, self.name, self.Q, self.initial_x, P0, n, n)
      else:
        self.filter = BatchedEKF(generated_dir, self.name, self.Q, self.initial_x, P0, n, n, batch=batch, device=device)

  RandomKalman.__name__ = f"RandomWideObs{n}Kalman" if zbig else (f"RandomAffine{n}Kalman" if affine else f"Random{n}Kalman")
  return RandomKalman


# module-level classes so that "module:Class" references (oracle/build_oracle.py) resolve
for _n in SIZES:
  globals()[f"Random{_n}Kalman"] = make(_n)
for _n in AFFINE_SIZES:
  globals()[f"RandomAffine{_n}Kalman"] = make(_n, affine=True)
for _n, _z in WIDE_OBS:
  globals()[f"RandomWideObs{_n}Kalman"] = make(_n, zbig=_z)


if __name__ == "__main__":
  if sys.argv[1].startswith("randz"):
    globals()[f"RandomWideObs{int(sys.argv[1].replace('randz', ''))}Kalman"].generate_code(sys.argv[2])
  elif sys.argv[1].startswith("randaff"):
    globals()[f"RandomAffine{int(sys.argv[1].replace('randaff', ''))}Kalman"].generate_code(sys.argv[2])
  else:
    globals()[f"Random{int(sys.argv[1].replace('rand', ''))}Kalman"].generate_code(sys.argv[2])
