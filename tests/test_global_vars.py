"""global_vars (reference: ekf_sym.py:129-132,166-171 -- run-time scalars settable through {name}_set_{var}).
CPU part: the generated sources declare the device global and export the setter.  GPU part: a model whose process
model uses a run-time gain must match the same model with the gain baked in as a literal."""
import os

import numpy as np
import pytest
import sympy as sp

from conftest import REPO


def _model(gain, n=2):
  """Chain of n integrators whose coupling is `gain` (n = 2: lane-per-filter kernels; n = 10: lane-group kernels, where the
  gain is evaluated in the scalar phase and reaches the covariance phase through the LDS slot); the observation uses it too."""
  state_sym = sp.MatrixSymbol('state', n, 1)
  state = sp.Matrix(state_sym)
  dt = sp.Symbol('dt')
  rate = sp.zeros(n, 1)
  for i in range(n - 1):
    rate[i] = gain * state[i + 1, 0]
  f_sym = state + dt * rate
  obs = [[sp.Matrix([state[0, 0] + (gain * state[n - 1, 0] if n > 2 else 0)]), 1, None]]
  return dict(f_sym=f_sym, dt_sym=dt, x_sym=state_sym, obs_eqs=obs, dim_x=n, dim_err=n)


@pytest.fixture(scope="module")
def libs():
  from rednose_amd.helpers.ekf_sym import gen_code
  folder = os.path.join(REPO, "generated")
  g = sp.Symbol('gain')
  gen_code(folder, "gv_runtime", global_vars=[g], **_model(g))
  gen_code(folder, "gv_literal", **_model(sp.Float(2.5)))
  gen_code(folder, "gv_runtime10", global_vars=[g], **_model(g, 10))
  gen_code(folder, "gv_literal10", **_model(sp.Float(2.5), 10))
  return folder


def test_setter_is_generated_and_exported(libs):
  import ctypes
  with open(os.path.join(libs, "gv_runtime.h"), encoding="utf-8") as f:
    assert "void gv_runtime_set_gain(double x);" in f.read()
  with open(os.path.join(libs, "gv_runtime.hip"), encoding="utf-8") as f:
    assert "__device__ double gain" in f.read()
  assert hasattr(ctypes.CDLL(os.path.join(libs, "libgv_runtime.so")), "gv_runtime_set_gain")


@pytest.mark.gpu
@pytest.mark.parametrize("dim,suffix", [(2, ""), (10, "10")])
def test_runtime_global_equals_literal(libs, dim, suffix):
  import torch
  from rednose_amd.helpers.ekf_sym import EKF_sym, BatchedEKF
  Q = np.diag(np.linspace(0.01, 4.0, dim)); x0 = np.linspace(0.5, 0.3, dim); P0 = np.eye(dim)
  a = EKF_sym(libs, "gv_runtime" + suffix, Q, x0, P0, dim, dim, global_vars=["gain"])
  a.set_global("gain", 2.5)
  n = 100
  rng = np.random.default_rng(0)
  X0 = rng.normal(size=(n, dim))
  fa = BatchedEKF(libs, "gv_runtime" + suffix, Q, x0, P0, dim, dim, batch=n); fa.init_state(X0, P0, 0.0)
  fb = BatchedEKF(libs, "gv_literal" + suffix, Q, x0, P0, dim, dim, batch=n); fb.init_state(X0, P0, 0.0)
  for i in range(1, 6):
    z = rng.normal(size=(n, 1))
    fa.predict_and_update_batch(0.01 * i, 1, z.copy(), np.array([[0.01]]))
    fb.predict_and_update_batch(0.01 * i, 1, z.copy(), np.array([[0.01]]))
  torch.cuda.synchronize()
  # a literal gain lets the compiler fold constants that the run-time gain multiplies at run time: equal to rounding
  assert torch.allclose(fa.x, fb.x, rtol=1e-12, atol=1e-14) and torch.allclose(fa.P, fb.P, rtol=1e-12, atol=1e-14)
  a.set_global("gain", 0.0)          # with a zero gain the position no longer integrates the velocity
  fa.init_state(X0, P0, 0.0)
  fa.predict(1.0)
  torch.cuda.synchronize()
  assert np.array_equal(fa.state(), X0)


# ------------------------------------------------------------------ extra_routines (ekf_sym.py:94-95, ekf.h:32)
@pytest.fixture(scope="module")
def extra_lib():
  from rednose_amd.helpers.ekf_sym import gen_code
  folder = os.path.join(REPO, "generated")
  mdl = _model(sp.Float(1.0))
  st = mdl["x_sym"]
  other = sp.MatrixSymbol('other', 2, 1)
  energy = sp.Matrix([[0.5 * st[1, 0]**2 + 9.81 * st[0, 0] + other[0, 0] * other[1, 0]]])
  gen_code(folder, "gv_extra", extra_routines=[("energy", energy, [st, other])], **mdl)
  return folder


def test_extra_routine_is_exported(extra_lib):
  import ctypes
  with open(os.path.join(extra_lib, "gv_extra.h"), encoding="utf-8") as f:
    assert "void gv_extra_energy(double *state, double *other, double *out);" in f.read()
  assert hasattr(ctypes.CDLL(os.path.join(extra_lib, "libgv_extra.so")), "gv_extra_energy")


@pytest.mark.gpu
def test_extra_routine_value(extra_lib):
  from rednose_amd.helpers import load_code
  ffi, lib = load_code(extra_lib, "gv_extra")
  x = np.array([2.0, 3.0]); o = np.array([0.5, 4.0]); out = np.zeros(1)
  lib.gv_extra_energy(ffi.cast("double *", x.ctypes.data), ffi.cast("double *", o.ctypes.data), ffi.cast("double *", out.ctypes.data))
  assert abs(out[0] - (0.5 * 9.0 + 9.81 * 2.0 + 2.0)) < 1e-12
