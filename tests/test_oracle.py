"""CPU tests that PIN THE ORACLE (oracle/ekf_oracle.c + the sympy C block) before anything trusts it.

Pins, strongest first:
  * the four literals of /root/reference/examples/test_kinematic_kf.py:52-55 (7 decimal places there);
  * tests/golden/*.npz -- trajectories / single steps produced by the reference's own Python path
    (oracle/make_golden.py: EKF_sym._predict_python/_update_python over reference-generated C);
  * oracle flavour `port` (our sympy front end) == flavour `ref` (reference gen_code) when both exist.
"""
import os

import numpy as np
import pytest

from conftest import assert_close, golden
import build_oracle
from oracle_lib import OracleLib

HAVE_REF = build_oracle.have_reference() or os.path.exists(os.path.join(os.path.dirname(build_oracle.__file__), "_ref", "liblive.so"))


def _kinematic_run(lib):
  g = golden("kinematic_stream.npz")
  Q = np.diag([0.1**2, 2.0**2])
  x = np.array([0.5, 0.0]); P = np.diag([1.0, 1.0])
  R = np.array([[0.1**2]])
  xs, Ps = [], []
  t_prev = None
  for t, meas in zip(g["ts"], g["zs"]):
    dt = 0.0 if t_prev is None else t - t_prev   # first call: filter_time := t (ekf_sym.cc:198-200)
    t_prev = t
    lib.predict(x, P, Q, dt)
    z = np.array([meas])
    lib.update(1, x, P, z, R)
    xs.append(x.copy()); Ps.append(P.copy())
  return np.array(xs), np.array(Ps), g


@pytest.mark.parametrize("flavour", ["auto", "port"])
def test_kinematic_known_answers(flavour):
  lib = OracleLib("kinematic", flavour)
  xs, Ps, g = _kinematic_run(lib)
  lit = g["literals"]
  std = np.sqrt(np.diag(Ps[-1]))
  # the reference asserts 7 decimal places; the oracle is good to ~1e-15
  assert abs(xs[-1][0] - lit[0]) < 1e-13 and abs(std[0] - lit[1]) < 1e-13
  assert abs(xs[-1][1] - lit[2]) < 1e-13 and abs(std[1] - lit[3]) < 1e-13
  assert_close(xs, g["xs"], rtol=1e-11, floor=1e-13, what="kinematic x trajectory vs reference numpy path")
  assert_close(Ps.reshape(len(Ps), -1), g["Ps"].reshape(len(Ps), -1), rtol=1e-11, floor=1e-13, what="kinematic P trajectory")


def test_kinematic6_is_three_copies_of_kinematic():
  """Each axis of the builder-defined 6-state model must reproduce the reference's pinned 2-state filter."""
  lib6 = OracleLib("kinematic6")
  g = golden("kinematic_stream.npz")
  Q = np.diag([0.1**2] * 3 + [2.0**2] * 3)
  x = np.array([0.5] * 3 + [0.0] * 3); P = np.eye(6)
  R = np.eye(3) * 0.1**2
  t_prev = None
  for t, meas in zip(g["ts"], g["zs"]):
    dt = 0.0 if t_prev is None else t - t_prev
    t_prev = t
    lib6.predict(x, P, Q, dt)
    z = np.array([meas] * 3)
    lib6.update(1, x, P, z, R)
  lit = g["literals"]
  for ax in range(3):
    assert abs(x[ax] - lit[0]) < 1e-12 and abs(x[3 + ax] - lit[2]) < 1e-12
    assert abs(np.sqrt(P[ax, ax]) - lit[1]) < 1e-12 and abs(np.sqrt(P[3 + ax, 3 + ax]) - lit[3]) < 1e-12


def test_live_single_steps_vs_reference_numpy():
  lib = OracleLib("live")
  g = golden("live_single_steps.npz")
  from examples.live_kf import LiveKalman
  n = g["x_in"].shape[0]
  for i in range(n):
    x = g["x_in"][i].copy(); P = g["P_in"][i].copy()
    lib.predict(x, P, LiveKalman.Q, g["predict_dt"][i])
    assert_close(x, g["predict_x"][i], what=f"predict x[{i}]")
    assert_close(P, g["predict_P"][i], rtol=1e-11, floor=1e-13, what=f"predict P[{i}]")
  for k in (3, 4, 9, 10, 12, 13, 14, 19):
    for i in range(n):
      x = g["x_in"][i].copy(); P = g["P_in"][i].copy(); z = g[f"upd{k}_z"][i].copy()
      lib.update(k, x, P, z, g[f"upd{k}_R"])
      assert_close(x, g[f"upd{k}_x"][i], rtol=1e-9, floor=1e-12, what=f"update_{k} x[{i}]")
      assert_close(P, g[f"upd{k}_P"][i], rtol=1e-8, floor=1e-10, what=f"update_{k} P[{i}]")
      assert_close(z, g[f"upd{k}_y"][i], rtol=1e-9, floor=1e-12, what=f"update_{k} y[{i}]")


def test_live_stream_vs_reference_numpy():
  """84-step IMU+GNSS stream with quaternion renormalisation after predict and update."""
  lib = OracleLib("live")
  g = golden("live_stream.npz")
  from examples.live_kf import LiveKalman
  L = LiveKalman
  kinds, ts, zs = g["kinds"], g["ts"], g["zs"]
  T = len(kinds)
  x = g["x0"][None].copy(); P = g["P0"][None].copy()
  dts = np.diff(np.concatenate([[ts[0]], ts]))
  z = np.zeros((T, 1, 3)); z[:, 0, :] = zs
  R = np.stack([L.obs_noise[int(k)] for k in kinds])
  xf = np.zeros((T, 1, 23)); Pf = np.zeros((T, 1, 22, 22)); xp = np.zeros((T, 1, 23))
  lib.batch_run(kinds, dts, x, P, z, R, L.Q, quat_idx=3, xp=xp, xf=xf, Pf=Pf)
  assert_close(xp[:, 0], g["x_pred"], rtol=1e-8, floor=1e-10, what="live stream predicted states")
  assert_close(xf[:, 0], g["xs"], rtol=1e-8, floor=1e-10, what="live stream states")
  assert_close(z[:, 0], g["ys"], rtol=1e-7, floor=1e-9, what="live stream residuals")
  assert_close(Pf[g["P_idx"], 0].reshape(len(g["P_idx"]), -1), g["Ps"].reshape(len(g["P_idx"]), -1), rtol=1e-7, floor=1e-9,
               what="live stream covariances")


def test_maha_gate_matches_reference_decisions():
  """ekf_c.c:88-94 gate (live_maha: kind 12 generated with MAHA_TEST=1) vs the reference's maha_test()."""
  lib = OracleLib("live_maha")
  g = golden("live_maha.npz")
  n = g["x"].shape[0]
  x = g["x"].copy(); P = g["P"].copy(); z = g["z"].copy()
  flags = np.zeros(n, dtype=np.uint8)
  from examples.live_kf import LiveKalman
  lib.batch_step(12, x, P, z, g["R"], LiveKalman.Q, 0.0, flags=flags, do_predict=False)
  assert np.array_equal(flags == 0, g["accepted"])
  # a gated update must leave the state essentially untouched (R inflated by 1e16)
  gated = flags == 1
  assert gated.sum() == 16
  assert np.allclose(x[gated], g["x"][gated], rtol=0, atol=1e-6)


def test_msckf_feature_updates_and_block_predict_vs_reference_numpy():
  """ekf_c.c:23-26 (block predict, MEDIM < EDIM) and :66-76 (null-space projection) against the reference's numpy
  path (ekf_sym.py:541-556, :576-591).  x and P do not depend on the null-space basis; the projected residual does
  (SVD basis there, LU kernel here), so y is compared through its basis-independent quadratic form."""
  g = golden("feature_stream.npz")
  o = OracleLib("feature")
  assert (o.D, o.E, o.M) == (15, 15, 6) and o.zdim(2) == 6
  R = np.eye(6) * 0.01**2
  for i in range(g["upd_x_in"].shape[0]):
    x, P, z = g["upd_x_in"][i].copy(), g["upd_P_in"][i].copy(), g["upd_z"][i].copy()
    o.update(2, x, P, z, R, ea=np.concatenate([g["upd_ea"][i], [0.0]]))
    assert_close(x, g["upd_x"][i], rtol=1e-9, floor=1e-11, what=f"feature update x[{i}]")
    assert_close(P, g["upd_P"][i], rtol=1e-8, floor=1e-10, what=f"feature update P[{i}]")
    # R is a multiple of the identity here: for an orthonormal basis |y|^2/r is the form y^T (A^T R A)^-1 y; the LU-kernel
    # basis A gives the same number through (A^T A)^-1
    He = np.zeros(18); o.call("He_2", g["upd_x_in"][i].copy(), g["upd_ea"][i].copy(), He)
    u, sv, _ = np.linalg.svd(He.reshape(6, 3)); A = u[:, 3:]
    hx = np.zeros(6); o.call("h_2", g["upd_x_in"][i].copy(), g["upd_ea"][i].copy(), hx)
    want = np.linalg.norm(A.T @ (g["upd_z"][i] - hx))
    assert abs(np.linalg.norm(g["upd_y"][i]) - want) < 1e-9 * max(1.0, want)
  Q = np.diag([0.05**2] * 3 + [0.5**2] * 3 + [0.0] * 9)
  x, P, t_prev = None, None, None
  for t in range(len(g["ts"])):
    if t == 0:
      x, P, dt = np.concatenate([[0.0, 0.0, 0.0, 1.0, 0.5, 0.0], np.zeros(9)]), np.diag([0.25] * 3 + [1.0] * 3 + [0.25] * 9), 0.0
    else:
      x, P, dt = g["x_after"][t - 1].copy(), g["P_after"][t - 1].copy(), g["ts"][t] - g["ts"][t - 1]
    o.predict(x, P, Q, dt)
    assert_close(x, g["xk_km1"][t], rtol=1e-10, floor=1e-12, what=f"predict x t={t}")
    assert_close(P, g["Pk_km1"][t], rtol=1e-10, floor=1e-12, what=f"predict P t={t}")
    k = int(g["kinds"][t]); Z = 3 if k == 1 else 6
    z = g["zs"][t, :Z].copy()
    o.update(k, x, P, z, np.eye(Z) * (0.2**2 if k == 1 else 0.01**2), ea=np.concatenate([g["eas"][t], [0.0]]))
    assert_close(x, g["xk_k"][t], rtol=1e-9, floor=1e-11, what=f"update x t={t}")
    assert_close(P, g["Pk_k"][t], rtol=1e-8, floor=1e-10, what=f"update P t={t}")


def test_thresholds_match_reference_table():
  from rednose_amd.helpers.chi2_lookup import chi2_ppf
  g = golden("live_maha.npz")
  for dim, want in zip((1, 2, 3, 6), g["thresholds"]):
    assert abs(chi2_ppf(0.95, dim) - want) < 1e-12 * want
  # the three constants the reference bakes into generated code
  assert abs(chi2_ppf(0.95, 1) - 3.8414588206941227) < 1e-13
  assert abs(chi2_ppf(0.95, 3) - 7.814727903251177) < 1e-13
  assert abs(chi2_ppf(0.95, 6) - 12.591587243743978) < 1e-13


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["kinematic", "live", "kinematic6", "kinematic9", "feature", "rand5", "rand17"])
def test_port_flavour_equals_ref_flavour(name):
  """Our model front end (examples/ + rednose_amd.codegen.spec) must generate the same functions as the reference."""
  a, b = OracleLib(name, "ref"), OracleLib(name, "port")
  assert (a.D, a.E, a.M) == (b.D, b.E, b.M)
  rng = np.random.default_rng(3)
  if name == "live":
    from examples.live_kf import LiveKalman
    x = LiveKalman.initial_x + rng.normal(size=23) * np.array([100] * 3 + [0.2] * 4 + [3] * 3 + [0.1] * 3 + [0.01] * 3 + [0.01] + [0.5] * 3 + [0.02] * 3)
    kinds = [3, 4, 9, 10, 12, 13, 14, 19]
  else:
    x = rng.normal(size=a.D)
    kinds = {"kinematic9": [1, 2, 3], "feature": [1, 2], "rand5": [1, 2, 3], "rand17": [1, 2, 3]}.get(name, [1])
  dt = 0.037
  def both(sym, *args, shape):
    oa, ob = np.zeros(shape), np.zeros(shape)
    a.call(sym, *args, oa); b.call(sym, *args, ob)
    assert_close(ob, oa, rtol=1e-13, floor=1e-15, what=f"{name}.{sym}")
  both("f_fun", x, dt, shape=a.D)
  both("F_fun", x, dt, shape=a.E * a.E)
  both("H_mod_fun", x, shape=a.D * a.E)
  dx = rng.normal(size=a.E) * 0.01
  both("err_fun", x, dx, shape=a.D)
  x2 = x + rng.normal(size=a.D) * 1e-3
  both("inv_err_fun", x, x2, shape=a.E)
  ea = np.array([0.3, -0.2, 7.0])       # landmark of the feature kind; ignored by kinds without extra args
  for k in kinds:
    Z = a.zdim(k)
    assert b.zdim(k) == Z
    both(f"h_{k}", x, ea, shape=Z)
    both(f"H_{k}", x, ea, shape=Z * a.D)
    if name == "feature" and k == 2:
      both("He_2", x, ea, shape=Z * 3)


def test_zero_dt_shortcut_is_guarded_symbolically():
  """The lane-group kernels skip the covariance phase of predict(dt = 0) only for models where that is the identity:
  f(x, 0) == x and F(x, 0) == I symbolically (ekf_c.c:15-28 evaluates f and F unconditionally)."""
  import examples.random_kf as R
  from examples.live_kf import LiveKalman
  from examples.kinematic9_kf import Kinematic9Kalman
  from rednose_amd.helpers.ekf_sym import gen_code
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    assert gen_code(d, **R.Random11Kalman.model(), compile=False).identity_at_dt0()
    assert gen_code(d, **Kinematic9Kalman.model(), compile=False).identity_at_dt0()
    for dim in R.AFFINE_SIZES:
      spec = gen_code(d, **getattr(R, f"RandomAffine{dim}Kalman").model(), compile=False)
      assert not spec.identity_at_dt0()
    with open(os.path.join(d, "randaff11.hip"), encoding="utf-8") as f:
      src = f.read()
    assert "dt_scalar == 0.0" not in src and "const bool do_pred = true;" in src
    with open(os.path.join(d, "rand11.hip"), encoding="utf-8") as f:
      src = f.read()
    assert "dt_scalar == 0.0" in src and "const bool do_pred = dt != 0.0;" in src
