"""GPU parity of the MSCKF path (windowed-camera example, examples/feature_kf.py): block-structured predict, null-space
projected feature updates, window shift.  Goldens come from the reference's numpy path (tests/golden/feature_stream.npz,
oracle/make_golden.py); random batches are checked against the oracle.  x and P do not depend on the null-space basis
(tolerance as for every other kind); the projected residual does, so it is compared through its norm (the reference's
numpy path and the HIP kernels both use orthonormal bases: equal up to a rotation)."""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["feature", "feature36"])
def env(request):
  """feature: 15 error states, 4 filters per wavefront; feature36: 36 error states, one filter per wavefront."""
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  from examples.feature_kf import FeatureKalman, WideFeatureKalman
  FK = {"feature": FeatureKalman, "feature36": WideFeatureKalman}[request.param]
  return torch, ensure_generated([request.param]), FK


def _filter(env, n):
  torch, gen, FK = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  return BatchedEKF(gen, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), 6, 6, batch=n, **FK.filter_kwargs())


def _gold(env):
  return golden("feature_stream.npz" if env[2].name == "feature" else "feature36_stream.npz")


def test_feature_updates_vs_reference_numpy(env):
  torch, gen, FK = env
  g = _gold(env)
  n = g["upd_x_in"].shape[0]
  f = _filter(env, n)
  f.init_state(g["upd_x_in"], g["upd_P_in"], 0.0)
  y = f.update(2, g["upd_z"].copy(), FK.obs_noise[2], extra_args=g["upd_ea"])
  torch.cuda.synchronize()
  assert_close(f.state(), g["upd_x"], rtol=1e-9, floor=1e-11, what="feature update x")
  assert_close(f.covs().reshape(n, -1), g["upd_P"].reshape(n, -1), rtol=1e-8, floor=1e-10, what="feature update P")
  y = y.cpu().numpy()
  # the golden residual is the numpy path's (an orthonormal null-space basis, ekf_sym.py:20-26); ours is in the basis of the reference's C
  # path, A = Hea^T.fullPivLu().kernel() (ekf_c.c:71), which is not orthonormal: |y_numpy|^2 = y^T (A^T A)^-1 y, A rebuilt from the oracle's Hea
  from oracle_lib import OracleLib
  o = OracleLib(FK.name)
  for i in range(n):
    Hea = np.zeros(6 * 3)
    o.call("He_2", g["upd_x_in"][i].copy(), np.ascontiguousarray(g["upd_ea"][i], dtype=np.float64), Hea)
    Aref = _fullpiv_kernel(Hea.reshape(6, 3).T)
    w = y[i, :3] @ np.linalg.solve(Aref.T @ Aref, y[i, :3])
    assert abs(w - g["upd_y"][i] @ g["upd_y"][i]) <= 1e-9 * max(w, 1e-30), f"filter {i}: basis-independent norm of the projected residual"
  assert np.array_equal(y[:, 3:], g["upd_z"][:, 3:])          # y has Z - 3 rows; the tail of z is left alone (ekf_c.c:120)
  assert not f.flags.cpu().numpy().any()


@pytest.mark.parametrize("n", [1, 4, 5, 13, 200])
def test_both_kinds_vs_oracle_strict(env, n):
  """Random states; fused predict+update and split launches; tiles of 16 filters in groups of 4."""
  torch, gen, FK = env
  from oracle_lib import OracleLib
  o = OracleLib(FK.name)
  D = FK.dim_state
  rng = np.random.default_rng(50 + n)
  x0 = np.tile(FK.initial_x, (n, 1)) + rng.normal(size=(n, D)) * 0.3
  A = rng.normal(size=(n, D, D)) * 0.2
  P0 = np.diag(FK.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)
  landmarks = np.array([2.0, 1.0, 8.0])[None] + rng.normal(size=(n, 3))
  f = _filter(env, n)
  for kind, Z in ((1, 3), (2, 6)):
    R = FK.obs_noise[kind]
    for fused in (True, False):
      z = rng.normal(size=(n, Z)) * 0.3
      f.init_state(x0, P0, 0.0)
      xr, Pr, zr = x0.copy(), P0.copy(), z.copy()
      o.batch_step(kind, xr, Pr, zr, R, FK.Q, 0.05, ea=landmarks)
      if fused:
        y = f.predict_and_update_batch(0.05, kind, z.copy(), R, extra_args=landmarks)
      else:
        f.predict(0.05)
        y = f.update(kind, z.copy(), R, extra_args=landmarks)
      torch.cuda.synchronize()
      what = f"kind {kind} n={n} fused={fused}"
      assert_close(f.state(), xr, rtol=1e-11, floor=1e-13, what=what + " x")
      assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-10, floor=1e-12, what=what + " P")
      if kind == 1:
        assert_close(y.cpu().numpy(), zr, atol=1e-14 * np.abs(z).max(), what=what + " y")
      else:
        # Feature-track kinds: the residual written back into z is A^T (z - h) with A = Hea^T.fullPivLu().kernel() (ekf_c.c:71-73,120),
        # which is NOT orthonormal.  The update runs in the reflectors' orthonormal basis (x and P above do not depend on the choice);
        # y is formed a second time the way Eigen forms it (codegen/lower.py: rn::nullspace_residual, the oracle's loops) -- compared
        # entry for entry with the oracle's, which is pinned to the reference's numpy path through the golden stream.
        yh, Zp = y.cpu().numpy(), Z - 3
        assert_close(yh[:, :Zp], zr[:, :Zp], rtol=1e-10, atol=1e-12 * max(1.0, np.abs(z).max()), what=what + " y in the reference's null-space basis")
        assert np.array_equal(yh[:, Zp:], z[:, Zp:]), what + ": the last 3 entries of z pass through (y has Z - 3 rows, ekf_c.c:120)"


from conftest import fullpiv_kernel as _fullpiv_kernel      # noqa: E402  (Eigen's FullPivLU::kernel() restated in numpy)


def test_stream_with_window_shifts_vs_reference_numpy(env):
  """The golden stream: POSITION fixes followed by augment(), FEATURE tracks in between, every step against the
  reference's numpy filter (including the state after each window shift)."""
  torch, gen, FK = env
  g = _gold(env)
  n = 6
  f = _filter(env, n)
  for t in range(len(g["ts"])):
    k = int(g["kinds"][t]); Z = 3 if k == 1 else 6
    z = np.tile(g["zs"][t, :Z], (n, 1))
    est = f.predict_and_update_batch(float(g["ts"][t]), k, z, FK.obs_noise[k], extra_args=np.tile(g["eas"][t], (n, 1)),
                                     augment=bool(g["augment"][t]), keep_estimate=True)
    torch.cuda.synchronize()
    for j in (0, n - 1):
      assert_close(est[0].cpu().numpy()[j], g["xk_km1"][t], rtol=1e-8, floor=1e-10, what=f"xk_km1 t={t}")
      assert_close(est[2].cpu().numpy()[j].reshape(1, -1), g["Pk_km1"][t].reshape(1, -1), rtol=1e-8, floor=1e-10, what=f"Pk_km1 t={t}")
      assert_close(est[1].cpu().numpy()[j], g["xk_k"][t], rtol=1e-8, floor=1e-10, what=f"xk_k t={t}")
      assert_close(est[3].cpu().numpy()[j].reshape(1, -1), g["Pk_k"][t].reshape(1, -1), rtol=1e-7, floor=1e-9, what=f"Pk_k t={t}")
      assert_close(f.state()[j], g["x_after"][t], rtol=1e-8, floor=1e-10, what=f"x after augment t={t}")
      assert_close(f.covs()[j].reshape(1, -1), g["P_after"][t].reshape(1, -1), rtol=1e-7, floor=1e-9, what=f"P after augment t={t}")
  assert f.get_augment_times()[-1] == pytest.approx(float(g["ts"][np.where(g["augment"])[0][-1]]))


def test_rank_deficient_projection_is_flagged_and_ignored(env):
  """A landmark on the optical axis of every window position (ray = (0, 0, z)) still gives a rank-3 Hea; a degenerate one
  needs zero rows: scale the problem so that Hea vanishes -> the update must be skipped, flag bit 4 set (the reference's
  numpy path ignores such measurements, ekf_sym.py:589-591)."""
  torch, gen, FK = env
  n = 3
  f = _filter(env, n)
  x0 = np.tile(FK.initial_x, (n, 1))
  P0 = np.tile(np.diag(FK.initial_P_diag), (n, 1, 1))
  f.init_state(x0, P0, 0.0)
  far = np.array([[1.0, 1.0, 1e200]] * n)             # rays ~ (0, 0, 1e200): d h / d landmark underflows to zero
  y = f.update(2, np.zeros((n, 6)), FK.obs_noise[2], extra_args=far)
  torch.cuda.synchronize()
  assert (f.flags.cpu().numpy() & 4).all()
  assert np.array_equal(f.state(), x0) and np.array_equal(f.covs(), P0)
  assert not np.abs(y.cpu().numpy()[:, :3]).any()


def test_scalar_abi_feature_update(env):
  """The drop-in host-pointer entry point {name}_update_{kind}(x, P, z, R, ea) of the reference, batch of one."""
  torch, gen, FK = env
  from rednose_amd.helpers.ekf_sym import EKF_sym
  g = _gold(env)
  f = EKF_sym(gen, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), 6, 6, **FK.filter_kwargs())
  assert f.feature_track_kinds == [2]
  for i in range(3):
    x, P, z = g["upd_x_in"][i].copy(), g["upd_P_in"][i].copy(), g["upd_z"][i].copy()
    f._updates[2](x, P, z, FK.obs_noise[2].copy(), g["upd_ea"][i].copy())       # pylint: disable=protected-access
    assert_close(x, g["upd_x"][i], rtol=1e-9, floor=1e-11)
    assert_close(P, g["upd_P"][i], rtol=1e-8, floor=1e-10)
  He = np.zeros((6, 3))
  f.Hes[2](g["upd_x_in"][0].copy(), g["upd_ea"][0].copy(), He)
  from oracle_lib import OracleLib
  want = np.zeros(18); OracleLib(FK.name).call("He_2", g["upd_x_in"][0].copy(), g["upd_ea"][0].copy(), want)
  assert_close(He.reshape(-1), want)


def test_unsupported_entry_points_fail_loudly(env):
  """A feature-track kind without its extra arguments must raise, not mis-compute."""
  torch, gen, FK = env
  from rednose_amd.helpers import KalmanError
  f = _filter(env, 4)
  with pytest.raises(KalmanError):          # the landmark is missing
    f.run(np.array([0.1]), np.array([2], dtype=np.int32), np.zeros((1, 4, 6)), {2: FK.obs_noise[2]})
  with pytest.raises(KalmanError):
    f.update(2, np.zeros((4, 6)), FK.obs_noise[2])              # extra arguments missing


@pytest.mark.parametrize("inplace", [False, True])
def test_smoother_main_block_vs_reference(env, inplace):
  """EKF_sym.rts_smooth of the reference on the MSCKF trajectory (window shifts included): only the main block of the
  covariance and the main states are smoothed, the window part of every estimate passes through
  (/root/reference/rednose/helpers/ekf_sym.py:675-686).  The predicted pair of the last step is handed over, as the
  reference takes it from the last estimate."""
  torch, gen, FK = env
  g = _gold(env)
  n = 7
  f = _filter(env, n)
  T = len(g["ts"])
  D = FK.dim_state
  xf = np.tile(g["xk_k"][:, None, :], (1, n, 1)); Pf = np.tile(g["Pk_k"][:, None], (1, n, 1, 1))
  last = (np.tile(g["xk_km1"][-1], (n, 1)), np.tile(g["Pk_km1"][-1], (n, 1, 1)))
  if inplace:
    xf, Pf = torch.as_tensor(xf, device=f.device), torch.as_tensor(Pf, device=f.device)
  xs, Ps = f.rts_smooth(xf, Pf, g["ts"], norm_quats=False, inplace=inplace, last_predicted=last)
  torch.cuda.synchronize()
  X, P = xs.cpu().numpy(), Ps.cpu().numpy()
  for j in (0, n - 1):
    assert_close(X[:, j], g["xs_smooth"], rtol=1e-8, floor=1e-10, what="MSCKF smoothed states")
    assert_close(P[:, j].reshape(T, -1), g["Ps_smooth"].reshape(T, -1), rtol=1e-7, floor=1e-9, what="MSCKF smoothed covariances")
  # what is not smoothed is the filtered estimate, bit for bit
  d1 = 6
  assert np.array_equal(X[:-1, 0, d1:], g["xk_k"][:-1, d1:])
  assert np.array_equal(P[:-1, 0, d1:, :], g["Pk_k"][:-1, d1:, :]) and np.array_equal(P[:-1, 0, :, d1:], g["Pk_k"][:-1, :, d1:])


def test_fused_run_with_landmarks_and_window_shifts(env):
  """The whole MSCKF stream of the reference's numpy path -- POSITION fixes followed by a window shift, FEATURE tracks with their
  per-observation landmark -- in ONE {name}_batch_run launch: extra arguments per filter and step, augment flags in the
  schedule.  Filtered trace (the estimate before each shift, like the reference's Estimate), final state after the last shift,
  projected residual norms.  Both layouts of the fused run: several rows per lane (15 states), one filter per wavefront (36)."""
  torch, gen, FK = env
  from rednose_amd.helpers import KalmanError
  g = _gold(env)
  n = 9
  f = _filter(env, n)
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  T = len(kinds)
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  eas = np.tile(g["eas"][:, None, :], (1, n, 1))
  Rs = {1: FK.obs_noise[1], 2: FK.obs_noise[2]}
  # (round 3: above 32 error states the fused run gives a filter the whole wavefront, one row of P per lane -- feature36 runs
  # through batch_run like the 15-state model)
  ys, tx, tP, fl = f.run(ts, kinds, zs.copy(), Rs, trace=True, flags=True, extra_args=eas, augment=g["augment"])
  torch.cuda.synchronize()
  X, P, Y = tx.cpu().numpy(), tP.cpu().numpy(), ys.cpu().numpy()
  assert not fl.cpu().numpy().any()
  for j in (0, n - 1):
    assert_close(X[:, j], g["xk_k"], rtol=1e-8, floor=1e-10, what="fused MSCKF run, filtered states")
    assert_close(P[:, j].reshape(T, -1), g["Pk_k"].reshape(T, -1), rtol=1e-7, floor=1e-9, what="fused MSCKF run, filtered covariances")
    assert_close(f.state()[j], g["x_after"][-1], rtol=1e-8, floor=1e-10, what="state after the last window shift")
    assert_close(f.covs()[j].reshape(1, -1), g["P_after"][-1].reshape(1, -1), rtol=1e-7, floor=1e-9)
  feat = kinds == 2
  assert_close(Y[~feat, 0, :3], g["ys"][~feat, :3], rtol=1e-8, atol=1e-10)
  assert f.get_augment_times()[-1] == float(ts[np.where(g["augment"])[0][-1]])
  # and the same schedule step by step gives the same estimates
  s = _filter(env, n)
  for t in range(T):
    ys_t = s.predict_and_update_batch(float(ts[t]), int(kinds[t]), zs[t, :, :FK.obs_noise[int(kinds[t])].shape[0]].copy(), Rs[int(kinds[t])],
                                      extra_args=eas[t] if kinds[t] == 2 else None, augment=bool(g["augment"][t]))
    if kinds[t] == 2:      # the feature tracks' residuals: the fused run's in the same (reference's, ekf_c.c:71) basis as the step kernels',
      #                      which test_both_kinds_vs_oracle_strict compares entry for entry with the oracle's
      assert_close(Y[t][:, :3], ys_t.cpu().numpy()[:, :3], rtol=1e-8, atol=1e-10, what=f"projected residual of step {t}, fused run vs step kernel")
  torch.cuda.synchronize()
  assert_close(s.state(), f.state(), rtol=1e-9, floor=1e-11)
  assert_close(s.covs().reshape(n, -1), f.covs().reshape(n, -1), rtol=1e-8, floor=1e-10)


def test_kalmanfilter_stream_with_landmarks_and_window_shifts(env):
  """KalmanFilter.predict_and_observe_stream with extra_args / augment through the fused run, against the reference's numpy
  stream (the per-step fallback for libraries without batch_run is covered by tests/test_gpu_random.py: 56 states)."""
  torch, gen, FK = env
  from rednose_amd.helpers.kalmanfilter import KalmanFilter
  g = _gold(env)
  n = 5
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]

  class KF(KalmanFilter):
    name = FK.name
    obs_noise = FK.obs_noise
  kf = KF()
  kf.filter = _filter(env, n)
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  eas = np.tile(g["eas"][:, None, :], (1, n, 1))
  res = kf.predict_and_observe_stream(ts, kinds, zs.copy(), extra_args=eas, augment=g["augment"])
  torch.cuda.synchronize()
  assert not isinstance(res, list), "both models have a fused run since round 3"
  for j in (0, n - 1):
    assert_close(kf.x[j], g["x_after"][-1], rtol=1e-8, floor=1e-10, what="stream API, state after the last window shift")
    assert_close(kf.P[j].reshape(1, -1), g["P_after"][-1].reshape(1, -1), rtol=1e-7, floor=1e-9)


@pytest.mark.parametrize("passes", [1, 2])
def test_smooth_shifts_the_augment_times_once(env, passes):
  """smooth(augment=...) over the whole batch: the window shifts of the schedule are recorded in get_augment_times() ONCE, whatever
  the number of passes (round-3 advice: run() shifted them once per pass and smooth() once more)."""
  torch, gen, FK = env
  g = _gold(env)
  n = 4
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  eas = np.tile(g["eas"][:, None, :], (1, n, 1))
  Rs = {1: FK.obs_noise[1], 2: FK.obs_noise[2]}
  r = _filter(env, n)
  before = list(r.get_augment_times())
  r.run(ts, kinds, zs.copy(), Rs, extra_args=eas, augment=g["augment"])
  want = list(r.get_augment_times())
  f = _filter(env, n)
  assert list(f.get_augment_times()) == before
  f.smooth(ts, kinds, zs.copy(), Rs, passes=passes, extra_args=eas, augment=g["augment"], norm_quats=False)
  torch.cuda.synchronize()
  assert list(f.get_augment_times()) == want and want != before
