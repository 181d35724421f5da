"""Asymmetric covariances through EVERY entry point (include/rednose_amd_filter.h, "Asymmetric covariances").

The reference never symmetrises P (ekf_c.c:24,101,115 multiply with both halves, S is solved as a general matrix), so a P with a
skew part is a legal input and every entry point has to say what it does with one:
  * step-granular entry points (batch_predict, batch_update_k, batch_predict_update_k, the _masked twins, batch_maha_k): the
    reference's result on the asymmetric matrix, entry for entry (oracle on identical inputs, 1e-10 of the row maximum);
  * batch_run (with and without the trace): the reference's result on (P + P^T) / 2;
  * batch_rts: gain and correction from the LOWER triangles of the covariances it is given, Ps[k] = Pf[k] + correction.
Three models = the three kernel structures: kinematic6 (lane per filter), rand24 (lane groups, rows in registers), live (lane groups,
register-lean, error-state with a quaternion).  The skew part is 1e-4 of sqrt(P_ii P_jj): four to six orders of magnitude above the
tolerances, so an entry point that quietly read one triangle -- or symmetrised where it must not -- fails.
"""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu

MODELS = ("kinematic6", "rand24", "live")
SKEW = 1e-4


def _setup(name, n, seed):
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  from oracle_lib import OracleLib
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  rng = np.random.default_rng(seed)
  gen = ensure_generated([name])
  if name == "live":
    from examples.live_kf import LiveKalman as M
    D, E, quat = 23, 22, 3
    g = golden("live_single_steps.npz")
    idx = rng.integers(0, g["x_in"].shape[0], size=n)
    x0 = g["x_in"][idx] + rng.normal(size=(n, 23)) * 1e-3
    x0[:, 3:7] /= np.linalg.norm(x0[:, 3:7], axis=1, keepdims=True)
    P0 = g["P_in"][idx] * rng.uniform(0.5, 2.0, size=(n, 1, 1))
    kinds = (4, 10, 12)
    f = BatchedEKF(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, quaternion_idxs=[3])
  else:
    if name == "kinematic6":
      from examples.kinematic6_kf import Kinematic6Kalman as M
      kinds = (1,)
    else:
      import examples.random_kf as R
      M = R.Random24Kalman
      kinds = (1, 2, 3)
    D = E = M.initial_x.shape[0]
    quat = -1
    x0 = M.initial_x[None] + rng.normal(size=(n, D)) * 0.3
    A = rng.normal(size=(n, E, E)) * 0.2
    P0 = np.diag(M.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)
    f = BatchedEKF(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n)
  P0 = 0.5 * (P0 + P0.transpose(0, 2, 1))
  return torch, M, f, OracleLib(name), rng, x0, P0, kinds, quat


def _skewed(P, rng, eps=SKEW):
  """P + eps * sqrt(P_ii P_jj) * (W - W^T): symmetric part unchanged, a skew part well above every tolerance of this file."""
  W = rng.normal(size=P.shape)
  dg = np.sqrt(np.einsum("...ii->...i", P))
  return P + eps * dg[..., :, None] * dg[..., None, :] * (W - np.swapaxes(W, -1, -2))


def _sym(P):
  return 0.5 * (P + np.swapaxes(P, -1, -2))


def _lower(P):
  L = np.tril(P)
  return L + np.swapaxes(np.tril(P, -1), -1, -2)


def _observations(o, M, rng, x0, kind, quat):
  Z = o.zdim(kind)
  hx = np.zeros((x0.shape[0], Z))
  for i in range(x0.shape[0]):
    o.call(f"h_{kind}", x0[i].copy(), np.zeros(4), hx[i])
  return hx + rng.normal(size=hx.shape) * np.sqrt(np.diag(np.atleast_2d(M.obs_noise[kind])))[None]


def _rel(a, b):
  return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", MODELS)
def test_step_granular_entry_points_follow_the_reference_on_asymmetric_P(name):
  n = 70
  torch, M, f, o, rng, x0, P0, kinds, quat = _setup(name, n, 11)
  Pa = _skewed(P0, rng)
  assert _rel(Pa, np.swapaxes(Pa, -1, -2)) > 1e-5
  E = P0.shape[-1]
  # batch_predict
  f.init_state(x0, Pa, 0.0)
  f.predict_dt(0.013)
  torch.cuda.synchronize()
  xr, Pr = x0.copy(), Pa.copy()
  for i in range(n):
    o.predict(xr[i], Pr[i], M.Q, 0.013)
  if quat >= 0:
    xr[:, quat:quat + 4] /= np.linalg.norm(xr[:, quat:quat + 4], axis=1, keepdims=True)
  assert_close(f.state(), xr, rtol=1e-10, floor=1e-12, what=f"{name} batch_predict x")
  assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-10, floor=1e-12, what=f"{name} batch_predict P")
  assert _rel(f.covs(), _sym(Pr)) > 1e-6, "the predicted covariance must have kept its skew part (the reference does)"
  for k in kinds:
    R = np.atleast_2d(M.obs_noise[k])
    z = _observations(o, M, rng, x0, k, quat)
    # batch_update_k, batch_predict_update_k (dt > 0 and dt = 0), and the masked twin on half of the filters
    for what, dt, fused in ((f"{name} update_{k}", None, False), (f"{name} predict_update_{k}", 0.01, True), (f"{name} predict_update_{k} dt=0", 0.0, True)):
      f.init_state(x0, Pa, 0.0)
      xr, Pr, zr = x0.copy(), Pa.copy(), z.copy()
      if fused:
        y = f.predict_and_update_batch(dt, k, z.copy(), R)
        o.batch_step(k, xr, Pr, zr, R, M.Q, dt, quat_idx=quat)
      else:
        y = f.update(k, z.copy(), R)
        o.batch_step(k, xr, Pr, zr, R, M.Q, 0.0, quat_idx=quat, do_predict=False)
      torch.cuda.synchronize()
      assert_close(f.state(), xr, rtol=1e-10, floor=1e-10, what=what + " x")
      assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-10, floor=1e-10, what=what + " P")
      assert_close(y.cpu().numpy(), zr, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(z).max()), what=what + " y")
      # the same call on the symmetrised matrix is a DIFFERENT answer: the test separates the two
      xs_, Ps_, zs_ = x0.copy(), _sym(Pa), z.copy()
      o.batch_step(k, xs_, Ps_, zs_, R, M.Q, 0.0 if dt is None else dt, quat_idx=quat, do_predict=fused)
      assert _rel(Pr, Ps_) > 1e-7
    act = (np.arange(n) % 2 == 0)
    fm = _setup(name, n, 11)[2]              # (a fresh orchestrator: masks switch it to per-filter timelines for good)
    fm.init_state(x0, Pa, 0.0)
    fm.predict_and_update_batch(np.full(n, 0.01), k, z.copy(), R, active=act)
    torch.cuda.synchronize()
    xr, Pr, zr = x0.copy(), Pa.copy(), z.copy()
    o.batch_step(k, xr, Pr, zr, R, M.Q, 0.01, quat_idx=quat)
    assert_close(fm.state()[act], xr[act], rtol=1e-10, floor=1e-10, what=f"{name} masked predict_update_{k} x")
    assert_close(fm.covs()[act].reshape(int(act.sum()), -1), Pr[act].reshape(int(act.sum()), -1), rtol=1e-10, floor=1e-10, what=f"{name} masked predict_update_{k} P")
    assert np.array_equal(fm.covs()[~act], Pa[~act]) and np.array_equal(fm.state()[~act], x0[~act]), "masked-out filters pass through bit for bit"
    # batch_maha_k: y^T (He P He^T + R)^-1 y with the asymmetric S solved as a general matrix
    f2 = _setup(name, n, 11)[2]
    f2.init_state(x0, Pa, 0.0)
    d2 = f2.maha_dist(k, z.copy(), R).cpu().numpy()
    Z = o.zdim(k)
    for i in (0, n // 2, n - 1):
      H = np.zeros(Z * x0.shape[1]); Hm = np.zeros(x0.shape[1] * E); hx = np.zeros(Z)
      o.call(f"H_{k}", x0[i].copy(), np.zeros(4), H); o.call("H_mod_fun", x0[i].copy(), Hm); o.call(f"h_{k}", x0[i].copy(), np.zeros(4), hx)
      He = H.reshape(Z, -1) @ Hm.reshape(-1, E)
      yv = z[i] - hx
      want = yv @ np.linalg.solve(He @ Pa[i] @ He.T + R, yv)
      assert abs(d2[i] - want) <= 1e-9 * max(1.0, abs(want)), f"{name} maha_{k} filter {i}: {d2[i]} vs {want}"


def _schedule(M, o, rng, x0, kinds, quat, T):
  n = x0.shape[0]
  zmax = max(o.zdim(k) for k in kinds)
  sched = np.array([kinds[t % len(kinds)] for t in range(T)], dtype=np.int32)
  ts = np.cumsum(np.where(np.arange(T) % 3 == 1, 0.0, 0.01)) + 0.01
  zs = np.zeros((T, n, zmax))
  Rt = np.zeros((T, zmax * zmax))
  for t, k in enumerate(sched):
    Z = o.zdim(int(k))
    zs[t, :, :Z] = _observations(o, M, rng, x0, int(k), quat)
    Rk = np.atleast_2d(M.obs_noise[int(k)])
    Rt[t, :Z * Z] = Rk.reshape(-1)
  return sched, ts, zs, Rt


@pytest.mark.parametrize("trace", [False, True], ids=["no_trace", "trace"])
@pytest.mark.parametrize("name", MODELS)
def test_fused_run_is_the_reference_on_the_symmetric_part(name, trace):
  n, T = 45, 9
  torch, M, f, o, rng, x0, P0, kinds, quat = _setup(name, n, 23)
  Pa = _skewed(P0, rng)
  sched, ts, zs, Rt = _schedule(M, o, rng, x0, kinds, quat, T)
  Rs = {int(k): np.atleast_2d(M.obs_noise[int(k)]) for k in kinds}
  f.init_state(x0, Pa, 0.0)
  ys, tx, tP, _ = f.run(ts, sched, zs.copy(), Rs, trace=trace)
  torch.cuda.synchronize()
  dts = np.diff(np.concatenate([[0.0], ts]))
  xr, Pr, zr = x0.copy(), _sym(Pa), zs.copy()
  xf, Pf = np.zeros((T, n, x0.shape[1])), np.zeros((T, n) + P0.shape[1:])
  o.batch_run(sched, dts, xr, Pr, zr, Rt, M.Q, quat_idx=quat, xf=xf, Pf=Pf)
  assert_close(f.state(), xr, rtol=1e-8, floor=1e-9, what=f"{name} run x")
  assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-8, floor=1e-9, what=f"{name} run P")
  if trace:
    assert_close(tx.cpu().numpy().reshape(T * n, -1), xf.reshape(T * n, -1), rtol=1e-8, floor=1e-9, what=f"{name} run trace x")
    assert_close(tP.cpu().numpy().reshape(T * n, -1), Pf.reshape(T * n, -1), rtol=1e-8, floor=1e-9, what=f"{name} run trace P")
    tPh = tP.cpu().numpy()
    assert _rel(tPh, np.swapaxes(tPh, -1, -2)) < 1e-9, "the trace comes back symmetric to rounding"
  # against the reference on the asymmetric matrix itself the difference is first order in the skew part -- and visible
  xa, Pa_, za = x0.copy(), Pa.copy(), zs.copy()
  o.batch_run(sched, dts, xa, Pa_, za, Rt, M.Q, quat_idx=quat)
  d = _rel(f.covs(), Pa_)
  assert 1e-9 < d < 1e3 * SKEW, f"{name}: distance to the run on the asymmetric matrix {d:.2e}"


@pytest.mark.parametrize("name", MODELS)
def test_run_exact_is_the_reference_on_the_asymmetric_matrix_itself(name):
  """run(exact=True) walks the schedule with the step-granular kernels (both halves of P, S as a general matrix): the reference's own
  result for a T-step schedule on a covariance with a skew part -- ekf_c.c:24,100-101,115 never symmetrise -- trace and flags included."""
  n, T = 45, 9
  torch, M, f, o, rng, x0, P0, kinds, quat = _setup(name, n, 29)
  Pa = _skewed(P0, rng)
  sched, ts, zs, Rt = _schedule(M, o, rng, x0, kinds, quat, T)
  Rs = {int(k): np.atleast_2d(M.obs_noise[int(k)]) for k in kinds}
  f.init_state(x0, Pa, 0.0)
  ys, tx, tP, fl = f.run(ts, sched, zs.copy(), Rs, trace=True, flags=True, exact=True)
  torch.cuda.synchronize()
  dts = np.diff(np.concatenate([[0.0], ts]))
  xa, Pa_, za = x0.copy(), Pa.copy(), zs.copy()
  xf, Pf = np.zeros((T, n, x0.shape[1])), np.zeros((T, n) + P0.shape[1:])
  fr = np.zeros((T, n), dtype=np.uint8)
  o.batch_run(sched, dts, xa, Pa_, za, Rt, M.Q, quat_idx=quat, flags=fr, xf=xf, Pf=Pf)
  assert_close(f.state(), xa, rtol=1e-10, floor=1e-11, what=f"{name} run(exact) x")
  assert_close(f.covs().reshape(n, -1), Pa_.reshape(n, -1), rtol=1e-10, floor=1e-10, what=f"{name} run(exact) P")
  assert_close(tx.cpu().numpy().reshape(T * n, -1), xf.reshape(T * n, -1), rtol=1e-10, floor=1e-11, what=f"{name} run(exact) trace x")
  assert_close(tP.cpu().numpy().reshape(T * n, -1), Pf.reshape(T * n, -1), rtol=1e-10, floor=1e-10, what=f"{name} run(exact) trace P")
  assert np.array_equal(fl.cpu().numpy() & 1, fr & 1)
  yh = ys.cpu().numpy()
  for t, k in enumerate(sched):
    Z = o.zdim(int(k))
    assert_close(yh[t][:, :Z], za[t][:, :Z], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(zs).max()), what=f"{name} run(exact) y[{t}]")
  # and it is NOT the symmetrised run: the two differ by the first-order effect of the skew part
  assert _rel(f.covs(), _sym(Pa)) > 0 and _rel(Pa_, f.covs()) < 1e-9


def _numpy_backward_step(o, M, name, quat, Xk, Pk_raw, dt, x1n, P1n_raw, newest, oldest):
  """One step of ekf_sym.py:651-690 under the contract of batch_rts: gain and correction from the lower triangles."""
  D, E = Xk.shape[0], Pk_raw.shape[0]
  Pk = _lower(Pk_raw)
  x1k = np.zeros(D); Fk = np.zeros(E * E)
  o.call("f_fun", Xk.copy(), float(dt), x1k); o.call("F_fun", Xk.copy(), float(dt), Fk)
  if quat >= 0:
    x1k[quat:quat + 4] /= np.linalg.norm(x1k[quat:quat + 4])
  Fk = Fk.reshape(E, E)
  P1k = Fk @ Pk @ Fk.T + dt * M.Q
  if newest:
    return x1k, P1k, None, None
  Ck = np.linalg.solve(P1k, Fk @ Pk.T).T
  if quat >= 0:
    delta = np.zeros(E); xkn = np.zeros(D)
    o.call("inv_err_fun", x1k.copy(), x1n.copy(), delta)
    o.call("err_fun", Xk.copy(), Ck @ delta, xkn)
    if not oldest:
      xkn[quat:quat + 4] /= np.linalg.norm(xkn[quat:quat + 4])
  else:
    xkn = Xk + Ck @ (x1n - x1k)
  Pkn = Pk_raw + Ck @ (_lower(P1n_raw) - P1k) @ Ck.T
  return x1k, P1k, xkn, Pkn


@pytest.mark.parametrize("with_last", [False, True], ids=["recomputed_last", "given_last"])
@pytest.mark.parametrize("name", MODELS)
def test_smoother_reads_lower_triangles(name, with_last):
  """batch_rts on an ASYMMETRIC trace (and, optionally, an asymmetric predicted pair of the last step), every backward step against
  the numpy restatement under the stated contract, restarted from the GPU's own estimate of step k + 1."""
  n, T = 19, 8
  torch, M, f, o, rng, x0, P0, kinds, quat = _setup(name, n, 37)
  sched, ts, zs, Rt = _schedule(M, o, rng, x0, kinds, quat, T)
  ts = np.cumsum(np.full(T, 0.01))             # every step advances time: the predicted pairs differ from the filtered ones
  Rs = {int(k): np.atleast_2d(M.obs_noise[int(k)]) for k in kinds}
  f.init_state(x0, P0, 0.0)
  _, tx, tP, _ = f.run(ts, sched, zs.copy(), Rs, trace=True)
  torch.cuda.synchronize()
  X = tx.cpu().numpy()
  Pf = _skewed(_sym(tP.cpu().numpy()), rng)        # what the smoother is handed: every filtered covariance with a skew part
  last = None
  if with_last:
    xl = np.zeros_like(X[0]); Pl = np.zeros_like(Pf[0])
    for j in range(n):
      x1k, P1k, _, _ = _numpy_backward_step(o, M, name, quat, X[T - 2, j], Pf[T - 2, j], ts[T - 1] - ts[T - 2], None, None, True, False)
      xl[j], Pl[j] = x1k, P1k
    Pl = _skewed(_sym(Pl), rng)
    last = (xl, Pl)
  xs, Ps = f.rts_smooth(X.copy(), Pf.copy(), ts, last_predicted=last)
  torch.cuda.synchronize()
  Xs, Pss = xs.cpu().numpy(), Ps.cpu().numpy()
  assert np.isfinite(Xs).all() and np.isfinite(Pss).all()
  for j in range(n):
    x1k, P1k, _, _ = _numpy_backward_step(o, M, name, quat, X[T - 2, j], Pf[T - 2, j], ts[T - 1] - ts[T - 2], None, None, True, False)
    if with_last:       # the newest estimate is the given pair, verbatim (ekf_sym.py:658-659), the state after its renormalisation
      want_x = last[0][j].copy()
      if quat >= 0:
        want_x[quat:quat + 4] /= np.linalg.norm(want_x[quat:quat + 4])
      assert_close(Xs[T - 1, j], want_x, rtol=1e-12, floor=1e-13, what=f"{name} newest state")
      assert np.array_equal(Pss[T - 1, j], last[1][j]), f"{name}: the given predicted covariance passes through bit for bit"
    else:
      assert_close(Xs[T - 1, j], x1k, rtol=1e-10, floor=1e-12, what=f"{name} newest state")
      assert_close(Pss[T - 1, j].reshape(1, -1), P1k.reshape(1, -1), rtol=1e-9, floor=1e-10, what=f"{name} newest covariance")
    for k in range(T - 2, -1, -1):
      _, P1k, xkn, Pkn = _numpy_backward_step(o, M, name, quat, X[k, j], Pf[k, j], ts[k + 1] - ts[k], Xs[k + 1, j], Pss[k + 1, j], False, k == 0)
      bound = max(1e-9, 50 * np.linalg.cond(P1k) * 2.2e-16)
      ex = _rel(Xs[k, j], xkn)
      eP = np.abs(Pss[k, j] - Pkn).max() / np.abs(Pkn).max()
      assert ex <= bound and eP <= bound, f"{name} filter {j} backward step {k}: x {ex:.2e} P {eP:.2e} bound {bound:.2e}"
      # ... and the contract is visible: the same step from the symmetrised inputs is a different matrix
      _, _, _, Pkn_sym = _numpy_backward_step(o, M, name, quat, X[k, j], _sym(Pf[k, j]), ts[k + 1] - ts[k], Xs[k + 1, j], _sym(Pss[k + 1, j]), False, k == 0)
    assert np.abs(Pkn_sym - Pkn).max() / np.abs(Pkn).max() > 1e-7
