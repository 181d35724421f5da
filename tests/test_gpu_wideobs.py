"""A mid-size lane-group model (10 states) whose first kind is a 9-dimensional observation: more observation entries per
wavefront tile than lanes in the fused run (8 filters x 9 = 72), an 81-entry innovation covariance factored in registers.
The reference puts no limit on ZDIM (rednose/templates/ekf_c.c:37 is templated on it); round 2's fused run asserted
FPW * zmax <= 64 at generation time and made such a library ungeneratable.  Single calls strictly against the oracle, the
fused run (trace included) and the step-granular path against the oracle's run."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  import examples.random_kf as R
  return torch, ensure_generated(["randz10"]), R.RandomWideObs10Kalman


def _filter(env, n):
  _, gen, M = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  return BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), M.dim, M.dim, batch=n)


def _states(M, rng, n):
  x0 = M.initial_x[None] + rng.normal(size=(n, M.dim)) * 0.3
  A = rng.normal(size=(n, M.dim, M.dim)) * 0.2
  return x0, np.diag(M.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)


@pytest.mark.parametrize("n", [1, 9, 130])
def test_single_calls_vs_oracle(env, n):
  torch, _, M = env
  from oracle_lib import OracleLib
  o = OracleLib(M.name)
  assert o.zdim(1) == 9
  rng = np.random.default_rng(90 + n)
  x0, P0 = _states(M, rng, n)
  f = _filter(env, n)
  for k in (1, 2, 3):
    Z = o.zdim(k)
    for fused in (True, False):
      z = rng.normal(size=(n, Z))
      f.init_state(x0, P0, 0.0)
      xr, Pr, zr = x0.copy(), P0.copy(), z.copy()
      o.batch_step(k, xr, Pr, zr, M.obs_noise[k], M.Q, 0.02)
      if fused:
        y = f.predict_and_update_batch(0.02, k, z.copy(), M.obs_noise[k])
      else:
        f.predict(0.02)
        y = f.update(k, z.copy(), M.obs_noise[k])
      torch.cuda.synchronize()
      what = f"{M.name} kind {k} n={n} fused={fused}"
      assert_close(f.state(), xr, rtol=1e-10, floor=1e-12, what=what + " x")
      assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-10, floor=1e-12, what=what + " P")
      assert_close(y.cpu().numpy(), zr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(z).max()), what=what + " y")


@pytest.mark.parametrize("n", [5, 8, 75])
def test_fused_run_with_more_observation_entries_than_lanes(env, n):
  torch, _, M = env
  from oracle_lib import OracleLib
  o = OracleLib(M.name)
  T, zmax = 15, 9
  rng = np.random.default_rng(n)
  x0, P0 = _states(M, rng, n)
  kinds = np.array([(1, 2, 1, 3)[t % 4] for t in range(T)], dtype=np.int32)
  ts = np.cumsum(rng.uniform(0.005, 0.03, size=T))
  zs = rng.normal(size=(T, n, zmax)) * 0.5
  Rs = {k: M.obs_noise[k] for k in (1, 2, 3)}
  f = _filter(env, n); f.init_state(x0, P0, 0.0)
  ys, tx, tP, _ = f.run(ts, kinds, zs.copy(), Rs, trace=True)
  s = _filter(env, n); s.init_state(x0, P0, 0.0)
  for t in range(T):
    Z = Rs[int(kinds[t])].shape[0]
    s.predict_and_update_batch(float(ts[t]), int(kinds[t]), zs[t, :, :Z].copy(), Rs[int(kinds[t])])
  torch.cuda.synchronize()
  xr, Pr, zr = x0.copy(), P0.copy(), zs.copy()
  Rt = np.zeros((T, zmax * zmax))
  for t, k in enumerate(kinds):
    Rt[t, :Rs[int(k)].size] = Rs[int(k)].reshape(-1)
  xf = np.zeros((T, n, M.dim)); Pf = np.zeros((T, n, M.dim, M.dim))
  o.batch_run(kinds, np.diff(np.concatenate([[0.0], ts])), xr, Pr, zr, Rt, M.Q, xf=xf, Pf=Pf)
  for name, got in (("fused run", f), ("step path", s)):
    assert_close(got.state(), xr, rtol=1e-8, floor=1e-10, what=f"{M.name} {name} x")
    assert_close(got.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-8, floor=1e-10, what=f"{M.name} {name} P")
  assert_close(tx.cpu().numpy().reshape(T * n, -1), xf.reshape(T * n, -1), rtol=1e-8, floor=1e-10, what="trace x")
  assert_close(tP.cpu().numpy().reshape(T * n, -1), Pf.reshape(T * n, -1), rtol=1e-8, floor=1e-10, what="trace P")
  # residuals: every one of the 9 entries of kind 1, and the padding of the narrower kinds left untouched
  yr = ys.cpu().numpy()
  for t, k in enumerate(kinds):
    Z = Rs[int(k)].shape[0]
    assert_close(yr[t, :, :Z], zr[t, :, :Z], rtol=1e-8, atol=1e-10, what=f"y[{t}]")
    assert np.array_equal(yr[t, :, Z:], zs[t, :, Z:]), f"step {t}: padding columns of a {Z}-dimensional kind were written"
