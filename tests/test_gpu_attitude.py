"""GPU parity of a small error-state filter with a quaternion (examples/attitude_kf.py: 7 states, 6 error states) -- the
lane-per-filter kernels with dim_x != dim_err, a state-dependent H_mod, multiplicative error injection and quaternion
renormalisation.  Goldens: the reference's numpy path with the C++ orchestration order (tests/golden/attitude_stream.npz);
random batches against the oracle."""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  from examples.attitude_kf import AttitudeKalman
  return torch, ensure_generated(["attitude"]), AttitudeKalman


def _filter(env, n):
  torch, gen, AK = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  return BatchedEKF(gen, "attitude", AK.Q, AK.initial_x, np.diag(AK.initial_P_diag), 7, 6, batch=n, quaternion_idxs=[0])


@pytest.mark.parametrize("n", [1, 65, 700])
def test_single_calls_vs_oracle_strict(env, n):
  torch, gen, AK = env
  from oracle_lib import OracleLib
  o = OracleLib("attitude")
  rng = np.random.default_rng(n)
  q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
  x0 = np.concatenate([q, rng.normal(size=(n, 3)) * 0.5], axis=1)
  A = rng.normal(size=(n, 6, 6)) * 0.2
  P0 = np.diag(AK.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)
  f = _filter(env, n)
  for k in (1, 2):
    R = AK.obs_noise[k]
    for fused in (True, False):
      z = rng.normal(size=(n, 3)) * (0.5 if k == 1 else 5.0)
      f.init_state(x0, P0, 0.0)
      xr, Pr, zr = x0.copy(), P0.copy(), z.copy()
      o.batch_step(k, xr, Pr, zr, R, AK.Q, 0.02, quat_idx=0)
      if fused:
        y = f.predict_and_update_batch(0.02, k, z.copy(), R)
      else:
        f.predict(0.02)
        y = f.update(k, z.copy(), R)
      torch.cuda.synchronize()
      what = f"kind {k} n={n} fused={fused}"
      assert_close(f.state(), xr, rtol=1e-11, floor=1e-13, what=what + " x")
      assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-11, floor=1e-13, what=what + " P")
      assert_close(y.cpu().numpy(), zr, rtol=1e-11, atol=1e-13 * np.abs(z).max(), what=what + " y")
      assert np.abs(np.linalg.norm(f.state()[:, :4], axis=1) - 1).max() < 1e-14


def test_stream_run_and_smoother_vs_reference(env):
  torch, gen, AK = env
  g = golden("attitude_stream.npz")
  n = 9
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  T = len(kinds)
  Rs = {1: AK.obs_noise[1], 2: AK.obs_noise[2]}
  s = _filter(env, n); s.init_state(g["x0"], g["P0"], None)
  for t in range(T):
    y = s.predict_and_update_batch(float(ts[t]), int(kinds[t]), np.tile(g["zs"][t], (n, 1)), Rs[int(kinds[t])])
    for j in (0, n - 1):
      assert_close(s.state()[j], g["xs"][t], rtol=1e-9, floor=1e-11, what=f"stream x t={t}")
      assert_close(s.covs()[j].reshape(1, -1), g["Ps"][t].reshape(1, -1), rtol=1e-8, floor=1e-10, what=f"stream P t={t}")
      assert_close(y.cpu().numpy()[j], g["ys"][t], rtol=1e-8, atol=1e-10, what=f"stream y t={t}")
  f = _filter(env, n); f.init_state(g["x0"], g["P0"], None)
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  ys, tx, tP, _ = f.run(ts, kinds, zs, Rs, trace=True)
  torch.cuda.synchronize()
  X, P = tx.cpu().numpy(), tP.cpu().numpy()
  for j in (0, n - 1):
    assert_close(X[:, j], g["xs"], rtol=1e-9, floor=1e-11, what="fused run x")
    assert_close(P[:, j].reshape(T, -1), g["Ps"].reshape(T, -1), rtol=1e-8, floor=1e-10, what="fused run P")
  xs, Ps = f.rts_smooth(tx, tP, ts, norm_quats=False)
  torch.cuda.synchronize()
  xs, Ps = xs.cpu().numpy(), Ps.cpu().numpy()
  for j in (0, n - 1):
    assert_close(xs[:, j], g["xs_smooth"], rtol=1e-7, floor=1e-9, what="smoothed x")
    assert_close(Ps[:, j].reshape(T, -1), g["Ps_smooth"].reshape(T, -1), rtol=1e-6, floor=1e-8, what="smoothed P")
