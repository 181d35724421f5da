"""n observations per call in the batched orchestrators: the reference's predict_and_update_batch predicts ONCE,
applies the n observations of the call one after the other, writes ONE checkpoint and returns y as a list of n
(/root/reference/rednose/helpers/ekf_sym.py:484-531, ekf_sym.cc:158-194).  BatchedEKF takes z (N, n, Z), R (Z, Z) | (n, Z, Z) |
(N, n, Z, Z), extra_args (n, EA) | (N, n, EA) and does the same for every filter of the batch.
Golden: tests/golden/multi_obs.npz (oracle/make_golden.py::multi_obs_goldens, the reference class once per filter)."""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  return ensure_generated(["kinematic9", "feature"])


def _k9():
  from examples.kinematic9_kf import Kinematic9Kalman as K9
  return K9


def test_per_filter_logs_with_multi_observation_calls_and_a_late_one_each(gen):
  """8 logs of the 9-state model, 36 calls each: own times, kinds 1 / 2 / 3, 1-3 observations per call with a different noise matrix
  per observation, one LATE multi-observation call per filter: its ring is rewound over 2-4 multi-observation checkpoints, the call is
  applied, and the overtaken calls are replayed with all their observations.  One group of masked launches per (kind, n) present at
  an arrival index; every filter follows the reference instance that was fed its log, residuals of every observation included."""
  import torch
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  K9 = _k9()
  g = golden("multi_obs.npz")
  NB, TB = g["A_t"].shape
  f = BatchedEKF(gen, "kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9, batch=NB, rewind_to_keep=64, per_filter=True)
  groups = 0
  for j in range(TB):
    est_at = j % 6 == 0
    for k in (1, 2, 3):
      Z = K9.obs_noise[k].shape[0]
      for n in (1, 2, 3):
        act = (g["A_kind"][:, j] == k) & (g["A_n"][:, j] == n)
        if not act.any():
          continue
        groups += 1
        z = g["A_z"][:, j, :n, :Z].copy()
        R = g["A_Rscale"][:, j, :n, None, None] * K9.obs_noise[k][None, None]
        ret = f.predict_and_update_batch(g["A_t"][:, j].copy(), k, z, R, active=act, keep_estimate=est_at)
        y = torch.stack(ret[6], 1) if est_at else ret
        assert tuple(y.shape) == (NB, n, Z)
        assert_close(y.cpu().numpy()[act].reshape(-1, Z), g["A_y"][act, j, :n, :Z].reshape(-1, Z), rtol=1e-7, atol=1e-9, what=f"arrival {j} kind {k} n {n} residuals")
        if est_at:
          a = j // 6
          assert_close(ret[0].cpu().numpy()[act], g["A_xk_km1"][act, a], rtol=1e-8, floor=1e-10, what=f"arrival {j} xk_km1")
          assert_close(ret[2].cpu().numpy()[act].reshape(act.sum(), -1), g["A_Pk_km1"][act, a].reshape(act.sum(), -1), rtol=1e-8, floor=1e-10, what=f"arrival {j} Pk_km1")
          assert_close(ret[1].cpu().numpy()[act], g["A_xk_k"][act, a], rtol=1e-8, floor=1e-10, what=f"arrival {j} xk_k")
          assert_close(ret[3].cpu().numpy()[act].reshape(act.sum(), -1), g["A_Pk_k"][act, a].reshape(act.sum(), -1), rtol=1e-8, floor=1e-10, what=f"arrival {j} Pk_k")
          assert_close(ret[7].cpu().numpy()[act].reshape(-1, Z), g["A_z"][act, j, :n, :Z].reshape(-1, Z), rtol=0, atol=0, what="Estimate.z is the observation, not the residual")
    assert_close(f.state(), g["A_x"][:, j], rtol=1e-8, floor=1e-10, what=f"arrival {j} x")
    assert_close(f.covs().reshape(NB, -1), g["A_P"][:, j].reshape(NB, -1), rtol=1e-8, floor=1e-10, what=f"arrival {j} P")
  assert groups > 2 * TB
  assert f._ring["nmax"] == 3      # pylint: disable=protected-access


@pytest.mark.parametrize("filt", [1, 5])
def test_shared_timeline_ring_rewinds_over_multi_observation_checkpoints(gen, filt):
  """ONE log (filter `filt` of the golden) on a batch of 6 identical filters with the shared-timeline ring: the late multi-observation
  call rewinds the whole batch, is applied, and the overtaken calls are replayed -- one checkpoint per CALL, whatever its n."""
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  K9 = _k9()
  g = golden("multi_obs.npz")
  TB, N = g["A_t"].shape[1], 6
  f = BatchedEKF(gen, "kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9, batch=N, rewind_to_keep=64)
  late = int(g["A_late"][filt])
  for j in range(TB):
    k, n = int(g["A_kind"][filt, j]), int(g["A_n"][filt, j])
    Z = K9.obs_noise[k].shape[0]
    z = np.tile(g["A_z"][filt, j, :n, :Z][None], (N, 1, 1))
    R = g["A_Rscale"][filt, j, :n, None, None] * K9.obs_noise[k][None]            # (n, Z, Z): the reference's own argument shape
    before = len(f.rewind_t)
    y = f.predict_and_update_batch(float(g["A_t"][filt, j]), k, z, R)
    assert y is not None and tuple(y.shape) == (N, n, Z)
    if j != late:
      assert len(f.rewind_t) == min(64, before + 1), "one checkpoint per call"
    assert_close(y.cpu().numpy().reshape(N, -1), np.tile(g["A_y"][filt, j, :n, :Z].reshape(1, -1), (N, 1)), rtol=1e-7, atol=1e-9, what=f"call {j} residuals")
    assert_close(f.state(), np.tile(g["A_x"][filt, j], (N, 1)), rtol=1e-8, floor=1e-10, what=f"call {j} x")
    assert_close(f.covs().reshape(N, -1), np.tile(g["A_P"][filt, j].reshape(1, -1), (N, 1)), rtol=1e-8, floor=1e-10, what=f"call {j} P")


_FEATURE_RESIDUALS = {}


@pytest.mark.parametrize("fused", [None, False])
def test_msckf_feature_tracks_four_per_timestamp(gen, fused):
  """The MSCKF example: every third call 3 position fixes + the window shift, the others 4 feature tracks (4 landmarks as extra_args,
  per filter).  fused=None: the n observations of a call are ONE batch_run launch (dts = dt, 0, 0, 0: predict(0) is the identity for this
  model); fused=False: one predict + update launch and n - 1 update launches.  Both against the reference instances, residuals in the
  reference's null-space basis included."""
  import torch
  import examples.feature_kf as F
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  FK = F.FeatureKalman
  g = golden("multi_obs.npz")
  NF, TF = g["B_x"].shape[:2]
  f = BatchedEKF(gen, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), F.DIM_MAIN, F.DIM_MAIN, batch=NF, **FK.filter_kwargs())
  f.multi_obs_fused = fused
  assert f._identity_dt0() and f._has_batch_run()      # pylint: disable=protected-access
  ZF = 2 * len(FK.observed)
  ys_seen = []
  for j in range(TF):
    k, n = int(g["B_kind"][j]), int(g["B_n"][j])
    est_at = j % 4 == 0
    if k == 1:
      ret = f.predict_and_update_batch(float(g["B_t"][j]), 1, g["B_z"][:, j, :3, :3].copy(), FK.obs_noise[1], augment=True, keep_estimate=est_at)
      Zy = 3
    else:
      ret = f.predict_and_update_batch(float(g["B_t"][j]), 2, g["B_z"][:, j, :n].copy(), np.eye(ZF) * 0.01**2, extra_args=g["B_ea"][:, j, :n].copy(),
                                       keep_estimate=est_at)
      Zy = ZF - 3
    y = torch.stack(ret[6], 1) if est_at else ret
    assert tuple(y.shape) == (NF, n, 3 if k == 1 else ZF)
    if k == 1:
      assert_close(y.cpu().numpy().reshape(NF * n, -1), g["B_y"][:, j, :n, :3].reshape(NF * n, -1), rtol=1e-6, atol=1e-8, what=f"call {j} position residuals")
    else:
      # feature tracks: the golden's residuals are in the numpy path's basis of the null space (an SVD's, ekf_sym.py:576-591), ours in the C
      # template's (Eigen's fullPivLu().kernel(), ekf_c.c:71-73; tests/test_gpu_msckf.py compares those entry for entry with the oracle).  What
      # does not depend on the basis is compared here: x and P after the call, below -- every one of the call's n updates feeds them.
      ys_seen.append(y.cpu().numpy()[:, :, :Zy].copy())
    if est_at:
      assert_close(ret[0].cpu().numpy(), g["B_xk_km1"][:, j // 4], rtol=1e-8, floor=1e-10, what=f"call {j} xk_km1")
      assert_close(ret[2].cpu().numpy().reshape(NF, -1), g["B_Pk_km1"][:, j // 4].reshape(NF, -1), rtol=1e-8, floor=1e-10, what=f"call {j} Pk_km1")
    assert_close(f.state(), g["B_x"][:, j], rtol=1e-8, floor=1e-10, what=f"call {j} x")
    assert_close(f.covs().reshape(NF, -1), g["B_P"][:, j].reshape(NF, -1), rtol=1e-8, floor=1e-10, what=f"call {j} P")
  assert tuple(f.flags_obs.shape) == (NF, int(g["B_n"][TF - 1]))
  # the two ways of serving a call give the same residuals (same basis in the fused run and in the step kernels)
  key = "fused" if fused is None else "stepwise"
  _FEATURE_RESIDUALS[key] = ys_seen
  if len(_FEATURE_RESIDUALS) == 2:
    for a, b in zip(_FEATURE_RESIDUALS["fused"], _FEATURE_RESIDUALS["stepwise"]):
      assert_close(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), rtol=1e-7, atol=1e-9, what="feature-track residuals, one batch_run launch vs update launches")


def test_multi_observation_call_equals_sequential_single_calls(gen):
  """z (N, n, Z) in one call == predict_and_update_batch(t, z[:, 0]) followed by update(z[:, i]) for the rest, bit for bit on the
  step-granular path, to rounding on the fused one; shapes the reference accepts for R ((n, Z, Z)) and shared (Z, Z); n = 1 as a 3-D
  array; wrong shapes are refused."""
  import torch
  from rednose_amd.helpers.ekf_sym import BatchedEKF, KalmanError
  K9 = _k9()
  rng = np.random.default_rng(5)
  N, n = 37, 3
  mk = lambda: BatchedEKF(gen, "kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9, batch=N)      # noqa: E731
  z = K9.initial_x[None, None, :3] + rng.normal(size=(N, n, 3)) * 0.1
  Rn = np.stack([K9.obs_noise[1] * s_ for s_ in (1.0, 0.5, 2.0)])
  a = mk()
  a.predict_and_update_batch(0.0, 1, z[:, 0].copy(), Rn[0])
  a.predict_and_update_batch(0.02, 1, z[:, 0].copy(), Rn[0])
  ya = [a.update(1, z[:, i].copy(), Rn[i]) for i in (1, 2)]
  b = mk()
  b.multi_obs_fused = False
  b.predict_and_update_batch(0.0, 1, z[:, 0].copy(), Rn[0])
  yb = b.predict_and_update_batch(0.02, 1, z.copy(), Rn)
  assert np.array_equal(a.state(), b.state()) and np.array_equal(a.covs(), b.covs())
  assert np.array_equal(yb[:, 1].cpu().numpy(), ya[0].cpu().numpy()) and np.array_equal(yb[:, 2].cpu().numpy(), ya[1].cpu().numpy())
  c = mk()
  c.predict_and_update_batch(0.0, 1, z[:, 0].copy(), Rn[0])
  yc = c.predict_and_update_batch(0.02, 1, z.copy(), Rn)             # one batch_run launch
  assert_close(c.state(), a.state(), rtol=1e-11, floor=1e-13, what="fused x")
  assert_close(c.covs().reshape(N, -1), a.covs().reshape(N, -1), rtol=1e-11, floor=1e-13, what="fused P")
  assert_close(yc.cpu().numpy().reshape(N, -1), yb.cpu().numpy().reshape(N, -1), rtol=1e-10, atol=1e-12, what="fused residuals")
  assert np.array_equal(c.flags_obs.cpu().numpy(), np.zeros((N, n), dtype=np.uint8))
  # n = 1 as (N, 1, Z) and a shared (Z, Z) noise
  d, e = mk(), mk()
  y1 = d.predict_and_update_batch(0.0, 1, z[:, :1].copy(), K9.obs_noise[1])
  y0 = e.predict_and_update_batch(0.0, 1, z[:, 0].copy(), K9.obs_noise[1])
  assert tuple(y1.shape) == (N, 1, 3) and np.array_equal(y1[:, 0].cpu().numpy(), y0.cpu().numpy()) and np.array_equal(d.covs(), e.covs())
  with pytest.raises(KalmanError):
    d.predict_and_update_batch(0.1, 1, z.copy(), np.stack([K9.obs_noise[1]] * 2))       # (2, Z, Z) for n = 3
  with pytest.raises(KalmanError):
    d.predict_and_update_batch(0.1, 1, z[:, :, :2].copy(), K9.obs_noise[1])              # wrong Z
  torch.cuda.synchronize()


def test_corner_shapes_of_a_multi_observation_call(gen):
  """n = 0 (the reference's loop body never runs: a predict and nothing else), per-filter noise (N, n, Z, Z) on the shared timeline (update
  launches: the fused run takes one noise matrix per step), the same observations for every filter ((1, n, Z)), and the Estimate of the fused
  and the step-granular service of one call."""
  import torch
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  K9 = _k9()
  rng = np.random.default_rng(9)
  N, n = 21, 2
  mk = lambda: BatchedEKF(gen, "kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9, batch=N)      # noqa: E731
  a, b = mk(), mk()
  for f in (a, b):
    f.predict_and_update_batch(0.0, 1, np.tile(K9.initial_x[None, :3], (N, 1)), K9.obs_noise[1])
  y0 = a.predict_and_update_batch(0.03, 1, np.zeros((N, 0, 3)), K9.obs_noise[1])
  b.predict(0.03)
  assert tuple(y0.shape) == (N, 0, 3) and np.array_equal(a.state(), b.state()) and np.array_equal(a.covs(), b.covs()) and a.filter_time == 0.03
  # per-filter noise
  z = K9.initial_x[None, None, :3] + rng.normal(size=(N, n, 3)) * 0.1
  Rf = K9.obs_noise[1][None, None] * rng.uniform(0.5, 2.0, size=(N, n, 1, 1))
  ya = a.predict_and_update_batch(0.05, 1, z.copy(), Rf)
  b.predict_and_update_batch(0.05, 1, z[:, 0].copy(), Rf[:, 0])
  yb1 = b.update(1, z[:, 1].copy(), Rf[:, 1])
  assert np.array_equal(a.state(), b.state()) and np.array_equal(a.covs(), b.covs()) and np.array_equal(ya[:, 1].cpu().numpy(), yb1.cpu().numpy())
  # (1, n, Z): the same observations for every filter
  z1 = K9.initial_x[None, None, :3] + rng.normal(size=(1, n, 3)) * 0.1
  c, d = mk(), mk()
  c.predict_and_update_batch(0.0, 1, z1.copy(), K9.obs_noise[1])
  d.predict_and_update_batch(0.0, 1, np.tile(z1, (N, 1, 1)), K9.obs_noise[1])
  assert np.array_equal(c.state(), d.state()) and np.array_equal(c.covs(), d.covs())
  # Estimate: fused (predict + one batch_run launch) against update launches
  e, f_ = mk(), mk()
  f_.multi_obs_fused = False
  est = [g_.predict_and_update_batch(0.02, 1, z.copy(), K9.obs_noise[1], keep_estimate=True) for g_ in (e, f_)]
  torch.cuda.synchronize()
  assert np.array_equal(est[0][0].cpu().numpy(), est[1][0].cpu().numpy()) and np.array_equal(est[0][2].cpu().numpy(), est[1][2].cpu().numpy())      # predicted pair: the same launch
  assert_close(est[0][1].cpu().numpy(), est[1][1].cpu().numpy(), rtol=1e-11, floor=1e-13, what="xk_k")
  assert_close(est[0][3].cpu().numpy().reshape(N, -1), est[1][3].cpu().numpy().reshape(N, -1), rtol=1e-11, floor=1e-13, what="Pk_k")
  assert len(est[0][6]) == n and tuple(est[0][7].shape) == (N, n, 3) and np.array_equal(est[0][7].cpu().numpy(), z)

