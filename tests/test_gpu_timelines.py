"""Per-filter timelines: N filters of one batch behave like N independent instances of the reference's
orchestrator (/root/reference/rednose/helpers/ekf_sym.py:418-482, ekf_sym.cc:83-156) -- own filter_time, own ring of
checkpoints, a late observation rewinds / replays only the filters it is late for, an observation that is too old is ignored for
that filter alone, filters without an observation in a call pass through bit for bit.
Golden: tests/golden/perfilter_timelines.npz, produced by oracle/make_golden.py running the reference class once per filter."""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  return ensure_generated(["kinematic", "kinematic9", "live"])


@pytest.mark.parametrize("copies", [1, 7])
def test_single_kind_logs_with_a_different_swapped_pair_per_filter(gen, copies):
  """12 logs of 700 observations (x `copies`: 84 filters span two wavefront tiles), a different out-of-order pair in each, the
  ring of 512 wraps, one observation 2.5 s late is ignored for its filter only."""
  import torch
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  g = golden("perfilter_timelines.npz")
  NA, T = g["A_t"].shape
  n = NA * copies
  tile = lambda a: np.concatenate([a] * copies, axis=0)      # noqa: E731
  f = BatchedEKF(gen, "kinematic", np.diag([0.1**2, 2.0**2]), np.array([0.5, 0.0]), np.eye(2), 2, 2, batch=n, rewind_to_keep=512, per_filter=True)
  R = np.array([[0.1**2]])
  keep = set(g["A_keep"].tolist())
  for j in range(T):
    y = f.predict_and_update_batch(tile(g["A_t"][:, j]), 1, tile(g["A_z"][:, j:j + 1]).copy(), R)
    assert y is not None
    fl = f.flags.cpu().numpy()
    assert np.array_equal((fl & 32) != 0, tile(g["A_none"][:, j])), f"arrival {j}: which filters ignored their observation"
    assert np.abs(f.filter_times().cpu().numpy() - tile(g["A_ft"][:, j])).max() < 1e-12, f"arrival {j}: filter times"
    if j in keep:
      a = j // 25
      assert_close(f.state(), tile(g["A_x"][:, a]), rtol=1e-9, floor=1e-11, what=f"arrival {j} x")
      assert_close(f.covs().reshape(n, -1), tile(g["A_P"][:, a]).reshape(n, -1), rtol=1e-9, floor=1e-11, what=f"arrival {j} P")
  assert_close(f.state(), tile(g["A_x_final"]), rtol=1e-9, floor=1e-11, what="final x")
  assert_close(f.covs().reshape(n, -1), tile(g["A_P_final"]).reshape(n, -1), rtol=1e-9, floor=1e-11, what="final P")
  X = f.state()
  for c in range(1, copies):
    assert np.array_equal(X[:NA], X[c * NA:(c + 1) * NA]), "copies of a log must stay bit-identical"


def test_three_kinds_masks_and_late_observations_lane_group_family(gen):
  """10 logs of the 9-state model: own times, own kind order, ticks without an observation, one late observation each.  One
  masked launch per kind present at an arrival index; every filter follows the reference instance that was fed its log."""
  import torch
  from examples.kinematic9_kf import Kinematic9Kalman as K9
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  g = golden("perfilter_timelines.npz")
  NB, TB = g["B_t"].shape
  f = BatchedEKF(gen, "kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9, batch=NB, rewind_to_keep=64, per_filter=True)
  launches = 0
  for j in range(TB):
    for k in (1, 2, 3):
      act = g["B_kind"][:, j] == k
      if not act.any():
        continue
      Z = K9.obs_noise[k].shape[0]
      y = f.predict_and_update_batch(np.nan_to_num(g["B_t"][:, j]), k, g["B_z"][:, j, :Z].copy(), K9.obs_noise[k], active=act)
      launches += 1
      fl = f.flags.cpu().numpy()
      assert np.array_equal((fl & 16) != 0, ~act) and not (fl & 32).any()
      assert_close(y.cpu().numpy()[act], g["B_y"][act, j, :Z], rtol=1e-7, atol=1e-9, what=f"arrival {j} kind {k} residuals")
    assert_close(f.state(), g["B_x"][:, j], rtol=1e-8, floor=1e-10, what=f"arrival {j} x")
    assert_close(f.covs().reshape(NB, -1), g["B_P"][:, j].reshape(NB, -1), rtol=1e-8, floor=1e-10, what=f"arrival {j} P")
  assert launches > TB


@pytest.mark.parametrize("model", ["kinematic6", "kinematic9", "live"])
def test_masked_step_touches_only_active_filters(gen, model):
  """The masked entry points of every kernel family: active filters get exactly what the unmasked launch gives them (per-filter
  dt included), masked-out ones keep x, P and z bit for bit and report flag bit 4."""
  import torch
  from examples import ensure_generated
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  import bench
  M = bench.model_class(model)
  g = ensure_generated([model])
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  n = 203
  rng = np.random.default_rng(3)
  quat = list(getattr(M, "quaternion_idxs", []))
  kind = 10 if model == "live" else 1
  Z = np.atleast_2d(M.obs_noise[kind]).shape[0]
  x0 = np.tile(M.initial_x, (n, 1)) + (0.0 if quat else 0.1) * rng.normal(size=(n, D))
  A = rng.normal(size=(n, E, E)) * 0.05
  P0 = np.diag(M.initial_P_diag)[None] * 1e-2 + A @ A.transpose(0, 2, 1)
  z = rng.normal(size=(n, Z))
  dt = rng.uniform(0.0, 0.05, size=n)
  act = rng.random(n) < 0.6
  act[:3] = (True, False, True)
  ref = BatchedEKF(g, M.name, M.Q, M.initial_x, P0[0], D, E, batch=n, quaternion_idxs=quat)
  ref.init_state(x0, P0, -dt)             # per-filter times -dt, one common step to t = 0: the UNMASKED entry point with a dt vector
  yr = ref.predict_and_update_batch(0.0, kind, z.copy(), M.obs_noise[kind]).cpu().numpy()
  assert not ref.per_filter
  assert not ref.flags.cpu().numpy().any()
  f = BatchedEKF(g, M.name, M.Q, M.initial_x, P0[0], D, E, batch=n, quaternion_idxs=quat, per_filter=True)
  f.init_state(x0, P0, np.zeros(n))
  y = f.predict_and_update_batch(dt, kind, z.copy(), M.obs_noise[kind], active=act).cpu().numpy()
  torch.cuda.synchronize()
  X, P, fl = f.state(), f.covs(), f.flags.cpu().numpy()
  assert np.array_equal(X[act], ref.state()[act]) and np.array_equal(P[act], ref.covs()[act]) and np.array_equal(y[act], yr[act])
  assert np.array_equal(X[~act], x0[~act]) and np.array_equal(P[~act], P0[~act]) and np.array_equal(y[~act], z[~act])
  assert np.array_equal(fl == 16, ~act) and not fl[act].any()
  ft = f.filter_times().cpu().numpy()
  assert np.array_equal(ft[act], dt[act]) and not ft[~act].any()
  # predict alone, masked
  f.predict(np.full(n, 0.1), active=~act)
  assert np.array_equal(f.state()[act], X[act]) and np.array_equal(f.covs()[act], P[act])
  assert not np.array_equal(f.covs()[~act], P0[~act])
  assert np.array_equal(f.filter_times().cpu().numpy()[~act], np.full((~act).sum(), 0.1))


def test_late_observation_without_a_ring_is_an_error(gen):
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  f = BatchedEKF(gen, "kinematic", np.diag([0.1**2, 2.0**2]), np.array([0.5, 0.0]), np.eye(2), 2, 2, batch=5, per_filter=True)
  f.predict_and_update_batch(np.array([1.0, 2.0, 3.0, 4.0, 5.0]), 1, np.zeros((5, 1)), np.array([[0.01]]))
  with pytest.raises(AssertionError):
    f.predict_and_update_batch(np.array([1.5, 2.5, 2.9, 4.5, 5.5]), 1, np.zeros((5, 1)), np.array([[0.01]]))


def test_estimate_and_flags_of_a_late_observation_are_its_own():
  """Round-3 advice on the per-filter path: after a per-filter rewind the returned Estimate must be the state right after the
  LATE observation (the reference captures `ret` before it fast-forwards, ekf_sym.py:473-479), not the fast-forwarded one, and the
  flags the caller reads must be the late observation's (here: its Mahalanobis gate fired), not those of the replayed ones."""
  import torch
  from examples import ensure_generated
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  gen6 = ensure_generated(["kinematic6_maha"])
  n = 3
  mk = lambda **kw: BatchedEKF(gen6, "kinematic6_maha", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n, maha_test_kinds=[1], **kw)   # noqa: E731
  rng = np.random.default_rng(3)
  R = K6.obs_noise[1]
  ts = 0.01 * np.arange(1, 11)
  zs = rng.normal(size=(10, n, 3)) * 0.1
  z_late = zs[4].copy()
  z_late[1] += 50.0                                   # a gross outlier for filter 1: its gate fires
  f = mk(rewind_to_keep=32, per_filter=True)
  for t, z in zip(ts, zs):
    f.predict_and_update_batch(np.full(n, t), 1, z.copy(), R)
  act = np.array([False, True, False])
  est = f.predict_and_update_batch(np.full(n, 0.055), 1, z_late.copy(), R, active=act, keep_estimate=True)
  torch.cuda.synchronize()
  assert (f.flags.cpu().numpy()[1] & 1) == 1, "the late observation's gate flag must survive the replay"
  # what the reference instance of filter 1 returns: observations up to 0.05, then the late one
  a = mk()
  for t, z in zip(ts[:5], zs[:5]):
    a.predict_and_update_batch(float(t), 1, z.copy(), R)
  a.predict_and_update_batch(0.055, 1, z_late.copy(), R)
  torch.cuda.synchronize()
  assert_close(est[1][1].cpu().numpy()[None], a.state()[1][None], rtol=1e-10, floor=1e-12, what="xk_k of the late observation")
  assert_close(est[3][1].cpu().numpy().reshape(1, -1), a.covs()[1].reshape(1, -1), rtol=1e-10, floor=1e-12, what="Pk_k of the late observation")
  # ... and the filter itself went on through the five observations it had overtaken
  for t, z in zip(ts[5:], zs[5:]):
    a.predict_and_update_batch(float(t), 1, z.copy(), R)
  torch.cuda.synchronize()
  assert_close(f.state()[1][None], a.state()[1][None], rtol=1e-9, floor=1e-11, what="state after the fast-forward")
  assert np.abs(f.state()[1] - est[1][1].cpu().numpy()).max() > 1e-6, "the two must differ for this test to mean anything"
