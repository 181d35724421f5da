"""GPU parity of the fused multi-step entry point {name}_batch_run against the oracle's batch_run (same schedule
conventions) and against the step-granular path (must agree bit for bit: same device code, same order)."""
import numpy as np
import pytest

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  return torch, ensure_generated(["kinematic6", "live", "live_maha"])


@pytest.mark.parametrize("n", [64, 301])
def test_kinematic6_run_vs_oracle_and_step_path(env, n):
  torch, gen = env
  from oracle_lib import OracleLib
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  o = OracleLib("kinematic6")
  rng = np.random.default_rng(n)
  T = 40
  x0 = rng.normal(size=(n, 6)); A = rng.normal(size=(n, 6, 6)) * 0.3; P0 = np.eye(6)[None] + A @ A.transpose(0, 2, 1)
  ts = np.cumsum(rng.uniform(0.005, 0.02, size=T))
  zs = rng.normal(size=(T, n, 3))
  R = K6.obs_noise[1]
  f = BatchedEKF(gen, "kinematic6", K6.Q, x0[0], P0[0], 6, 6, batch=n); f.init_state(x0, P0, 0.0)
  ys, tx, tP, _ = f.run(ts, np.ones(T, dtype=np.int32), zs.copy(), {1: R}, trace=True)
  torch.cuda.synchronize()
  s = BatchedEKF(gen, "kinematic6", K6.Q, x0[0], P0[0], 6, 6, batch=n); s.init_state(x0, P0, 0.0)
  for t in range(T):
    y = s.predict_and_update_batch(ts[t], 1, zs[t].copy(), R)
    # same generated device functions in both kernels, FMA contraction decided per kernel by hipcc: equal to the last bits, not
    # necessarily bit for bit (errors do not accumulate here beyond the filter's own contraction: 40 steps)
    assert_close(y.cpu().numpy(), ys[t].cpu().numpy(), rtol=1e-11, atol=1e-12, what=f"y at t={t}")
    assert_close(s.x.cpu().numpy(), tx[t].cpu().numpy(), rtol=1e-11, floor=1e-12, what=f"x at t={t}")
    assert_close(s.P.cpu().numpy().reshape(n, -1), tP[t].cpu().numpy().reshape(n, -1), rtol=1e-11, floor=1e-12, what=f"P at t={t}")
  assert torch.equal(tx[-1], f.x) and torch.equal(tP[-1], f.P)
  xr, Pr, zr = x0.copy(), P0.copy(), zs.copy()
  xf = np.zeros((T, n, 6)); Pf = np.zeros((T, n, 6, 6))
  o.batch_run(np.ones(T, dtype=np.int32), np.diff(np.concatenate([[0.0], ts])), xr, Pr, zr, np.tile(R.reshape(1, 9), (T, 1)), K6.Q, xf=xf, Pf=Pf)
  assert_close(f.state(), xr, rtol=1e-10, floor=1e-12); assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-10, floor=1e-12)
  assert_close(tx.cpu().numpy().reshape(T * n, -1), xf.reshape(T * n, -1), rtol=1e-10, floor=1e-12)
  assert_close(ys.cpu().numpy().reshape(T * n, -1), zr.reshape(T * n, -1), rtol=1e-9, atol=1e-12)


def test_live_run_vs_reference_stream_and_step_path(env):
  torch, gen = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.live_kf import LiveKalman as L
  g = golden("live_stream.npz")
  n = 5
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  T = len(kinds)
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  Rs = {int(k): L.obs_noise[int(k)] for k in set(kinds.tolist())}
  mk = lambda: BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])  # noqa: E731
  f = mk(); f.init_state(g["x0"], g["P0"], None)
  ys, tx, tP, _ = f.run(ts, kinds, zs.copy(), Rs, trace=True)
  torch.cuda.synchronize()
  s = mk(); s.init_state(g["x0"], g["P0"], None)
  for t in range(T):
    y = s.predict_and_update_batch(float(ts[t]), int(kinds[t]), zs[t].copy(), Rs[int(kinds[t])])
    # the step-granular kernels (three-phase structure) and the fused run (state-resident structure) are different
    # instruction streams: same algebra, different FMA contraction -> equal to rounding, not bit for bit
    assert_close(s.x.cpu().numpy(), tx[t].cpu().numpy(), rtol=1e-9, floor=1e-11, what=f"fused vs step-granular x at t={t}")
    assert_close(s.P.cpu().numpy().reshape(n, -1), tP[t].cpu().numpy().reshape(n, -1), rtol=1e-8, floor=1e-10, what=f"fused vs step-granular P at t={t}")
  X = tx.cpu().numpy()
  for j in range(n):
    assert_close(X[:, j], g["xs"], rtol=1e-8, floor=1e-10, what="fused live stream vs reference numpy path")
  assert_close(tP.cpu().numpy()[g["P_idx"], 0].reshape(len(g["P_idx"]), -1), g["Ps"].reshape(len(g["P_idx"]), -1), rtol=1e-7, floor=1e-9)


def test_live_maha_run_flags(env):
  """Config 4 forward pass in miniature: gated GNSS kind inside a fused run, 2 % gross outliers, decisions vs oracle."""
  torch, gen = env
  from oracle_lib import OracleLib
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.live_kf import LiveKalman as L
  g = golden("live_stream.npz")
  o = OracleLib("live_maha")
  n = 37
  rng = np.random.default_rng(4)
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  T = len(kinds)
  zs = np.tile(g["zs"][:, None, :], (1, n, 1)) + rng.normal(size=(T, n, 3)) * 1e-3
  pos_steps = np.where(kinds == 12)[0]
  out_t = pos_steps[2:]
  bad = rng.random(size=(len(out_t), n)) < 0.3
  for a, t in enumerate(out_t):
    zs[t, bad[a]] += rng.normal(size=(bad[a].sum(), 3)) * 5000.0
  Rs = {int(k): L.obs_noise[int(k)] for k in set(kinds.tolist())}
  f = BatchedEKF(gen, "live_maha", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3], maha_test_kinds=[12])
  f.init_state(g["x0"], g["P0"], None)
  ys, _, _, fl = f.run(ts, kinds, zs.copy(), Rs, flags=True)
  torch.cuda.synchronize()
  xr = np.tile(g["x0"], (n, 1)); Pr = np.tile(g["P0"], (n, 1, 1)); zr = zs.copy()
  Rt = np.zeros((T, 9))
  for t, k in enumerate(kinds):
    Rt[t] = L.obs_noise[int(k)].reshape(-1)
  flr = np.zeros((T, n), dtype=np.uint8)
  o.batch_run(kinds, np.diff(np.concatenate([[ts[0]], ts])), xr, Pr, zr, Rt, L.Q, quat_idx=3, flags=flr)
  got = fl.cpu().numpy()
  assert np.array_equal(got & 1, flr)
  assert flr[out_t][bad].all() and flr.sum() >= bad.sum()
  assert_close(f.state(), xr, rtol=1e-7, floor=1e-9)
  assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-6, floor=1e-8)


def test_degenerate_sizes(env):
  """Empty batch / empty schedule / single-step smoothing must be no-ops, not crashes (C ABI level)."""
  import ctypes
  torch, gen = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=5)
  lib = f._lib
  p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
  kd = torch.ones(1, dtype=torch.int32, device=f.device); dd = torch.zeros(1, dtype=torch.float64, device=f.device)
  zz = torch.zeros((1, 5, 3), dtype=torch.float64, device=f.device); Rd = torch.eye(3, dtype=torch.float64, device=f.device).reshape(1, 9).contiguous()
  x_before = f.state().copy()
  assert lib.kinematic6_batch_run(p(f.x), p(f.P), p(f.Q), p(kd), p(dd), 0, p(zz), p(Rd), 5, 0, None, None, None, None, None, None) == 0     # T = 0
  assert lib.kinematic6_batch_run(p(f.x), p(f.P), p(f.Q), p(kd), p(dd), 1, p(zz), p(Rd), 0, 0, None, None, None, None, None, None) == 0     # n = 0
  torch.cuda.synchronize()
  assert np.array_equal(f.state(), x_before)
  # the Python wrapper treats an empty schedule the same way (no IndexError on ts[0] / ts[-1])
  ys, tx0, tP0, fl0 = f.run(np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros((0, 5, 3)), {1: K6.obs_noise[1]}, trace=True)
  assert tuple(ys.shape) == (0, 5, 3) and tx0 is None and np.array_equal(f.state(), x_before)
  # single-estimate smoothing: nothing to smooth, the filtered pair comes back
  tx = torch.randn((1, 5, 6), dtype=torch.float64, device=f.device); tP = torch.eye(6, dtype=torch.float64, device=f.device).repeat(1, 5, 1, 1)
  xs, Ps = f.rts_smooth(tx, tP, np.array([0.0]))
  torch.cuda.synchronize()
  assert torch.equal(xs, tx) and torch.equal(Ps, tP)


@pytest.mark.parametrize("packed", [False, True])
def test_degenerate_sizes_lane_group(env, packed):
  """The two-wavefront fused run and the register-broadcast smoother (live: k_run2 / k_rts4 and their packed-triangle forms) on the smallest
  inputs: an empty schedule, ONE step, a one-estimate and a two-estimate trace, one filter and a ragged tile -- each against the step-granular
  path / the same call on a larger batch, none a crash."""
  torch, gen = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.live_kf import LiveKalman as L
  g = golden("live_stream.npz")
  kinds, ts = g["kinds"].astype(np.int32), g["ts"]
  Rs = {int(k): L.obs_noise[int(k)] for k in set(kinds.tolist())}
  mk = lambda n: BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])  # noqa: E731
  il = np.tril_indices(22)
  lower = (lambda P: P) if packed else (lambda P: P[..., il[0], il[1]])
  ref = mk(9); ref.init_state(g["x0"], g["P0"], None)
  zs9 = np.tile(g["zs"][:, None, :], (1, 9, 1))
  _, rx, rP, _ = ref.run(ts[:6], kinds[:6], zs9[:6].copy(), Rs, trace=True, packed=packed)
  sx, sP = ref.rts_smooth(rx, rP, ts[:6], packed=packed)
  torch.cuda.synchronize()
  for n in (1, 3):                                     # one filter; a ragged tile of the 8-filter fused run / the 4-filter smoother
    f = mk(n); f.init_state(g["x0"], g["P0"], None)
    x0 = f.state().copy()
    ys, tx0, _, _ = f.run(np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros((0, n, 3)), Rs, trace=True, packed=packed)      # T = 0
    assert tuple(ys.shape)[0] == 0 and tx0 is None and np.array_equal(f.state(), x0)
    _, tx, tP, _ = f.run(ts[:1], kinds[:1], zs9[:1, :n].copy(), Rs, trace=True, packed=packed)                                 # T = 1
    torch.cuda.synchronize()
    assert torch.equal(tx[0], rx[0, :n]) and torch.equal(tP[0], rP[0, :n])
    assert np.array_equal(f.state(), tx[0].cpu().numpy())
    x1, P1 = f.rts_smooth(tx, tP, ts[:1], packed=packed)                                                                     # one estimate: comes back
    assert torch.equal(x1, tx) and torch.equal(P1, tP)
    _, tx5, tP5, _ = f.run(ts[1:6], kinds[1:6], zs9[1:6, :n].copy(), Rs, trace=True, packed=packed)
    fx, fP = torch.cat([tx, tx5]), torch.cat([tP, tP5])
    assert torch.equal(fx, rx[:, :n]) and torch.equal(fP, rP[:, :n])                                                         # a run continued = the run
    x6, P6 = f.rts_smooth(fx, fP, ts[:6], packed=packed)
    assert torch.equal(x6, sx[:, :n]) and np.array_equal(lower(P6.cpu().numpy()), lower(sP[:, :n].cpu().numpy()))
    x2, P2 = f.rts_smooth(fx[4:], fP[4:], ts[4:6], packed=packed)                                                            # two estimates: one backward step
    torch.cuda.synchronize()
    assert torch.equal(x2[1], x6[5]) and np.array_equal(lower(P2[1].cpu().numpy()), lower(P6[5].cpu().numpy()))
    assert_close(x2[0].cpu().numpy(), x6[4].cpu().numpy(), rtol=1e-13, floor=1e-15, what="two-estimate smoothing, oldest state")
    assert_close(lower(P2[0].cpu().numpy()).reshape(n, -1), lower(P6[4].cpu().numpy()).reshape(n, -1), rtol=1e-13, floor=1e-15, what="two-estimate smoothing, oldest covariance")


def test_million_filter_batch_properties(env):
  """Beyond the 256 MiB Infinity Cache (755 MB of state): grid-stride path, every filter identical -> results identical."""
  torch, gen = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  n = 1 << 20
  f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n)
  z = torch.full((n, 3), 0.25, dtype=torch.float64, device=f.device)
  for i in range(3):
    f.predict_and_update_batch(0.01 * i, 1, z.clone(), K6.obs_noise[1])
  torch.cuda.synchronize()
  assert bool((f.x == f.x[0]).all()) and bool((f.P == f.P[0]).all()) and bool(torch.isfinite(f.P).all())
