"""The untraced fused run of lane-per-filter models (k_run_blk: blocks of K steps in registers, emit_small.run_kernel_blk) against
the traced kernel (k_run), the step path and the oracle: same schedules, schedule lengths around the block size (one step, one
short of a block, exactly one, one more, several blocks and a ragged last one), ragged filter tiles, several kinds per schedule,
gated observations (per-lane flags)."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  return torch, ensure_generated(["kinematic", "kinematic6", "kinematic6_maha", "rand3", "rand5"])


def _block(f):
  return int(getattr(f._lib, f"{f.name}_run_unroll")())      # pylint: disable=protected-access


def _lengths(K):
  return sorted({1, 2, K - 1, K, K + 1, 2 * K, 3 * K + 2})


def _make(gen, model, n):
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  if model == "kinematic":
    from examples.kinematic_kf import KinematicKalman as M
    D, name, Rs, Q = 2, "kinematic", {1: np.atleast_2d(M.obs_noise[1])}, M.Q
    x0 = np.array([0.5, 0.0])
    P0 = np.eye(2)
  elif model in ("kinematic6", "kinematic6_maha"):
    from examples.kinematic6_kf import Kinematic6Kalman as M
    D, name, Rs, Q = 6, model, {1: M.obs_noise[1]}, M.Q
    x0, P0 = M.initial_x, np.diag(M.initial_P_diag)
  else:
    import examples.random_kf as R
    M = getattr(R, f"Random{model[4:]}Kalman")
    D, name, Rs, Q = M.dim, M.name, {k: M.obs_noise[k] for k in (1, 2, 3)}, M.Q
    x0, P0 = M.initial_x, np.diag(M.initial_P_diag)
  extra = {"maha_test_kinds": [1]} if model == "kinematic6_maha" else {}
  mk = lambda: BatchedEKF(gen, name, Q, x0, P0, D, D, batch=n, **extra)      # noqa: E731
  return mk, D, Rs, x0, P0


@pytest.mark.parametrize("model,n", [("kinematic", 1), ("kinematic", 65), ("kinematic", 1000), ("kinematic6", 64), ("kinematic6", 301),
                                     ("kinematic6_maha", 130), ("rand3", 70), ("rand5", 41)])
def test_untraced_run_equals_traced_run(env, model, n):
  torch, gen = env
  mk, D, Rs, xi, Pi = _make(gen, model, n)
  rng = np.random.default_rng(n + D)
  x0 = xi[None] + rng.normal(size=(n, D)) * 0.3
  A = rng.normal(size=(n, D, D)) * 0.2
  P0 = Pi[None] + A @ A.transpose(0, 2, 1)
  kset = sorted(Rs)
  zmax = max(R.shape[0] for R in Rs.values())
  a = mk()
  K = _block(a)
  assert K >= 4, "the blocked kernel was not generated for this model"
  for T in _lengths(K):
    kinds = np.array([kset[t % len(kset)] for t in range(T)], dtype=np.int32)
    ts = np.cumsum(rng.uniform(0.005, 0.03, size=T))
    # a few far-off observations so that a gated kind (kinematic6_maha) raises flags in some lanes and not in others
    zs = rng.normal(size=(T, n, zmax)) * np.where(rng.uniform(size=(T, n, 1)) < 0.1, 30.0, 0.5)
    a.init_state(x0, P0, 0.0)
    ya, _, _, fa = a.run(ts, kinds, zs.copy(), Rs, flags=True)             # k_run_blk
    b = mk(); b.init_state(x0, P0, 0.0)
    yb, tx, tP, fb = b.run(ts, kinds, zs.copy(), Rs, trace=True, flags=True)   # k_run
    torch.cuda.synchronize()
    what = f"{model} n={n} T={T}"
    assert torch.equal(fa, fb), what + " flags"
    if model == "kinematic6_maha" and T >= K:
      assert int((fa != 0).sum()) > 0 and int((fa == 0).sum()) > 0
    assert_close(ya.cpu().numpy().reshape(T * n, -1), yb.cpu().numpy().reshape(T * n, -1), rtol=1e-10, atol=1e-11, what=what + " y")
    assert_close(a.state(), b.state(), rtol=1e-10, floor=1e-12, what=what + " x")
    assert_close(a.covs().reshape(n, -1), b.covs().reshape(n, -1), rtol=1e-10, floor=1e-12, what=what + " P")


def test_untraced_run_vs_oracle_and_untouched_neighbours(env):
  """Against the oracle's batch_run, on a z array with guard rows before and after the schedule: the kernel reads rows past the end
  of a short last block clamped, and must neither read garbage into results nor write outside (T, n, zmax)."""
  torch, gen = env
  from oracle_lib import OracleLib
  from examples.kinematic_kf import KinematicKalman as M
  n = 334            # ragged last tile (5 x 64 + 14); even, so that row 1 of the guarded array below starts on a 16-byte boundary
  mk, D, Rs, xi, Pi = _make(gen, "kinematic", n)
  f = mk()
  K = _block(f)
  o = OracleLib("kinematic")
  rng = np.random.default_rng(5)
  for T in (K + 3, 4 * K):
    x0 = xi[None] + rng.normal(size=(n, D)) * 0.3
    P0 = np.tile(Pi, (n, 1, 1))
    ts = np.cumsum(rng.uniform(0.005, 0.03, size=T))
    zs = rng.normal(size=(T, n, 1))
    f.init_state(x0, P0, 0.0)
    guard = torch.full((T + 2, n, 1), 777.0, dtype=torch.float64, device=f.device)
    guard[1:T + 1] = torch.from_numpy(zs).to(f.device)
    fl = torch.full((T + 2, n), 99, dtype=torch.uint8, device=f.device)
    kd = torch.ones(T, dtype=torch.int32, device=f.device)
    dd = torch.from_numpy(np.diff(np.concatenate([[0.0], ts]))).to(f.device)
    Rd = torch.from_numpy(np.tile(np.atleast_2d(M.obs_noise[1]).reshape(1, 1), (T, 1))).to(f.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    rc = f._lib.kinematic_batch_run(p(f.x), p(f.P), p(f.Q), p(kd), p(dd), T, p(guard[1]), p(Rd), n, 0, p(fl[1]), None, None, None, None, None)      # pylint: disable=protected-access
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((guard[0] == 777.0).all()) and bool((guard[T + 1] == 777.0).all())
    assert bool((fl[0] == 99).all()) and bool((fl[T + 1] == 99).all()) and bool((fl[1:T + 1] == 0).all())
    xr, Pr, zr = x0.copy(), P0.copy(), zs.copy()
    o.batch_run(np.ones(T, dtype=np.int32), np.diff(np.concatenate([[0.0], ts])), xr, Pr, zr, np.tile(np.atleast_2d(M.obs_noise[1]).reshape(1, 1), (T, 1)), M.Q)
    assert_close(f.state(), xr, rtol=1e-9, floor=1e-11, what=f"x T={T}")
    assert_close(f.covs().reshape(n, -1), Pr.reshape(n, -1), rtol=1e-9, floor=1e-11, what=f"P T={T}")
    assert_close(guard[1:T + 1].cpu().numpy().reshape(T * n, -1), zr.reshape(T * n, -1), rtol=1e-9, atol=1e-11, what=f"y T={T}")
