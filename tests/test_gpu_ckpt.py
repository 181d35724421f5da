"""{name}_batch_predict_update_{kind}_ckpt -- the fused step that writes its own checkpoint (k_stepc_{kind}) -- against the plain fused step
through the C ABI: the same bits in x, P, y and the flags; the checkpoint holds the observations as they came and the filtered pair; nothing is
written past it.  Both kernel families, records of odd length (kinematic9: 81 doubles), an MSCKF model's ordinary kind, ragged batches.  The
orchestrators' rewind rings go through this entry point for single-observation calls (tests/test_gpu_parity.py, test_gpu_multi_obs.py,
test_gpu_cpp.py: the reference class's swapped-sample logs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  assert torch.cuda.is_available()
  from examples import ensure_generated
  return torch, ensure_generated(["kinematic6", "kinematic9", "live_maha", "feature"])


def _cases():
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  from examples.kinematic9_kf import Kinematic9Kalman as K9
  from examples.live_kf import LiveKalman as L
  from examples.feature_kf import FeatureKalman as F
  return [("kinematic6", K6, 6, 6, [1], {}), ("kinematic9", K9, 9, 9, [1, 2], {}),
          ("live_maha", L, 23, 22, [4, 10, 12], dict(quaternion_idxs=[3], maha_test_kinds=[12])), ("feature", F, None, None, [1], None)]


@pytest.mark.parametrize("case", range(4))
@pytest.mark.parametrize("n", [1, 67, 300])
def test_checkpointing_step_equals_step_plus_copies(env, case, n):
  torch, gen = env
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  name, M, D, E, kinds, kw = _cases()[case]
  if kw is None:                               # the MSCKF test model: its ordinary (position) kind
    f = BatchedEKF(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), 6, 6, batch=n, **M.filter_kwargs())
  else:
    f = BatchedEKF(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, **kw)
  D, E = f.dim_x, f.dim_err
  dev = f.device
  g = torch.Generator(device=dev).manual_seed(100 * case + n)
  x_init = torch.as_tensor(np.asarray(M.initial_x, dtype=np.float64), device=dev)
  for kind in kinds:
    Z = f.zdims[kind]
    x0 = x_init[None] + 0.01 * torch.randn((n, D), generator=g, device=dev, dtype=torch.float64) * torch.clamp(x_init.abs(), min=1.0)[None] * (x_init.abs() < 10.0)[None]
    A = 0.1 * torch.randn((n, E, E), generator=g, device=dev, dtype=torch.float64) * torch.sqrt(torch.as_tensor(np.asarray(M.initial_P_diag, dtype=np.float64), device=dev))[None, :, None]
    P0 = torch.diag(torch.as_tensor(np.asarray(M.initial_P_diag, dtype=np.float64), device=dev))[None] + A @ A.transpose(1, 2)
    z0 = torch.randn((n, Z), generator=g, device=dev, dtype=torch.float64)
    if name == "live_maha" and kind == 12:
      z0 = x0[:, :3] + z0 * 3.0 + (torch.rand((n, 1), generator=g, device=dev, dtype=torch.float64) < 0.3) * 1e7      # some gated (the prior is ~1e4 m wide)
    _, Rd, per = f._obs_args(kind, z0.clone(), np.atleast_2d(M.obs_noise[kind]))      # pylint: disable=protected-access
    p = f._p      # pylint: disable=protected-access
    res = []
    for ckpt in (False, True):
      x, P, z = x0.clone(), P0.clone(), z0.clone()
      fl = torch.full((n,), 99, dtype=torch.uint8, device=dev)
      if ckpt:
        cx = torch.full((n + 1, D), 7.0, dtype=torch.float64, device=dev)
        cP = torch.full((n + 1, E, E), 7.0, dtype=torch.float64, device=dev)
        cz = torch.full((n + 1, Z), 7.0, dtype=torch.float64, device=dev)
        f._call(f"batch_predict_update_{kind}_ckpt", p(x), p(P), p(f.Q), None, 0.01, p(z), p(Rd), per, None, n, f.norm_quats, p(fl), p(cx), p(cP), p(cz), f._stream())      # pylint: disable=protected-access
      else:
        f._call(f"batch_predict_update_{kind}", p(x), p(P), p(f.Q), None, 0.01, p(z), p(Rd), per, None, n, f.norm_quats, p(fl), f._stream())      # pylint: disable=protected-access
      torch.cuda.synchronize()
      res.append((x, P, z, fl))
    (x1, P1, y1, f1), (x2, P2, y2, f2) = res
    what = f"{name} kind {kind} n {n}"
    assert torch.equal(x1, x2) and torch.equal(P1, P2) and torch.equal(y1, y2) and torch.equal(f1, f2), what + ": checkpointing step vs plain step"
    assert torch.equal(cx[:n], x2) and torch.equal(cP[:n], P2) and torch.equal(cz[:n], z0), what + ": checkpoint"
    assert bool((cx[n] == 7.0).all()) and bool((cP[n] == 7.0).all()) and bool((cz[n] == 7.0).all()), what + ": guard rows"
    assert torch.isfinite(x2).all()
    if name == "live_maha" and kind == 12 and n > 1:
      assert int((f2 & 1).sum()) > 0


def test_checkpoint_pointers_are_checked(env):
  torch, gen = env
  from rednose_amd.helpers import KalmanError
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=4)
  z = torch.zeros((4, 3), dtype=torch.float64, device=f.device)
  _, Rd, per = f._obs_args(1, z, K6.obs_noise[1])      # pylint: disable=protected-access
  p = f._p      # pylint: disable=protected-access
  cz = torch.zeros_like(z)
  with pytest.raises(KalmanError):      # the checkpoint may not alias the state
    f._call("batch_predict_update_1_ckpt", p(f.x), p(f.P), p(f.Q), None, 0.0, p(z), p(Rd), per, None, 4, 0, None, p(f.x), p(f.P), p(cz), f._stream())      # pylint: disable=protected-access
  with pytest.raises(KalmanError):      # ... nor be absent
    f._call("batch_predict_update_1_ckpt", p(f.x), p(f.P), p(f.Q), None, 0.0, p(z), p(Rd), per, None, 4, 0, None, None, None, None, f._stream())      # pylint: disable=protected-access
