"""Stress of the packed-triangle trace pipeline (batch_run_tri / batch_rts_tri) against the full-matrix one over many random batch shapes, schedules
with zero time differences, and models (live: 22 states; rand13 / rand17: odd record lengths): the packed results must be the full ones' lower
triangles bit for bit every time (the forward kernel's packed row stores lean on the wavefront's lockstep and on store order).  One-off confidence
run behind tests/test_gpu_tri.py; prints a summary line per model.   python tools/stress_tri.py [cases per model]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated, model_class_of
from rednose_amd.helpers.ekf_sym import BatchedEKF

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gen = ensure_generated(["live", "rand13", "rand17"])
for name in ("live", "rand13", "rand17"):
  M = model_class_of(name)
  D, E = int(M.initial_x.shape[0]), int(M.initial_P_diag.shape[0])
  quat = list(getattr(M, "quaternion_idxs", [3] if name == "live" else []))
  kinds_all = sorted(M.obs_noise) if name != "live" else [4, 10, 12]
  zmax = max(np.atleast_2d(M.obs_noise[k]).shape[0] for k in kinds_all)
  rng = np.random.default_rng(len(name))
  il = np.tril_indices(E)
  bad = 0
  for case in range(cases):
    n = int(rng.integers(1, 300))
    T = int(rng.integers(2, 40))
    f = BatchedEKF(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, quaternion_idxs=quat)
    x0 = np.tile(M.initial_x, (n, 1))
    if name != "live":
      x0 = x0 + rng.normal(size=(n, D)) * 0.2
    A = rng.normal(size=(n, E, E)) * 0.05 * np.sqrt(M.initial_P_diag)[None, :, None]
    P0 = np.diag(M.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)
    ks = rng.choice(kinds_all, size=T).astype(np.int32)
    dts = rng.uniform(0.002, 0.02, size=T)
    dts[rng.random(T) < 0.4] = 0.0
    ts = np.cumsum(dts)
    zs = rng.normal(size=(T, n, zmax)) * 0.01
    if name == "live":
      zs[ks == 12] += M.initial_x[:3]
      zs[ks == 10] += np.array([0.0, 0.0, 0.0])
    Rs = {int(k): np.atleast_2d(M.obs_noise[int(k)]) for k in kinds_all}
    res = {}
    for packed in (False, True):
      f.init_state(x0, P0, 0.0)
      _, tx, tP, _ = f.run(ts, ks, zs.copy(), Rs, trace=True, packed=packed)
      xs, Ps = f.rts_smooth(tx, tP, ts, packed=packed)
      torch.cuda.synchronize()
      res[packed] = (tP.cpu().numpy(), xs.cpu().numpy(), Ps.cpu().numpy())
    ok = (np.array_equal(res[True][0], res[False][0][:, :, il[0], il[1]]) and np.array_equal(res[True][1], res[False][1]) and
          np.array_equal(res[True][2], res[False][2][:, :, il[0], il[1]]) and np.isfinite(res[True][2]).all())
    if not ok:
      bad += 1
      print(f"  MISMATCH {name} case {case}: n={n} T={T}")
  print(f"{name}: {cases} random shapes, {bad} mismatches")
