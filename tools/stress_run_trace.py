"""Stress check of the fused run's trace against the step-granular path (GPU vs GPU) over random batch sizes and schedules.
For lane-per-filter models the two paths run the same device code in the same order: any difference is a bug (race, stale
staging, lifetime)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated
import examples.random_kf as R
from rednose_amd.helpers.ekf_sym import BatchedEKF

names = sys.argv[1:] or ["rand8", "rand3", "rand5"]
REPS = int(os.environ.get("RN_STRESS_REPS", "40"))
MODE = os.environ.get("RN_STRESS_MODE", "")       # "sync": synchronize after every launch; "keep": never free a tensor
keep = []
bad = 0
for name in names:
  M = getattr(R, f"Random{name[4:]}Kalman")
  gen = ensure_generated([name])
  rng = np.random.default_rng(1)
  for rep in range(REPS):
    n = int(rng.integers(1, 400)); T = int(rng.integers(2, 30))
    D = M.dim
    x0 = M.initial_x[None] + rng.normal(size=(n, D)) * 0.3
    A = rng.normal(size=(n, D, D)) * 0.2
    P0 = np.diag(M.initial_P_diag)[None] + A @ A.transpose(0, 2, 1)
    kinds = rng.integers(1, 4, size=T).astype(np.int32)
    ts = np.cumsum(rng.uniform(0.005, 0.03, size=T))
    zs = rng.normal(size=(T, n, 3)) * 0.5
    Rs = {k: M.obs_noise[k] for k in (1, 2, 3)}
    f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, D, batch=n); f.init_state(x0, P0, 0.0)
    if os.environ.get('RN_STRESS_VERBOSE'):
      print(name, 'rep', rep, 'n', n, 'T', T, flush=True)
    ys, tx, tP, _ = f.run(ts, kinds, zs.copy(), Rs, trace=True)
    if MODE == "keep":
      keep.extend([f, ys, tx, tP])
    if os.environ.get('RN_STRESS_VERBOSE'):
      torch.cuda.synchronize()
    s = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, D, batch=n); s.init_state(x0, P0, 0.0)
    for t in range(T):
      Z = Rs[int(kinds[t])].shape[0]
      y = s.predict_and_update_batch(float(ts[t]), int(kinds[t]), zs[t, :, :Z].copy(), Rs[int(kinds[t])])
      if MODE == "keep":
        keep.append(y)
      if MODE == "sync":
        torch.cuda.synchronize()
      if not (torch.equal(s.x, tx[t]) and torch.equal(s.P, tP[t])):
        dx = (s.x - tx[t]).abs().max().item(); dP = (s.P - tP[t]).abs().max().item()
        if dx > 1e-9 or dP > 1e-9:
          bad += 1
          print(f"MISMATCH {name} rep {rep} n={n} T={T} t={t}: |dx|={dx:.3e} |dP|={dP:.3e}")
          badf = torch.nonzero((s.x - tx[t]).abs().amax(dim=1) > 1e-9).flatten().tolist()
          print("   bad filters:", badf[:20], "count", len(badf), " kinds so far", kinds[:t + 1].tolist())
          # who is right?  replay this rep on the oracle
          sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
          from oracle_lib import OracleLib
          o = OracleLib(name)
          xr, Pr, zr = x0.copy(), P0.copy(), zs.copy()
          Rt = np.zeros((T, 9))
          for tt, k in enumerate(kinds):
            Rt[tt, :Rs[int(k)].size] = Rs[int(k)].reshape(-1)
          xf = np.zeros((T, n, D)); Pf = np.zeros((T, n, D, D))
          o.batch_run(kinds, np.diff(np.concatenate([[0.0], ts])), xr, Pr, zr, Rt, M.Q, xf=xf, Pf=Pf)
          print("   |step - oracle| at t:", np.abs(s.x.cpu().numpy() - xf[t]).max(), " |fused trace - oracle| at t:", np.abs(tx[t].cpu().numpy() - xf[t]).max(),
                " |fused final - oracle final|:", np.abs(f.x.cpu().numpy() - xr).max())
          X = tx.cpu().numpy()
          bt, bi, bk = np.where(np.abs(X - xf) > 1e-9)
          print("   all bad (t, i, k):", list(zip(bt.tolist(), bi.tolist(), bk.tolist()))[:40], "total", len(bt))
          for (a, b, c) in list(zip(bt.tolist(), bi.tolist(), bk.tolist()))[:6]:
            hits = np.argwhere(np.abs(xf - X[a, b, c]) < 1e-12)
            print("     value", X[a, b, c], "want", xf[a, b, c], "equals oracle trace at (t,i,k):", hits[:4].tolist())
          i0 = badf[0]
          print("   filter", i0, "step x", s.x[i0].cpu().numpy()[:4], "trace x", tx[t, i0].cpu().numpy()[:4], "oracle", xf[t, i0][:4])
          break
print("mismatches:", bad)
