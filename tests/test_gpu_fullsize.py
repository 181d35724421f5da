"""BASELINE configs 3 and 4 at their STATED size: live ESKF, 16 384 filters, the 10 s IMU@100 Hz + GNSS@10 Hz
stream = 2 100 steps per filter; config 4 adds the Mahalanobis gate on ECEF_POS with 2 % gross outliers and the RTS backward
pass over the whole stream (swept in batch chunks: the filtered trace is 140 GB otherwise).

Checkers: (1) the reference's own numpy path on one 2 100-step stream (tests/golden/live_stream_2100.npz, oracle/make_golden.py);
(2) the oracle (C restatement of ekf_c.c over the reference-generated sympy C, OpenMP over filters) on identical per-filter
streams: all 16 384 filters at the final step, a 1 024-filter subset every 100 steps; gate decisions of all 34 M steps;
(3) for the smoother, the host restatement of ekf_sym.py:651-690 (EKF_sym.rts_smooth bound to the oracle library) on the
oracle's own estimates of a few filters over all 2 100 steps.
Tolerance: 1e-8 of the row maximum for x and P of the forward pass; the smoother's budget is set by the
conditioning of the predicted covariance it solves with (see test_rts_error_budget in test_gpu_rts.py)."""
import json
import os

import numpy as np
import pytest

from conftest import REPO, assert_close, golden

pytestmark = pytest.mark.gpu

N, T_FULL = 16384, 2100


def _schedule(total):
  kinds, ts, tick = [], [], 0
  while len(kinds) < total:
    t = 0.01 * tick
    kinds += [4, 10]; ts += [t, t]
    if tick % 10 == 9:
      kinds.append(12); ts.append(t)
    tick += 1
  return np.array(kinds[:total], dtype=np.int32), np.array(ts[:total])


def _setup(name, n, seed):
  import torch
  from examples import ensure_generated
  from examples.live_kf import LiveKalman as L
  from oracle_lib import OracleLib
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  gen = ensure_generated([name])
  kw = dict(maha_test_kinds=[12]) if name == "live_maha" else {}
  f = BatchedEKF(gen, name, L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3], **kw)
  o = OracleLib(name)
  rng = np.random.default_rng(seed)
  x0 = np.tile(L.initial_x, (n, 1))
  e = rng.uniform(-0.05, 0.05, size=(n, 3))
  q = np.concatenate([np.ones((n, 1)), e / 2], axis=1)
  x0[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
  hacc = np.zeros(3)
  o.call("h_10", L.initial_x.copy(), np.zeros(1), hacc)
  return torch, L, f, o, rng, x0, hacc


def _observations(L, rng, hacc, kinds, n, outlier_frac=0.0):
  T = len(kinds)
  zs = rng.normal(size=(T, n, 3))
  zs[kinds == 4] *= 0.025
  zs[kinds == 10] = zs[kinds == 10] * 0.5 + hacc
  zs[kinds == 12] = zs[kinds == 12] * 5.0 + L.initial_x[:3]
  if outlier_frac > 0:
    gi = np.where(kinds == 12)[0]
    sel = rng.random(size=(len(gi), n)) < outlier_frac
    zs[gi] += sel[..., None] * rng.normal(size=(len(gi), n, 3)) * 500.0
  return zs


def _Rtable(L, kinds):
  Rt = np.zeros((len(kinds), 9))
  for t, k in enumerate(kinds):
    Rt[t] = np.asarray(L.obs_noise[int(k)]).reshape(-1)
  return Rt


def _report(key, **vals):
  """Measured numbers of the full-size runs next to the profiles (gpurun_out/ is merged back by gpurun)."""
  d = os.path.join(REPO, "gpurun_out")
  if os.path.isdir(d):
    fn = os.path.join(d, "fullsize_parity.json")
    rec = {}
    if os.path.exists(fn):
      with open(fn, encoding="utf-8") as fh:
        rec = json.load(fh)
    rec[key] = vals
    with open(fn, "w", encoding="utf-8") as fh:
      json.dump(rec, fh, indent=1)


def test_config3_reference_stream_2100_steps():
  """One 2 100-step stream of the reference's numpy path, replicated over a batch: step-granular launches AND the fused run."""
  g = golden("live_stream_2100.npz")
  torch, L, f, _, _, _, _ = _setup("live", 96, 0)
  kinds, ts, idx = g["kinds"].astype(np.int32), g["ts"], g["idx"]
  n = f.batch
  zs = np.tile(g["zs"][:, None, :], (1, n, 1))
  Rs = {int(k): L.obs_noise[int(k)] for k in (4, 10, 12)}
  # fused run in segments ending at the kept steps
  f.init_state(g["x0"], g["P0"], None)
  prev = 0
  for a, stop in enumerate(idx):
    if stop + 1 > prev:
      f.run(ts[prev:stop + 1], kinds[prev:stop + 1], zs[prev:stop + 1].copy(), Rs)
      prev = stop + 1
    X, P = f.state(), f.covs()
    for j in (0, n - 1):
      assert_close(X[j], g["xs"][a], rtol=1e-8, floor=1e-8, what=f"fused run, state at step {stop}")
      assert_close(P[j].reshape(1, -1), g["Ps"][a].reshape(1, -1), rtol=1e-8, floor=1e-8, what=f"fused run, covariance at step {stop}")
  # step-granular launches over the whole stream
  s = _setup("live", 96, 0)[2]
  s.init_state(g["x0"], g["P0"], None)
  zd = torch.as_tensor(zs, device=s.device)
  for t in range(len(kinds)):
    s.predict_and_update_batch(float(ts[t]), int(kinds[t]), zd[t], Rs[int(kinds[t])])
  X, P = s.state(), s.covs()
  assert_close(X[0], g["xs"][-1], rtol=1e-8, floor=1e-8, what="step-granular, final state")
  assert_close(P[0].reshape(1, -1), g["Ps"][-1].reshape(1, -1), rtol=1e-8, floor=1e-8, what="step-granular, final covariance")
  assert np.array_equal(X, np.tile(X[0], (n, 1))), "identical filters must stay identical"


def test_config3_full_size_vs_oracle():
  """16 384 filters x 2 100 steps, every filter its own attitude error and noise: the fused run in 100-step segments against
  the oracle on identical inputs; 1 024 filters compared at every segment boundary, all of them at the end."""
  torch, L, f, o, rng, x0, hacc = _setup("live", N, 2025)
  kinds, ts = _schedule(T_FULL)
  Rs = {int(k): L.obs_noise[int(k)] for k in (4, 10, 12)}
  P0 = np.diag(L.initial_P_diag)
  f.init_state(x0, P0, None)
  xr, Pr = x0.copy(), np.tile(P0, (N, 1, 1))
  sub = np.sort(rng.choice(N, size=1024, replace=False))
  t_prev = ts[0]
  worst_x = worst_P = 0.0
  for lo in range(0, T_FULL, 100):
    hi = lo + 100
    zs = _observations(L, rng, hacc, kinds[lo:hi], N)
    f.run(ts[lo:hi], kinds[lo:hi], zs.copy(), Rs)
    dts = np.diff(np.concatenate([[t_prev], ts[lo:hi]]))
    t_prev = ts[hi - 1]
    o.batch_run(kinds[lo:hi], dts, xr, Pr, zs, _Rtable(L, kinds[lo:hi]), L.Q, quat_idx=3)
    X = f.x[torch.as_tensor(sub, device=f.device)].cpu().numpy()
    P = f.P[torch.as_tensor(sub, device=f.device)].cpu().numpy().reshape(len(sub), -1)
    worst_x = max(worst_x, (np.abs(X - xr[sub]) / np.abs(xr[sub]).max(axis=1, keepdims=True)).max())
    worst_P = max(worst_P, (np.abs(P - Pr[sub].reshape(len(sub), -1)) / np.abs(Pr[sub]).reshape(len(sub), -1).max(axis=1, keepdims=True)).max())
    assert_close(X, xr[sub], rtol=1e-8, floor=1e-8, what=f"subset states after step {hi}")
    assert_close(P, Pr[sub].reshape(len(sub), -1), rtol=1e-8, floor=1e-8, what=f"subset covariances after step {hi}")
  X, P = f.state(), f.covs().reshape(N, -1)
  assert_close(X, xr, rtol=1e-8, floor=1e-8, what="all 16 384 final states")
  assert_close(P, Pr.reshape(N, -1), rtol=1e-8, floor=1e-8, what="all 16 384 final covariances")
  assert np.abs(np.linalg.norm(X[:, 3:7], axis=1) - 1).max() < 1e-14
  _report("config3", filters=N, steps=T_FULL, worst_rel_err_x=worst_x, worst_rel_err_P=worst_P, tolerance=1e-8)


def test_config4_full_size_gate_and_smoother():
  """live with the gate, 2 % outliers, 16 384 x 2 100: smooth() sweeps the batch in chunks of 2 048 filters (forward run
  keeping the trace + gate flags, backward pass).  Gate decisions of all 34 M steps and the final filtered state of all filters
  against the oracle; the smoothed trajectory of 6 filters over all steps against the host restatement of the reference's
  rts_smooth on the oracle's estimates; for every filter: finite, unit quaternions, trace(P_smoothed) <= trace(P_filtered)."""
  from rednose_amd.helpers.ekf_sym import EKF_sym
  torch, L, f, o, rng, x0, hacc = _setup("live_maha", N, 4242)
  kinds, ts = _schedule(T_FULL)
  Rs = {int(k): L.obs_noise[int(k)] for k in (4, 10, 12)}
  P0 = np.diag(L.initial_P_diag)
  zs = _observations(L, rng, hacc, kinds, N, outlier_frac=0.02)
  f.init_state(x0, P0, None)
  pick = np.array([0, 1, 2047, 2048, 9000, N - 1])
  got = dict(flags=np.zeros((T_FULL, N), dtype=np.uint8), xs={}, Ps={}, ok=True, tr_ok=True)

  def on_chunk(lo, hi, xs, Ps, ys, fl):
    got["flags"][:, lo:hi] = fl.cpu().numpy()
    got["ok"] = got["ok"] and bool(torch.isfinite(xs).all()) and bool(torch.isfinite(Ps).all())
    qn = torch.linalg.norm(xs[1:, :, 3:7], dim=-1)
    got["ok"] = got["ok"] and bool(((qn - 1).abs() < 1e-13).all())
    for j in pick:
      if lo <= j < hi:
        got["xs"][int(j)] = xs[:, j - lo].cpu().numpy()
        got["Ps"][int(j)] = Ps[:, j - lo].cpu().numpy()

  f.smooth(ts, kinds, torch.as_tensor(zs, device=f.device), Rs, chunk=2048, on_chunk=on_chunk, flags=True)
  torch.cuda.synchronize()
  assert got["ok"], "non-finite or un-normalised smoothed estimates"
  # ---- forward pass of ALL filters vs oracle: gate decisions and final state ----
  xr, Pr, zr = x0.copy(), np.tile(P0, (N, 1, 1)), zs.copy()
  flr = np.zeros((T_FULL, N), dtype=np.uint8)
  o.batch_run(kinds, np.diff(np.concatenate([[ts[0]], ts])), xr, Pr, zr, _Rtable(L, kinds), L.Q, quat_idx=3, flags=flr)
  flips = int(np.sum((got["flags"] & 1) != flr))
  gnss = kinds == 12
  assert flips == 0, f"{flips} of {flr.size} gate decisions differ from the oracle"
  assert 0.02 < flr[gnss].mean() < 0.12 and not flr[~gnss].any()
  assert not (got["flags"] & 2).any()
  assert_close(f.state(), xr, rtol=1e-8, floor=1e-8, what="final filtered states, all filters")
  assert_close(f.covs().reshape(N, -1), Pr.reshape(N, -1), rtol=1e-8, floor=1e-8, what="final filtered covariances, all filters")
  # ---- smoother of the picked filters vs the reference algorithm on the oracle's estimates ----
  m = len(pick)
  xs0, Ps0, zp = x0[pick].copy(), np.tile(P0, (m, 1, 1)), zs[:, pick].copy()
  xp = np.zeros((T_FULL, m, 23)); Pp = np.zeros((T_FULL, m, 22, 22)); xf = np.zeros_like(xp); Pf = np.zeros_like(Pp)
  o.batch_run(kinds, np.diff(np.concatenate([[ts[0]], ts])), xs0, Ps0, zp, _Rtable(L, kinds), L.Q, quat_idx=3, xp=xp, Pp=Pp, xf=xf, Pf=Pf)
  host = EKF_sym(os.path.dirname(o.path), "live_maha", L.Q, L.initial_x, P0, 23, 22, quaternion_idxs=[3], maha_test_kinds=[12])
  worst = {}
  for a, j in enumerate(pick):
    est = [(xp[t, a], xf[t, a], Pp[t, a], Pf[t, a], ts[t], int(kinds[t]), None, None, None) for t in range(T_FULL)]
    xs_ref, Ps_ref = host.rts_smooth(est, norm_quats=True)
    X, P = got["xs"][int(j)], got["Ps"][int(j)]
    ex = (np.abs(X - xs_ref) / np.abs(xs_ref).max(axis=1, keepdims=True)).max()
    eP = (np.abs(P - Ps_ref).reshape(T_FULL, -1) / np.abs(Ps_ref).reshape(T_FULL, -1).max(axis=1, keepdims=True)).max()
    worst[int(j)] = (float(ex), float(eP))
    assert_close(X, xs_ref, rtol=1e-6, floor=1e-6, what=f"smoothed states of filter {j}")
    assert_close(P.reshape(T_FULL, -1), Ps_ref.reshape(T_FULL, -1), rtol=1e-5, floor=1e-5, what=f"smoothed covariances of filter {j}")
    tr_s = np.trace(P, axis1=1, axis2=2); tr_f = np.trace(Pf[:, a], axis1=1, axis2=2)
    assert (tr_s[:-1] <= tr_f[:-1] * (1 + 1e-9)).all()
  _report("config4", filters=N, steps=T_FULL, gate_flips=flips, gated_fraction_of_gnss=float(flr[gnss].mean()),
          smoother_worst_rel_err={str(k): v for k, v in worst.items()})
