#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON PATH in this container.

What runs: /root/reference/rednose/helpers/ekf_sym.py `EKF_sym` (the pure-Python orchestrator) with
its numpy math selected (`_predict_python` :533-559, `_update_python` :561-624 -- the two lines the
reference leaves commented at :346-349), `rts_smooth` :651-690 and `maha_test` :626-649, on top of
the reference-GENERATED sympy C (oracle/_ref/lib{name}.so; the f/F/h/H/H_mod/err functions are the
reference's own codegen output).  cffi is absent, so oracle/cffi_shim provides the three cffi calls
the class uses.  Nothing of ours is in the numerical path of these vectors except gcc.

The fixtures pin: (1) oracle/ekf_oracle.c (tests/test_oracle.py) and (2) the HIP kernels
(tests/test_gpu_parity.py).  /root/reference does not exist on the GPU box, hence committed vectors.

Run:  python oracle/make_golden.py        (needs /root/reference; rewrites tests/golden/)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "cffi_shim"), REF, HERE, REPO]

import build_oracle  # noqa: E402
from rednose.helpers.ekf_sym import EKF_sym as RefEKF  # noqa: E402  (the reference class)

GOLD = os.path.join(REPO, "tests", "golden")
REFDIR = os.path.join(HERE, "_ref")


def ref_filter(name, Q, x0, P0, D, E, numpy_math=True, **kw):
  if not os.path.exists(os.path.join(REFDIR, f"lib{name}.so")):
    build_oracle.build(name, "ref")
  f = RefEKF(REFDIR, name, Q, x0, P0, D, E, **kw)
  if numpy_math:
    f._predict = f._predict_python   # pylint: disable=protected-access
    f._update = f._update_python     # pylint: disable=protected-access
  return f


def kinematic_stream():
  """The known-answer scenario of /root/reference/examples/test_kinematic_kf.py:11-48, full trajectory."""
  Q = np.diag([0.1**2, 2.0**2]); x0 = np.array([0.5, 0.0]); P0 = np.diag([1.0, 1.0]); R = np.array([[[0.1**2]]])
  out = {}
  for tag, numpy_math in (("numpy", True), ("c", False)):
    np.random.seed(0)
    f = ref_filter("kinematic", Q, x0, P0, 2, 2, numpy_math=numpy_math)
    dt = 0.01
    ts = np.arange(0, 5, step=dt)
    xs, Ps, zs = [], [], []
    x = 0.0
    for t, v in zip(ts, np.sin(ts * 5)):
      meas = np.random.normal(x, 0.1)
      zs.append(meas)
      f.predict_and_update_batch(t, 1, np.array([[meas]]), R)
      xs.append(f.state().copy()); Ps.append(f.covs().copy())
      x += v * dt
    out[tag] = (np.array(xs), np.array(Ps))
    out["ts"], out["zs"] = ts, np.array(zs)
  xs, Ps = out["numpy"]
  print("kinematic final (reference numpy path):", xs[-1], np.sqrt(np.diag(Ps[-1])))
  print("  literals in test_kinematic_kf.py:52-55: -0.010866289677966417 0.04477103863330089 -0.8553720537261753 0.6695762270974388")
  print("  |numpy - C restatement| max:", np.abs(out["numpy"][0] - out["c"][0]).max(), np.abs(out["numpy"][1] - out["c"][1]).max())
  np.savez_compressed(os.path.join(GOLD, "kinematic_stream.npz"), ts=out["ts"], zs=out["zs"], xs=xs, Ps=Ps,
                      literals=np.array([-0.010866289677966417, 0.04477103863330089, -0.8553720537261753, 0.6695762270974388]))


def compare_rewind():
  """/root/reference/examples/test_compare.py:88-120: samples 20 and 40 swapped => rewind + fast-forward."""
  Q = np.diag([0.1**2, 2.0**2]); x0 = np.array([0.5, 0.0]); P0 = np.diag([1.0, 1.0]); R = np.array([[[0.1**2]]])
  np.random.seed(0)
  f = ref_filter("compare", Q, x0, P0, 2, 2)
  dt = 0.01
  ts = np.arange(0, 5, step=dt)
  xs_true = np.empty(ts.shape)
  x = 0.0
  for i, v in enumerate(np.sin(ts * 5)):
    xs_true[i] = x
    x += v * dt
  a, b = 20, 40
  ts[a], ts[b] = ts[b], ts[a]
  xs_true[a], xs_true[b] = xs_true[b], xs_true[a]
  xs, Ps, fts, zs = [], [], [], []
  for t, xt in zip(ts, xs_true):
    meas = np.random.normal(xt, 0.1)
    zs.append(meas)
    f.predict_and_update_batch(t, 1, np.array([[meas]]), R)
    xs.append(f.state().copy()); Ps.append(f.covs().copy()); fts.append(f.get_filter_time())
  np.savez_compressed(os.path.join(GOLD, "compare_rewind.npz"), ts=ts, zs=np.array(zs), xs=np.array(xs), Ps=np.array(Ps),
                      filter_times=np.array(fts))


def _live_setup():
  sys.path.insert(0, REPO)
  from examples.live_kf import LiveKalman  # our restatement of the model constants (x0, P0, Q, R)
  return LiveKalman


def _random_spd(rng, diag):
  E = diag.shape[0]
  A = rng.normal(size=(E, E)) * 0.15
  C = np.eye(E) + A @ A.T
  s = np.sqrt(diag)
  return (C * s[:, None]) * s[None, :]


def live_single_steps():
  """Random state / covariance, one predict and one update per kind: reference numpy math, single calls."""
  L = _live_setup()
  rng = np.random.default_rng(7)
  f = ref_filter("live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22)
  kinds = [3, 4, 9, 10, 12, 13, 14, 19]
  recs = {}
  n = 6
  X = np.tile(L.initial_x, (n, 1))
  X[:, 0:3] += rng.normal(size=(n, 3)) * 100
  q = rng.normal(size=(n, 4)) * 0.3 + np.array([1, 0, 0, 0]); X[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
  X[:, 7:10] = rng.normal(size=(n, 3)) * 5
  X[:, 10:13] = rng.normal(size=(n, 3)) * 0.1
  X[:, 13:16] = rng.normal(size=(n, 3)) * 0.01
  X[:, 16] = 1 + rng.normal(size=n) * 0.01
  X[:, 17:20] = rng.normal(size=(n, 3)) * 0.5
  X[:, 20:23] = rng.normal(size=(n, 3)) * 0.02
  Pdiag = np.array(L.initial_P_diag) * np.array([1e-4] * 3 + [1e-3] * 3 + [1e-2] * 3 + [1.0] * 13)
  Ps = np.stack([_random_spd(rng, Pdiag) for _ in range(n)])
  recs["x_in"], recs["P_in"] = X, Ps
  dts = np.array([0.0, 0.01, 0.05, 0.1, 0.01, 0.02])
  xo, Po = [], []
  for i in range(n):
    x, P = f._predict_python(X[i].reshape(-1, 1).copy(), Ps[i].copy(), dts[i])  # pylint: disable=protected-access
    xo.append(x.flatten()); Po.append(P)
  recs["predict_dt"], recs["predict_x"], recs["predict_P"] = dts, np.array(xo), np.array(Po)
  for k in kinds:
    Z = 1 if k == 3 else 3
    R = np.atleast_2d(L.obs_noise.get(k, np.eye(Z) * 0.1))
    zs = rng.normal(size=(n, Z))
    xo, Po, yo = [], [], []
    for i in range(n):
      h = np.zeros((Z, 1)); f.hs[k](X[i].copy(), np.zeros(1), h)
      z = h.flatten() + zs[i] * np.sqrt(np.diag(R)) * 2
      zs[i] = z
      x, P, y = f._update_python(X[i].reshape(-1, 1).copy(), Ps[i].copy(), k, z.copy(), R.copy(), extra_args=np.zeros(1))  # pylint: disable=protected-access
      xo.append(np.asarray(x).flatten()); Po.append(P); yo.append(y)
    recs[f"upd{k}_z"], recs[f"upd{k}_R"] = zs, R
    recs[f"upd{k}_x"], recs[f"upd{k}_P"], recs[f"upd{k}_y"] = np.array(xo), np.array(Po), np.array(yo)
  np.savez_compressed(os.path.join(GOLD, "live_single_steps.npz"), **recs)


def live_stream(n_ticks=40, out_name="live_stream.npz", keep_every=7, seed=2025):
  """IMU@100 Hz (gyro then accel at the same t) + ECEF_POS every 10th tick, stationary device, reference numpy
  path with quaternion_idxs=[3].  To get the C++ orchestrator's behaviour (renormalise after predict AND after
  update, /root/reference/rednose/helpers/ekf_sym.cc:207,213) from the Python class, each observation is applied
  as  f.predict(t)  [normalises, ekf_sym.py:461]  followed by predict_and_update_batch(t, ...) whose internal
  predict then has dt == 0 and is an exact identity (F = I, dt*Q = 0)."""
  L = _live_setup()
  rng = np.random.default_rng(seed)
  f = ref_filter("live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, quaternion_idxs=[3])
  x_true = L.initial_x.copy()
  x0 = L.initial_x.copy()
  e = rng.uniform(-0.05, 0.05, size=3)
  q = np.array([1.0, e[0] / 2, e[1] / 2, e[2] / 2]); x0[3:7] = q / np.linalg.norm(q)
  P0 = np.diag(L.initial_P_diag)
  f.init_state(x0, P0, None)
  hacc = np.zeros((3, 1)); f.hs[10](x_true.copy(), np.zeros(1), hacc)
  sched, zs, ts = [], [], []
  for tick in range(n_ticks):
    t = 0.01 * tick
    sched.append(4); ts.append(t); zs.append(rng.normal(size=3) * 0.025)
    sched.append(10); ts.append(t); zs.append(hacc.flatten() + rng.normal(size=3) * 0.5)
    if tick % 10 == 9:
      sched.append(12); ts.append(t); zs.append(x_true[0:3] + rng.normal(size=3) * 5)
  est = []
  xs, Ps, ys, xps = [], [], [], []
  for k, t, z in zip(sched, ts, zs):
    f.predict(t)
    xps.append(f.state().copy())
    r = f.predict_and_update_batch(t, k, np.array([z]), np.array([L.obs_noise[k]]))
    est.append(r)
    xs.append(f.state().copy()); Ps.append(f.covs().copy()); ys.append(np.asarray(r[6][0]).flatten())
  xs, Ps = np.array(xs), np.array(Ps)
  keep = np.arange(0, len(sched), keep_every).tolist() + [len(sched) - 1]
  if out_name == "live_stream.npz":
    np.savez_compressed(os.path.join(GOLD, out_name), x0=x0, P0=P0, kinds=np.array(sched), ts=np.array(ts),
                        zs=np.array(zs), ys=np.array(ys), xs=xs, x_pred=np.array(xps), P_idx=np.array(keep), Ps=Ps[keep])
  else:      # long stream (BASELINE config 3: 10 s, 2 100 steps): states / covariances at the kept steps only
    np.savez_compressed(os.path.join(GOLD, out_name), x0=x0, P0=P0, kinds=np.array(sched, dtype=np.int32), ts=np.array(ts),
                        zs=np.array(zs), idx=np.array(keep), xs=xs[keep], Ps=Ps[keep])
  print("live stream:", len(sched), "steps; final pos err", xs[-1][:3] - x_true[:3], "quat", xs[-1][3:7])
  return f, est


def rts_goldens():
  """rts_smooth (ekf_sym.py:651-690) on (a) the kinematic stream, (b) a live stream, reference numpy path.
  rts_smooth aliases and mutates its input estimates (xk_n = xk_k, Pk_n = Pk_k) -- deep copies go in."""
  import copy
  Q = np.diag([0.1**2, 2.0**2]); x0 = np.array([0.5, 0.0]); P0 = np.diag([1.0, 1.0]); R = np.array([[[0.1**2]]])
  np.random.seed(0)
  f = ref_filter("kinematic", Q, x0, P0, 2, 2)
  dt = 0.01
  ts = np.arange(0, 1.2, step=dt)
  est = []
  x = 0.0
  for t, v in zip(ts, np.sin(ts * 5)):
    est.append(f.predict_and_update_batch(t, 1, np.array([[np.random.normal(x, 0.1)]]), R))
    x += v * dt
  pack = lambda es: dict(xk_km1=np.array([e[0] for e in es]), xk_k=np.array([e[1] for e in es]),  # noqa: E731
                         Pk_km1=np.array([e[2] for e in es]), Pk_k=np.array([e[3] for e in es]),
                         t=np.array([e[4] for e in es]))
  inp = pack(est)
  xs_s, Ps_s = f.rts_smooth(copy.deepcopy(est), norm_quats=False)
  np.savez_compressed(os.path.join(GOLD, "kinematic_rts.npz"), xs_smooth=xs_s, Ps_smooth=Ps_s, **inp)

  fl, est_l = live_stream(n_ticks=30)
  inp = pack(est_l)
  xs_s, Ps_s = fl.rts_smooth(copy.deepcopy(est_l), norm_quats=True)
  keep = np.arange(0, len(est_l), 5)
  np.savez_compressed(os.path.join(GOLD, "live_rts.npz"), xs_smooth=xs_s, Ps_smooth_idx=keep, Ps_smooth=Ps_s[keep],
                      xk_km1=inp["xk_km1"], xk_k=inp["xk_k"], t=inp["t"], Pk_km1=inp["Pk_km1"].astype(np.float64),
                      Pk_k=inp["Pk_k"])


def kinematic9_goldens(T=60):
  """Mixed-kind stream (POSITION / RANGE / VELOCITY) of the 9-state constant-acceleration example through the
  reference's numpy predict/update and rts_smooth: golden for the mid-size (7 filters per wavefront) kernels."""
  import copy
  import importlib.util   # by path: `examples` on this script's sys.path is the reference's package
  spec = importlib.util.spec_from_file_location("rn_amd_kinematic9_kf", os.path.join(REPO, "examples", "kinematic9_kf.py"))
  mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
  K9, ANCHOR = mod.Kinematic9Kalman, mod.ANCHOR
  rng = np.random.default_rng(9)
  f = ref_filter("kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9)
  truth = np.array([0.5, 0.5, 0.5, 1.0, -0.5, 0.2, 0.3, 0.1, -0.2])
  t, est, kinds, zs = 0.0, [], [], []
  for i in range(T):
    dt = float(rng.uniform(0.01, 0.05))
    t += dt
    truth[0:3] += dt * truth[3:6]; truth[3:6] += dt * truth[6:9]
    k = (1, 2, 3)[i % 3]
    if k == 1:
      z = truth[0:3] + rng.normal(size=3) * 0.1
    elif k == 2:
      z = np.array([np.linalg.norm(truth[0:3] - np.array(ANCHOR))]) + rng.normal(size=1) * 0.2
    else:
      z = truth[3:6] + rng.normal(size=3) * 0.3
    est.append(f.predict_and_update_batch(t, k, np.array([z]), np.array([K9.obs_noise[k]])))
    kinds.append(k); zs.append(np.concatenate([z, np.zeros(3 - len(z))]))
  xs_s, Ps_s = f.rts_smooth(copy.deepcopy(est), norm_quats=False)
  # "multiple forward and backwards passes of the data" (/root/reference/README.md:41-45) with the reference class: each further
  # pass restarts the filter from the oldest smoothed estimate of the previous pass and runs the same observations again
  ts_all = [e[4] for e in est]
  passes = [(xs_s.copy(), Ps_s.copy())]
  for _ in range(2):
    f.init_state(passes[-1][0][0].copy(), passes[-1][1][0].copy(), None)
    est_p = [f.predict_and_update_batch(tt, kk, np.array([zz[:K9.obs_noise[kk].shape[0]]]), np.array([K9.obs_noise[kk]]))
             for tt, kk, zz in zip(ts_all, kinds, zs)]
    xp, Pp = f.rts_smooth(copy.deepcopy(est_p), norm_quats=False)
    passes.append((xp.copy(), Pp.copy()))
  np.savez_compressed(os.path.join(GOLD, "kinematic9_multipass.npz"), xs_pass2=passes[1][0], Ps_pass2=passes[1][1],
                      xs_pass3=passes[2][0], Ps_pass3=passes[2][1])
  np.savez_compressed(os.path.join(GOLD, "kinematic9_stream.npz"), kinds=np.array(kinds), zs=np.array(zs),
                      ts=np.array([e[4] for e in est]), xk_km1=np.array([e[0] for e in est]), xk_k=np.array([e[1] for e in est]),
                      Pk_km1=np.array([e[2] for e in est]), Pk_k=np.array([e[3] for e in est]),
                      ys=np.array([np.concatenate([np.ravel(e[6][0]), np.zeros(3 - len(np.ravel(e[6][0])))]) for e in est]),
                      xs_smooth=xs_s, Ps_smooth=Ps_s)


def feature_goldens(T=45, cls_name="FeatureKalman", out_name="feature_stream.npz", n_upd=24):
  """MSCKF stream of the windowed-camera example (examples/feature_kf.py) through the reference's numpy path:
  block-structured predict (ekf_sym.py:541-556), null-space projected FEATURE updates (:576-591), POSITION updates
  and augment() window shifts (:365-391).  Also single feature updates from random states."""
  import importlib.util
  spec = importlib.util.spec_from_file_location("rn_amd_feature_kf", os.path.join(REPO, "examples", "feature_kf.py"))
  mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
  FK = getattr(mod, cls_name)
  rng = np.random.default_rng(21)
  kw = FK.filter_kwargs()
  D = FK.dim_state
  ZF = 2 * len(FK.observed)
  f = ref_filter(FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), mod.DIM_MAIN, mod.DIM_MAIN, **kw)
  landmarks = np.array([[2.0, 1.0, 8.0], [-1.5, 0.5, 6.0], [0.5, -1.0, 10.0], [3.0, 2.0, 12.0]])
  truth_p, truth_v = np.zeros(3), np.array([1.0, 0.5, 0.0])
  window = [np.zeros(3)] * FK.n_window
  recs = dict(kinds=[], ts=[], zs=[], eas=[], augment=[], xk_k=[], Pk_k=[], xk_km1=[], Pk_km1=[], ys=[])
  ests = []
  t = 0.0
  for i in range(T):
    t += 0.05
    truth_p = truth_p + 0.05 * truth_v
    if i % 3 == 0:
      kind, ea, aug = 1, np.zeros(3), True                         # position fix, then push it into the window
      z = truth_p + rng.normal(size=3) * 0.2
    else:
      kind, aug = 2, False
      ea = landmarks[i % len(landmarks)] + rng.normal(size=3) * 0.05
      rays = [landmarks[i % len(landmarks)] - window[w] for w in FK.observed]
      z = np.concatenate([[r[0] / r[2], r[1] / r[2]] for r in rays]) + rng.normal(size=ZF) * 0.01
    R = FK.obs_noise[kind] if kind == 1 else np.eye(ZF) * 0.01**2
    est = f.predict_and_update_batch(t, kind, np.array([z]), np.array([R]), [list(ea)] if kind == 2 else [[]], augment=aug)
    if aug:
      window = window[1:] + [truth_p.copy()]
    ests.append(est)
    y = np.ravel(est[6][0])
    recs["kinds"].append(kind); recs["ts"].append(t); recs["zs"].append(np.concatenate([z, np.zeros(ZF - len(z))]))
    recs["eas"].append(ea); recs["augment"].append(aug)
    recs["xk_km1"].append(est[0]); recs["xk_k"].append(est[1]); recs["Pk_km1"].append(est[2]); recs["Pk_k"].append(est[3])
    recs["ys"].append(np.concatenate([y, np.zeros(ZF - len(y))]))
    recs.setdefault("x_after", []).append(f.state().copy()); recs.setdefault("P_after", []).append(f.covs().copy())
  # single FEATURE updates from random states (reference numpy update, no predict)
  n = n_upd
  xs = np.tile(FK.initial_x, (n, 1)) + rng.normal(size=(n, D)) * 0.3
  Ps = []
  for _ in range(n):
    A = rng.normal(size=(D, D)) * 0.2
    Ps.append(np.diag(FK.initial_P_diag) + A @ A.T)
  Ps = np.array(Ps)
  eas = landmarks[rng.integers(0, 4, size=n)] + rng.normal(size=(n, 3)) * 0.2
  zs = rng.normal(size=(n, ZF)) * 0.3
  ux, uP, uy = [], [], []
  for i in range(n):
    x1, P1, y1 = f._update_python(xs[i].reshape(-1, 1).copy(), Ps[i].copy(), 2, zs[i].copy(), np.eye(ZF) * 0.01**2, extra_args=eas[i].copy())  # pylint: disable=protected-access
    ux.append(np.ravel(x1)); uP.append(P1); uy.append(np.ravel(y1))
  out = {k: np.array(v) for k, v in recs.items()}
  # the reference's smoother on this MSCKF trajectory: only the main block / main states are smoothed (ekf_sym.py:675-686).
  # It aliases and mutates its input -- a deep copy goes in.
  import copy
  xs_smooth, Ps_smooth = f.rts_smooth(copy.deepcopy(ests), norm_quats=False)
  np.savez_compressed(os.path.join(GOLD, out_name), **out, upd_x_in=xs, upd_P_in=Ps, upd_ea=eas, upd_z=zs,
                      upd_x=np.array(ux), upd_P=np.array(uP), upd_y=np.array(uy), xs_smooth=xs_smooth, Ps_smooth=Ps_smooth)
  print("feature stream: final x[:6]", out["x_after"][-1][:6], "y dims", {len(np.ravel(e)) for e in uy})


def attitude_goldens(T=60):
  """7/6-state attitude ESKF (examples/attitude_kf.py): gyro every step, gravity direction every third, a rotating body;
  reference numpy path with quaternion_idxs=[0], C++ orchestration order obtained as in live_stream (predict(t), then
  predict_and_update_batch at the same t), then rts_smooth with norm_quats (it normalises x[3:7] whatever the model,
  ekf_sym.py:666-667 -- for this state layout that slice is NOT the quaternion, which the golden therefore also pins)."""
  import copy
  import importlib.util
  spec = importlib.util.spec_from_file_location("rn_amd_attitude_kf", os.path.join(REPO, "examples", "attitude_kf.py"))
  mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
  AK = mod.AttitudeKalman
  rng = np.random.default_rng(77)
  f = ref_filter("attitude", AK.Q, AK.initial_x, np.diag(AK.initial_P_diag), 7, 6, quaternion_idxs=[0])
  x0 = AK.initial_x.copy()
  e = rng.uniform(-0.1, 0.1, size=3)
  q = np.array([1.0, e[0] / 2, e[1] / 2, e[2] / 2]); x0[0:4] = q / np.linalg.norm(q)
  P0 = np.diag(AK.initial_P_diag)
  f.init_state(x0, P0, None)
  w_true = np.array([0.3, -0.2, 0.5])
  q_true = np.array([1.0, 0.0, 0.0, 0.0])
  g_w = np.array(AK.gravity_world)

  def rot(qq):
    a, b, c, d = qq
    return np.array([[a*a + b*b - c*c - d*d, 2*(b*c - a*d), 2*(b*d + a*c)],
                     [2*(b*c + a*d), a*a - b*b + c*c - d*d, 2*(c*d - a*b)],
                     [2*(b*d - a*c), 2*(c*d + a*b), a*a - b*b - c*c + d*d]])

  kinds, ts, zs, est, xs, Ps, ys = [], [], [], [], [], [], []
  t = 0.0
  for i in range(T):
    t += 0.02
    wq = np.array([0.0, *w_true])
    dq = 0.5 * np.array([-q_true[1]*wq[1] - q_true[2]*wq[2] - q_true[3]*wq[3],
                          q_true[0]*wq[1] + q_true[2]*wq[3] - q_true[3]*wq[2],
                          q_true[0]*wq[2] - q_true[1]*wq[3] + q_true[3]*wq[1],
                          q_true[0]*wq[3] + q_true[1]*wq[2] - q_true[2]*wq[1]])
    q_true = q_true + 0.02 * dq; q_true /= np.linalg.norm(q_true)
    if i % 3 == 2:
      k, z = 2, rot(q_true).T @ g_w + rng.normal(size=3) * 0.3
    else:
      k, z = 1, w_true + rng.normal(size=3) * 0.02
    f.predict(t)
    r = f.predict_and_update_batch(t, k, np.array([z]), np.array([AK.obs_noise[k]]))
    est.append(r); kinds.append(k); ts.append(t); zs.append(z)
    xs.append(f.state().copy()); Ps.append(f.covs().copy()); ys.append(np.asarray(r[6][0]).flatten())
  xs_s, Ps_s = f.rts_smooth(copy.deepcopy(est), norm_quats=False)
  np.savez_compressed(os.path.join(GOLD, "attitude_stream.npz"), x0=x0, P0=P0, kinds=np.array(kinds), ts=np.array(ts), zs=np.array(zs),
                      ys=np.array(ys), xs=np.array(xs), Ps=np.array(Ps), xk_km1=np.array([e_[0] for e_ in est]),
                      Pk_km1=np.array([e_[2] for e_ in est]), xs_smooth=xs_s, Ps_smooth=Ps_s)
  print("attitude: final quat", xs[-1][:4], "truth", q_true, "rate", xs[-1][4:])


def maha_goldens():
  """Gate DECISIONS of the reference's maha_test (ekf_sym.py:626-649; threshold = chi2_ppf(0.95, Z) from the
  reference's lookup table) on live ECEF_POS observations with and without gross outliers."""
  L = _live_setup()
  rng = np.random.default_rng(99)
  f = ref_filter("live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22)
  n = 64
  X = np.tile(L.initial_x, (n, 1)); X[:, 0:3] += rng.normal(size=(n, 3)) * 3
  Pdiag = np.array(L.initial_P_diag) * np.array([1e-6] * 3 + [1e-3] * 19)
  Ps = np.stack([_random_spd(rng, Pdiag) for _ in range(n)])
  R = L.obs_noise[12]
  z = X[:, 0:3] + rng.normal(size=(n, 3)) * 8
  z[::4] += rng.normal(size=(len(z[::4]), 3)) * 500
  ok = np.array([f.maha_test(X[i].reshape(-1, 1).copy(), Ps[i].copy(), 12, z[i].copy(), R.copy(), extra_args=np.zeros(1)) for i in range(n)])
  from rednose.helpers.chi2_lookup import chi2_ppf as ref_chi2
  np.savez_compressed(os.path.join(GOLD, "live_maha.npz"), x=X, P=Ps, z=z, R=R, accepted=ok,
                      thresholds=np.array([ref_chi2(0.95, d) for d in (1, 2, 3, 6)]))
  print("maha accepted", ok.sum(), "of", n)


def perfilter_timelines():
  """N independent instances of the reference's orchestrator, each fed its OWN out-of-order log (SURVEY.md 8f row 1;
  /root/reference/rednose/helpers/ekf_sym.py:418-482 = ekf_sym.cc:83-156): what a batched orchestrator with per-filter
  timelines must reproduce filter by filter.
  Part A, `compare` (2 states, one kind): the stream of /root/reference/examples/test_compare.py:88-120 extended to 700
  observations (the ring of 512 checkpoints wraps), every filter with a DIFFERENT pair of samples swapped (filter 0: none;
  some with two swaps; swaps beyond sample 512), one filter with an observation more than max_rewind_age old (ignored: None).
  Part B, `kinematic9` (9 states, three kinds of 3 / 1 / 3 dimensions): every filter has its own times, its own order of
  kinds, ticks without an observation, and one late observation at a place of its own."""
  Q = np.diag([0.1**2, 2.0**2]); x0 = np.array([0.5, 0.0]); P0 = np.diag([1.0, 1.0]); R = np.array([[[0.1**2]]])
  dt, T, NA = 0.01, 700, 12
  rng = np.random.default_rng(41)
  ts0 = np.arange(T) * dt
  truth = np.concatenate([[0.0], np.cumsum(np.sin(ts0 * 5) * dt)[:-1]])
  swaps = [[], [(20, 40)], [(21, 22)], [(100, 180)], [(5, 6), (300, 350)], [(250, 251), (400, 460)], [(511, 530)], [(600, 640)],
           [(650, 699)], [(10, 90), (520, 560)], [(333, 334)], [(40, 60)]]
  tA = np.tile(ts0, (NA, 1)); zA = np.empty((NA, T)); noneA = np.zeros((NA, T), dtype=bool)
  xA = np.empty((NA, T, 2)); PA = np.empty((NA, T, 2, 2)); ftA = np.empty((NA, T))
  for i in range(NA):
    order = np.arange(T)
    for a, b in swaps[i]:
      order[a], order[b] = order[b], order[a]
    tA[i] = ts0[order]
    if i == NA - 1:
      tA[i, 300] = ts0[300] - 2.5           # 2.5 s behind the filter: older than max_rewind_age = 1 s -> ignored
    zA[i] = truth[order] + rng.normal(size=T) * 0.1
    f = ref_filter("compare", Q, x0, P0, 2, 2)
    for j in range(T):
      ret = f.predict_and_update_batch(float(tA[i, j]), 1, np.array([[zA[i, j]]]), R)
      noneA[i, j] = ret is None
      xA[i, j], PA[i, j], ftA[i, j] = f.state(), f.covs(), f.get_filter_time()
  assert noneA.sum() == 1 and noneA[NA - 1, 300]
  # ---- part B
  import importlib.util
  spec = importlib.util.spec_from_file_location("rn_amd_kinematic9_kf_b", os.path.join(REPO, "examples", "kinematic9_kf.py"))
  mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
  K9, ANCHOR = mod.Kinematic9Kalman, mod.ANCHOR
  NB, TB = 10, 48
  tB = np.full((NB, TB), np.nan); kB = np.zeros((NB, TB), dtype=np.int32); zB = np.zeros((NB, TB, 3))
  xB = np.empty((NB, TB, 9)); PB = np.empty((NB, TB, 9, 9)); yB = np.zeros((NB, TB, 3))
  for i in range(NB):
    f = ref_filter("kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9)
    tr = np.array([0.5, 0.5, 0.5, 1.0, -0.5, 0.2, 0.3, 0.1, -0.2]) + rng.normal(size=9) * 0.05
    times = np.cumsum(rng.uniform(0.01, 0.05, size=TB)) + rng.uniform(0, 0.3)
    late_at = int(rng.integers(5, TB - 2))
    back = int(rng.integers(2, 5))
    times[late_at] = 0.5 * (times[late_at - back] + times[late_at - back + 1])      # arrives `back` observations late
    skip = rng.random(TB) < 0.2                                                     # ticks without an observation for this filter
    skip[late_at] = False
    t_prev = 0.0
    for j in range(TB):
      if skip[j]:
        xB[i, j], PB[i, j] = f.state(), f.covs()
        continue
      t = float(times[j])
      k = int(rng.integers(1, 4))
      # truth propagated to t from scratch (constant acceleration): observations need only be plausible
      p = tr[0:3] + t * tr[3:6] + 0.5 * t * t * tr[6:9]; v = tr[3:6] + t * tr[6:9]
      if k == 1:
        z = p + rng.normal(size=3) * 0.1
      elif k == 2:
        z = np.array([np.linalg.norm(p - np.array(ANCHOR))]) + rng.normal(size=1) * 0.2
      else:
        z = v + rng.normal(size=3) * 0.3
      ret = f.predict_and_update_batch(t, k, np.array([z]), np.array([K9.obs_noise[k]]))
      assert ret is not None
      tB[i, j], kB[i, j] = t, k
      zB[i, j, :len(z)] = z
      yB[i, j, :len(z)] = np.ravel(ret[6][0])
      xB[i, j], PB[i, j] = f.state(), f.covs()
      t_prev = t
  np.savez_compressed(os.path.join(GOLD, "perfilter_timelines.npz"), A_t=tA, A_z=zA, A_none=noneA, A_x=xA[:, ::25], A_P=PA[:, ::25],
                      A_keep=np.arange(T)[::25], A_x_final=xA[:, -1], A_P_final=PA[:, -1], A_ft=ftA,
                      B_t=tB, B_kind=kB, B_z=zB, B_x=xB, B_P=PB, B_y=yB)
  print("per-filter timelines: part A", NA, "x", T, "ignored", int(noneA.sum()), "; part B", NB, "x", TB, "observations", int((kB > 0).sum()))


def multi_obs_goldens():
  """n observations per call (the reference's predict_and_update_batch proper, ekf_sym.py:484-531 = ekf_sym.cc:158-194): ONE predict, n
  sequential updates, ONE checkpoint, y a list of n.
  Part A, `kinematic9`, per-filter logs: 8 instances of the reference class, each fed its own log of 36 calls -- own times, kinds
  1 / 2 / 3 at random, n in {1, 2, 3} observations per call (position fixes: mostly 3) with a DIFFERENT noise matrix per observation,
  ONE LATE multi-observation call per filter (rewind over 2-4 multi-observation checkpoints, then fast-forward through them).
  Part B, `feature` (MSCKF), shared timeline: 4 instances with the same call times / kinds and their own observations: every third call
  3 position fixes + window shift, the others 4 feature tracks (n = 4 landmarks as extra_args)."""
  import importlib.util
  rng = np.random.default_rng(77)
  spec = importlib.util.spec_from_file_location("rn_amd_kinematic9_kf_m", os.path.join(REPO, "examples", "kinematic9_kf.py"))
  mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
  K9, ANCHOR = mod.Kinematic9Kalman, mod.ANCHOR
  NB, TB, NM = 8, 36, 3
  tA = np.zeros((NB, TB)); kA = np.zeros((NB, TB), dtype=np.int32); nA = np.zeros((NB, TB), dtype=np.int32)
  zA = np.zeros((NB, TB, NM, 3)); sA = np.ones((NB, TB, NM)); yA = np.zeros((NB, TB, NM, 3))
  xA = np.empty((NB, TB, 9)); PA = np.empty((NB, TB, 9, 9)); x1A = np.empty((NB, TB, 9)); P1A = np.empty((NB, TB, 9, 9))
  xkA = np.empty((NB, TB, 9)); PkA = np.empty((NB, TB, 9, 9)); lateA = np.zeros(NB, dtype=np.int32)
  for i in range(NB):
    f = ref_filter("kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9)
    tr = np.array([0.5, 0.5, 0.5, 1.0, -0.5, 0.2, 0.3, 0.1, -0.2]) + rng.normal(size=9) * 0.05
    times = np.cumsum(rng.uniform(0.01, 0.05, size=TB)) + rng.uniform(0, 0.3)
    late_at = int(rng.integers(6, TB - 2))
    back = int(rng.integers(2, 5))
    times[late_at] = 0.5 * (times[late_at - back] + times[late_at - back + 1])
    lateA[i] = late_at
    for j in range(TB):
      t = float(times[j])
      k = int(rng.integers(1, 4))
      n = 3 if (k == 1 and rng.random() < 0.7) else int(rng.integers(1, 4))
      if j == late_at:
        n = max(n, 2)                      # the late call carries several observations
      p = tr[0:3] + t * tr[3:6] + 0.5 * t * t * tr[6:9]; v = tr[3:6] + t * tr[6:9]
      Z = K9.obs_noise[k].shape[0]
      zs, Rs = [], []
      for m in range(n):
        sc = float(rng.uniform(0.5, 2.0))
        if k == 1:
          z = p + rng.normal(size=3) * 0.1 * np.sqrt(sc)
        elif k == 2:
          z = np.array([np.linalg.norm(p - np.array(ANCHOR))]) + rng.normal(size=1) * 0.2 * np.sqrt(sc)
        else:
          z = v + rng.normal(size=3) * 0.3 * np.sqrt(sc)
        zs.append(z); Rs.append(K9.obs_noise[k] * sc)
        zA[i, j, m, :Z] = z; sA[i, j, m] = sc
      ret = f.predict_and_update_batch(t, k, np.array(zs), np.array(Rs), [[]] * n)      # (the default [[]] serves ONE observation: ekf_sym.py:518 indexes extra_args[i])
      assert ret is not None and len(ret[6]) == n
      tA[i, j], kA[i, j], nA[i, j] = t, k, n
      for m in range(n):
        yA[i, j, m, :Z] = np.ravel(ret[6][m])
      x1A[i, j], xkA[i, j], P1A[i, j], PkA[i, j] = ret[0], ret[1], ret[2], ret[3]
      xA[i, j], PA[i, j] = f.state(), f.covs()
  # ---- part B: MSCKF feature tracks, shared timeline
  spec = importlib.util.spec_from_file_location("rn_amd_feature_kf_m", os.path.join(REPO, "examples", "feature_kf.py"))
  fmod = importlib.util.module_from_spec(spec); spec.loader.exec_module(fmod)
  FK = fmod.FeatureKalman
  NF, TF, NT = 4, 24, 4
  ZF = 2 * len(FK.observed)
  D = FK.dim_state
  landmarks = np.array([[2.0, 1.0, 8.0], [-1.5, 0.5, 6.0], [0.5, -1.0, 10.0], [3.0, 2.0, 12.0]])
  tB = 0.05 * (1 + np.arange(TF)); kB = np.where(np.arange(TF) % 3 == 0, 1, 2).astype(np.int32); nB = np.where(kB == 1, 3, NT).astype(np.int32)
  zB = np.zeros((NF, TF, NT, ZF)); eaB = np.zeros((NF, TF, NT, 3)); yB = np.zeros((NF, TF, NT, ZF))
  xB = np.empty((NF, TF, D)); PB = np.empty((NF, TF, D, D)); x1B = np.empty((NF, TF, D)); P1B = np.empty((NF, TF, D, D))
  for i in range(NF):
    f = ref_filter(FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), fmod.DIM_MAIN, fmod.DIM_MAIN, **FK.filter_kwargs())
    truth_p, truth_v = np.zeros(3), np.array([1.0, 0.5, 0.0]) + rng.normal(size=3) * 0.05
    window = [np.zeros(3)] * FK.n_window
    for j in range(TF):
      truth_p = truth_p + 0.05 * truth_v
      if kB[j] == 1:
        zs = [truth_p + rng.normal(size=3) * 0.2 for _ in range(3)]
        for m in range(3):
          zB[i, j, m, :3] = zs[m]
        ret = f.predict_and_update_batch(float(tB[j]), 1, np.array(zs), np.array([FK.obs_noise[1]] * 3), [[]] * 3, augment=True)
        window = window[1:] + [truth_p.copy()]
      else:
        zs, eas = [], []
        for m in range(NT):
          ea = landmarks[m] + rng.normal(size=3) * 0.05
          rays = [landmarks[m] - window[w] for w in FK.observed]
          zs.append(np.concatenate([[r[0] / r[2], r[1] / r[2]] for r in rays]) + rng.normal(size=ZF) * 0.01)
          eas.append(list(ea))
          zB[i, j, m], eaB[i, j, m] = zs[m], ea
        ret = f.predict_and_update_batch(float(tB[j]), 2, np.array(zs), np.array([np.eye(ZF) * 0.01**2] * NT), eas)
      for m in range(int(nB[j])):
        y = np.ravel(ret[6][m])
        yB[i, j, m, :len(y)] = y
      x1B[i, j], P1B[i, j] = ret[0], ret[2]
      xB[i, j], PB[i, j] = f.state(), f.covs()
  # ---- part C: per-filter logs like part A, but the noise of observation m of a call is the kind's matrix times SCALE_C[m] for every filter -- what a
  # caller with one noise matrix per observation shared by the batch passes (the C++ per-filter entry point takes host matrices)
  SCALE_C = np.array([1.0, 0.5, 2.0])
  NC, TC = 6, 30
  tC = np.zeros((NC, TC)); kC = np.zeros((NC, TC), dtype=np.int32); nC = np.zeros((NC, TC), dtype=np.int32)
  zC = np.zeros((NC, TC, NM, 3)); yC = np.zeros((NC, TC, NM, 3)); xC = np.empty((NC, TC, 9)); PC = np.empty((NC, TC, 9, 9)); lateC = np.zeros(NC, dtype=np.int32)
  for i in range(NC):
    f = ref_filter("kinematic9", K9.Q, K9.initial_x, np.diag(K9.initial_P_diag), 9, 9)
    tr = np.array([0.5, 0.5, 0.5, 1.0, -0.5, 0.2, 0.3, 0.1, -0.2]) + rng.normal(size=9) * 0.05
    times = np.cumsum(rng.uniform(0.01, 0.05, size=TC)) + rng.uniform(0, 0.3)
    late_at = int(rng.integers(6, TC - 2))
    back = int(rng.integers(2, 5))
    times[late_at] = 0.5 * (times[late_at - back] + times[late_at - back + 1])
    lateC[i] = late_at
    for j in range(TC):
      t = float(times[j])
      k = int(rng.integers(1, 4))
      n = int(rng.integers(1, 4))
      if j == late_at:
        n = max(n, 2)
      p = tr[0:3] + t * tr[3:6] + 0.5 * t * t * tr[6:9]; v = tr[3:6] + t * tr[6:9]
      Z = K9.obs_noise[k].shape[0]
      zs, Rs = [], []
      for m in range(n):
        sc = float(SCALE_C[m])
        if k == 1:
          z = p + rng.normal(size=3) * 0.1 * np.sqrt(sc)
        elif k == 2:
          z = np.array([np.linalg.norm(p - np.array(ANCHOR))]) + rng.normal(size=1) * 0.2 * np.sqrt(sc)
        else:
          z = v + rng.normal(size=3) * 0.3 * np.sqrt(sc)
        zs.append(z); Rs.append(K9.obs_noise[k] * sc)
        zC[i, j, m, :Z] = z
      ret = f.predict_and_update_batch(t, k, np.array(zs), np.array(Rs), [[]] * n)
      assert ret is not None and len(ret[6]) == n
      tC[i, j], kC[i, j], nC[i, j] = t, k, n
      for m in range(n):
        yC[i, j, m, :Z] = np.ravel(ret[6][m])
      xC[i, j], PC[i, j] = f.state(), f.covs()
  np.savez_compressed(os.path.join(GOLD, "multi_obs.npz"), C_t=tC, C_kind=kC, C_n=nC, C_z=zC, C_y=yC, C_x=xC, C_P=PC[:, ::5], C_late=lateC, C_scale=SCALE_C, A_t=tA, A_kind=kA, A_n=nA, A_z=zA, A_Rscale=sA, A_y=yA, A_x=xA, A_P=PA,
                      A_xk_km1=x1A[:, ::6], A_Pk_km1=P1A[:, ::6], A_xk_k=xkA[:, ::6], A_Pk_k=PkA[:, ::6], A_late=lateA,
                      B_t=tB, B_kind=kB, B_n=nB, B_z=zB, B_ea=eaB, B_y=yB, B_x=xB, B_P=PB, B_xk_km1=x1B[:, ::4], B_Pk_km1=P1B[:, ::4])
  print("multi-observation calls: part A", NB, "x", TB, "calls,", int(nA.sum()), "observations, late calls at", lateA.tolist(),
        "; part B", NF, "x", TF, "calls,", int(nB.sum()) * NF, "observations")


if __name__ == "__main__":
  os.makedirs(GOLD, exist_ok=True)
  if len(sys.argv) > 1:          # python oracle/make_golden.py <function> ...: regenerate only those fixtures
    for fn_name in sys.argv[1:]:
      globals()[fn_name]()
    sys.exit(0)
  kinematic_stream()
  compare_rewind()
  live_single_steps()
  live_stream()
  live_stream(n_ticks=1000, out_name="live_stream_2100.npz", keep_every=100, seed=2026)
  rts_goldens()
  maha_goldens()
  kinematic9_goldens()
  attitude_goldens()
  feature_goldens()
  feature_goldens(T=15, cls_name="WideFeatureKalman", out_name="feature36_stream.npz", n_upd=6)
  perfilter_timelines()
  multi_obs_goldens()
  for fn in sorted(os.listdir(GOLD)):
    print(fn, os.path.getsize(os.path.join(GOLD, fn)))
