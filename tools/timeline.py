#!/usr/bin/env python3
"""Phase timeline of the three-phase step kernels (debugging aid, GPU only).

  RN_TUNE=wide_timeline=1[,other knobs] RN_GEN_DIR=/some/dir python tools/timeline.py [model] [batch]

Builds the model with the timeline knob (rednose_amd/codegen/tuning.py), runs a few steps, then one launch per kind of the
live IMU / GNSS pattern, and prints for that launch, averaged over the first 256 workgroups, when (microseconds after the
workgroup's start, 100 MHz wall clock) each phase boundary was reached, plus the spread of workgroup start times."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
  import torch
  from examples import ensure_generated, GENERATED_DIR
  from examples.live_kf import LiveKalman as L
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
  gen = ensure_generated(["live"], folder=GENERATED_DIR)
  f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])
  tl = getattr(f._lib, "live_debug_timeline")
  rng = np.random.default_rng(0)
  x0 = np.tile(L.initial_x, (n, 1))
  f.init_state(x0, np.diag(L.initial_P_diag), None)
  z = {k: torch.as_tensor(rng.normal(size=(n, 3)) * s, device=f.device) for k, s in ((4, 0.025), (10, 0.5), (12, 5.0))}
  z[10] = z[10] + torch.tensor([0.0, 0.0, 0.0], device=f.device)
  steps = {k: f.bind_step(k, L.obs_noise[k]) for k in (4, 10, 12)}
  for i in range(6):
    steps[4](z[4].clone(), 0.01)
  torch.cuda.synchronize()
  buf = (ctypes.c_ulonglong * (256 * 64 * 2))()
  import re
  with open(os.path.join(gen, "live.hip"), encoding="utf-8") as fh:
    ft = int(re.search(r"constexpr int FT2 = (\d+);", fh.read()).group(1))
  ngroups = ft // 2
  for kind, dt in ((4, 0.01), (10, 0.0), (12, 0.0), (4, 0.01)):
    zz = z[kind].clone()
    torch.cuda.synchronize()
    steps[kind](zz, dt)
    torch.cuda.synchronize()
    assert tl(ctypes.cast(buf, ctypes.c_void_p)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 2).astype(np.float64)
    nb = min(256, (n + ft - 1) // ft)
    a = a[:nb]
    wall = a[:, :, 1] / 100.0          # microseconds
    cyc = a[:, :, 0]
    t0 = wall[:, 0]
    rel = wall - t0[:, None]
    names = {0: "start", 1: "x/z landed", 2: "phase 1 done", 3: "phase 2 done", 63: "end"}
    print(f"--- kind {kind} dt {dt}: {nb} workgroups; start spread {t0.max() - t0.min():.2f} us; "
          f"clock {(cyc[:, 63] - cyc[:, 0]).mean() / max(1e-9, (wall[:, 63] - wall[:, 0]).mean()):.0f} cycles/us")
    order = [0, 1, 2] + [4 + i for i in range(4 * ngroups)] + [3, 63]
    prev = 0.0
    for idx in order:
      v = rel[:, idx]
      if idx >= 4 and idx < 36:
        p, q = divmod(idx - 4, 4)
        nm = f"group {p} " + ("P landed", "predict done", "update done", "stores issued")[q]
      else:
        nm = names[idx]
      m = v.mean()
      print(f"  {nm:28s} {m:8.2f} us  (+{m - prev:6.2f})   min {v.min():7.2f} max {v.max():7.2f}")
      prev = m
    print(f"  last workgroup ends {(wall[:, 63].max() - t0.min()):.2f} us after the first one starts")
    # every workgroup's first tile: start / end on the 100 MHz clock
    bb = (ctypes.c_ulonglong * (4096 * 2))()
    assert getattr(f._lib, "live_debug_blocks")(ctypes.cast(bb, ctypes.c_void_p)) == 0
    b = np.frombuffer(bb, dtype=np.uint64).reshape(4096, 2).astype(np.float64) / 100.0
    nblk = min(4096, (n + ft - 1) // ft)
    b = b[:nblk]
    b0 = b[:, 0].min()
    st, en = b[:, 0] - b0, b[:, 1] - b0
    print(f"  all {nblk} workgroups: starts {np.percentile(st, [0, 50, 90, 100]).round(2)} us, ends {np.percentile(en, [0, 50, 90, 100]).round(2)} us, "
          f"duration {np.percentile(en - st, [0, 50, 100]).round(2)} us")


def rts():
  """Phase timeline of one backward step of the smoother (k_rts_group), averaged over the first 256 workgroups."""
  import torch
  from examples import ensure_generated, GENERATED_DIR
  from examples.live_kf import LiveKalman as L
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  n, T = 16384, 40
  gen = ensure_generated(["live"], folder=GENERATED_DIR)
  f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])
  g = torch.Generator(device="cuda").manual_seed(0)
  tx = torch.as_tensor(L.initial_x, device="cuda").repeat(T, n, 1).contiguous()
  A = torch.randn((n, 22, 22), dtype=torch.float64, device="cuda", generator=g) * 0.01
  P = torch.diag(torch.as_tensor(L.initial_P_diag, device="cuda")) * 1e-2 + A @ A.transpose(1, 2)
  tP = P.repeat(T, 1, 1, 1).contiguous()
  ts = np.arange(T) * 0.01
  for _ in range(2):
    f.rts_smooth(tx, tP, ts)
  torch.cuda.synchronize()
  buf = (ctypes.c_ulonglong * (256 * 16))()
  assert getattr(f._lib, "live_debug_rts_timeline")(ctypes.cast(buf, ctypes.c_void_p)) == 0
  a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.float64) / 100.0
  rel = a[:, :10] - a[:, :1]
  names = ["step start", "x landed", "scalars (f, F) done", "predict done (B = Pk1_k)", "difference / output issued", "Cholesky done",
           "solves done", "state update done", "T = Ck Dm done", "Pk_n done (step end)"]
  if "rts3=0" not in os.environ.get("RN_TUNE", ""):      # k_rts3 (emit_rts3.py): eleven stamps
    rel = a[:, :11] - a[:, :1]
    names = ["step start", "Pk copy + x issued", "scalars done, Pk landed", "predict done (rows of Pk1_k)", "Pk1_n read back, D / Pk1_k in LDS",
             "Cholesky done", "solves done", "state update done", "T = Ck D done", "U = T Ck^T done", "Pk_n stored (step end)"]
  print("--- smoother, one backward step (us after the step's start)")
  prev = 0.0
  for i, nm in enumerate(names):
    m = rel[:, i].mean()
    print(f"  {nm:30s} {m:8.2f} us (+{m - prev:6.2f})  min {rel[:, i].min():7.2f} max {rel[:, i].max():7.2f}")
    prev = m


def run():
  """Phase timeline of the fused multi-step run (k_run of the lane-group family): the last three steps of a live IMU / GNSS
  schedule, averaged over the first 256 workgroups (one wavefront each)."""
  import torch
  from examples import ensure_generated, GENERATED_DIR
  from examples.live_kf import LiveKalman as L
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
  trace = len(sys.argv) > 3 and sys.argv[3] == "trace"
  gen = ensure_generated(["live"], folder=GENERATED_DIR)
  f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])
  T = 126
  kinds = np.tile(np.array([4, 10, 12], dtype=np.int32), T // 3)
  ts = np.repeat(np.arange(1, T // 3 + 1) * 0.01, 3)
  rng = np.random.default_rng(0)
  zs = torch.as_tensor(rng.normal(size=(T, n, 3)) * 0.02, device=f.device)
  Rs = {k: np.atleast_2d(L.obs_noise[k]) for k in (4, 10, 12)}
  f.init_state(np.tile(L.initial_x, (n, 1)), np.diag(L.initial_P_diag), 0.0)
  for _ in range(2):
    f.filter_time = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f.run(ts, kinds, zs.clone(), Rs, trace=trace)
    e1.record()
    torch.cuda.synchronize()
  print(f"--- fused run, live, {n} filters x {T} steps{' with trace' if trace else ''}: {e0.elapsed_time(e1) * 1e3 / T:.2f} us per step")
  buf = (ctypes.c_ulonglong * (256 * 64 * 2))()
  assert getattr(f._lib, "live_debug_timeline")(ctypes.cast(buf, ctypes.c_void_p)) == 0
  a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 2).astype(np.float64)[:min(256, (n + 7) // 8)]
  wall = a[:, :60, 1].reshape(-1, 3, 20) / 100.0
  if len(sys.argv) > 4 and sys.argv[4] == "run2":      # k_run2: matrix wavefront stamps 0-9, scalar wavefront stamps 10-19, both against the matrix wavefront's step start
    nm = {0: "M  step start (after B1)", 1: "M  predict done", 2: "M  after B2", 3: "M  G in the buffer", 4: "M  S, gate, K, dx done (before B3)", 5: "M  after B3",
          6: "M  P - K G done", 7: "M  Joseph coefficients (+B4)", 8: "M  + D K^T done", 9: "M  trace out (before B1)",
          10: " S step start (after B1)", 11: " S h / He done (before B2)", 12: " S after B2", 13: " S after B3", 14: " S injection, flags", 15: " S y / x trace out, next z in",
          16: " S f / F of the next step", 17: " S after B4", 19: " S before B1"}
    for s_ in range(3):
      t = T - 3 + s_
      rel = wall[:, t % 3, :] - wall[:, t % 3, :1]
      dtv = ts[t] - (ts[t - 1] if t else 0.0)
      print(f"  step {t}: kind {kinds[t]} dt {dtv:.2f}")
      rows = sorted(((rel[:, i].mean(), i) for i in nm if np.all(wall[:, t % 3, i] > 0)))
      for m, i in rows:
        print(f"    {nm[i]:44s} {m:8.2f} us   min {rel[:, i].min():7.2f} max {rel[:, i].max():7.2f}")
    return
  names = {0: "step start", 1: "scalars of predict (f, F)", 8: "  predict: rows of A = P F^T -> image", 9: "  predict: columns of A read, rows of P' formed",
           2: "predict, covariance (end)", 3: "scalars of the kind (h, He)", 10: "  update: G -> buffer", 11: "  update: S, Cholesky, gate",
           12: "  update: K rows solved", 13: "  update: P - K G", 14: "  update: Joseph coefficients, K^T -> buffer",
           15: "  update: + D K^T", 4: "update, covariance (end)", 5: "injection", 6: "y / trace out", 7: "next z in the slot"}
  order = [0, 1, 8, 9, 2, 3, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7]
  for s in range(3):
    t = T - 3 + s
    rel = wall[:, t % 3, :] - wall[:, t % 3, :1]
    dtv = ts[t] - (ts[t - 1] if t else 0.0)
    print(f"  step {t}: kind {kinds[t]} dt {dtv:.2f}")
    prev = 0.0
    for i in order:
      if i in (8, 9) and dtv == 0.0:
        continue
      m = rel[:, i].mean()
      print(f"    {names[i]:48s} {m:8.2f} us (+{m - prev:6.2f})  min {rel[:, i].min():7.2f} max {rel[:, i].max():7.2f}")
      prev = m


if __name__ == "__main__":
  if len(sys.argv) > 1 and sys.argv[1] == "rts":
    rts()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "run":
    run()
    sys.exit(0)
  main()
