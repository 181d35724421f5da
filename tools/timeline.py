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
  for kind, dt in ((4, 0.01), (10, 0.0), (12, 0.0), (4, 0.01)):
    zz = z[kind].clone()
    torch.cuda.synchronize()
    steps[kind](zz, dt)
    torch.cuda.synchronize()
    assert tl(ctypes.cast(buf, ctypes.c_void_p)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 2).astype(np.float64)
    nb = min(256, (n + 15) // 16)
    a = a[:nb]
    wall = a[:, :, 1] / 100.0          # microseconds
    cyc = a[:, :, 0]
    t0 = wall[:, 0]
    rel = wall - t0[:, None]
    names = {0: "start", 1: "x/z landed", 2: "phase 1 done", 3: "phase 2 done", 63: "end"}
    print(f"--- kind {kind} dt {dt}: {nb} workgroups; start spread {t0.max() - t0.min():.2f} us; "
          f"clock {(cyc[:, 63] - cyc[:, 0]).mean() / max(1e-9, (wall[:, 63] - wall[:, 0]).mean()):.0f} cycles/us")
    order = [0, 1, 2] + [4 + i for i in range(4 * 8)] + [3, 63]
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


if __name__ == "__main__":
  main()
