"""Per-call wall time of the PER-FILTER timeline path (BatchedEKF(per_filter=True): every filter on its own clock, own checkpoint ring) on kinematic6:
all filters active with their own times, no late observations; then 1 % of the filters late by one call (rewind + replay for those)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated
from examples.kinematic6_kf import Kinematic6Kalman as K6
from rednose_amd.helpers.ekf_sym import BatchedEKF

gen = ensure_generated(["kinematic6"])
dev = torch.device("cuda:0")
for n in (4096, 65536):
  for ring in (0, 8):
    f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n, device=dev, per_filter=True, **({"rewind_to_keep": ring} if ring else {}))
    R = np.ascontiguousarray(K6.obs_noise[1], dtype=np.float64)
    z = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    off = np.linspace(0.0, 0.005, n)
    t = 0.0
    K = 200
    for phase in ("warm", "in order", "1 % late"):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(50 if phase == "warm" else K):
        t += 0.01
        tt = t + off
        if phase == "1 % late" and ring:
          tt = tt.copy()
          tt[::100] -= 0.015          # behind the previous call of those filters: rewind one checkpoint, apply, replay one
        f.predict_and_update_batch(tt, 1, z, R)
      torch.cuda.synchronize()
      if phase != "warm" and (ring or phase == "in order"):
        print(f"per-filter timelines, kinematic6 x {n}, ring {ring}, {phase}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per call")
