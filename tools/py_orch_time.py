"""Per-call wall time of BatchedEKF.predict_and_update_batch on the headline workload (kinematic6 x 65 536, device-resident observations), beside the
bound entry point bench.py loops over (BatchedEKF.bind_step): what the Python orchestrator adds to a 9 us launch."""
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
n, K = 65536, 2000
dev = torch.device("cuda:0")
for ring in (0, 8):
  f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n, device=dev, **({"rewind_to_keep": ring} if ring else {}))
  z = torch.zeros((n, 3), dtype=torch.float64, device=dev)
  R = np.ascontiguousarray(K6.obs_noise[1], dtype=np.float64)
  t = 0.0
  for _ in range(300):
    t += 0.01
    f.predict_and_update_batch(t, 1, z, R)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(K):
    t += 0.01
    f.predict_and_update_batch(t, 1, z, R)
  torch.cuda.synchronize()
  print(f"BatchedEKF.predict_and_update_batch, kinematic6 x {n}, rewind ring {ring}: {(time.perf_counter() - t0) / K * 1e6:.2f} us per call (wall, {K} calls)")
