"""Timing of the live fused multi-step run alone (8 192 filters x 252 steps of the IMU / GNSS pattern, no trace unless `trace`
is given); RN_GEN_DIR selects an A/B build.  (Counters: the live_run_notrace section of profiles/pmc_workload.py.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated
from examples.live_kf import LiveKalman as L
from rednose_amd.helpers.ekf_sym import BatchedEKF

trace = len(sys.argv) > 1 and sys.argv[1] == "trace"
gen = ensure_generated(["live"])
n, T = 8192, 252
f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])
kinds = np.tile(np.array([4, 10, 12], dtype=np.int32), T // 3)
ts = np.repeat(np.arange(1, T // 3 + 1) * 0.01, 3)
rng = np.random.default_rng(0)
zs = torch.as_tensor(rng.normal(size=(T, n, 3)) * 0.02, device=f.device)
Rs = {k: np.atleast_2d(L.obs_noise[k]) for k in (4, 10, 12)}
out = None
if trace:
  out = (torch.empty((T, n, 23), dtype=torch.float64, device=f.device), torch.empty((T, n, 22, 22), dtype=torch.float64, device=f.device))
for rep in range(3):
  f.init_state(np.tile(L.initial_x, (n, 1)), np.diag(L.initial_P_diag), 0.0)
  z = zs.clone()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  f.run(ts, kinds, z, Rs, out=out) if trace else f.run(ts, kinds, z, Rs)
  e1.record()
  torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"{ms:.2f} ms  {n * T / ms / 1e3:.1f} M steps/s  {ms * 1e3 / T:.2f} us per step and wavefront")
