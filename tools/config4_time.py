"""Timing of ONE chunk of BASELINE config 4 (live_maha, 8 192 filters x 2 100 steps): forward fused run writing the filtered trace
+ gate flags, backward smoother in place -- what bench.py's config4_extra launches per chunk.  RN_GEN_DIR / RN_TUNE select an A/B
build (e.g. RN_TUNE=rts_dt0=0: the full solve on every backward step); C4_PACKED=1: the packed-triangle trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from examples.live_kf import LiveKalman as L
from rednose_amd.helpers.ekf_sym import BatchedEKF

n, T = 8192, 2100
dev = torch.device("cuda:0")
gen = bench.gen_dir(["live_maha"])
f = BatchedEKF(gen, "live_maha", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, device=dev, quaternion_idxs=[3], maha_test_kinds=[12])
gdev = torch.Generator(device=dev).manual_seed(4242)
hacc = bench.live_true_accel(L, gen, "live_maha")
x0 = bench.live_x0(torch, L, n, dev, gdev)
kinds, ts = bench.live_schedule(T)
zs = bench.live_observations(torch, L, hacc, kinds, n, dev, gdev, outlier_frac=0.02)
Rs = {int(k): np.atleast_2d(L.obs_noise[int(k)]) for k in set(kinds.tolist())}
packed = bool(int(os.environ.get("C4_PACKED", "0")))      # the packed-triangle trace (batch_run_tri / batch_rts_tri)
tx = torch.empty((T, n, 23), dtype=torch.float64, device=dev)
tP = torch.empty((T, n, 253) if packed else (T, n, 22, 22), dtype=torch.float64, device=dev)
for rep in range(3):
  f.init_state(x0, np.diag(L.initial_P_diag), None)
  zc = zs.clone()
  e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
  e[0].record()
  f.run(ts, kinds, zc, Rs, flags=True, out=(tx, tP), packed=packed)
  e[1].record()
  if os.environ.get("RN_C4_GAP_MS"):      # (experiment: idle time between the two passes -- is the backward pass's time a function of what ran just before it?)
    import time
    torch.cuda.synchronize()
    time.sleep(float(os.environ["RN_C4_GAP_MS"]) * 1e-3)
    e[1].record()
  f._rts_on(tx, tP, ts, n, None, packed=packed)      # pylint: disable=protected-access
  e[2].record()
  torch.cuda.synchronize()
fw, bw = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
print(f"config 4 chunk ({os.environ.get('RN_TUNE', 'default')}{', packed trace' if packed else ''}): forward {fw:.2f} ms = {n * T / fw / 1e3:.1f} M steps/s ({n * T * 4105 / fw / 1e6 / 8000:.3f} of 8 TB/s), "
      f"backward {bw:.2f} ms = {n * (T - 1) / bw / 1e3:.1f} M steps/s ({n * (T - 1) * 8112 / bw / 1e6 / 8000:.3f}), finite {bool(torch.isfinite(tP[0]).all())}")
