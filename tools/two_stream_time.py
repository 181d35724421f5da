"""Experiment: the headline step (kinematic6, 65 536 filters) issued as S sub-batch launches on S HIP streams instead of one launch:
sub-batches run concurrently, each one's dependent chain of launches on its own stream, so the load phase of one overlaps the
store phase of another and the inter-kernel boundary of one stream is covered by the others.
  python tools/two_stream_time.py [streams ...]     (default: 1 2 4)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from examples.kinematic6_kf import Kinematic6Kalman as M
from rednose_amd.helpers.ekf_sym import BatchedEKF

n, K, W = 65536, 2000, 100
dev = torch.device("cuda:0")
gen = bench.gen_dir(["kinematic6"])
R = np.atleast_2d(M.obs_noise[1])
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
  m = n // S
  streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
  filters, steps, zs = [], [], []
  for s in range(S):
    with torch.cuda.stream(streams[s]):
      f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), 6, 6, batch=m, device=dev)
      x0, P0, sched = bench.kinematic_stream(torch, M, m, K + W, dev, s)
      f.init_state(x0, P0, None)
      filters.append(f)
      steps.append(f.bind_step(1, R))           # binds the CURRENT stream (this sub-batch's)
      zs.append([z.clone() for (_, _, z) in sched])
  torch.cuda.synchronize()
  def run(lo, hi):
    for i in range(lo, hi):
      for s in range(S):
        steps[s](zs[s][i], 0.01)
  run(0, W)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  run(W, W + K)
  torch.cuda.synchronize()
  el = time.perf_counter() - t0
  ok = all(bool(torch.isfinite(f.x).all()) for f in filters)
  print(f"{S} stream(s) x {m} filters: {el / K * 1e6:.3f} us per step of {n} filters = {n * K / el / 1e9:.3f} G steps/s, "
        f"{720.0 * n / (el / K) / 1e9 / 8000:.3f} of the HBM roofline end to end, finite {ok}")
