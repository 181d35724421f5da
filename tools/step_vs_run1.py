"""Experiment (round 6): what would the live step cost in the fused run's two-wavefront structure?  `batch_run` with T = 1 IS a step of that
structure (k_run2: rows of P in registers, 8 filters per workgroup, matrix + scalar wavefront; reads (P + P^T) / 2), so timing it against
`batch_predict_update_k` (the three-phase lean kernel, the reference's arithmetic on any P) on the same 16 384 live filters gives the answer
without writing the kernel.  Launches alternate kinds like the bench's dt > 0 section (every launch advances time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from examples.live_kf import LiveKalman as L
from rednose_amd.helpers.ekf_sym import BatchedEKF

n = int(os.environ.get("N", 16384))
dev = torch.device("cuda:0")
gen = bench.gen_dir(["live"])
f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, device=dev, quaternion_idxs=[3])
gdev = torch.Generator(device=dev).manual_seed(1)
x0 = bench.live_x0(torch, L, n, dev, gdev)
K = 200
for kind in (4, 10, 12):
  R = np.atleast_2d(L.obs_noise[kind])
  zs = [torch.randn((n, 3), dtype=torch.float64, device=dev, generator=gdev) * 0.01 for _ in range(8)]
  if kind == 12:
    zs = [z + torch.as_tensor(L.initial_x[:3], device=dev) for z in zs]
  if kind == 10:
    zs = [z + torch.as_tensor(bench.live_true_accel(L, gen), device=dev) for z in zs]
  step = f.bind_step(kind, R)
  kd = torch.full((1,), kind, dtype=torch.int32, device=dev)
  dd = torch.full((1,), 0.01, dtype=torch.float64, device=dev)
  Rd = torch.zeros((1, 9), dtype=torch.float64, device=dev)
  Rd[0, :9] = torch.as_tensor(R.reshape(-1), device=dev)
  res = {}
  for name in ("step", "run1", "step", "run1"):
    f.init_state(x0, np.diag(L.initial_P_diag), 0.0)
    zc = [z.clone().reshape(1, n, 3) for z in zs for _ in range(K // 8 + 1)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
      if name == "step":
        step(zc[i][0], 0.01)
      else:
        f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), 1, f._p(zc[i]), f._p(Rd), n, f.norm_quats, None, None, None, None, None, f._stream())      # pylint: disable=protected-access
    e1.record()
    torch.cuda.synchronize()
    assert torch.isfinite(f.x).all()
    res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / K)
  print(f"live kind {kind}, {n} filters, dt = 0.01 every launch: batch_predict_update {min(res['step']):.2f} us ({n * 8160 / min(res['step']) / 8e6:.3f} of 8 TB/s), "
        f"batch_run T=1 {min(res['run1']):.2f} us ({n * 8160 / min(res['run1']) / 8e6:.3f})")
