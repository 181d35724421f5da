"""Timing of the live RTS backward pass alone (random SPD trace, 16 384 filters x 60 steps); RN_GEN_DIR selects an A/B build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated
from examples.live_kf import LiveKalman as L
from rednose_amd.helpers.ekf_sym import BatchedEKF

gen = ensure_generated(["live"])
n, T = 16384, 60
f = BatchedEKF(gen, "live", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, quaternion_idxs=[3])
g = torch.Generator(device="cuda").manual_seed(0)
tx = torch.as_tensor(L.initial_x, device="cuda").repeat(T, n, 1).contiguous()
A = torch.randn((n, 22, 22), dtype=torch.float64, device="cuda", generator=g) * 0.01
P = torch.diag(torch.as_tensor(L.initial_P_diag, device="cuda")) * 1e-2 + A @ A.transpose(1, 2)
tP = P.repeat(T, 1, 1, 1).contiguous()
ts = np.arange(T) * 0.01
for rep in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  xs, Ps = f.rts_smooth(tx, tP, ts, inplace=False)
  e1.record()
  torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"{ms:.2f} ms  {n * (T - 1) / ms / 1e3:.1f} M steps/s  {ms * 1e3 / (8 * (T - 1)):.2f} us per step and wavefront")
