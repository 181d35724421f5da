"""Timing of the MSCKF feature-track step (feature36, 16 384 filters, fused predict + kind-2 update); RN_TUNE / RN_GEN_DIR select an A/B build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples import ensure_generated, GENERATED_DIR
from examples.feature_kf import WideFeatureKalman as FK
from rednose_amd.helpers.ekf_sym import BatchedEKF

gen = ensure_generated(["feature36"], folder=GENERATED_DIR)
dev = torch.device("cuda:0")
nf, Kf = 16384, 100
ff = BatchedEKF(gen, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), 6, 6, batch=nf, device=dev, **FK.filter_kwargs())
lm = torch.tensor([2.0, 1.0, 8.0], dtype=torch.float64, device=dev) + torch.randn((nf, 3), dtype=torch.float64, device=dev)
zf = [0.05 * torch.randn((nf, 6), dtype=torch.float64, device=dev) for _ in range(8)]
for i in range(10):
  ff.predict_and_update_batch(0.01 * (i + 1), 2, zf[i % 8].clone(), FK.obs_noise[2], extra_args=lm)
zc = [zf[i % 8].clone() for i in range(Kf)]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(Kf):
  ff.predict_and_update_batch(0.01 * (i + 11), 2, zc[i], FK.obs_noise[2], extra_args=lm)
e1.record()
torch.cuda.synchronize()
assert torch.isfinite(ff.x).all() and torch.isfinite(ff.P).all()
us = e0.elapsed_time(e1) / Kf * 1e3
bf = 8.0 * (2 * (36 + 36 * 36) + 6 + 3 + 3)
print(f"{os.environ.get('RN_TUNE', 'default')}: {us:.1f} us per launch, {bf * nf / (us * 1e-6) / 8e12 * 100:.1f} % of 8 TB/s, checksum {float(ff.x.sum()):.12g}")
