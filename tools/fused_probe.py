"""Why does bench.py see the 2-state fused run at ~0.52 ms when tools/ab_run.cpp sees 0.40 ms?  Times the same library call with
differently prepared observation buffers."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples import ensure_generated
from examples.kinematic_kf import KinematicKalman as M
from rednose_amd.helpers.ekf_sym import BatchedEKF

gen = ensure_generated(["kinematic"])
n, T, dev = 65536, 2000, "cuda:0"
f = BatchedEKF(gen, "kinematic", M.Q, M.initial_x, np.diag(M.initial_P_diag), 2, 2, batch=n, device=dev)
kd = torch.ones(T, dtype=torch.int32, device=dev)
dd = torch.full((T,), 0.01, dtype=torch.float64, device=dev)
Rd = torch.from_numpy(np.tile(np.atleast_2d(M.obs_noise[1]).reshape(1, 1), (T, 1))).to(dev)
zs = torch.randn((T, n, 1), dtype=torch.float64, device=dev) * 0.1
buf = torch.empty_like(zs)


def once(z):
  f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), T, f._p(z), f._p(Rd), n, 0, None, None, None, None, None, f._stream())
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1)


for label, prep in (("clone", lambda: zs.clone()), ("copy_ into one buffer", lambda: buf.copy_(zs)), ("reuse y as z", lambda: buf),
                    ("uniform", lambda: buf.uniform_(-1.7, 1.7)), ("randn*1", lambda: buf.normal_()), ("clone again", lambda: zs.clone())):
  ts = [once(prep()) for _ in range(4)]
  print(f"{label:24s} " + " ".join(f"{t:.4f}" for t in ts), flush=True)
