"""The 2-state fused run (65 536 x 2 000) timed from Python; run with and without
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 to see whether the HIP runtime bundled with torch accounts for the gap to tools/ab_run."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.kinematic_kf import KinematicKalman as M
from rednose_amd.helpers.ekf_sym import BatchedEKF
rt = ctypes.c_int(0)
ctypes.CDLL(None).hipRuntimeGetVersion(ctypes.byref(rt))
n, T, dev = 65536, 2000, "cuda:0"
f = BatchedEKF(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "generated"), "kinematic", M.Q, M.initial_x, np.diag(M.initial_P_diag), 2, 2, batch=n, device=dev)
kd = torch.ones(T, dtype=torch.int32, device=dev)
dd = torch.full((T,), 0.01, dtype=torch.float64, device=dev)
Rd = torch.full((T, 1), 0.01, dtype=torch.float64, device=dev)
z = torch.randn((T, n, 1), dtype=torch.float64, device=dev)
ts = []
for _ in range(4):
  f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), T, f._p(z), f._p(Rd), n, 0, None, None, None, None, None, f._stream())
  e1.record()
  torch.cuda.synchronize()
  ts.append(e0.elapsed_time(e1))
print("hip runtime", rt.value, "LD_PRELOAD", os.environ.get("LD_PRELOAD"), " ".join(f"{t:.4f}" for t in ts), flush=True)
