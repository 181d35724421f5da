"""Third stage of the Python-vs-harness gap (profiles/tuning_notes.md): a torch process is as fast as the C++ harness when it
drives the library through raw ctypes (tools/notorch_probe.py, RN_PROBE_MODE=tensors), tools/preload_probe.py -- same process
contents, but through BatchedEKF, torch events and torch-generated observations -- is 30 % slower on the 2-state fused run.  One
switch at a time:  PP_Z = h2d | randn   (who wrote the observations),  PP_CALL = raw | batched   (ctypes.CDLL + raw pointers, or
BatchedEKF._call),  PP_EV = hip | torch   (hipEvent through ctypes, or torch.cuda.Event),  PP_STATE = random | init (x0 / P0 of
the filters: random per filter, or init_state's identical filters)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from examples.kinematic_kf import KinematicKalman as M      # noqa: E402
from rednose_amd.helpers.ekf_sym import BatchedEKF           # noqa: E402

Z, CALL, EV, STATE = (os.environ.get(k, d) for k, d in (("PP_Z", "h2d"), ("PP_CALL", "raw"), ("PP_EV", "hip"), ("PP_STATE", "random")))
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
gen = os.path.join(HERE, "..", os.environ.get("RN_GEN", "generated"))
n, T, dev = 65536, 2000, "cuda:0"
rng = np.random.default_rng(0)
kd = torch.ones(T, dtype=torch.int32, device=dev)
dd = torch.full((T,), 0.01, dtype=torch.float64, device=dev)
Rd = torch.full((T, 1), 0.01, dtype=torch.float64, device=dev)
if Z == "randn":
  z = torch.randn((T, n, 1), dtype=torch.float64, device=dev)
else:
  z = torch.empty((T, n, 1), dtype=torch.float64, device=dev)
  zh = rng.normal(size=(T, n, 1))
  assert hip.hipMemcpy(ctypes.c_void_p(z.data_ptr()), zh.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(zh.nbytes), 1) == 0
f = BatchedEKF(gen, "kinematic", M.Q, M.initial_x, np.diag(M.initial_P_diag), 2, 2, batch=n, device=dev)
raw = ctypes.CDLL(os.path.join(gen, "libkinematic.so")).kinematic_batch_run
raw.restype = ctypes.c_int
p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
ts = []
for _ in range(5):
  if STATE == "init":
    f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
  else:
    f.init_state(M.initial_x[None] + 0.1 * rng.normal(size=(n, 2)), np.diag(M.initial_P_diag), 0.0)
  torch.cuda.synchronize()
  if EV == "torch":
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
  else:
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
    hip.hipEventRecord(e0, None)
  if CALL == "batched":
    f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), T, f._p(z), f._p(Rd), n, 0, None, None, None, None, None, f._stream())
  else:
    assert raw(p(f.x), p(f.P), p(f.Q), p(kd), p(dd), ctypes.c_int64(T), p(z), p(Rd), ctypes.c_int64(n), 0, None, None, None, None, None, None) == 0
  if EV == "torch":
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
  else:
    hip.hipEventRecord(e1, None)
    hip.hipEventSynchronize(e1)
    ms = ctypes.c_float()
    hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
    ts.append(ms.value)
print(f"z={Z:5s} call={CALL:7s} events={EV:5s} state={STATE:6s}  " + " ".join(f"{t:.4f}" for t in ts), flush=True)
