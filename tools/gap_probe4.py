"""Fourth stage of the Python-vs-harness gap: ONE torch process, the headline step launch (kinematic6, 65 536 filters) timed through raw
ctypes on torch-allocated buffers after each of a sequence of actions -- which one moves the launch from 8.5 to 9.0 us?"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
gen = os.path.join(HERE, "..", os.environ.get("RN_GEN", "generated"))
dev = "cuda:0"
n = 65536
rng = np.random.default_rng(0)


def raw_buf(arr):
  arr = np.ascontiguousarray(arr)
  t = torch.empty(arr.nbytes, dtype=torch.uint8, device=dev)
  assert hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), 1) == 0
  return t


torch.zeros(1, device=dev)
x6, P6 = raw_buf(rng.normal(size=(n, 6)) * 0.1), raw_buf(np.tile(np.eye(6), (n, 1, 1)))
Q6, z6, R6 = raw_buf(np.diag([0.01] * 3 + [4.0] * 3)), raw_buf(rng.normal(size=(n, 3))), raw_buf(np.eye(3) * 0.01)
step = ctypes.CDLL(os.path.join(gen, "libkinematic6.so")).kinematic6_batch_predict_update_1
step.restype = ctypes.c_int
p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))


def measure(label, stream=None):
  out = []
  for _ in range(3):
    hip.hipDeviceSynchronize()
    hip.hipEventRecord(e0, stream)
    for _ in range(1000):
      step(p(x6), p(P6), p(Q6), None, ctypes.c_double(0.01), p(z6), p(R6), 0, None, ctypes.c_int64(n), 0, None, stream)
    hip.hipEventRecord(e1, stream)
    hip.hipEventSynchronize(e1)
    ms = ctypes.c_float()
    hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
    out.append(ms.value)
  print(f"{label:70s} " + " ".join(f"{t:.3f}" for t in out) + " us/launch", flush=True)


measure("baseline: torch context, torch-allocated buffers, raw ctypes calls")
a = torch.ones(1000, device=dev) * 2.0
torch.cuda.synchronize()
measure("after a torch elementwise kernel")
b = torch.randn((1000, 1000), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
measure("after torch.randn")
c = torch.as_tensor(np.ones((1000, 6))).to(dev).expand(1000, 6).contiguous().clone()
torch.cuda.synchronize()
measure("after a host-to-device copy through torch + clone")
ev = torch.cuda.Event(enable_timing=True); ev.record(); torch.cuda.synchronize()
measure("after a torch.cuda.Event record")
s_ = torch.cuda.current_stream(dev).cuda_stream
measure(f"same, launching on torch.cuda.current_stream() = {s_}", ctypes.c_void_p(s_))
from examples.kinematic6_kf import Kinematic6Kalman as K6      # noqa: E402
from rednose_amd.helpers.ekf_sym import BatchedEKF            # noqa: E402
measure("after importing rednose_amd / the model (sympy)")
f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n, device=dev)
torch.cuda.synchronize()
measure("after constructing BatchedEKF")
bs = f.bind_step(1, np.atleast_2d(K6.obs_noise[1]))
zt = torch.randn((n, 3), dtype=torch.float64, device=dev)
for _ in range(10):
  bs(zt, 0.01)
torch.cuda.synchronize()
measure("after 10 launches through BatchedEKF.bind_step")
f.init_state(K6.initial_x[None] + 0.1 * rng.normal(size=(n, 6)), np.diag(K6.initial_P_diag), 0.0)
hip.hipDeviceSynchronize()
ts = []
for _ in range(3):
  torch.cuda.synchronize()
  hip.hipEventRecord(e0, None)
  for _ in range(1000):
    bs(zt, 0.01)
  hip.hipEventRecord(e1, None)
  hip.hipEventSynchronize(e1)
  ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1); ts.append(ms.value)
print(f"{'1000 launches through bind_step on BatchedEKF-owned x / P':70s} " + " ".join(f"{t:.3f}" for t in ts) + " us/launch", flush=True)
measure("raw buffers again")
