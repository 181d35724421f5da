"""The 2-state fused run (65 536 x 2 000) and the headline step launch (kinematic6, 65 536) timed from a Python process that never
imports torch: buffers from hipMalloc of the SYSTEM runtime (/opt/rocm/lib), library calls through ctypes, HIP events.
Companion of tools/preload_probe.py / tools/ab_run.cpp for the Python-vs-harness gap (profiles/tuning_notes.md)."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RN_PROBE_MODE: none (default) = no torch in the process; import = `import torch` only; init = + a HIP context created by torch
# (one tensor on the device); tensors = + every buffer is a torch tensor (torch's caching allocator) instead of a hipMalloc block.
MODE = os.environ.get("RN_PROBE_MODE", "none")
torch = None
if MODE != "none":
  import torch
  hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)      # the runtime torch bundles (already loaded)
  if MODE in ("init", "tensors"):
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
else:
  hip = ctypes.CDLL(os.environ.get("RN_HIP_LIB", "/opt/rocm/lib/libamdhip64.so.7"), mode=ctypes.RTLD_GLOBAL)
  assert "torch" not in sys.modules
_keep = []


def ck(e):
  assert e == 0, e


def dmalloc(nbytes):
  if MODE == "tensors":
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    _keep.append(t)
    return ctypes.c_void_p(t.data_ptr())
  p = ctypes.c_void_p()
  ck(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)))
  return p


def h2d(p, arr):
  arr = np.ascontiguousarray(arr)
  ck(hip.hipMemcpy(p, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), 1))


def timed(fn, reps):
  e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
  ck(hip.hipEventCreate(ctypes.byref(e0)))
  ck(hip.hipEventCreate(ctypes.byref(e1)))
  out = []
  for _ in range(reps):
    ck(hip.hipDeviceSynchronize())
    ck(hip.hipEventRecord(e0, None))
    fn()
    ck(hip.hipEventRecord(e1, None))
    ck(hip.hipEventSynchronize(e1))
    ms = ctypes.c_float()
    ck(hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1))
    out.append(ms.value)
  return out


if os.environ.get("RN_PROBE_BEKF") and MODE == "tensors":
  # does merely CONSTRUCTING the orchestrator (library loaded through load_code, a handful of small torch tensors, init_state's copies)
  # change the timing of the raw launches below?  (tools/gap_probe3.py: every variant of a process that had one was slow)
  sys.path.insert(0, os.path.join(HERE, ".."))
  from examples.kinematic_kf import KinematicKalman as _M
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  _f = BatchedEKF(os.path.join(HERE, "..", os.environ.get("RN_GEN", "generated")), "kinematic", _M.Q, _M.initial_x, np.diag(_M.initial_P_diag), 2, 2,
                  batch=65536, device="cuda:0")
  if os.environ.get("RN_PROBE_BEKF") == "2":
    del _f
  torch.cuda.synchronize()
rt = ctypes.c_int(0)
hip.hipRuntimeGetVersion(ctypes.byref(rt))
gen = os.path.join(HERE, "..", os.environ.get("RN_GEN", "generated"))
rng = np.random.default_rng(0)

# ---- 2-state fused run ----
n, T = 65536, 2000
lib = ctypes.CDLL(os.path.join(gen, "libkinematic.so"))
run = lib.kinematic_batch_run
run.restype = ctypes.c_int
x, P, Q = dmalloc(n * 2 * 8), dmalloc(n * 4 * 8), dmalloc(4 * 8)
kd, dd, Rd, z = dmalloc(T * 4), dmalloc(T * 8), dmalloc(T * 8), dmalloc(T * n * 8)
h2d(x, rng.normal(size=(n, 2)) * 0.1)
h2d(P, np.tile(np.eye(2), (n, 1, 1)))
h2d(Q, np.diag([0.01, 4.0]))
h2d(kd, np.ones(T, dtype=np.int32))
h2d(dd, np.full(T, 0.01))
h2d(Rd, np.full(T, 0.01))
h2d(z, rng.normal(size=(T, n, 1)))
i64 = ctypes.c_int64
ts = timed(lambda: ck(run(x, P, Q, kd, dd, i64(T), z, Rd, i64(n), 0, None, None, None, None, None, None)), 5)
print(f"mode {MODE} nocache {os.environ.get('PYTORCH_NO_HIP_MEMORY_CACHING')}, hip runtime", rt.value, "fused 2-state run ms:", " ".join(f"{t:.4f}" for t in ts), flush=True)

# ---- headline step launch ----
n = 65536
lib6 = ctypes.CDLL(os.path.join(gen, "libkinematic6.so"))
step = lib6.kinematic6_batch_predict_update_1
step.restype = ctypes.c_int
x6, P6, Q6, z6, R6 = dmalloc(n * 6 * 8), dmalloc(n * 36 * 8), dmalloc(36 * 8), dmalloc(n * 3 * 8), dmalloc(9 * 8)
h2d(x6, rng.normal(size=(n, 6)) * 0.1)
h2d(P6, np.tile(np.eye(6), (n, 1, 1)))
h2d(Q6, np.diag([0.01] * 3 + [4.0] * 3))
h2d(z6, rng.normal(size=(n, 3)))
h2d(R6, np.eye(3) * 0.01)
dbl = ctypes.c_double


def many():
  for _ in range(1000):
    step(x6, P6, Q6, None, dbl(0.01), z6, R6, 0, None, i64(n), 0, None, None)


ts = timed(many, 4)
print(f"mode {MODE}, headline step us/launch:", " ".join(f"{t:.3f}" for t in ts), flush=True)
