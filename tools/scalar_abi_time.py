"""Latency of the reference's scalar host-pointer ABI ({name}_predict / {name}_update_{kind} / {name}_f_fun ..., what the reference's EKF_sym binds:
rednose/helpers/ekf_sym.py:149-165) executed as a batch of one on the GPU: microseconds per call, host buffers in, host buffers out.
   python tools/scalar_abi_time.py [generated-dir]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.kinematic_kf import KinematicKalman as K      # noqa: E402
from examples.live_kf import LiveKalman as L                # noqa: E402

gen = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "generated")
dp = ctypes.POINTER(ctypes.c_double)
ptr = lambda a: a.ctypes.data_as(dp)      # noqa: E731


def timeit(fn, n=2000, warm=200):
  for _ in range(warm):
    fn()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  return (time.perf_counter() - t0) / n * 1e6


for name, M, D, E, kind, Z in (("kinematic", K, 2, 2, 1, 1), ("live", L, 23, 22, 4, 3), ("live", L, 23, 22, 12, 3)):
  lib = ctypes.CDLL(os.path.join(gen, f"lib{name}.so"))
  x = np.array(M.initial_x, dtype=np.float64)
  P = np.diag(np.asarray(M.initial_P_diag if hasattr(M, "initial_P_diag") else np.ones(E), dtype=np.float64)).copy() if name != "kinematic" else np.eye(2)
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  R = np.ascontiguousarray(np.atleast_2d(M.obs_noise[kind]), dtype=np.float64)
  pred = getattr(lib, f"{name}_predict"); pred.argtypes = [dp, dp, dp, ctypes.c_double]; pred.restype = None
  upd = getattr(lib, f"{name}_update_{kind}"); upd.argtypes = [dp] * 5; upd.restype = None
  ff = getattr(lib, f"{name}_f_fun"); ff.argtypes = [dp, ctypes.c_double, dp]; ff.restype = None
  err = getattr(lib, f"{name}_last_error"); err.restype = ctypes.c_int
  x0, P0 = x.copy(), P.copy()
  z = np.zeros(Z)
  out = np.zeros(D)

  def step():
    x[:] = x0; P[:] = P0
    z[:] = x0[:Z] if kind == 12 else 0.0
    pred(ptr(x), ptr(P), ptr(Q), 0.01)
    upd(ptr(x), ptr(P), ptr(z), ptr(R), None)
  t_f = timeit(lambda: ff(ptr(x0), 0.01, ptr(out)))
  t_p = timeit(lambda: pred(ptr(x), ptr(P), ptr(Q), 0.0))
  t_s = timeit(step)
  assert err() == 0, err()
  print(f"{name} kind {kind}: f_fun {t_f:.1f} us, predict {t_p:.1f} us, predict + update {t_s:.1f} us per call   (x[:3] after a step {x[:3]})")
