"""ctypes access to the CPU ORACLE libraries (test infrastructure, not the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Resolution order for lib{name}.so: oracle/_ref (reference-generated sympy block, prebuilt in the
container that has /root/reference) then oracle/_port (buildable anywhere); built on demand.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
  sys.path.insert(0, HERE)

import build_oracle  # noqa: E402  pylint: disable=wrong-import-position

_dp = ctypes.c_void_p


def _ptr(a):
  return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def find_or_build(name, flavour="auto", suffix="", cflags=None):
  order = ["_ref", "_port"] if flavour == "auto" else ["_" + flavour]
  for sub in order:
    fn = os.path.join(HERE, sub, f"lib{name}{suffix}.so")
    if os.path.exists(fn):
      return fn, sub[1:]
  # a variant with other compiler flags (suffix) of a model whose generated glue is already there: compile that glue --
  # no code generation, works on the GPU box (where /root/reference does not exist), and flags like -march=native are
  # resolved on the machine that will run the code
  if suffix:
    for sub in order:
      glue = os.path.join(HERE, sub, f"{name}_glue.c")
      if os.path.exists(glue):
        return build_oracle.compile_glue(name, sub, cflags=cflags, suffix=suffix), sub[1:]
  fl = flavour if flavour != "auto" else ("ref" if build_oracle.have_reference() else "port")
  return build_oracle.build(name, fl, cflags=cflags, suffix=suffix, verbose=False), fl


class OracleLib:
  def __init__(self, name, flavour="auto", suffix="", cflags=None):
    self.name = name
    self.path, self.flavour = find_or_build(name, flavour, suffix, cflags)
    self.dll = ctypes.CDLL(self.path)
    d = (ctypes.c_int * 3)()
    getattr(self.dll, f"{name}_oracle_dims")(d)
    self.D, self.E, self.M = int(d[0]), int(d[1]), int(d[2])
    self._zdim = getattr(self.dll, f"{name}_oracle_zdim")
    self._zdim.restype = ctypes.c_int

  def zdim(self, kind):
    z = self._zdim(int(kind))
    if z < 0:
      raise KeyError(kind)
    return z

  def _fn(self, sym, argtypes, restype=None):
    f = getattr(self.dll, f"{self.name}_{sym}")
    f.argtypes = argtypes
    f.restype = restype
    return f

  # --- reference scalar ABI (in-place, one filter) ---
  def predict(self, x, P, Q, dt):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    self._fn("predict", [_dp, _dp, _dp, ctypes.c_double])(_ptr(x), _ptr(P), _ptr(Q), float(dt))

  def update(self, kind, x, P, z, R, ea=None):
    ea = np.zeros(4) if ea is None else np.ascontiguousarray(ea, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    self._fn(f"update_{kind}", [_dp] * 5)(_ptr(x), _ptr(P), _ptr(z), _ptr(R), _ptr(ea))

  def call(self, sym, *args):
    """Generic sympy routine: numpy arrays -> pointers, floats -> double; last array is the output."""
    at = [(_dp if isinstance(a, np.ndarray) else ctypes.c_double) for a in args]
    self._fn(sym, at)(*[(_ptr(a) if isinstance(a, np.ndarray) else float(a)) for a in args])

  # --- oracle-only batch drivers ---
  def batch_step(self, kind, x, P, z, R, Q, dt, quat_idx=-1, flags=None, do_predict=True, ea=None):
    n = x.shape[0]
    R = np.ascontiguousarray(R, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    dt = np.ascontiguousarray(np.atleast_1d(dt), dtype=np.float64)
    r_shared = int(R.ndim == 2)
    f = self._fn("oracle_batch_step", [ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int, _dp, ctypes.c_int, _dp])
    ea = None if ea is None else np.ascontiguousarray(ea, dtype=np.float64).reshape(n, 3)
    f(int(kind), _ptr(x), _ptr(P), _ptr(z), _ptr(R), r_shared, _ptr(Q), _ptr(dt), int(dt.size == 1), n,
      int(quat_idx), _ptr(flags), int(do_predict), _ptr(ea))

  def batch_run(self, kinds, dts, x, P, z, R, Q, quat_idx=-1, flags=None, xp=None, Pp=None, xf=None, Pf=None):
    """z: (T, n, zmax) in/out, R: (T, zmax, zmax)."""
    T, n, zmax = z.shape
    kinds = np.ascontiguousarray(kinds, dtype=np.int32)
    dts = np.ascontiguousarray(dts, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    f = self._fn("oracle_batch_run", [_dp, _dp, ctypes.c_int64, _dp, _dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int64,
                                      ctypes.c_int, _dp, _dp, _dp, _dp, _dp])
    f(_ptr(kinds), _ptr(dts), T, _ptr(x), _ptr(P), _ptr(z), zmax, _ptr(R), _ptr(Q), n, int(quat_idx), _ptr(flags),
      _ptr(xp), _ptr(Pp), _ptr(xf), _ptr(Pf))
