"""Minimal stand-in for the `cffi` package, built on ctypes (ORACLE / test infrastructure only).

cffi is not installed in this image and there is no network.  The reference imports it at
module import time (/root/reference/rednose/helpers/__init__.py:3) and its pure-Python
orchestrator `EKF_sym` uses exactly three cffi features
(/root/reference/rednose/helpers/ekf_sym.py:258-336):
  ffi.cdef(header)       -- header = the `void ...;` prototype lines
  ffi.dlopen(path)       -- attribute access returns callables, dir() lists the symbols
  ffi.cast("double *", int_address) / ffi.cast("double", value)
This shim provides those three on top of ctypes so the reference's own Python code can be
imported and run here to produce golden vectors (oracle/make_golden.py).  It is never
imported by rednose_amd/.
"""
import ctypes
import re

_PROTO = re.compile(r"^\s*void\s+(\w+)\s*\((.*)\)\s*;\s*$")


def _argtypes(arglist):
  out = []
  for a in [s.strip() for s in arglist.split(",") if s.strip()]:
    if a == "void":
      continue
    out.append(ctypes.c_void_p if "*" in a else ctypes.c_double)
  return out


class _Lib:
  def __init__(self, path, protos):
    self._dll = ctypes.CDLL(path)
    self._names = []
    for name, argtypes in protos.items():
      try:
        fn = getattr(self._dll, name)
      except AttributeError:
        continue
      fn.restype = None
      fn.argtypes = argtypes
      setattr(self, name, fn)
      self._names.append(name)

  def __dir__(self):
    return list(self._names)


class FFI:
  def __init__(self):
    self._protos = {}

  def cdef(self, header):
    for line in header.split("\n"):
      m = _PROTO.match(line)
      if m:
        self._protos[m.group(1)] = _argtypes(m.group(2))

  def dlopen(self, path):
    return _Lib(path, self._protos)

  @staticmethod
  def cast(ctype, value):
    ctype = ctype.replace(" ", "")
    if ctype == "double*":
      return ctypes.c_void_p(int(value))
    if ctype == "double":
      return ctypes.c_double(float(value))
    raise NotImplementedError(ctype)
