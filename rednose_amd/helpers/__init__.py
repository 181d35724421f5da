"""Loader shim for generated filter libraries (the drop-in boundary, Python side).

Mirrors /root/reference/rednose/helpers/__init__.py:18-31 (`load_code`): a generated directory
holds `{name}.h` + `lib{name}.so`; the header's prototype lines are what gets bound.  cffi is used
when importable; otherwise the same prototypes are bound with ctypes and a tiny `ffi` facade
(`cast`) keeps reference-style calling code working unchanged.

Beyond the reference: prototype lines starting with `int ` (the batched device-pointer entry
points declared in include/rednose_amd_filter.h) are bound as well.
"""
import ctypes
import os
import platform
import re

TEMPLATE_DIR = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'templates'))

_PROTO = re.compile(r"^\s*(void|int|const char \*)\s*(\w+)\s*\((.*)\)\s*;\s*$")

_CTYPES = {
  "double": ctypes.c_double,
  "int": ctypes.c_int,
  "int64_t": ctypes.c_int64,
  "int32_t": ctypes.c_int32,
  "long": ctypes.c_long,
}


class KalmanError(Exception):
  pass


def _arg_ctype(decl):
  decl = decl.strip()
  if "*" in decl:
    return ctypes.c_void_p
  base = decl.replace("const", "").split()
  return _CTYPES[base[0]]


def parse_prototypes(header_text):
  """-> {symbol: (restype, [argtypes])} for every one-line C prototype in `header_text`."""
  protos = {}
  for line in header_text.split("\n"):
    m = _PROTO.match(line)
    if not m:
      continue
    ret, sym, args = m.groups()
    argtypes = [_arg_ctype(a) for a in args.split(",") if a.strip() and a.strip() != "void"]
    restype = None if ret == "void" else (ctypes.c_int if ret == "int" else ctypes.c_char_p)
    protos[sym] = (restype, argtypes)
  return protos


class CtypesFFI:
  """The two `ffi` operations reference-style callers use (ekf_sym.py:266-336 of the reference)."""

  @staticmethod
  def cast(ctype, value):
    ctype = ctype.replace(" ", "")
    if ctype.endswith("*"):
      return ctypes.c_void_p(int(value))
    if ctype == "double":
      return ctypes.c_double(float(value))
    if ctype in ("int", "int64_t"):
      return int(value)
    raise NotImplementedError(ctype)

  @staticmethod
  def string(value):
    """cffi's ffi.string for a `const char *` return value (ctypes already hands back bytes)."""
    return value if isinstance(value, bytes) else bytes(value or b"")


class CtypesLib:
  def __init__(self, shared_fn, protos):
    self._dll = ctypes.CDLL(shared_fn)
    self._symbols = []
    for sym, (restype, argtypes) in protos.items():
      try:
        fn = getattr(self._dll, sym)
      except AttributeError as e:
        raise KalmanError(f"{shared_fn} does not export {sym} declared in its header") from e
      fn.restype = restype
      fn.argtypes = argtypes
      setattr(self, sym, fn)
      self._symbols.append(sym)

  def __dir__(self):
    return list(self._symbols)


def lib_paths(folder, name):
  shared_ext = "dylib" if platform.system() == "Darwin" else "so"
  return os.path.join(folder, f"lib{name}.{shared_ext}"), os.path.join(folder, f"{name}.h")


def load_code(folder, name, backend=None):
  """Returns (ffi, lib) exactly like the reference's load_code.

  backend: None = cffi when importable, else ctypes (what the reference-style EKF_sym uses); "ctypes" = always the ctypes
  binding.  BatchedEKF asks for "ctypes": it marshals device pointers, streams and NULLs as ctypes values, which a cffi
  library object rejects (cdata pointers / ffi.NULL required) -- cffi is a hard dependency of every real rednose
  environment, so the batched path must not depend on which of the two is installed."""
  shared_fn, header_fn = lib_paths(folder, name)
  if not os.path.exists(shared_fn):
    raise KalmanError(f"generated filter library missing: {shared_fn} (run the model's generate_code / rednose_amd.build first)")
  with open(header_fn, encoding='utf-8') as f:
    header = f.read()
  if backend not in (None, "cffi", "ctypes"):
    raise ValueError(backend)
  if backend == "ctypes" or os.environ.get("RN_LOADER") == "ctypes":
    return CtypesFFI(), CtypesLib(shared_fn, parse_prototypes(header))

  try:
    from cffi import FFI  # type: ignore  # pylint: disable=import-outside-toplevel
    if not hasattr(FFI, "cdef"):
      raise ImportError
    keep = "\n".join(line for line in header.split("\n") if line.startswith(("void ", "int ", "const char *")))
    keep = keep.replace("int64_t", "long long")
    ffi = FFI()
    ffi.cdef(keep)
    return ffi, ffi.dlopen(shared_fn)
  except ImportError:
    if backend == "cffi":
      raise
    return CtypesFFI(), CtypesLib(shared_fn, parse_prototypes(header))
