import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
  return np.load(os.path.join(GOLDEN, name))


def assert_close(got, want, rtol=1e-12, floor=1e-14, what="", atol=0.0):
  """|got - want| <= rtol*|want| + floor*max|want| over the last axis (per-row scale), SURVEY.md section 8c.
  `atol` is for residuals y = z - h(x), whose rounding error scales with |z|, not |y|."""
  got = np.asarray(got, dtype=np.float64)
  want = np.asarray(want, dtype=np.float64)
  assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
  scale = np.max(np.abs(want), axis=-1, keepdims=True) if want.ndim else np.abs(want)
  tol = rtol * np.abs(want) + floor * scale + atol + 1e-300
  err = np.abs(got - want)
  bad = err > tol
  if np.any(bad):
    idx = np.unravel_index(np.argmax(err / tol), err.shape)
    raise AssertionError(f"{what}: {bad.sum()} of {bad.size} entries off; worst at {idx}: got {got[idx]!r} want {want[idx]!r} "
                         f"err {err[idx]:.3e} tol {tol[idx]:.3e}")


@pytest.fixture(scope="session")
def repo_root():
  return REPO


@pytest.fixture(autouse=True)
def _poison_lds(request):
  """Before every GPU test, fill the LDS of all CUs with NaN patterns (tools/lds_poison.hip): a kernel that reads LDS it never
  wrote then produces NaNs deterministically instead of depending on what the previous kernel left there."""
  if request.node.get_closest_marker("gpu") is None:
    yield
    return
  import ctypes
  lib = os.path.join(REPO, "tools", "liblds_poison.so")
  if os.path.exists(lib):
    import torch
    if torch.cuda.is_available():
      ctypes.CDLL(lib).lds_poison(None)
      torch.cuda.synchronize()
  yield
