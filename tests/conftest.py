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


def fullpiv_kernel(M):
  """Basis of the right null space of M (rows x cols) the way Eigen's FullPivLU::kernel() builds it (ekf_c.c:71 calls it on Hea^T):
  Gaussian elimination with full pivoting, P M Q = L U, U = [U1 U2], kernel vectors Q [-U1^-1 U2 ; I].  Pivot search column by column,
  first maximum (Eigen's maxCoeff visitor); all min(rows, cols) steps unless the corner is exactly zero; a pivot counts towards the rank
  when it exceeds epsilon x min(rows, cols) x the largest pivot met.  oracle/ekf_oracle.c:fullpiv_kernel and rn::nullspace_residual
  (rednose_amd/codegen/lower.py) are the same restatement in C / HIP."""
  import numpy as np
  U = np.array(M, dtype=np.float64)
  rows, cols = U.shape
  perm = list(range(cols))
  piv = []
  for k in range(min(rows, cols)):
    sub = np.abs(U[k:, k:])
    pc, pr = np.unravel_index(np.argmax(sub.T), sub.T.shape)      # column-major scan, first maximum
    best = sub[pr, pc]
    pr, pc = pr + k, pc + k
    if best == 0.0:
      break
    piv.append(best)
    U[[k, pr]] = U[[pr, k]]
    U[:, [k, pc]] = U[:, [pc, k]]
    perm[k], perm[pc] = perm[pc], perm[k]
    for i in range(k + 1, rows):
      U[i, k:] -= U[i, k] / U[k, k] * U[k, k:]
  rank = 0
  while rank < len(piv) and piv[rank] > 2.220446049250313e-16 * min(rows, cols) * max(piv):
    rank += 1
  nk = cols - rank
  ker = np.zeros((cols, nk))
  for c in range(nk):
    v = np.linalg.solve(np.triu(U[:rank, :rank]), -U[:rank, rank + c])
    for i in range(rank):
      ker[perm[i], c] = v[i]
    ker[perm[rank + c], c] = 1.0
  return ker
