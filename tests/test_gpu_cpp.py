"""C++ host side: rednose_amd::EKFSymBatch (include/rednose_amd/ekf_sym_batch.hpp, the batched counterpart of the reference's
C++ EKFSym) driven by a small C++ program over the known-answer stream of /root/reference/examples/test_kinematic_kf.py."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, golden


def _build():
  src = os.path.join(REPO, "tests", "cpp", "test_ekf_sym_batch.cpp")
  exe = os.path.join(REPO, "tests", "cpp", "test_ekf_sym_batch")
  if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(REPO, "include", "rednose_amd", "ekf_sym_batch.hpp"))):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(REPO, "include"), src, "-o", exe, "-ldl"], check=True)
  return exe


def test_cpp_driver_compiles():
  assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_orchestrator_known_answers(tmp_path):
  from examples import ensure_generated
  gen = ensure_generated(["kinematic"])
  g = golden("kinematic_stream.npz")
  stream = tmp_path / "stream.txt"
  with open(stream, "w", encoding="utf-8") as f:
    for t, z in zip(g["ts"], g["zs"]):
      f.write(f"{float(t)!r} {float(z)!r}\n")
  out = subprocess.run([_build(), gen, str(stream), "130"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  head = out[0].split()
  assert head[1] == "500" and head[3] == "1" and head[5] == "1"          # 500 steps, late observation rejected, unknown kind threw
  lit = g["literals"]
  maha = out[-1].split()
  assert maha[0] == "maha" and abs(float(maha[1]) - float(maha[2])) < 1e-12 * float(maha[2]) and maha[4] == "1"
  for line in out[1:-1]:
    v = [float(t) for t in line.replace("x ", "").replace("std ", "").split()]
    for got, want in zip((v[0], v[2], v[1], v[3]), lit):
      assert round(abs(got - want), 7) == 0                                # the reference's assertAlmostEqual
    assert abs(v[0] - g["xs"][-1][0]) < 1e-10 and abs(v[1] - g["xs"][-1][1]) < 1e-10
