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


def _build_plugin_host():
  src = os.path.join(REPO, "tests", "cpp", "test_ekf_plugin.cpp")
  exe = os.path.join(REPO, "tests", "cpp", "test_ekf_plugin")
  hdr = os.path.join(REPO, "include", "rednose_amd", "ekf_plugin.h")
  if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    # -rdynamic: the library's weak ekf_register must bind to the host's definition, as it does when the reference's
    # ekf_load.cc lives in a shared object of its own
    subprocess.run(["g++", "-O2", "-std=c++17", "-rdynamic", "-I", os.path.join(REPO, "include"), src, "-o", exe, "-ldl"], check=True)
  return exe


def _plugin_host(which):
  """`lookalike`: the host with its own registry; `reference_loader`: the same driver linked against the reference's own
  rednose/helpers/ekf_load.cc object (oracle/_ref/ref_ekf_load.o, built by __graft_entry__.build() where /root/reference exists and
  shipped to the GPU box) -- ekf_load_and_register / ekf_lookup are then the reference's code, not a restatement."""
  if which == "lookalike":
    return _build_plugin_host()
  import __graft_entry__ as ge
  exe = ge.build_reference_loader()
  if exe is None:
    pytest.skip("oracle/_ref/ref_ekf_load.o is not there: run __graft_entry__.build() where /root/reference is present")
  return exe


@pytest.mark.parametrize("host", ["lookalike", "reference_loader"])
@pytest.mark.parametrize("name,kinds,nfeat", [("kinematic", "1", 0), ("live", "3 4 9 10 12 13 14 19", 0), ("feature", "1 2", 1)])
def test_plugin_descriptor_loads_like_the_reference_host(name, kinds, nfeat, host):
  """ekf_get() / struct EKF / self-registration (rednose/helpers/ekf.h:14-42, ekf_load.cc:22-39) of a generated library, through
  a host program written like the reference's loader and through the reference's loader itself.  No device needed: only the
  descriptor is read.  (The reference's registry ends up holding the descriptor twice -- once from the library's constructor, once
  from ekf_load_and_register, ekf_load.cc:38 -- as it does for the reference's own libraries.)"""
  from examples import ensure_generated
  gen = ensure_generated([name])
  out = subprocess.run([_plugin_host(host), gen, name], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  reg = 1 if host == "lookalike" else 2
  assert out[0] == f"name {name} kinds {kinds} feature_kinds {nfeat} registered {reg} same 1", out
  assert out[1].startswith("complete 1"), out


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["lookalike", "reference_loader"])
def test_plugin_known_answers_through_descriptor(tmp_path, host):
  """The calls EKFSym makes (ekf->predict, ekf->updates.at(kind)) over /root/reference/examples/test_kinematic_kf.py's stream, the
  library found and registered by the reference's own ekf_load_and_register / ekf_lookup in the `reference_loader` build."""
  from examples import ensure_generated
  gen = ensure_generated(["kinematic"])
  g = golden("kinematic_stream.npz")
  stream = tmp_path / "stream.txt"
  with open(stream, "w", encoding="utf-8") as f:
    for t, z in zip(g["ts"], g["zs"]):
      f.write(f"{float(t)!r} {float(z)!r}\n")
  out = subprocess.run([_plugin_host(host), gen, "kinematic", str(stream)], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  v = out[-1].split()
  assert v[0] == "steps" and v[1] == "500"
  got = (float(v[3]), float(v[6]), float(v[4]), float(v[7]))
  for a, want in zip(got, g["literals"]):
    assert round(abs(a - want), 7) == 0


@pytest.mark.gpu
def test_cpp_orchestrator_rewind_ring(tmp_path):
  """/root/reference/examples/test_compare.py:103-120 through the C++ class: samples 20 and 40 arrive swapped; the checkpoint
  ring (rewind_to_keep = 512) rewinds the batch, applies the late observation and replays -- states after every arrival as the
  reference's orchestrators produce them (tests/golden/compare_rewind.npz)."""
  from examples import ensure_generated
  gen = ensure_generated(["kinematic"])
  g = golden("compare_rewind.npz")
  stream = tmp_path / "stream.txt"
  with open(stream, "w", encoding="utf-8") as f:
    for t, z in zip(g["ts"], g["zs"]):
      f.write(f"{float(t)!r} {float(z)!r}\n")
  out = subprocess.run([_build(), gen, str(stream), "70", "rewind"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  assert out[-1] == "too_old_dropped 1 untouched 1"
  rows = np.array([[float(v) for v in line.split()] for line in out[:-1]])
  assert rows.shape[0] == len(g["ts"]) and (rows[:, 0] == 1).all()
  assert np.abs(rows[:, 1] - g["filter_times"]).max() < 1e-12
  for col in (2, 4):
    assert np.abs(rows[:, col:col + 2] - g["xs"]).max() < 1e-9
  assert (np.diff(g["ts"]) < 0).any(), "the stream must contain a late observation"


@pytest.mark.gpu
@pytest.mark.parametrize("filt", [0, 3])
def test_cpp_orchestrator_n_observations_per_call(tmp_path, filt):
  """EKFSym::predict_and_update_batch with vectors of observations (ekf_sym.cc:83-117,158-194) through the C++ class: calls of 1-3
  observations with a different noise matrix each, ONE predict and ONE checkpoint per call, a late multi-observation call rewinding over
  multi-observation checkpoints -- against the reference instance that was fed the same log (tests/golden/multi_obs.npz part A)."""
  from examples import ensure_generated
  from examples.kinematic9_kf import Kinematic9Kalman as K9
  gen = ensure_generated(["kinematic9"])
  g = golden("multi_obs.npz")
  TB = g["A_t"].shape[1]
  stream = tmp_path / "multi.txt"
  num = lambda a: " ".join(repr(float(v)) for v in np.ravel(a))      # noqa: E731
  with open(stream, "w", encoding="utf-8") as f:
    f.write("\n".join([num(K9.Q), num(K9.initial_x), num(np.diag(K9.initial_P_diag)), num(K9.obs_noise[1]), num(K9.obs_noise[2]), num(K9.obs_noise[3])]) + "\n")
    for j in range(TB):
      k, n = int(g["A_kind"][filt, j]), int(g["A_n"][filt, j])
      Z = K9.obs_noise[k].shape[0]
      f.write(f"{float(g['A_t'][filt, j])!r} {k} {n}")
      for m in range(n):
        f.write(" " + num(g["A_z"][filt, j, m, :Z]) + " " + repr(float(g["A_Rscale"][filt, j, m])))
      f.write("\n")
  out = subprocess.run([_build(), gen, str(stream), "33", "multi"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  assert out[-1] == "mismatch_threw 1"
  assert len(out) == TB + 1
  assert (np.diff(g["A_t"][filt]) < 0).any(), "the log must contain a late call"
  for j, line in enumerate(out[:-1]):
    v = np.array([float(t) for t in line.split()])
    k, n = int(g["A_kind"][filt, j]), int(g["A_n"][filt, j])
    Z = K9.obs_noise[k].shape[0]
    assert v[0] == 1 and len(v) == 2 + 18 + n * Z
    for off in (2, 11):
      assert np.abs(v[off:off + 9] - g["A_x"][filt, j]).max() < 1e-8 * max(1.0, np.abs(g["A_x"][filt, j]).max()), f"call {j}: state"
    assert np.abs(v[20:].reshape(n, Z) - g["A_y"][filt, j, :n, :Z]).max() < 1e-7, f"call {j}: residuals"


@pytest.mark.gpu
def test_cpp_per_filter_timelines_with_n_observations_per_call(tmp_path):
  """predict_and_update_batch_per_filter with vectors of observations: 6 filters on their own clocks, calls of 1-3 observations (one noise matrix per
  observation, shared by the batch), ring entries that hold whole calls, one late multi-observation call per filter that rewinds over 2-4 such
  entries and replays them with all their observations -- against the reference instances fed the same logs (tests/golden/multi_obs.npz part C)."""
  from examples import ensure_generated
  from examples.kinematic9_kf import Kinematic9Kalman as K9
  gen = ensure_generated(["kinematic9"])
  g = golden("multi_obs.npz")
  NC, TC = g["C_t"].shape
  stream = tmp_path / "timelines_multi.txt"
  num = lambda a: " ".join(repr(float(v)) for v in np.ravel(a))      # noqa: E731
  with open(stream, "w", encoding="utf-8") as f:
    f.write("\n".join([num(K9.Q), num(K9.initial_x), num(np.diag(K9.initial_P_diag)), num(K9.obs_noise[1]), num(K9.obs_noise[2]), num(K9.obs_noise[3]),
                       num(g["C_scale"]), f"{NC} {TC}"]) + "\n")
    for j in range(TC):
      for i in range(NC):
        f.write(f"{float(g['C_t'][i, j])!r} {int(g['C_kind'][i, j])} {int(g['C_n'][i, j])} " + num(g["C_z"][i, j]) + "\n")
  out = subprocess.run([_build(), gen, str(stream), "0", "timelines_multi"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  assert out[-1] == "too_many_threw 1" and len(out) == NC * TC + 1
  rows = np.array([[float(v) for v in line.split()] for line in out[:-1]]).reshape(TC, NC, 19)
  assert any((np.diff(g["C_t"][i]) < 0).any() for i in range(NC)), "the logs must contain late calls"
  for j in range(TC):
    for i in range(NC):
      k, n = int(g["C_kind"][i, j]), int(g["C_n"][i, j])
      Z = K9.obs_noise[k].shape[0]
      assert np.abs(rows[j, i, 1:10] - g["C_x"][i, j]).max() < 1e-8 * max(1.0, np.abs(g["C_x"][i, j]).max()), f"arrival {j} filter {i}: state"
      got = rows[j, i, 10:].reshape(3, 3)[:n, :Z]
      assert np.abs(got - g["C_y"][i, j, :n, :Z]).max() < 1e-7, f"arrival {j} filter {i}: residuals"
  # filter times: the newest time each filter has seen so far (a late call leaves it where the replay ends)
  for i in range(NC):
    assert np.abs(rows[:, i, 0] - np.maximum.accumulate(g["C_t"][i])).max() < 1e-12


@pytest.mark.gpu
def test_cpp_set_global_and_extra_routine():
  import sympy as sp
  from rednose_amd.helpers.ekf_sym import gen_code
  import test_global_vars as tg
  folder = os.path.join(REPO, "generated")
  gsym = sp.Symbol('gain')
  gen_code(folder, "gv_runtime", global_vars=[gsym], **tg._model(gsym))
  out = subprocess.run([_build(), folder, "-", "3", "globals"], check=True, capture_output=True, text=True).stdout.strip().split()
  # x0 + dt * gain * x1 with gain = 2.5, dt = 0.1
  assert abs(float(out[1]) - (0.5 + 0.1 * 2.5 * 0.3)) < 1e-14 and abs(float(out[2]) - 0.3) < 1e-15
  assert out[4] == "1" and out[6] == "1"


@pytest.mark.gpu
def test_cpp_orchestrator_per_filter_timelines(tmp_path):
  """EKFSymBatch::predict_and_update_batch_per_filter: 12 filters, each fed its own out-of-order log of 700 observations (a
  different swapped pair per filter, the ring of 512 wraps, one observation too old for its filter) -- every filter must follow
  the reference instance that was fed its log (tests/golden/perfilter_timelines.npz, oracle/make_golden.py)."""
  from examples import ensure_generated
  gen = ensure_generated(["kinematic"])
  g = golden("perfilter_timelines.npz")
  NA, T = g["A_t"].shape
  stream = tmp_path / "logs.txt"
  with open(stream, "w", encoding="utf-8") as f:
    for j in range(T):
      f.write(" ".join(f"{float(g['A_t'][i, j])!r} {float(g['A_z'][i, j])!r}" for i in range(NA)) + "\n")
  out = subprocess.run([_build(), gen, str(stream), str(NA), "timelines"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  rows = np.array([[float(v) for v in line.split()] for line in out]).reshape(T, NA, 4)
  assert np.array_equal(rows[:, :, 0].T != 0, g["A_none"])
  assert np.abs(rows[:, :, 1].T - g["A_ft"]).max() < 1e-12
  keep = g["A_keep"]
  assert np.abs(rows[keep][:, :, 2:4].transpose(1, 0, 2) - g["A_x"]).max() < 1e-9
  assert np.abs(rows[-1, :, 2:4] - g["A_x_final"]).max() < 1e-9


@pytest.mark.gpu
def test_cpp_per_filter_path_is_robust():
  """EKFSymBatch per-filter timelines: the host-side table of noise matrices stays bounded under a time-varying R (400 calls, a
  new R each: entries no live ring slot references are dropped), an observation too old for one filter's ring sets flag bits 4 | 5
  for that filter only, and a refused call (late observation, no ring) throws before any state -- rings, filter times, x, P -- is
  touched."""
  from examples import ensure_generated
  gen = ensure_generated(["kinematic"])
  out = subprocess.run([_build(), gen, "-", "5", "robustness"], check=True, capture_output=True, text=True).stdout.strip().split("\n")
  a = out[0].split()
  assert a[0] == "table_max" and int(a[1]) <= 130, out[0]                      # 400 distinct matrices went through a ring of 8 x 5 slots
  assert a[3] == "1" and [int(v) for v in a[5:]] == [0, 0, 48, 0, 0], out[0]
  b = out[1].split()
  assert b[1] == "1" and b[3] == "1" and abs(float(b[5]) - 1.8) < 1e-12, out[1]
