"""CPU test: every generated library loads without a GPU and exports every symbol its committed header declares
(include/kinematic.h, include/kinematic6.h, include/kinematic9.h, include/live.h, include/feature.h), plus the generic ABI of include/rednose_amd_filter.h.
No compute call is made here; compute without a device must fail loudly, which is checked."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO

INCLUDE = os.path.join(REPO, "include")
MODELS = ["kinematic", "kinematic6", "kinematic9", "live", "feature"]


@pytest.fixture(scope="module")
def gen_dir():
  from examples import ensure_generated
  return ensure_generated(MODELS + ["live_maha"])            # hipcc cross-compiles gfx950 without a GPU


@pytest.mark.parametrize("name", MODELS)
def test_library_exports_committed_header(gen_dir, name):
  from rednose_amd.helpers import parse_prototypes
  hdr_fn = os.path.join(INCLUDE, f"{name}.h")
  assert os.path.exists(hdr_fn), "run __graft_entry__.build() to refresh include/"
  with open(hdr_fn, encoding="utf-8") as f:
    committed = f.read()
  with open(os.path.join(gen_dir, f"{name}.h"), encoding="utf-8") as f:
    assert f.read() == committed, f"include/{name}.h is stale against the generator"
  protos = parse_prototypes(committed)
  assert len(protos) > 15
  dll = ctypes.CDLL(os.path.join(gen_dir, f"lib{name}.so"))
  for sym in protos:
    assert hasattr(dll, sym), f"lib{name}.so does not export {sym}"
  # the reference's scalar symbols must all be there with `void` return (ekf_sym.py:149-165)
  for sym in ("predict", "f_fun", "F_fun", "err_fun", "inv_err_fun", "H_mod_fun"):
    assert protos[f"{name}_{sym}"][0] is None
  d = (ctypes.c_int * 3)()
  getattr(dll, f"{name}_dims")(d)
  assert tuple(d) == {"kinematic": (2, 2, 2), "kinematic6": (6, 6, 6), "kinematic9": (9, 9, 9), "live": (23, 22, 22),
                      "feature": (15, 15, 6)}[name]
  md = (ctypes.c_int * 5)()
  getattr(dll, f"{name}_msckf_dims")(md)
  assert tuple(md) == ((6, 6, 3, 3, 3) if name == "feature" else (tuple(d)[0], tuple(d)[1], 0, 0, 0))
  assert hasattr(dll, f"{name}_batch_augment") == (name == "feature")
  if name == "feature":        # feature-track kind: extra-argument Jacobian exported, 3 extra arguments
    assert hasattr(dll, "feature_He_2") and not hasattr(dll, "feature_He_1")
    assert dll.feature_kind_eadim(2) == 3 and dll.feature_kind_eadim(1) == 0


def test_generic_header_macros_cover_generated_symbols(gen_dir):
  """Every symbol family documented in rednose_amd_filter.h exists in a generated library, and vice versa."""
  with open(os.path.join(INCLUDE, "rednose_amd_filter.h"), encoding="utf-8") as f:
    text = f.read()
  tri = {"batch_run_tri", "batch_rts_tri", "batch_tri_unpack", "batch_tri_pack"}      # packed-triangle trace: libraries with has_tri_trace() == 1 (live below)
  documented = {a + b for a, b in re.findall(r"RN_FN\(name, (\w+?)(?:##k(?:##(\w+))?)?\)", text)} - {"sym", "batch_augment"} - tri   # augment: MSCKF models only
  from rednose_amd.helpers import parse_prototypes
  with open(os.path.join(gen_dir, "kinematic6.h"), encoding="utf-8") as f:
    protos = parse_prototypes(f.read())
  generated = set()
  for sym in protos:
    s = sym[len("kinematic6_"):]
    generated.add(re.sub(r"_\d+(_masked|_ckpt)?$", lambda m: "_" + (m.group(1) or ""), s))      # the kind number is part of the symbol
  assert generated == documented, (sorted(generated - documented), sorted(documented - generated))
  with open(os.path.join(gen_dir, "live.h"), encoding="utf-8") as f:
    live = parse_prototypes(f.read())
  assert all(f"live_{t}" in live for t in tri) and not any(f"kinematic6_{t}" in protos for t in tri)
  dll = ctypes.CDLL(os.path.join(gen_dir, "liblive.so"))
  assert dll.live_has_tri_trace() == 1 and ctypes.CDLL(os.path.join(gen_dir, "libkinematic6.so")).kinematic6_has_tri_trace() == 0


def test_compute_without_device_fails_loudly(gen_dir):
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is present")
  from rednose_amd.helpers import KalmanError
  from rednose_amd.helpers.ekf_sym import EKF_sym, BatchedEKF
  f = EKF_sym(gen_dir, "kinematic", np.eye(2), np.zeros(2), np.eye(2), 2, 2)
  with pytest.raises(KalmanError):
    f.predict_and_update_batch(0.0, 1, np.zeros((1, 1)), np.ones((1, 1, 1)))
  with pytest.raises(KalmanError):
    BatchedEKF(gen_dir, "kinematic", np.eye(2), np.zeros(2), np.eye(2), 2, 2, batch=8)


@pytest.mark.parametrize("name", ["kinematic", "kinematic6", "kinematic9", "live", "live_maha", "feature"])
def test_step_kernels_do_not_spill(gen_dir, name):
  """Every library is built with hipcc's per-kernel resource report next to it ({name}.kernels.txt).  The step kernels of the
  shipped models must not touch scratch memory: a spill costs a lone wavefront microseconds per access, and the one
  lane-per-filter build that spilled (8 error states) also produced wrong results."""
  fn = os.path.join(gen_dir, f"{name}.kernels.txt")
  assert os.path.exists(fn)
  rows = {}
  with open(fn, encoding="utf-8") as f:
    for line in f:
      if line.startswith("#") or line.startswith("kernel"):
        continue
      parts = line.split()
      rows[parts[0]] = dict(zip(("vgprs", "agprs", "scratch", "lds", "spills", "occ"), map(int, parts[1:7])))
  steps = {k: v for k, v in rows.items() if k.startswith("k_step_") or k == "k_predict"}
  assert steps, rows.keys()
  for k, v in steps.items():
    # zero: round 3 allowed the live accelerometer kernel 4 registers in scratch; with the slot coefficients of the matrix phase
    # read in one batch (emit_wide2._in_registers) it needs 246-256 registers and none in scratch
    assert v["scratch"] == 0 and v["spills"] == 0, f"{name}: {k} spills ({v})"
    assert v["lds"] <= 65536
  # fused multi-step kernels and smoothers: whatever ships must not touch scratch memory either (gen_code drops batch_run / falls
  # back to another smoother otherwise: rednose_amd/helpers/ekf_sym.py)
  for k, v in rows.items():
    if k.startswith(("k_run", "k_rts")):
      assert v["scratch"] == 0, f"{name}: {k} uses scratch memory ({v})"


def test_no_generated_library_touches_scratch_memory(gen_dir):
  """Every library that __graft_entry__.build() generated, not only the BASELINE / example models above: no kernel of any of them uses
  scratch memory -- the 10-state test model with a 9-dimensional observation kind included (its innovation covariance is factored in LDS
  since round 5: emit_wide2._wide_obs_update; it shipped with 516 B of scratch per lane before) and the 36-state MSCKF model.  (Spills
  into accumulation registers -- the fused runs of the dense 17- / 24- / 40-state test models -- cost moves, not memory accesses.)"""
  import glob
  files = sorted(glob.glob(os.path.join(gen_dir, "*.kernels.txt")))
  assert len(files) >= 6, files
  for fn in files:
    with open(fn, encoding="utf-8") as f:
      for line in f:
        if line.startswith("#") or line.startswith("kernel"):
          continue
        parts = line.split()
        if parts[0].startswith("k_"):
          assert int(parts[3]) == 0, f"{os.path.basename(fn)}: {parts[0]} uses {parts[3]} B of scratch per lane"


@pytest.mark.parametrize("name", ["live", "live_maha", "rand8", "randz10"])
def test_dpp_sources_respect_wait_states(name):
  """k_rts4 takes cross-lane operands with inline-assembly `v_fmac_f64_dpp` / `v_mov_b64_dpp ... row_newbcast`.  A VGPR read through
  DPP needs two wait states after the VALU instruction that wrote it; hipcc's hazard pass does not look inside inline assembly, so
  whether an FMAC's DPP source was produced just in front of it is a property of each build.  rednose_amd.build.dpp_hazards walks
  the disassembly of the shipped code object (compile_filter runs the same check and gen_code falls back to `no_rts4` on a hit);
  the detector itself is checked on a synthetic listing and by asking for more wait states than the hardware needs."""
  from examples import ensure_generated
  from rednose_amd import build as rb
  gen = ensure_generated([name])
  with open(os.path.join(gen, f"{name}.kernels.txt"), encoding="utf-8") as f:
    assert "k_rts4" in f.read(), f"{name}: expected the register-broadcast smoother in this library"
  dis = rb.disassemble(os.path.join(gen, f"lib{name}.so"))
  if dis is None:
    pytest.skip("llvm-objdump not available")
  n_dpp = sum(1 for ln in dis.split("\n") if "row_newbcast:" in ln)
  assert n_dpp > 100, n_dpp
  hz = rb.dpp_hazards(dis)
  assert not hz, f"{name}: {len(hz)} DPP read-after-write hazards in k_rts4, first {hz[0]}"
  assert rb.dpp_hazards(dis, wait_states=8), "detector found nothing even at 8 wait states: it is not reading this listing"


def test_dpp_hazard_detector_on_a_synthetic_listing():
  from rednose_amd import build as rb
  head = "0000000000001000 <_ZN12_GLOBAL__N_16k_rts4EPKdS1_>:\n"
  def ins(text, addr):
    return f"\t{text}// {addr:012X}: 00000000\n"
  dpp = "v_fmac_f64_dpp v[4:5], -v[2:3], v[90:91] row_newbcast:1 row_mask:0xf bank_mask:0xf"
  bad0 = head + ins("v_fma_f64 v[2:3], v[6:7], v[8:9], v[2:3]", 0x1000) + ins(dpp, 0x1008)
  bad1 = head + ins("v_mul_f64 v[2:3], v[6:7], v[8:9]", 0x1000) + ins("s_nop 0", 0x1008) + ins(dpp, 0x100c)
  ok_nop = head + ins("v_mul_f64 v[2:3], v[6:7], v[8:9]", 0x1000) + ins("s_nop 1", 0x1008) + ins(dpp, 0x100c)
  ok_two = head + ins("v_mul_f64 v[2:3], v[6:7], v[8:9]", 0x1000) + ins("v_add_f64 v[10:11], v[6:7], v[8:9]", 0x1008) + ins("ds_read_b64 v[20:21], v30", 0x1010) + ins(dpp, 0x1018)
  ok_other = head + ins("v_mul_f64 v[90:91], v[6:7], v[8:9]", 0x1000) + ins(dpp, 0x1008)      # src1 / accumulator are ordinary reads
  ok_load = head + ins("ds_read_b64 v[2:3], v30", 0x1000) + ins(dpp, 0x1008)                   # not a VALU write: s_waitcnt orders it
  half = head + ins("v_mov_b32_e32 v3, v7", 0x1000) + ins(dpp, 0x1004)                          # one half of the pair is enough
  assert len(rb.dpp_hazards(bad0)) == 1 and len(rb.dpp_hazards(bad1)) == 1 and len(rb.dpp_hazards(half)) == 1
  assert not rb.dpp_hazards(ok_nop) and not rb.dpp_hazards(ok_two) and not rb.dpp_hazards(ok_other) and not rb.dpp_hazards(ok_load)
  other_kernel = bad0.replace("k_rts4", "k_step")
  assert not rb.dpp_hazards(other_kernel)


def test_every_generated_library_has_its_smoother_and_fused_run_as_documented(gen_dir):
  """gen_code ships a library WITHOUT batch_rts (a warning, not an error) when no smoother variant fits the register file, and without the fused
  run for the dense 32- / 56-state test models (README "Limits").  Which libraries those are is pinned here, so that a change that pushes a
  smoother over the register file (the identity-gain body did that to the 56-state model's one-wavefront build until its state phase went
  through LDS) shows up on the CPU, not as a KalmanError on the GPU box."""
  import glob
  libs = sorted(glob.glob(os.path.join(gen_dir, "lib*.so")))
  assert len(libs) >= 20
  no_run = set()
  for lib in libs:
    name = os.path.basename(lib)[3:-3]
    if name.startswith("gv_"):          # (libraries of tests/test_global_vars.py: whatever that test generated)
      continue
    dll = ctypes.CDLL(lib)
    assert hasattr(dll, f"{name}_batch_rts"), f"lib{name}.so was built without batch_rts (see {name}.kernels.txt)"
    with open(os.path.join(gen_dir, f"{name}.kernels.txt"), encoding="utf-8") as f:
      assert any(ln.startswith("k_rts") for ln in f), name
    fn = getattr(dll, f"{name}_has_batch_run")
    fn.restype = ctypes.c_int
    if not fn():
      no_run.add(name)
  assert no_run == {"rand32", "rand56"}, no_run


def test_compiled_python_binding_builds_and_fails_loudly_without_a_device(gen_dir):
  """rednose_amd/helpers/_ekf_sym_batch*.so -- pybind11 over the C++ orchestrator EKFSymBatch, the analogue of the reference's Cython module
  (rednose/helpers/ekf_sym_pyx.pyx) -- builds in-tree with hipcc, imports without a GPU, exposes the EKF_sym_pyx surface, and constructing a
  filter without a device raises (no CPU path behind it)."""
  from rednose_amd import build as rb
  path = rb.build_python_binding()
  assert os.path.exists(path) and os.path.dirname(path).endswith(os.path.join("rednose_amd", "helpers"))
  from rednose_amd.helpers import _ekf_sym_batch as m
  from rednose_amd.helpers.ekf_sym_pyx import EKF_sym_pyx
  for meth in ("init_state", "state", "covs", "set_filter_time", "get_filter_time", "set_global", "reset_rewind", "predict", "predict_and_update_batch"):
    assert hasattr(m.EKFSymBatch, meth) and hasattr(EKF_sym_pyx, meth), meth          # ekf_sym_pyx.pyx:85-179
  for meth in ("augment", "get_augment_times", "rts_smooth", "maha_test"):            # :181-192 -- NotImplementedError there and here
    assert hasattr(EKF_sym_pyx, meth)
  import torch
  if not torch.cuda.is_available():
    with pytest.raises(RuntimeError, match="device|hipMalloc"):
      EKF_sym_pyx(gen_dir, "kinematic", np.diag([0.01, 4.0]), np.array([0.5, 0.0]), np.eye(2), 2, 2)


def test_loader_backends(gen_dir):
  """load_code binds the same prototypes through either backend: cffi when importable (what the reference uses), ctypes
  otherwise or on request.  BatchedEKF always asks for ctypes (it passes ctypes pointers); a forced "cffi" without cffi
  installed must raise instead of silently switching."""
  from rednose_amd.helpers import load_code, CtypesFFI
  ffi, lib = load_code(gen_dir, "kinematic", backend="ctypes")
  assert isinstance(ffi, CtypesFFI)
  d = (ctypes.c_int * 3)()
  lib.kinematic_dims(ctypes.cast(d, ctypes.c_void_p))
  assert tuple(d) == (2, 2, 2)
  assert ffi.string(lib.kinematic_last_error_string()) == b""
  assert "kinematic_h_1" in dir(lib) and "kinematic_batch_run" in dir(lib)
  try:
    import cffi  # noqa: F401
    have_cffi = hasattr(cffi, "FFI") and hasattr(cffi.FFI, "cdef")
  except ImportError:
    have_cffi = False
  if have_cffi:
    ffi2, lib2 = load_code(gen_dir, "kinematic", backend="cffi")
    dd = ffi2.new("int[3]")
    lib2.kinematic_dims(dd)
    assert tuple(dd) == (2, 2, 2)
  else:
    with pytest.raises(ImportError):
      load_code(gen_dir, "kinematic", backend="cffi")
    assert isinstance(load_code(gen_dir, "kinematic")[0], CtypesFFI)
