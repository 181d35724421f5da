"""Compile generated filter sources for gfx950 with hipcc (explicit, in-tree; no JIT cache).

Stands in for the reference's SCons step (`RednoseCompileFilter`,
/root/reference/site_scons/site_tools/rednose_filter.py:26-37): <gen script> -> {name}.cpp ->
lib{name}.so beside the header.  SCons is not available here and the target is a single GPU arch.
"""
import hashlib
import os
import shutil
import subprocess

from rednose_amd.helpers import TEMPLATE_DIR

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wno-unused-value"]


def find_hipcc():
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found: rednose_amd builds HIP kernels for gfx950 and has no CPU fallback")


def source_digest(*texts):
  h = hashlib.sha256()
  for t in texts:
    h.update(t.encode("utf-8"))
  for hdr in ("ekf_hip_rt.h", "ekf_hip_rts.h"):
    with open(os.path.join(TEMPLATE_DIR, hdr), "rb") as f:
      h.update(f.read())
  h.update((" ".join(HIPCC_FLAGS) + os.environ.get("RN_HIPCC_FLAGS", "")).encode())
  return h.hexdigest()


def compile_filter(folder, name, extra_flags=(), verbose=False):
  src = os.path.join(folder, f"{name}.hip")
  lib = os.path.join(folder, f"lib{name}.so")
  extra_flags = list(extra_flags) + os.environ.get("RN_HIPCC_FLAGS", "").split()     # A/B experiments only
  cmd = [find_hipcc()] + HIPCC_FLAGS + extra_flags + ["-I", TEMPLATE_DIR, "-x", "hip", src, "-o", lib]
  if verbose:
    print(" ".join(cmd))
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr[-6000:]}")
  return lib
