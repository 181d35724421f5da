"""Compile generated filter sources for gfx950 with hipcc (explicit, in-tree; no JIT cache).

Stands in for the reference's SCons step (`RednoseCompileFilter`,
/root/reference/site_scons/site_tools/rednose_filter.py:26-37): <gen script> -> {name}.cpp ->
lib{name}.so beside the header.  SCons is not available here and the target is a single GPU arch.
"""
import hashlib
import os
import re
import shutil
import subprocess

from rednose_amd.helpers import TEMPLATE_DIR

# -amdgpu-kernarg-preload-count: the first kernel arguments arrive in SGPRs with the dispatch instead of through an s_load the
# wavefront waits for before it can issue its first tile load (gfx950 supports the preload; the code keeps the s_load prologue for
# firmware that does not).  Same call, results bit-identical: headline launch 8.62 -> 8.46 us, live gyro launch 37.43 -> 37.13 us.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wno-unused-value",
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]


# Libraries that contain the register-broadcast smoother (codegen/emit_rts4.py) are scheduled with the GCN register-pressure trackers: k_rts4 lives
# within a few registers of its two-wavefronts-per-SIMD budget, and the generic trackers' estimate left it with ~10 spilled loop invariants (44 B of
# scratch) that the exact ones avoid.  Same libraries, other kernels: k_run 176 -> 164 accumulation registers, k_step_10<true> 253 -> 248, the rest
# unchanged (generated/live_maha.kernels.txt before / after).
RTS4_FLAGS = ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"]


def model_flags(source_text):
  """Extra hipcc flags a generated source asks for (a function of the text, so the source digest covers them)."""
  return list(RTS4_FLAGS) if "void k_rts4(" in source_text else []


def find_hipcc():
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found: rednose_amd builds HIP kernels for gfx950 and has no CPU fallback")


def source_digest(*texts):
  h = hashlib.sha256()
  for t in texts:
    h.update(t.encode("utf-8"))
  for hdr in ("ekf_hip_rt.h", "ekf_hip_rts.h", "ekf_plugin.h"):
    with open(os.path.join(TEMPLATE_DIR, hdr), "rb") as f:
      h.update(f.read())
  h.update((" ".join(HIPCC_FLAGS) + os.environ.get("RN_HIPCC_FLAGS", "")).encode())
  return h.hexdigest()


def compile_filter(folder, name, extra_flags=(), verbose=False):
  src = os.path.join(folder, f"{name}.hip")
  lib = os.path.join(folder, f"lib{name}.so")
  with open(src, encoding="utf-8") as f:
    extra_flags = list(extra_flags) + model_flags(f.read())
  extra_flags += os.environ.get("RN_HIPCC_FLAGS", "").split()     # A/B experiments only
  cmd = [find_hipcc()] + HIPCC_FLAGS + extra_flags + ["-Rpass-analysis=kernel-resource-usage", "-I", TEMPLATE_DIR, "-x", "hip", src, "-o", lib]
  if verbose:
    print(" ".join(cmd))
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr[-6000:]}")
  usage = kernel_resources(res.stderr)
  with open(os.path.join(folder, f"{name}.kernels.txt"), "w", encoding="utf-8") as f:
    f.write("# per-kernel resources reported by hipcc (-Rpass-analysis=kernel-resource-usage) for lib%s.so\n" % name)
    f.write("%-60s %6s %6s %8s %8s %7s %6s\n" % ("kernel", "vgprs", "agprs", "scratch", "lds", "spills", "occ"))
    for k, u in usage.items():
      f.write("%-60s %6d %6d %8d %8d %7d %6d\n" % (k[:60], u["vgprs"], u["agprs"], u["scratch"], u["lds"], u["vgpr_spill"], u["occupancy"]))
  if verbose:
    for k, u in usage.items():
      if u["scratch"] or u["vgpr_spill"]:
        print(f"note: {k} uses {u['scratch']} B of scratch per lane ({u['vgpr_spill']} spilled VGPRs)")
  compile_filter.last_usage = usage
  return lib


def spilled_kernels(usage, prefixes=("k_step", "k_run", "k_predict", "k_rts", "k_maha")):
  """Names of shipped kernels that use scratch memory or spilled registers."""
  return [k for k, u in usage.items() if k.startswith(prefixes) and (u["scratch"] > 0 or u["vgpr_spill"] > 0)]


def kernel_resources(remarks):
  """Parse hipcc's kernel-resource-usage remarks: {demangled-ish kernel name: dict(vgprs, agprs, scratch, lds, vgpr_spill,
  occupancy)}.  Written next to every library as {name}.kernels.txt: register spills cost a lone wavefront microseconds per
  access, and the one build that spilled in a lane-per-filter kernel also produced wrong results."""
  out, cur = {}, None
  keys = (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
          ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"))
  for line in remarks.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
      mangled = m.group(1)
      cur = mangled
      mm = re.search(r"(?:N_1|2rn)(\d+)(k_\w+)", mangled)        # Itanium mangling: <length><identifier>
      if mm:
        ident = mm.group(2)[:int(mm.group(1))]
        rest = mm.group(2)[int(mm.group(1)):]
        # template arguments: <bool DO_PREDICT>, <int TF> (filters per tile of the lane-per-filter kernels), or both
        mt = re.match(r"I(?:Lb([01])E)?(?:Li(\d+)E)?E", rest)
        targs = []
        if mt and mt.group(1) is not None:
          targs.append("true" if mt.group(1) == "1" else "false")
        if mt and mt.group(2) is not None:
          targs.append(mt.group(2))
        cur = ident + (f"<{','.join(targs)}>" if targs else "")
      out[cur] = dict(vgprs=0, agprs=0, scratch=0, lds=0, vgpr_spill=0, occupancy=0)
      continue
    if cur is None:
      continue
    for key, pat in keys:
      m = re.search(pat, line)
      if m:
        out[cur][key] = int(m.group(1))
  return out
