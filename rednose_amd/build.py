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


def build_python_binding(verbose=False):
  """Compile rednose_amd/csrc/ekf_sym_batch_py.cpp -- the pybind11 binding of rednose_amd::EKFSymBatch, the analogue of the reference's Cython
  module ekf_sym_pyx (rednose/helpers/ekf_sym_pyx.pyx, built there by SCons: rednose/SConscript) -- in-tree, next to the Python face that
  imports it (rednose_amd/helpers/ekf_sym_pyx.py).  Host code only (the kernels live in the generated filter libraries the class dlopens);
  hipcc because the header includes the HIP runtime API.  Skipped when the module is newer than its sources.  Returns the module's path."""
  import sysconfig
  import pybind11
  here = os.path.dirname(os.path.abspath(__file__))
  repo = os.path.dirname(here)
  src = os.path.join(here, "csrc", "ekf_sym_batch_py.cpp")
  hdr = os.path.join(repo, "include", "rednose_amd", "ekf_sym_batch.hpp")
  out = os.path.join(here, "helpers", "_ekf_sym_batch" + sysconfig.get_config_var("EXT_SUFFIX"))
  if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
    return out
  cmd = [find_hipcc(), "-O2", "-std=c++17", "-shared", "-fPIC", "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
         "-I", os.path.join(repo, "include"), src, "-o", out, "-ldl"]
  if verbose:
    print(" ".join(cmd))
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr[-4000:]}")
  return out


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
  if any("k_rts4" in k for k in usage):      # inline-assembly DPP operands: hipcc's hazard pass does not see them (dpp_hazards)
    dis = disassemble(lib)
    for k in usage:
      if "k_rts4" in k:        # each kernel on its own (k_rts4, k_rts4_tri): <length><identifier>E of the Itanium mangling
        hz = dpp_hazards(dis, kernel=f"{len(k)}{k}E") if dis is not None else []
        usage[k]["dpp_hazards"] = len(hz)
        if hz and verbose:
          print(f"note: {k} has {len(hz)} DPP read-after-write hazard(s), first: {hz[0]}")
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


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
  """Disassembly (llvm-objdump -d) of the gfx950 code object inside a HIP shared library, or None when the LLVM tools are absent."""
  import tempfile
  objdump = os.path.join(LLVM_BIN, "llvm-objdump")
  if not os.path.exists(objdump):
    return None
  with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "lib.hipfb"), os.path.join(d, "lib.co")
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True, capture_output=True)
    subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    return subprocess.run([objdump, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def _vgprs(operand):
  """VGPR numbers an assembly operand names: v7, -v7, |v7|, v[6:7] (anything else: none)."""
  m = re.fullmatch(r"[-|]*v(\d+)\|?", operand)
  if m:
    return {int(m.group(1))}
  m = re.fullmatch(r"[-|]*v\[(\d+):(\d+)\]\|?", operand)
  if m:
    return set(range(int(m.group(1)), int(m.group(2)) + 1))
  return set()


def dpp_hazards(disassembly, kernel="k_rts4", wait_states=2):
  """DPP read-after-VALU-write hazards of one kernel: a list of (address, instruction, offending earlier instruction).

  On gfx9 a VALU instruction that reads a VGPR through DPP (its src0) needs TWO wait states after the VALU instruction that wrote
  that VGPR (LLVM GCNHazardRecognizer::checkDPPHazards, DppVgprWaitStates = 2); every instruction issued in between is one wait
  state, `s_nop N` is N + 1.  hipcc inserts them for code it generates but not inside inline assembly, and emit_rts4's
  RN4_FMAC / RN4_FNMAC carry none (only RN4_BC has its `s_nop 1`): whether the DPP source of a `v_fmac_f64_dpp` was produced by
  the instruction in front of it is up to hipcc's scheduling of THIS model's expressions.  compile_filter runs this check on
  every library with k_rts4 and gen_code falls back to the smoother without DPP (`no_rts4`) when it finds anything;
  tests/test_abi.py runs it over the shipped models.  The walk is linear over the kernel's text (a branch target starts with
  the history of the instruction printed before it, which is the fall-through predecessor; a hazard across a taken branch
  would need the writer to be the last VALU instruction before an s_cbranch -- the branch itself is then one wait state and at
  least one more instruction stands between it and any DPP read in the code emit_rts4 prints)."""
  out, cur, hist = [], None, []
  for line in disassembly.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
      cur, hist = m.group(1), []
      continue
    if cur is None or kernel not in cur:
      continue
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if not m:
      continue
    op, args, addr = m.group(1), m.group(2), m.group(3)
    ops = [a.strip() for a in args.split(",")] if args else []
    if op.endswith("_dpp") or " row_newbcast:" in args or " quad_perm:" in args or " row_shr:" in args or " row_shl:" in args or " row_bcast:" in args:
      src0 = _vgprs(ops[1].split(" ")[0]) if len(ops) > 1 else set()
      for back, (w_op, w_args, w_regs) in enumerate(reversed(hist[-wait_states:])):
        if w_regs & src0:
          out.append((addr, f"{op} {args}", f"{w_op} {w_args} ({back} wait state(s) before)"))
    if op == "s_nop":
      hist += [("s_nop", "", set())] * (int(args, 0) + 1)
    elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):      # a VALU write of a VGPR (compares / lane reads write SGPRs)
      hist.append((op, args, _vgprs(ops[0].split(" ")[0]) if ops else set()))
    else:
      hist.append((op, args, set()))
    del hist[:-8]
  return out


def spilled_kernels(usage, prefixes=("k_step", "k_run", "k_predict", "k_rts", "k_maha")):
  """Names of shipped kernels that use scratch memory or spilled registers, or (k_rts4) read a DPP source too soon after it was written."""
  return [k for k, u in usage.items() if k.startswith(prefixes) and (u["scratch"] > 0 or u["vgpr_spill"] > 0 or u.get("dpp_hazards", 0) > 0)]


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
