"""Development harness of the two-wavefront fused run (codegen/emit_run2.py): emits ONLY k_run2 of a model (with the device functions it
calls) into a small translation unit and compiles it -- hipcc's register / scratch / occupancy report in seconds instead of the minutes a
whole filter library takes.   python tools/run2_dev.py [model] [out_dir]      (RUN2_ASM=1: ISA text instead of a library)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rts4_dev import spec_of
from rednose_amd import build as rb
from rednose_amd.codegen import emit_run2, emit_wide3, tuning
from rednose_amd.codegen.lower import SINCOS_FAST
from rednose_amd.helpers import TEMPLATE_DIR


def unit_text(spec):
  D, E, M = spec.dim_x, spec.dim_err, spec.dim_main_err
  GL, R, FPW = emit_wide3.layout(spec)      # (the smoothers' constants; k_run2 has its own)
  src = ['#include "ekf_hip_rt.h"', SINCOS_FAST, "", "namespace {", f"constexpr int DIM = {D};", f"constexpr int EDIM = {E};", f"constexpr int MEDIM = {M};",
         f"constexpr int GLR = {GL};", f"constexpr int RPL = {R};", f"constexpr int FPWR = {FPW};", ""]
  for var in spec.global_vars:
    src.append(f"__device__ double {var.name} = 0.0;")
  with tuning.using_model(spec):
    src.append(emit_run2.kernels(spec))
  src.append("}  // namespace")
  src.append(f"""extern "C" int run2_dev_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R,
    int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, void *stream) {{
  const double *ea = nullptr; const int32_t *augment = nullptr;
{emit_run2.launch_run()}
  return (int)hipGetLastError();
}}""")
  return "\n".join(src)


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else "live_maha"
  out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/run2_dev"
  os.makedirs(out, exist_ok=True)
  spec = spec_of(name)
  text = unit_text(spec)
  import re
  drop = os.environ.get("RUN2_DROP", "").split(",")      # bisecting register pressure: drop whole phases from the text
  if "predict" in drop:
    text = re.sub(r"\n\s*predict_rows(_qd)?_r2\(row0.*?\);", "", text)
  if "scalar" in drop:
    text = re.sub(r"\n\s*(if \(p[0n]\) |else )?scal_(predict|keep|inject)_r2\(.*?\);", "", text).replace("if (!bad) fl = scal_inject", "if (!bad) fl = 0; //")
    text = re.sub(r"case (\d+): scal_obs_\d+_r2\(.*?\); break;", r"case \1: break;", text)
  for kd in drop:
    if kd.startswith("k"):
      text = re.sub(r"case %s: update_%s_rows_r2\(.*?\); done = true; break;" % (kd[1:], kd[1:]), "", text)
  if "genq" in drop:      # (experiment) only the diagonal-Q predict
    text = re.sub(r"\} else \{\n\s*int qz = 0;\n.*?\n.*?predict_rows_r2\(.*?\);\n\s*\}", "}", text, flags=re.S)
  if "ubar" in drop:      # (experiment) no barriers inside the update functions
    text = text.replace("rn::wg_barrier();      // dx, flags -> the scalar wavefront", "").replace("if (he_release) rn::wg_barrier();      // He, y", "// He, y")
  if "trace" in drop:
    text = text.replace("if (tP != nullptr) {", "if (false) {")
  fn = os.path.join(out, f"{name}_run2.hip")
  with open(fn, "w", encoding="utf-8") as f:
    f.write(text)
  t0 = time.time()
  cmd = [rb.find_hipcc()] + rb.HIPCC_FLAGS + rb.model_flags(text) + os.environ.get("RN_HIPCC_FLAGS", "").split() + \
        ["-Rpass-analysis=kernel-resource-usage", "-I", TEMPLATE_DIR, "-x", "hip", fn, "-o", os.path.join(out, f"lib{name}_run2.so")]
  if os.environ.get("RUN2_ASM"):
    cmd = [c for c in cmd if c not in ("-shared",)] + ["-S", "--cuda-device-only"]
    cmd[cmd.index("-o") + 1] = os.path.join(out, f"{name}_run2.s")
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode:
    print(res.stderr[-6000:])
    sys.exit(1)
  for k, u in rb.kernel_resources(res.stderr).items():
    print(k, u)
  print(f"{time.time() - t0:.1f} s, {len(text.splitlines())} lines -> {fn}")


if __name__ == "__main__":
  main()
