"""Development harness of the register-broadcast smoother (codegen/emit_rts4.py): emits ONLY k_rts4 of a model (with the device functions it
calls) into a small translation unit and compiles it, so that hipcc's register / scratch / occupancy report for a change is back in
seconds instead of the minutes a whole filter library takes.   python tools/rts4_dev.py [model] [out_dir]"""
import os
import pickle
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rednose_amd import build as rb
from rednose_amd.codegen import emit_rts4, tuning
from rednose_amd.codegen.emit_common import routine_device_function
from rednose_amd.helpers import TEMPLATE_DIR


def spec_of(name, cache="/tmp/rts4_dev"):
  os.makedirs(cache, exist_ok=True)
  fn = os.path.join(cache, f"{name}.spec.pkl")
  if os.path.exists(fn):
    with open(fn, "rb") as f:
      return pickle.load(f)
  import examples
  from rednose_amd.helpers import ekf_sym
  got = {}

  class Found(Exception):
    pass

  def grab(spec, **kw):
    got["spec"] = spec
    raise Found()
  orig = ekf_sym.emit
  ekf_sym.emit = grab
  try:
    examples.model_table()[name]("/tmp/rts4_dev/none")
  except Found:
    pass
  finally:
    ekf_sym.emit = orig
  with open(fn, "wb") as f:
    pickle.dump(got["spec"], f)
  return got["spec"]


def unit_text(spec):
  D, E, M = spec.dim_x, spec.dim_err, spec.dim_main_err
  src = (["#define RN_RTS_TL 1"] if os.environ.get("RTS4_TL") else []) + ['#include "ekf_hip_rt.h"', '#include "ekf_hip_rts.h"', "", "namespace {",
         f"constexpr int DIM = {D};", f"constexpr int EDIM = {E};", f"constexpr int MEDIM = {M};", ""]
  for var in spec.global_vars:
    src.append(f"__device__ double {var.name} = 0.0;")
  for r in spec.routines():
    if r.name in ("err_fun", "inv_err_fun"):
      src.append(routine_device_function(r)[0])
  with tuning.using_model(spec):
    src.append(emit_rts4.kernel(spec))
    if os.environ.get("RTS4_TRI"):
      src.append(emit_rts4.kernel(spec, tri=True))
  src.append("}  // namespace")
  src.append(f"""extern "C" int rts4_dev_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream) {{
{emit_rts4.launch(spec)}
  return (int)hipGetLastError();
}}""")
  if os.environ.get("RTS4_TL"):
    src.append("""extern "C" int rts4_dev_timeline(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rn::g_rts_tl), sizeof(unsigned long long) * 256 * 16, 0, hipMemcpyDeviceToHost);
}""")
  text = "\n".join(src)
  if "rn::sincos_fast(" in text:      # models with trigonometric terms: the helper emit() splices in (codegen/emit.py)
    from rednose_amd.codegen.lower import SINCOS_FAST
    text = text.replace('#include "ekf_hip_rts.h"\n', '#include "ekf_hip_rts.h"\n' + SINCOS_FAST, 1)
  return text


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else "live_maha"
  out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rts4_dev"
  os.makedirs(out, exist_ok=True)
  spec = spec_of(name)
  text = unit_text(spec)
  fn = os.path.join(out, f"{name}_rts4.hip")
  with open(fn, "w", encoding="utf-8") as f:
    f.write(text)
  t0 = time.time()
  cmd = [rb.find_hipcc()] + rb.HIPCC_FLAGS + ([] if os.environ.get("RTS4_NO_MODEL_FLAGS") else rb.model_flags(text)) + os.environ.get("RN_HIPCC_FLAGS", "").split() + ["-Rpass-analysis=kernel-resource-usage", "-I", TEMPLATE_DIR, "-x", "hip", fn, "-o",
                                                  os.path.join(out, f"lib{name}_rts4.so")]
  if os.environ.get("RTS4_ASM"):
    cmd = [c for c in cmd if c not in ("-shared",)] + ["-S", "--cuda-device-only"]
    cmd[cmd.index("-o") + 1] = os.path.join(out, f"{name}_rts4.s")
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode:
    print(res.stderr[-5000:])
    sys.exit(1)
  for k, u in rb.kernel_resources(res.stderr).items():
    print(k, u)
  print(f"{time.time() - t0:.1f} s, {len(text.splitlines())} lines -> {fn}")


if __name__ == "__main__":
  main()
