#!/usr/bin/env python3
"""Build the CPU ORACLE libraries (test infrastructure, not the product).

Two flavours, same C restatement of the Eigen template (oracle/ekf_oracle.c), differing only in
where the sympy-generated f/F/h/H/H_mod/err C99 block comes from:

  ref   oracle/_ref/lib{name}.so   -- the block is produced by the REFERENCE's own gen_code
        (/root/reference/rednose/helpers/ekf_sym.py:29-217 + sympy_helpers.py:122-162), run here
        unmodified in a subprocess with oracle/cffi_shim on PYTHONPATH (cffi is not installed).
        Only possible where /root/reference exists (this container); the built .so travels to the
        GPU box with the snapshot (git-ignored, not gpurun-ignored).
  port  oracle/_port/lib{name}.so  -- the block is printed by sympy's C99 codegen from the routine
        list of rednose_amd.codegen.spec (no CSE, like the reference).  Buildable anywhere; used when
        oracle/_ref is absent.  tests/test_oracle.py checks ref == port where both exist.

Exports per library: the reference's scalar C-ABI ({name}_predict, {name}_update_{kind},
{name}_f_fun ... -- ekf_sym.py:149-171) plus oracle-only batch drivers ({name}_oracle_*) used by the
parity tests and by bench.py's cpu_baseline leg.

Usage:  python oracle/build_oracle.py [--flavour ref|port|auto] [models ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
SHIM = os.path.join(HERE, "cffi_shim")

# name -> how to obtain the symbolic model
#   ("refscript", path under /root/reference, generated name, extra gen kwargs)
#   ("model", "module:Class", extra gen kwargs)   -- our own model definition (examples/)
MODELS = {
  "kinematic": dict(refscript="examples/kinematic_kf.py", model="examples.kinematic_kf:KinematicKalman"),
  "compare": dict(refscript="examples/test_compare.py", model="examples.kinematic_kf:KinematicKalman", rename="compare"),
  "live": dict(refscript="examples/live_kf.py", model="examples.live_kf:LiveKalman"),
  "live_maha": dict(refscript="examples/live_kf.py", model="examples.live_kf:LiveKalman", rename="live_maha",
                    maha_test_kinds=[12]),
  "kinematic6": dict(model="examples.kinematic6_kf:Kinematic6Kalman"),
  "kinematic9": dict(model="examples.kinematic9_kf:Kinematic9Kalman"),
  "attitude": dict(model="examples.attitude_kf:AttitudeKalman"),
  "feature": dict(model="examples.feature_kf:FeatureKalman"),
  "feature36": dict(model="examples.feature_kf:WideFeatureKalman"),
  **{f"rand{n}": dict(model=f"examples.random_kf:Random{n}Kalman") for n in (3, 5, 8, 11, 13, 17, 24, 32, 40, 56)},
  **{f"randaff{n}": dict(model=f"examples.random_kf:RandomAffine{n}Kalman") for n in (5, 11)},
  "randz10": dict(model="examples.random_kf:RandomWideObs10Kalman"),
  "rand13_maha": dict(model="examples.random_kf:Random13Kalman", rename="rand13_maha", maha_test_kinds=[1, 3]),
  "kinematic6_maha": dict(model="examples.kinematic6_kf:Kinematic6Kalman", rename="kinematic6_maha",
                          maha_test_kinds=[1]),
}

CFLAGS_REF = ["-g", "-fPIC", "-O2"]          # the reference's flags, /root/reference/SConstruct:26-28


def have_reference():
  return os.path.isdir(os.path.join(REF, "rednose"))


_RUNNER = r'''
import runpy, sys, importlib, json
cfg = json.loads(sys.argv[1])
import rednose.helpers.ekf_sym as E
_orig = E.gen_code
def patched(folder, name, *a, **kw):
  if cfg.get("rename"):
    name = cfg["rename"]
  if cfg.get("maha_test_kinds"):
    kw["maha_test_kinds"] = cfg["maha_test_kinds"]
  return _orig(folder, name, *a, **kw)
E.gen_code = patched
if cfg.get("refscript"):
  sys.argv = [cfg["refscript"], "x", cfg["out"]]
  runpy.run_path(cfg["refscript"], run_name="__main__")
else:
  modname, clsname = cfg["model"].split(":")
  cls = getattr(importlib.import_module(modname), clsname)
  mdl = cls.model()
  name = mdl.pop("name")
  patched(cfg["out"], name, mdl.pop("f_sym"), mdl.pop("dt_sym"), mdl.pop("x_sym"), mdl.pop("obs_eqs"),
          mdl.pop("dim_x"), mdl.pop("dim_err"), **mdl)
'''


def run_reference_codegen(cfg, out_dir):
  """Run the reference's gen_code in a subprocess -> {name}.cpp/.h in out_dir."""
  import json
  cfg = dict(cfg, out=out_dir)
  if cfg.get("refscript"):
    cfg["refscript"] = os.path.join(REF, cfg["refscript"])
  env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
             PYTHONPATH=os.pathsep.join([SHIM, REPO, REF]))
  subprocess.run([sys.executable, "-c", _RUNNER, json.dumps(cfg)], check=True, env=env, cwd=out_dir)


def port_codegen(cfg, name, out_dir):
  """Same artefacts as the reference would write, printed from OUR routine list with sympy's C99 codegen."""
  import importlib
  sys.path.insert(0, REPO)
  from sympy.utilities import codegen
  import sympy as sp
  from rednose_amd.codegen.spec import build_spec
  modname, clsname = cfg["model"].split(":")
  cls = getattr(importlib.import_module(modname), clsname)
  mdl = cls.model()
  mdl["name"] = name
  if cfg.get("maha_test_kinds"):
    mdl["maha_test_kinds"] = cfg["maha_test_kinds"]
  spec = build_spec(**mdl)
  routines = []
  for r in spec.routines():
    cr = codegen.make_routine(r.name, r.expr, language="C99")
    ordered = []
    for a in r.args:
      if a is None:
        ordered.append(codegen.InputArgument(sp.Symbol('unused'), dimensions=[1, 1]))
        continue
      hit = [ca for ca in cr.arguments if str(ca.name) == str(a.name)]
      ordered.append(hit[0] if hit else codegen.InputArgument(a, dimensions=[1, 1]))
    ordered += [ca for ca in cr.arguments if isinstance(ca, codegen.OutputArgument)]
    cr.arguments = ordered
    routines.append(cr)
  [(_, c_code), (_, c_header)] = codegen.get_code_generator('C', 'ekf', 'C99').write(routines, "ekf")
  strip = lambda s: "\n".join(x for x in s.split("\n") if x and x[0] != '#')  # noqa: E731
  c_code, c_header = strip(c_code), strip(c_header)
  lines = [f"#define DIM {spec.dim_x}", f"#define EDIM {spec.dim_err}", f"#define MEDIM {spec.dim_main_err}"]
  post = []
  hdr = []
  for k in spec.kinds:
    lines.append(f"const static double MAHA_THRESH_{k.kind} = {k.maha_thresh!r};")
    hdr.append(f"void {name}_update_{k.kind}(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);")
    he = f"He_{k.kind}" if k.He_sym is not None else "NULL"
    post.append(f"  update<{k.zdim}, 3, {int(k.maha_test)}>(in_x, in_P, h_{k.kind}, H_{k.kind}, {he}, in_z, in_R, in_ea, MAHA_THRESH_{k.kind});")
  for line in c_header.split("\n"):
    if line.startswith("void "):
      hdr.append(f"void {name}_{line[5:line.index(')') + 1]};")
  hdr.append(f"void {name}_predict(double *in_x, double *in_P, double *in_Q, double dt);")
  with open(os.path.join(out_dir, f"{name}.cpp"), "w", encoding="utf-8") as f:
    f.write("\n".join(lines) + "\n/******  sympy C99 block (port)  ******/\n" + c_code + "\n#include <eigen3/Eigen/Dense>\n" + "\n".join(post) + "\n")
  with open(os.path.join(out_dir, f"{name}.h"), "w", encoding="utf-8") as f:
    f.write("#pragma once\nextern \"C\" {\n" + "\n".join(hdr) + "\n}")


def parse_generated(cpp_text):
  dims = {k: int(re.search(rf"#define {k} (\d+)", cpp_text).group(1)) for k in ("DIM", "EDIM", "MEDIM")}
  thresh = {int(k): float(v) for k, v in re.findall(r"MAHA_THRESH_(\d+) = ([0-9.eE+-]+);", cpp_text)}
  updates = {}
  for z, maha, k, he in re.findall(r"update<(\d+), 3, (\d+)>\(in_x, in_P, h_(\d+), H_\d+, (\w+),", cpp_text):
    updates[int(k)] = (int(z), int(maha), he)
  start = cpp_text.index("/*****")
  end = cpp_text.index("#include <eigen3/Eigen/Dense>")
  return dims, thresh, updates, cpp_text[start:end]


def emit_glue(name, dims, thresh, updates, sympy_block, header_text):
  kinds = sorted(updates)
  out = ["/* GENERATED by oracle/build_oracle.py -- oracle glue, do not edit, not committed */",
         "#include <math.h>", "#include <stdint.h>", "#include <string.h>", "#include \"ekf_oracle.h\"", ""]
  out.append(sympy_block)
  out.append(f"static const oracle_model MDL = {{ {dims['DIM']}, {dims['EDIM']}, {dims['MEDIM']}, f_fun, F_fun, err_fun, inv_err_fun, H_mod_fun }};")
  out.append(f"void {name}_predict(double *in_x, double *in_P, double *in_Q, double dt) {{ oracle_predict(&MDL, in_x, in_P, in_Q, dt); }}")
  for k in kinds:
    z, maha, he = updates[k]
    out.append(f"void {name}_update_{k}(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea) {{"
               f" oracle_update(&MDL, {z}, {maha}, {thresh[k]!r}, h_{k}, H_{k}, {he}, in_x, in_P, in_z, in_R, in_ea); }}")
  # thin wrappers for the sympy routines, from the reference-format header
  for line in header_text.split("\n"):
    m = re.match(rf"void {name}_(\w+)\((.*)\);", line)
    if not m or m.group(1) == "predict" or m.group(1).startswith("update_"):
      continue
    fn, args = m.groups()
    call = ", ".join(a.split()[-1].replace("*", "") for a in args.split(","))
    out.append(f"void {name}_{fn}({args}) {{ {fn}({call}); }}")
  # introspection + batch drivers (oracle only)
  out.append(f"void {name}_oracle_dims(int *d) {{ d[0] = {dims['DIM']}; d[1] = {dims['EDIM']}; d[2] = {dims['MEDIM']}; }}")
  out.append(f"int {name}_oracle_zdim(int kind) {{ switch (kind) {{ " + " ".join(f"case {k}: return {updates[k][0]};" for k in kinds) + " default: return -1; } }")
  out.append(f"""
static int upd_dispatch(int kind, double *x, double *P, double *z, const double *R, double *ea) {{
  switch (kind) {{
{chr(10).join(f"    case {k}: return oracle_update(&MDL, {updates[k][0]}, {updates[k][1]}, {thresh[k]!r}, h_{k}, H_{k}, {updates[k][2]}, x, P, z, R, ea);" for k in kinds)}
    default: return -1;
  }}
}}
/* one predict(dt) [+ renorm] + one update(kind) [+ renorm] on n filters; x:(n,D) P:(n,E,E) z:(n,Z) in/out y
 * R:(Z,Z) shared or (n,Z,Z); kind < 0 => predict only; quat_idx < 0 => no renormalisation (ekf_sym.cc:196-219) */
void {name}_oracle_batch_step(int kind, double *x, double *P, double *z, const double *R, int r_shared,
                              const double *Q, const double *dt, int dt_shared, int64_t n, int quat_idx,
                              unsigned char *flags, int do_predict, const double *ea) {{
  const int D = {dims['DIM']}, E = {dims['EDIM']};
  int Z = 0;
  switch (kind) {{ {" ".join(f"case {k}: Z = {updates[k][0]}; break;" for k in kinds)} default: break; }}
  #pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {{
    double *xi = x + i * D, *Pi = P + i * E * E;
    if (do_predict) {{
      oracle_predict(&MDL, xi, Pi, Q, dt_shared ? dt[0] : dt[i]);
      if (quat_idx >= 0) oracle_normalize_quat(xi, quat_idx);
    }}
    if (kind >= 0) {{
      double eal[4] = {{0}};
      if (ea) memcpy(eal, ea + i * 3, sizeof(double) * 3);      /* extra args of feature-track kinds: (n, 3) */
      int g = upd_dispatch(kind, xi, Pi, z + i * Z, r_shared ? R : R + i * Z * Z, eal);
      if (quat_idx >= 0) oracle_normalize_quat(xi, quat_idx);
      if (flags) flags[i] = (unsigned char)g;
    }}
  }}
}}
/* T steps with a shared schedule: kinds[t], dts[t]; z laid out (T, n, ZMAX), R (T, ZMAX, ZMAX) shared per step.
 * Optional trace buffers (each may be NULL): xp (T,n,D) Pp (T,n,E,E) after predict; xf, Pf after update. */
void {name}_oracle_batch_run(const int *kinds, const double *dts, int64_t T, double *x, double *P, double *z, int zmax,
                             const double *R, const double *Q, int64_t n, int quat_idx, unsigned char *flags,
                             double *xp, double *Pp, double *xf, double *Pf) {{
  const int D = {dims['DIM']}, E = {dims['EDIM']};
  #pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {{
    double *xi = x + i * D, *Pi = P + i * E * E;
    for (int64_t t = 0; t < T; t++) {{
      oracle_predict(&MDL, xi, Pi, Q, dts[t]);
      if (quat_idx >= 0) oracle_normalize_quat(xi, quat_idx);
      if (xp) memcpy(xp + (t * n + i) * D, xi, sizeof(double) * D);
      if (Pp) memcpy(Pp + (t * n + i) * E * E, Pi, sizeof(double) * E * E);
      double eal[4] = {{0}};
      int g = upd_dispatch(kinds[t], xi, Pi, z + (t * n + i) * zmax, R + t * zmax * zmax, eal);
      if (quat_idx >= 0) oracle_normalize_quat(xi, quat_idx);
      if (flags) flags[t * n + i] = (unsigned char)g;
      if (xf) memcpy(xf + (t * n + i) * D, xi, sizeof(double) * D);
      if (Pf) memcpy(Pf + (t * n + i) * E * E, Pi, sizeof(double) * E * E);
    }}
  }}
}}
""")
  return "\n".join(out)


def build(name, flavour="auto", cflags=None, suffix="", verbose=True):
  cfg = MODELS[name]
  if flavour == "auto":
    flavour = "ref" if have_reference() else "port"
  if flavour == "ref" and not have_reference():
    raise RuntimeError("/root/reference is not present: only the 'port' flavour can be built here")
  out_dir = os.path.join(HERE, "_ref" if flavour == "ref" else "_port")
  os.makedirs(out_dir, exist_ok=True)
  with tempfile.TemporaryDirectory() as tmp:
    if flavour == "ref":
      run_reference_codegen(cfg, tmp)
    else:
      port_codegen(cfg, name, tmp)
    with open(os.path.join(tmp, f"{name}.cpp"), encoding="utf-8") as f:
      cpp = f.read()
    with open(os.path.join(tmp, f"{name}.h"), encoding="utf-8") as f:
      hdr = f.read()
  dims, thresh, updates, block = parse_generated(cpp) if flavour == "ref" else _parse_port(cpp)
  glue = emit_glue(name, dims, thresh, updates, block, hdr)
  glue_fn = os.path.join(out_dir, f"{name}_glue.c")
  with open(glue_fn, "w", encoding="utf-8") as f:
    f.write(glue)
  with open(os.path.join(out_dir, f"{name}.h"), "w", encoding="utf-8") as f:
    f.write(hdr)
  lib = os.path.join(out_dir, f"lib{name}{suffix}.so")
  cmd = ["gcc", "-std=gnu11"] + (cflags or CFLAGS_REF) + ["-fopenmp", "-shared", "-I", HERE, glue_fn,
                                                         os.path.join(HERE, "ekf_oracle.c"), "-o", lib, "-lm"]
  subprocess.run(cmd, check=True)
  if verbose:
    print(f"[oracle] built {lib} ({flavour}; DIM={dims['DIM']} EDIM={dims['EDIM']} kinds={sorted(updates)})")
  return lib


def compile_glue(name, sub, cflags=None, suffix=""):
  """gcc on an existing {sub}/{name}_glue.c (written by build()) with other flags -> {sub}/lib{name}{suffix}.so."""
  out_dir = os.path.join(HERE, sub)
  lib = os.path.join(out_dir, f"lib{name}{suffix}.so")
  cmd = ["gcc", "-std=gnu11"] + (cflags or CFLAGS_REF) + ["-fopenmp", "-shared", "-I", HERE, os.path.join(out_dir, f"{name}_glue.c"),
                                                         os.path.join(HERE, "ekf_oracle.c"), "-o", lib, "-lm"]
  subprocess.run(cmd, check=True)
  return lib


def _parse_port(cpp_text):
  dims = {k: int(re.search(rf"#define {k} (\d+)", cpp_text).group(1)) for k in ("DIM", "EDIM", "MEDIM")}
  thresh = {int(k): float(v) for k, v in re.findall(r"MAHA_THRESH_(\d+) = ([0-9.eE+-]+);", cpp_text)}
  updates = {int(k): (int(z), int(m), he) for z, m, k, he in re.findall(r"update<(\d+), 3, (\d+)>\(in_x, in_P, h_(\d+), H_\d+, (\w+),", cpp_text)}
  start = cpp_text.index("/******  sympy C99 block (port)")
  end = cpp_text.index("#include <eigen3/Eigen/Dense>")
  return dims, thresh, updates, cpp_text[start:end]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--flavour", default="auto", choices=["auto", "ref", "port", "both"])
  ap.add_argument("models", nargs="*", default=list(MODELS))
  a = ap.parse_args()
  for m in a.models:
    if a.flavour == "both":
      if have_reference():
        build(m, "ref")
      build(m, "port")
    else:
      build(m, a.flavour)


if __name__ == "__main__":
  main()
