"""Assemble a generated filter library: `{name}.hip` (kernels + C ABI) and `{name}.h` (prototypes).

This is the replacement for the BACK half of the reference's gen_code
(/root/reference/rednose/helpers/ekf_sym.py:118-217).  What is kept from it is the contract:
  * `{name}.h` holds one prototype per line; every reference symbol is there with the reference's
    exact signature -- {name}_update_{kind} (:149), the sympy routine wrappers {name}_f_fun ... (:155-161),
    {name}_predict (:162), {name}_set_{var} (:166-171) -- so `load_code` style loaders keep working;
  * `lib{name}.so` is self-contained and lives beside the header.
What is new: `int {name}_batch_*` entry points over DEVICE pointers (include/rednose_amd_filter.h),
and every symbol, scalar ones included, executes on the GPU.  The C++ plugin hook of the reference
(:186-203: `const EKF {name} = {...}` + `ekf_lib_init` -> `extern "C" void *ekf_get()`, rednose/helpers/ekf.h:14-42)
is emitted too (plugin_text below, templates/ekf_plugin.h), so the reference's `ekf_load_and_register`
(ekf_load.cc:22-39) and through it `EKFSym` can load a rednose_amd library unmodified.
"""
import sympy as sp

from rednose_amd.codegen import emit_small
from rednose_amd.codegen.emit_common import routine_device_function

SMALL_MAX_E = 7    # lane-per-filter register budget: x, P and the update's temporaries in VGPRs/AGPRs without spilling.  At 8
                   # error states hipcc spills (12 VGPRs in a step kernel, 58 in the fused run) and the fused run's trace came out
                   # wrong for single filters on some runs (tools/stress_run_trace.py); 8 goes to the lane-group kernels


# Fallback structures gen_code may ask for when the first build of a model does not fit the register file (they are arguments
# of ONE emit() call, not process state: nothing about a model survives its gen_code call):
#   force_wide         the lane-per-filter build spilled registers -> lane-group family
#   no_model_defaults  the per-model tuning defaults (two wavefronts per SIMD) spilled -> general structure
#   rts_one_wave       the smoother spilled under the two-wavefronts-per-SIMD budget -> full register file
#   no_rts             the smoother still touches scratch -> library without batch_rts (forward filter unaffected)
#   no_rts4            the smoother with register-broadcast operands (emit_rts4, two wavefronts per SIMD) spilled, or one of its DPP reads follows
#                      the write of its source too closely (build.dpp_hazards) -> rn::k_rts_group
#   no_run_blk         the blocked fused run of a lane-per-filter model (emit_small.run_kernel_blk) spilled -> k_run serves untraced runs too
#   no_tri             one of the packed-triangle trace kernels (k_run2_tri / k_rts4_tri) spilled -> library without batch_run_tri / batch_rts_tri
#   no_run2            the fused run with a scalar wavefront beside the matrix wavefront (emit_run2, two wavefronts per SIMD) spilled -> k_run (emit_wide3)
#   no_run             the fused multi-step run of a model above 32 error states touches scratch -> library without batch_run
#                      (status ERR_UNSUPPORTED, the step-granular entry points cover such models)
FALLBACKS = ("force_wide", "no_model_defaults", "no_rts4", "rts_one_wave", "no_rts", "no_run2", "no_run", "no_run_blk", "no_tri")
_active = frozenset()      # fallbacks of the emit() call in progress


def family(spec, fallbacks=None):
  """Lane-per-filter kernels up to SMALL_MAX_E error states; lane-group kernels above, and for every MSCKF model
  (the null-space projection of feature-track kinds is emitted for the lane-group family only)."""
  fb = _active if fallbacks is None else fallbacks
  if any(k.He_sym is not None for k in spec.kinds) or "force_wide" in fb:
    return "wide"
  from rednose_amd.codegen import tuning
  return "small" if spec.dim_err <= min(SMALL_MAX_E, tuning.current().small_max_e) else "wide"


def _align2(n):
  return n + (n & 1)


def ea_len(k):
  """Number of extra arguments of kind k (0: the `ea` pointer is ignored)."""
  import sympy as sp
  return 0 if k.ea_sym is None else int(sp.Matrix(k.ea_sym).shape[0])


def ea_req(k):
  return " && ea" if ea_len(k) else ""


def emit(spec, fallbacks=()):
  """-> (header_text, hip_text).  `fallbacks`: see FALLBACKS."""
  global _active      # pylint: disable=global-statement
  from rednose_amd.codegen import tuning
  assert set(fallbacks) <= set(FALLBACKS), fallbacks
  _active = frozenset(fallbacks)
  try:
    with tuning.using_model(spec, enabled="no_model_defaults" not in _active):
      return _emit(spec)
  finally:
    _active = frozenset()


def plugin_text(spec):
  """The reference's plugin descriptor (ekf_sym.py:186-203): name, kinds, feature kinds and the scalar entry points by kind /
  by name, published through ekf_get() and, when the host defines it, ekf_register() at load time (ekf.h:35-42)."""
  name = spec.name
  feat = [k.kind for k in spec.kinds if k.He_sym is not None]
  L = ["", "// ---- C++ plugin hook (/root/reference/rednose/helpers/ekf.h:14-42, emitted by the reference at ekf_sym.py:186-203) ----",
       '#include "ekf_plugin.h"', "namespace {", "const EKF& rn_plugin_descriptor() {", "  static const EKF e = [] {", "    EKF d;",
       f'    d.name = "{name}";', f"    d.kinds = {{ {', '.join(str(k.kind) for k in spec.kinds)} }};",
       f"    d.feature_kinds = {{ {', '.join(str(k) for k in feat)} }};"]
  for fn in ("f_fun", "F_fun", "err_fun", "inv_err_fun", "H_mod_fun", "predict"):
    L.append(f"    d.{fn} = {name}_{fn};")
  for k in spec.kinds:
    L.append(f"    d.hs[{k.kind}] = {name}_h_{k.kind}; d.Hs[{k.kind}] = {name}_H_{k.kind}; d.updates[{k.kind}] = {name}_update_{k.kind};")
  for k in feat:
    L.append(f"    d.Hes[{k}] = {name}_He_{k};")
  for var in spec.global_vars:
    L.append(f'    d.sets["{var.name}"] = {name}_set_{var.name};')
  for r in spec.extra_routines:
    L.append(f'    d.extra_routines["{r[0]}"] = reinterpret_cast<extra_routine_t>({name}_{r[0]});')
  L += ["    return d;", "  }();", "  return e;", "}", "}  // namespace",
        'extern "C" void *ekf_get() { return (void *)&rn_plugin_descriptor(); }',
        "static void __attribute__((constructor)) rn_plugin_register(void) { if (ekf_register) ekf_register(&rn_plugin_descriptor()); }", ""]
  return "\n".join(L)


def _emit(spec):
  name = spec.name
  D, E, M = spec.dim_x, spec.dim_err, spec.dim_main_err
  EE = E * E
  fam = family(spec)
  if E > 64:
    raise NotImplementedError(f"{E} error states: the lane-group kernels hold one row of P per lane of a wavefront (<= 64)")
  has_run = "no_run" not in _active     # fused multi-step run: rows of P stay in VGPRs (emit_wide3: several rows per lane up to 32 error states, one above)
  import types
  from rednose_amd.codegen import tuning
  if fam == "wide":
    from rednose_amd.codegen import emit_run2, emit_wide2, emit_wide3
    use_run2 = has_run and tuning.current().run2 and "no_run2" not in _active and emit_run2.applicable(spec)
    from rednose_amd.codegen import emit_rts4 as _r4
    # packed-triangle trace: both structures or neither (batch_run_tri writes what batch_rts_tri reads)
    use_tri = (use_run2 and emit_run2.tri_trace(spec) and tuning.current().rts4 and not ({"no_tri", "no_rts4", "no_rts"} & set(_active)))
    if not has_run:
      use_tri = False
      fam_mod = types.SimpleNamespace(
        kernels=lambda sp_: emit_wide2.kernels(sp_) + "\n" + emit_wide2.maha_kernels(sp_),
        launch_predict=emit_wide2.launch_predict, launch_step=emit_wide2.launch_step, launch_step_ckpt=emit_wide2.launch_step_ckpt, launch_run=None,
        launch_maha=emit_wide2.launch_maha)
    else:
      # step-granular kernels: three-phase structure (emit_wide2); fused multi-step run: state resident in registers, several
      # rows of P per lane (emit_wide3)
      fam_mod = types.SimpleNamespace(
        kernels=lambda sp_: emit_wide2.kernels(sp_) + "\n" + emit_wide3.kernels(sp_, with_run=not use_run2) + "\n" +
                            (emit_run2.kernels(sp_, tri=use_tri) + "\n" if use_run2 else "") + emit_wide2.maha_kernels(sp_),
        launch_predict=emit_wide2.launch_predict, launch_step=emit_wide2.launch_step, launch_step_ckpt=emit_wide2.launch_step_ckpt,
        launch_run=emit_run2.launch_run if use_run2 else emit_wide3.launch_run,
        launch_maha=emit_wide2.launch_maha)
  else:
    use_tri = False
    fam_mod = types.SimpleNamespace(
      kernels=lambda sp_: emit_small.kernels(sp_) + "\n" + emit_small.maha_kernels(sp_),
      launch_predict=emit_small.launch_predict, launch_step=emit_small.launch_step, launch_step_ckpt=emit_small.launch_step_ckpt,
      launch_run=lambda: emit_small.launch_run(spec),
      launch_maha=emit_small.launch_maha)

  hdr = ["#pragma once", "#include <stdint.h>", "#ifdef __cplusplus", 'extern "C" {', "#endif"]
  src = [f"// GENERATED by rednose_amd.helpers.ekf_sym.gen_code for filter '{name}' -- do not edit.",
         f"// DIM={D} EDIM={E} MEDIM={M} kinds={[k.kind for k in spec.kinds]} family={fam}",
         *(["#define RN_RTS_TL 1"] if (fam == "wide" and tuning.current().wide_timeline) else []),
         *(["#define RN_EXACT_MATH 1"] if tuning.current().exact_math else []),
         '#include "ekf_hip_rt.h"', '#include "ekf_hip_rts.h"', "", "namespace {",
         f"constexpr int DIM = {D};", f"constexpr int EDIM = {E};", f"constexpr int MEDIM = {M};", ""]

  # run-time scalars of the model (reference: file-scope doubles + set_{var}, ekf_sym.py:129-132,166-171): device
  # globals written by {name}_set_{var}; like the reference they are per LIBRARY, not per filter instance
  for var in spec.global_vars:
    src.append(f"__device__ double {var.name} = 0.0;")

  # ---- the reference's per-routine functions (scalar ABI + used by the smoother) ------------------
  wrappers = []
  for r in spec.routines():
    text, params, n_out = routine_device_function(r)
    src.append(text)
    ptr_params = [(n, sz) for kind, n, sz in params if kind == "ptr"]
    c_sig = ", ".join((f"double *{n}" if kind == "ptr" else f"double {n}") for kind, n, _ in params)
    k_sig = ", ".join((f"const double* {n}" if kind == "ptr" else f"double {n}") for kind, n, _ in params)
    call = ", ".join(n for _, n, _ in params)
    src.append(f"__global__ void k_fn_{r.name}({k_sig}, double* out) {{ {r.name}({call}, out); }}")
    # host wrapper: arguments packed into the pinned staging buffer, one single-thread launch that works on it in place, the output unpacked
    offs, cur = {}, 0
    for n, sz in ptr_params:
      offs[n] = cur
      cur += _align2(sz)
    out_off = cur
    total = cur + _align2(n_out)
    w = [f"void {name}_{r.name}({c_sig}, double *out) {{",
         "  rn::Scratch& s = rn::scratch();", "  std::lock_guard<std::mutex> hold(s.mu);",
         f"  if (s.ensure({total}) != rn::OK) return;"]
    for n, sz in ptr_params:
      if n.startswith("unused"):
        continue
      w.append(f"  s.put({offs[n]}, {n}, {sz});")
    kargs = ", ".join((f"s.dev + {offs[n]}" if kind == "ptr" else n) for kind, n, _ in params)
    w.append(f"  hipLaunchKernelGGL(k_fn_{r.name}, dim3(1), dim3(1), 0, 0, {kargs}, s.dev + {out_off});")
    w.append(f"  if (s.wait(\"{r.name}\", __LINE__) != rn::OK) return;")
    w.append(f"  s.get(out, {out_off}, {n_out});")
    w.append("}")
    wrappers.append("\n".join(w))
    hdr.append(f"void {name}_{r.name}({c_sig}, double *out);")

  # ---- kernels ---------------------------------------------------------------------------------
  src.append(fam_mod.kernels(spec))
  # smoother.  Lane-per-filter models: rn::k_rts (state and covariance of a filter in one lane's registers).  Lane-group models: k_rts4
  # (emit_rts4: 8 .. 22 error states) or rn::k_rts_group (MSCKF models -- their main block is smoothed, ekf_sym.py:675-686 --, larger models,
  # and the fallback of k_rts4).
  group_rts = fam == "wide"
  from rednose_amd.codegen import emit_rts4
  use_rts4 = (group_rts and emit_rts4.applicable(spec) and tuning.current().rts4 and "no_rts4" not in _active and "no_rts" not in _active)
  use_tri = use_tri and use_rts4
  if use_rts4:
    src.append(emit_rts4.kernel(spec))
    if use_tri:
      src.append(emit_rts4.kernel(spec, tri=True))
  has_rts = (group_rts or (fam == "small" and spec.dim_main == spec.dim_x and spec.dim_main_err == spec.dim_err)) and "no_rts" not in _active
  if has_rts:
    quat = "".join(f" rn::normalize_quat<{D}>(x, {q});" for q in spec.quaternion_idxs)
    grp = f"""
  // lane-group smoother (k_rts_group): slot layout and phase functions of the three-phase step kernels
  static constexpr int DM = {spec.dim_main};
  static constexpr int EM = {M};
  static constexpr int SLOT = ::SLOT;
  static constexpr int OFF_X = SLOT_OFF_X;
  static constexpr int OFF_DT = SLOT_OFF_DT;
  static __device__ __forceinline__ void scal(const double* xin, double dt, double* sl, int norm) {{ scal_predict(xin, dt, sl, norm); }}
  static constexpr int WAVES = {2 if (M <= 22 and "rts_one_wave" not in _active) else 1};       // wavefronts per SIMD the register budget is set for (see k_rts_group)
  static __device__ __forceinline__ void mat_predict(const double (&row)[{M}], double* sB, const double* gQc, const double* sl, int cc, bool act,
                                                     double (&y)[{M}]) {{ mat_predict_rts(row, sB, gQc, sl, cc, act, y); }}""" if group_rts else ""
    src.append(f"""
// adapter handed to the hand-written smoother kernels (templates/ekf_hip_rts.h)
struct RtsModel {{
  static constexpr int D = {D};
  static constexpr int E = {E};
  static constexpr bool ID0 = {'true' if emit_rts4.dt0_path(spec) else 'false'};      // predict(dt = 0) is the identity: such steps take Ck = I (ekf_hip_rts.h)
  static __device__ __forceinline__ void f(const double* x, double dt, double* out) {{ f_fun(x, dt, out); }}
  static __device__ __forceinline__ void F(const double* x, double dt, double* out) {{ F_fun(x, dt, out); }}
  static __device__ __forceinline__ void err(const double* nom, const double* delta, double* out) {{ err_fun(nom, delta, out); }}
  static __device__ __forceinline__ void inv_err(const double* nom, const double* tru, double* out) {{ inv_err_fun(nom, tru, out); }}
  static __device__ __forceinline__ void normalize(double (&x)[{D}]) {{{quat} (void)x; }}
{grp}
}};
""")
  if spec.N > 0:
    d1, d2, d3, d4 = spec.dim_main, spec.dim_main_err, spec.dim_augment, spec.dim_augment_err
    src.append(f"""
// ---- MSCKF window shift (/root/reference/rednose/helpers/ekf_sym.py:365-391): the oldest augmented state drops out, the
// first {d3} main states become the newest one; P follows with rows/columns [{d2}, {d2 + d4}) deleted and the first {d4}
// re-appended.  One wavefront per filter, state staged through LDS (the permutation reads what it overwrites).
__device__ __forceinline__ int aug_src_err(int i) {{
  const int r = i < {E - d4} ? i : i - {E - d4};
  return r < {d2} ? r : r + {d4};
}}
__global__ __launch_bounds__(64) void k_augment(double* __restrict__ gx, double* __restrict__ gP, const int64_t n) {{
  __shared__ double s_P[{EE}];
  __shared__ double s_x[{D}];
  const int lane = threadIdx.x;
  for (int64_t f = blockIdx.x; f < n; f += gridDim.x) {{
    for (int i = lane; i < {D}; i += 64) s_x[i] = gx[f * {D} + i];
    for (int i = lane; i < {EE}; i += 64) s_P[i] = gP[f * {EE} + i];
    rn::wave_lds_sync();
    for (int i = lane; i < {D}; i += 64) {{
      const int src = i < {d1} ? i : (i < {D - d3} ? i + {d3} : i - {D - d3});
      gx[f * {D} + i] = s_x[src];
    }}
    for (int i = lane; i < {EE}; i += 64) {{
      const int r = i / {E}, c = i % {E};
      gP[f * {EE} + i] = s_P[aug_src_err(r) * {E} + aug_src_err(c)];
    }}
    rn::wave_lds_sync();
  }}
}}
""")
  src.append("}  // namespace\n")

  # ---- C ABI -----------------------------------------------------------------------------------
  abi = ['extern "C" {', ""]
  abi.append(f"void {name}_dims(int *dims) {{ dims[0] = {D}; dims[1] = {E}; dims[2] = {M}; }}")
  hdr.append(f"void {name}_dims(int *dims);")
  cases = " ".join(f"case {k.kind}: return {k.zdim};" for k in spec.kinds)
  abi.append(f"int {name}_kind_zdim(int kind) {{ switch (kind) {{ {cases} default: return -1; }} }}")
  hdr.append(f"int {name}_kind_zdim(int kind);")
  mcases = " ".join(f"case {k.kind}: return {int(k.maha_test)};" for k in spec.kinds)
  abi.append(f"int {name}_kind_maha(int kind) {{ switch (kind) {{ {mcases} default: return -1; }} }}")
  hdr.append(f"int {name}_kind_maha(int kind);")
  abi.append(f"int {name}_num_kinds(void) {{ return {len(spec.kinds)}; }}")
  hdr.append(f"int {name}_num_kinds(void);")
  abi.append(f"void {name}_kinds(int *out) {{ " + " ".join(f"out[{i}] = {k.kind};" for i, k in enumerate(spec.kinds)) + " }")
  hdr.append(f"void {name}_kinds(int *out);")
  abi.append(f"int {name}_last_error(void) {{ return rn::err().code; }}")
  hdr.append(f"int {name}_last_error(void);")
  abi.append(f"const char *{name}_last_error_string(void) {{ return rn::err().msg; }}")
  hdr.append(f"const char *{name}_last_error_string(void);")
  abi.append(f"void {name}_clear_error(void) {{ rn::err() = rn::ErrorState(); }}")
  hdr.append(f"void {name}_clear_error(void);")
  abi.append("")

  for var in spec.global_vars:
    abi.append(f"""void {name}_set_{var.name}(double x) {{
  if (hipMemcpyToSymbol(HIP_SYMBOL({var.name}), &x, sizeof(double), 0, hipMemcpyHostToDevice) != hipSuccess)
    rn::fail(rn::ERR_HIP, (int)hipGetLastError(), "{name}_set_{var.name}", __LINE__);
}}""")
    hdr.append(f"void {name}_set_{var.name}(double x);")
  from rednose_amd.codegen import tuning as _tn
  if fam == "wide" and _tn.current().wide_timeline:
    abi.append(f"""int {name}_debug_timeline(unsigned long long *out) {{
  RN_HIP(hipDeviceSynchronize());
  RN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * 256 * 64 * 2, 0, hipMemcpyDeviceToHost));
  return rn::OK;
}}""")
    hdr.append(f"int {name}_debug_timeline(unsigned long long *out);")
    abi.append(f"""int {name}_debug_rts_timeline(unsigned long long *out) {{
  RN_HIP(hipDeviceSynchronize());
  RN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rn::g_rts_tl), sizeof(unsigned long long) * 256 * 16, 0, hipMemcpyDeviceToHost));
  return rn::OK;
}}
int {name}_debug_blocks(unsigned long long *out) {{
  RN_HIP(hipDeviceSynchronize());
  RN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tlb), sizeof(unsigned long long) * 4096 * 2, 0, hipMemcpyDeviceToHost));
  return rn::OK;
}}""")
    hdr.append(f"int {name}_debug_rts_timeline(unsigned long long *out);")
    hdr.append(f"int {name}_debug_blocks(unsigned long long *out);")
  # batched, device pointers
  # Every step-granular entry point exists twice: plain, and `_masked` with a per-filter `active` byte (0 = this filter has no
  # observation in this call: its x, P and z pass through untouched and flag bit 4 is set) -- what a batch of filters on
  # INDEPENDENT timelines needs (each filter of the reference is its own instance with its own filter_time, ekf_sym.cc:83-117);
  # together with the per-filter dt vector a call then advances exactly the filters that have something to do.
  for sfx, act_param, act_decl in (("", "", "  const uint8_t *active = nullptr;\n"), ("_masked", "const uint8_t *active, ", "")):
    abi.append(f"""int {name}_batch_predict{sfx}(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, {act_param}void *stream) {{
  RN_REQUIRE(n >= 0 && x && P && Q, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P), rn::ERR_ALIGN);
{act_decl}{fam_mod.launch_predict()}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
    hdr.append(f"int {name}_batch_predict{sfx}(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, {act_param}void *stream);")
    for k in spec.kinds:
      abi.append(f"""int {name}_batch_update_{k.kind}{sfx}(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, {act_param}void *stream) {{
  RN_REQUIRE(n >= 0 && x && P && z && R{ea_req(k)}, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && (!r_per_filter || rn::aligned16(R)), rn::ERR_ALIGN);
{act_decl}{fam_mod.launch_step(k.kind, False)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}
int {name}_batch_predict_update_{k.kind}{sfx}(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, {act_param}void *stream) {{
  RN_REQUIRE(n >= 0 && x && P && Q && z && R{ea_req(k)}, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && (!r_per_filter || rn::aligned16(R)), rn::ERR_ALIGN);
{act_decl}{fam_mod.launch_step(k.kind, True)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
      hdr.append(f"int {name}_batch_update_{k.kind}{sfx}(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, {act_param}void *stream);")
      hdr.append(f"int {name}_batch_predict_update_{k.kind}{sfx}(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, {act_param}void *stream);")

  # the fused step that also writes the call's checkpoint (k_stepc_{kind}): what a rewind ring keeps of a call -- the observations as they came and the
  # filtered pair (ekf_sym.cc:142-156, 191) -- leaves with the step's own stores instead of three copies behind it
  for k in spec.kinds:
    abi.append(f"""int {name}_batch_predict_update_{k.kind}_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream) {{
  RN_REQUIRE(n >= 0 && x && P && Q && z && R && ckpt_x && ckpt_P && ckpt_z{ea_req(k)}, rn::ERR_ARG);
  RN_REQUIRE(ckpt_x != x && ckpt_P != P && ckpt_z != z, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && (!r_per_filter || rn::aligned16(R)) && rn::aligned16(ckpt_x) && rn::aligned16(ckpt_P) && rn::aligned16(ckpt_z), rn::ERR_ALIGN);
  const uint8_t *active = nullptr;
{fam_mod.launch_step_ckpt(k.kind)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
    hdr.append(f"int {name}_batch_predict_update_{k.kind}_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);")

  abi.append(f"""int {name}_batch_ring_copy(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride, int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n, int to_ring, void *stream) {{
  RN_REQUIRE(n >= 0 && rec >= 0 && rec <= ring_stride && rec <= flat_stride && ring && flat && slot, rn::ERR_ARG);
  if (n == 0 || rec == 0) return rn::OK;
  hipLaunchKernelGGL(rn::k_ring_copy, dim3((unsigned)(n < 65536 ? n : 65536)), dim3(64), 0, (hipStream_t)stream, ring, ring_stride, flat, flat_stride, rec, slot, active, n, to_ring);
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
  hdr.append(f"int {name}_batch_ring_copy(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride, int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n, int to_ring, void *stream);")
  abi.append(f"""int {name}_batch_flags_set(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream) {{
  RN_REQUIRE(n >= 0 && flags && mask, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  hipLaunchKernelGGL(rn::k_flags_set, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream, flags, mask, value, n);
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
  hdr.append(f"int {name}_batch_flags_set(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream);")

  if hasattr(fam_mod, "launch_maha"):
    for k in spec.kinds:
      abi.append(f"""int {name}_batch_maha_{k.kind}(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream) {{
  RN_REQUIRE(n >= 0 && x && P && z && R && d2{ea_req(k)}, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && (!r_per_filter || rn::aligned16(R)), rn::ERR_ALIGN);
{fam_mod.launch_maha(k.kind)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
      hdr.append(f"int {name}_batch_maha_{k.kind}(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);")
  if spec.N > 0:
    abi.append(f"""int {name}_batch_augment(double *x, double *P, int64_t n, void *stream) {{
  RN_REQUIRE(n >= 0 && x && P, rn::ERR_ARG);
  if (n == 0) return rn::OK;
  hipLaunchKernelGGL(k_augment, dim3(rn::grid_for_tiles(n)), dim3(64), 0, (hipStream_t)stream, x, P, n);
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
    hdr.append(f"int {name}_batch_augment(double *x, double *P, int64_t n, void *stream);")
  abi.append(f"void {name}_msckf_dims(int *dims) {{ dims[0] = {spec.dim_main}; dims[1] = {spec.dim_main_err}; dims[2] = {spec.dim_augment}; dims[3] = {spec.dim_augment_err}; dims[4] = {spec.N}; }}")
  hdr.append(f"void {name}_msckf_dims(int *dims);")
  eacases = " ".join(f"case {k.kind}: return {ea_len(k)};" for k in spec.kinds)
  abi.append(f"int {name}_kind_eadim(int kind) {{ switch (kind) {{ {eacases} default: return -1; }} }}")
  hdr.append(f"int {name}_kind_eadim(int kind);")
  zmax = max(k.zdim for k in spec.kinds)
  abi.append(f"int {name}_zmax(void) {{ return {zmax}; }}")
  hdr.append(f"int {name}_zmax(void);")
  # steps per loop iteration of the kernel an untraced batch_run launches (instruction accounting of bench.py)
  unroll = (emit_small.run_block(spec) or emit_small.run_unroll(spec)) if fam == "small" else 1
  abi.append(f"int {name}_run_unroll(void) {{ return {unroll}; }}")
  hdr.append(f"int {name}_run_unroll(void);")
  # 0: this model's fused multi-step kernel did not fit the register file (fallback no_run) -- {name}_batch_run returns ERR_UNSUPPORTED
  # and callers walk a schedule with the step-granular entry points (BatchedEKF.run does)
  abi.append(f"int {name}_has_batch_run(void) {{ return {int(has_run)}; }}")
  hdr.append(f"int {name}_has_batch_run(void);")
  # 1: predict(dt = 0) is the identity on (x, P) for this model, symbolically (FilterSpec.identity_at_dt0) -- a batch_run step with dt = 0 is
  # then an update alone, which is how the orchestrators serve the n observations of ONE predict_and_update_batch call (the reference predicts
  # once and updates n times, ekf_sym.cc:172-180) in one launch; 0: they issue batch_update_k launches instead
  abi.append(f"int {name}_predict_identity_at_dt0(void) {{ return {int(spec.identity_at_dt0())}; }}")
  hdr.append(f"int {name}_predict_identity_at_dt0(void);")
  abi.append(f"""int {name}_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream) {{
  RN_REQUIRE(n >= 0 && T >= 0 && x && P && Q && kinds && dts && z && R, rn::ERR_ARG);
  if (n == 0 || T == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && rn::aligned16(trace_x) && rn::aligned16(trace_P), rn::ERR_ALIGN);
{fam_mod.launch_run() if has_run else '  (void)norm_quats; (void)flags; (void)stream; (void)ea; (void)augment; return rn::fail(rn::ERR_UNSUPPORTED, 0, "batch_run: not generated for this model (its fused run did not fit the register file)", __LINE__);'}
  {'RN_HIP(hipGetLastError());' if has_run else ''}
  return rn::OK;
}}""")
  hdr.append(f"int {name}_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);")
  # Packed-triangle trace (opt-in; models with k_run2 AND k_rts4): batch_run_tri writes the lower triangle of every filtered covariance
  # (row-major, E (E + 1) / 2 doubles) where batch_run writes E^2, batch_rts_tri smooths such a trace into packed smoothed covariances; the fused
  # run's covariance is symmetric by contract and batch_rts reads lower triangles only (include/rednose_amd_filter.h), so nothing is lost.
  abi.append(f"int {name}_has_tri_trace(void) {{ return {int(use_tri)}; }}")
  hdr.append(f"int {name}_has_tri_trace(void);")
  if use_tri:
    from rednose_amd.codegen import emit_run2 as _r2
    abi.append(f"""int {name}_batch_run_tri(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream) {{
  RN_REQUIRE(n >= 0 && T >= 0 && x && P && Q && kinds && dts && z && R, rn::ERR_ARG);
  if (n == 0 || T == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(x) && rn::aligned16(P) && rn::aligned16(z) && rn::aligned16(trace_x) && rn::aligned16(trace_P), rn::ERR_ALIGN);
{_r2.launch_run(tri=True)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}
int {name}_batch_tri_unpack(const double *tri, double *full, int64_t count, void *stream) {{
  RN_REQUIRE(count >= 0 && tri && full, rn::ERR_ARG);
  if (count == 0) return rn::OK;
  hipLaunchKernelGGL(rn::k_tri_unpack<{E}>, dim3(8192), dim3(256), 0, (hipStream_t)stream, tri, full, count);
  RN_HIP(hipGetLastError());
  return rn::OK;
}}
int {name}_batch_tri_pack(const double *full, double *tri, int64_t count, void *stream) {{
  RN_REQUIRE(count >= 0 && tri && full, rn::ERR_ARG);
  if (count == 0) return rn::OK;
  hipLaunchKernelGGL(rn::k_tri_pack<{E}>, dim3(8192), dim3(256), 0, (hipStream_t)stream, full, tri, count);
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
    hdr.append(f"int {name}_batch_run_tri(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);")
    hdr.append(f"int {name}_batch_tri_unpack(const double *tri, double *full, int64_t count, void *stream);")
    hdr.append(f"int {name}_batch_tri_pack(const double *full, double *tri, int64_t count, void *stream);")

  if has_rts:
    if use_rts4:
      launch = emit_rts4.launch(spec)
    elif group_rts:
      GLr = 16 if M <= 16 else (32 if M <= 32 else 64)
      launch = f"""  const int64_t tiles = (n + {64 // GLr - 1}) / {64 // GLr};
  hipLaunchKernelGGL(rn::k_rts_group<RtsModel>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     xf, Pf, ts, T, Q, n, norm_quats, xs, Ps, x_last, P_last);"""
    else:
      launch = """  const int64_t tiles = (n + 1) / 2;
  hipLaunchKernelGGL(rn::k_rts<RtsModel>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     xf, Pf, ts, T, Q, n, norm_quats, xs, Ps, x_last, P_last);"""
    abi.append(f"""int {name}_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream) {{
  RN_REQUIRE(n >= 0 && T >= 0 && xf && Pf && ts && Q && xs && Ps, rn::ERR_ARG);
  if (n == 0 || T == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(xf) && rn::aligned16(Pf) && rn::aligned16(xs) && rn::aligned16(Ps) && rn::aligned16(x_last) && rn::aligned16(P_last), rn::ERR_ALIGN);
{launch}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
    hdr.append(f"int {name}_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);")
    if use_tri:
      abi.append(f"""int {name}_batch_rts_tri(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream) {{
  RN_REQUIRE(n >= 0 && T >= 0 && xf && Pf && ts && Q && xs && Ps, rn::ERR_ARG);
  if (n == 0 || T == 0) return rn::OK;
  RN_REQUIRE(rn::aligned16(xf) && rn::aligned16(Pf) && rn::aligned16(xs) && rn::aligned16(Ps) && rn::aligned16(x_last) && rn::aligned16(P_last), rn::ERR_ALIGN);
{emit_rts4.launch(spec, tri=True)}
  RN_HIP(hipGetLastError());
  return rn::OK;
}}""")
      hdr.append(f"int {name}_batch_rts_tri(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);")

  # the reference's scalar host-pointer ABI, executed as a batch of one on the GPU
  xo, Po, Qo = 0, _align2(D), _align2(D) + _align2(EE)
  abi.append(f"""
// ---- reference scalar ABI (/root/reference/rednose/helpers/ekf_sym.py:149-165): HOST pointers, in place ----
void {name}_predict(double *in_x, double *in_P, double *in_Q, double dt) {{
  rn::Scratch& s = rn::scratch();
  std::lock_guard<std::mutex> hold(s.mu);
  if (s.ensure({Qo + _align2(EE)}) != rn::OK) return;
  s.put({xo}, in_x, {D}); s.put({Po}, in_P, {EE}); s.put({Qo}, in_Q, {EE});
  if ({name}_batch_predict(s.dev + {xo}, s.dev + {Po}, s.dev + {Qo}, nullptr, dt, 1, 0, nullptr) != rn::OK) return;
  if (s.wait("{name}_predict", __LINE__) != rn::OK) return;
  s.get(in_x, {xo}, {D}); s.get(in_P, {Po}, {EE});
}}""")
  hdr.append(f"void {name}_predict(double *in_x, double *in_P, double *in_Q, double dt);")
  for k in spec.kinds:
    Z = k.zdim
    zo = Po + _align2(EE)
    Ro = zo + _align2(Z)
    eo = Ro + _align2(Z * Z)
    EA = ea_len(k)
    tot = eo + _align2(EA)
    ea_put = f" s.put({eo}, in_ea, {EA});" if EA else ""
    ea_arg = f"s.dev + {eo}" if EA else "nullptr"
    abi.append(f"""void {name}_update_{k.kind}(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea) {{
  (void)in_ea;
  rn::Scratch& s = rn::scratch();
  std::lock_guard<std::mutex> hold(s.mu);
  if (s.ensure({tot}) != rn::OK) return;
  s.put({xo}, in_x, {D}); s.put({Po}, in_P, {EE}); s.put({zo}, in_z, {Z}); s.put({Ro}, in_R, {Z * Z});{ea_put}
  if ({name}_batch_update_{k.kind}(s.dev + {xo}, s.dev + {Po}, s.dev + {zo}, s.dev + {Ro}, 0, {ea_arg}, 1, 0, nullptr, nullptr) != rn::OK) return;
  if (s.wait("{name}_update_{k.kind}", __LINE__) != rn::OK) return;
  s.get(in_x, {xo}, {D}); s.get(in_P, {Po}, {EE}); s.get(in_z, {zo}, {Z});
}}""")
    hdr.append(f"void {name}_update_{k.kind}(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);")
  abi += wrappers
  abi.append('}  // extern "C"')
  abi.append(plugin_text(spec))

  hdr += ["#ifdef __cplusplus", "}", "#endif", ""]
  text = "\n".join(src) + "\n" + "\n".join(abi) + "\n"
  if "rn::nullspace_residual<" in text:      # feature-track kinds: the residual in the reference's null-space basis (codegen/lower.py)
    from rednose_amd.codegen.lower import NULLSPACE_RESIDUAL
    text = text.replace('#include "ekf_hip_rts.h"\n', '#include "ekf_hip_rts.h"\n' + NULLSPACE_RESIDUAL, 1)
  if "rn::sincos_fast(" in text:      # the model has trigonometric terms: codegen/lower.py printed them through this helper
    from rednose_amd.codegen.lower import SINCOS_FAST
    text = text.replace('#include "ekf_hip_rts.h"\n', '#include "ekf_hip_rts.h"\n' + SINCOS_FAST, 1)
  return "\n".join(hdr), text
