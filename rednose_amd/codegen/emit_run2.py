"""Kernel family W, fused multi-step run with a SCALAR WAVEFRONT beside the matrix wavefront: `k_run2`.

`k_run` (emit_wide3) alternates, inside one wavefront, phases that use all 64 lanes (the covariance algebra on register rows) with
phases in which one lane per filter walks a chain of dependent fp64 instructions (f, F, h, He, the error injection): 2.6-3.1 us of
a 10-12 us traced step of live, with the LDS, the memory pipes and 56 of 64 lanes idle, and with the matrix phases' 132 row
registers parked in AGPRs around them (profiles/tuning_notes.md).  LDS (an E x E image per filter) allows four such wavefronts per
CU, one per SIMD: nothing else can issue while a chain waits.

Here a workgroup is TWO wavefronts on the same eight filters and the same LDS block:
  * wavefront 0 (matrix) keeps the rows of P in registers and runs emit_wide3's predict_rows / update_*_rows, nothing else;
  * wavefront 1 (scalar) keeps x in registers (one lane per filter) and runs the scalar phase functions, the observation /
    residual / flag / state-trace traffic, and the error injection.
They meet at workgroup barriers (s_waitcnt lgkmcnt(0) + s_barrier: LDS only -- a full fence would drain the matrix wavefront's
32 KB of trace stores at every barrier), per step t:

  B1  scalar: F(t), dt in the slot            | matrix: step t - 1 complete (image free)
      scalar: h, He, y of step t   (*)        | matrix: P <- F P F^T + dt Q
  B2  scalar: He, y, `bad` in the slot / LDS  | matrix: predict done
      scalar: waits                           | matrix: G, S, gate, K, dx -> slot
  B3  dx, flags in the slot
      scalar: x <- x (+) dx, flags / y / x trace out, next z in, f / F of step t + 1
                                              | matrix: P -= K G, Joseph coefficients
  B4  (only if step t + 1 has no predict) He is dead
      scalar: h, He, y of step t + 1  (*)     | matrix: P += D K^T, rows -> image -> trace
  (*) once per step: before B1 when the step has no predict (the matrix wavefront is still in the tail of the step before),
      after it otherwise (under the predict).

The slot cannot overlay F and He any more (both are live between B1 and B2); x leaves the slot for the scalar wavefront's
registers instead (the state trace and the final store stage it through the F region, which is dead then), so eight filters'
slots, images and G / K^T buffers stay under 40 KB: four workgroups = eight wavefronts per CU, two per SIMD, at <= 256 registers.
Arithmetic, order of operations and results are k_run's (same device functions against another slot layout): the parity tests
of the fused run apply unchanged (tests/test_gpu_run.py, test_gpu_random.py, test_gpu_fullsize.py, test_gpu_asymmetric.py) and
tests/test_emit_host.py runs the kernel on the host with a thread per lane of both wavefronts.
Reference: EKF_sym.predict_and_update_batch's loop body, ekf_sym.py:473-538, over a schedule; ekf_c.c:8-121.
"""
from rednose_amd.codegen import emit_wide3 as w3

LDS_BUDGET = 40960      # bytes per workgroup for four workgroups per CU (160 KB)


class Run2Layout:
  """Per-filter slot (doubles): [F | dx | staged x] [He] [z / y] [dt] [flags].  No x: it lives in the scalar wavefront."""

  def __init__(self, spec, f_vars, he_vars_by_kind):
    D, E = spec.dim_x, spec.dim_err
    self.zmax = max(k.zdim for k in spec.kinds)
    self.nf = len(f_vars)
    self.nh = max([len(v) for v in he_vars_by_kind.values()] + [0])
    self.OFF_X = None
    self.OFF_F = self.OFF_DX = self.OFF_XS = 0
    self.OFF_HE = max(self.nf, E, D)
    self.OFF_Y = self.OFF_HE + self.nh
    self.OFF_DT = self.OFF_Y + self.zmax
    self.OFF_FL = self.OFF_DT + 1
    self.OFF_RF = self.OFF_RP = self.OFF_YP = -(1 << 20)      # feature-track kinds stay with k_run (applicable())
    n = self.OFF_FL + 1
    self.SLOT = n + 1 - (n & 1)


def lds_bytes(spec):
  E = spec.dim_err
  _, _, FPW = w3.layout(spec)
  lay, _, _ = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  zmax = max(k.zdim for k in spec.kinds)
  return 8 * (FPW * E * E + 2 + FPW * zmax * E + FPW * lay.SLOT) + 16


def applicable(spec):
  """Models of the 8-lanes-per-filter layout (<= 22 error states) without feature-track kinds, extra arguments or a window shift,
  whose workgroup fits a quarter of a CU's LDS."""
  GL, _, FPW = w3.layout(spec)
  zmax = max(k.zdim for k in spec.kinds)
  plain = all(k.He_sym is None and k.ea_sym is None for k in spec.kinds)
  from rednose_amd.codegen import tuning
  return GL == 8 and plain and spec.N == 0 and FPW * zmax <= 64 and lds_bytes(spec) <= LDS_BUDGET and not tuning.current().wide_timeline


def kernels(spec):
  """Scalar phase functions against Run2Layout (suffix _r2, x as a register array), the matrix functions of emit_wide3 against it, k_run2."""
  from rednose_amd.codegen import emit_wide2 as w2
  scal_text, lay = w2.device_functions(spec, lay_cls=Run2Layout, sfx="_r2", xreg=True)
  out = [f"constexpr int SLOT_R2 = {lay.SLOT};   // two-wavefront fused run: doubles per scalar slot", "", scal_text, "",
         w3.predict_fn(spec, lay_cls=Run2Layout, sfx="_r2"), w3.predict_fn(spec, qdiag=True, lay_cls=Run2Layout, sfx="_r2")]
  for k in spec.kinds:
    out.append(w3.update_fn(spec, k, lay_cls=Run2Layout, sfx="_r2", two_wave=True))
  out.append(run_kernel(spec))
  return "\n".join(out)


def run_kernel(spec):
  from rednose_amd.codegen import tuning
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  GL, R, FPW = w3.layout(spec)
  lay, _, _ = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  zmax = max(k.zdim for k in spec.kinds)
  rows = ", ".join(f"row{s}" for s in range(R))
  idx = ", ".join(f"rr{s}, rc{s}, ok{s}" for s in range(R))
  nlc = chr(10)
  scal_cases = nlc.join(f"          case {k.kind}: scal_obs_{k.kind}_r2(xr, sl, sl + {lay.OFF_Y}); break;" for k in spec.kinds)
  # every case works on its own opaque copies of the slot / buffer addresses: identical loads in all cases (y, R, the rows of G) would
  # otherwise be hoisted in front of the switch and live across it -- each kind alone fits 256 registers, all of them together spilled
  mat_cases = nlc.join(f"            case {k.kind}: {{ double* slk = sl; double* sGk = s_G + gg * {zmax * E}; const double* gRk = gR + t * {zmax * zmax}; "
                       'asm volatile("" : "+v"(slk), "+v"(sGk), "+s"(gRk)); '
                       f"update_{k.kind}_rows_r2({rows}, gRk, sP, sGk, slk, slk, {idx}, he_release); done = true; break; }}"
                       for k in spec.kinds)
  img = nlc.join(f"        if (ok{s}) {{\n#pragma unroll\n          for (int j = 0; j < {E}; j++) sP[rr{s} * {E} + j] = row{s}[j];\n        }}" for s in range(R))
  id0_guard = "true" if not spec.identity_at_dt0() else "dt != 0.0"
  id0_next = "true" if not spec.identity_at_dt0() else "dtn != 0.0"
  nt_trace = "true" if tuning.current().nt_trace else "false"
  qd_decl = nlc.join(f"  const double qd{s} = gQ[((c + {GL * s}) < {E} ? (c + {GL * s}) : 0) * {E + 1}];" for s in range(R))
  qd_args = ", ".join(f"qd{s}" for s in range(R))
  decl_rows = nlc.join(f"      double row{s}[{E}];" for s in range(R))
  decl_idx = nlc.join(f"      const int rr{s} = c + {GL * s}; const bool ok{s} = live && rr{s} < {E}; const int rc{s} = rr{s} < {E} ? rr{s} : 0;" for s in range(R))
  load_rows = nlc.join(f"#pragma unroll\n      for (int j = 0; j < {E}; j++) row{s}[j] = 0.5 * (sP[rc{s} * {E} + j] + sP[j * {E} + rc{s}]);" for s in range(R))
  stage_x = f"""if (c == 0 && live) {{
#pragma unroll
          for (int i = 0; i < {D}; i++) sl[{lay.OFF_XS} + i] = xr[i];
        }}
        rn::wave_lds_sync();"""
  return f"""
// ---- fused multi-step run, matrix wavefront + scalar wavefront per tile of {FPW} filters (emit_run2.py): same interface as k_run ----
__global__ __launch_bounds__(128, 2) void k_run2(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* __restrict__ gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    double* __restrict__ tx, double* __restrict__ tP, const double* __restrict__ gea, const int32_t* __restrict__ augs) {{
  (void)gea; (void)augs;
  __shared__ __attribute__((aligned(16))) double s_P[FPWR * {EE} + 2];      // one image of P per filter
  __shared__ __attribute__((aligned(16))) double s_G[FPWR * {zmax * E}];     // G, then K^T
  __shared__ __attribute__((aligned(16))) double s_sl[FPWR * SLOT_R2];
  __shared__ int s_bad;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane / GLR;
  const int c = lane % GLR;
  const int64_t tiles = (n + FPWR - 1) / FPWR;
  if (wave == 0) {{
    // ================================ matrix wavefront ================================
    int qoff = 0;
    for (int i = lane; i < {EE}; i += 64) qoff |= (i / {E} != i % {E}) && (gQ[i] != 0.0);
    const bool qdiag = !__any(qoff);
{qd_decl}
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
      const int64_t base = tile * FPWR;
      const int cnt = (n - base) < FPWR ? (int)(n - base) : FPWR;
      const int gg = g < cnt ? g : 0;
      const bool live = g < cnt;
      double* sP = s_P + gg * {EE};
      double* sl = s_sl + gg * SLOT_R2;
{decl_idx}
      int lb = lane;
      asm volatile("" : "+v"(lb));
      rn::copy_g2l<FPWR * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lb);
      rn::wave_lds_sync();
{decl_rows}
{load_rows}
      rn::wave_lds_sync();
      rn::wg_barrier();                                   // B1 of step 0
      for (int64_t t = 0; t < T; t++) {{
        const int kind = kinds[t];
        const double dt = dts[t];
        const bool do_pred = {id0_guard};
        bool he_release = false;                          // B4: the next step has no predict, its h / He go under this step's tail
        if (t + 1 < T) {{ const double dtn = dts[t + 1]; he_release = !({id0_next}); }}
        if (do_pred) {{
          if (qdiag) {{
            predict_rows_qd_r2({rows}, sP, {qd_args}, sl, {idx});
          }} else {{
            int qz = 0;
            asm volatile("" : "+v"(qz));
            predict_rows_r2({rows}, sP, gQ + qz, sl, {idx});
          }}
        }}
        rn::wg_barrier();                                 // B2
        const int bad = __builtin_amdgcn_readfirstlane(s_bad);
        bool done = false;
        if (!bad) {{
          switch (kind) {{
{mat_cases}
            default: break;
          }}
        }}
        if (!done) {{
          rn::wg_barrier();                               // B3
          if (he_release) rn::wg_barrier();               // B4
        }}
        if (tP != nullptr) {{
          int lz = lane;
          asm volatile("" : "+v"(lz));
{img}
          rn::wave_lds_sync();
          rn::copy_l2g<FPWR * {EE}, {nt_trace}>(tP + (t * n + base) * {EE}, cnt * {EE}, s_P, lz);
          rn::wave_lds_sync();
        }}
        rn::wg_barrier();                                 // B1 of step t + 1
      }}
{img}
      rn::wave_lds_sync();
      int le = lane;
      asm volatile("" : "+v"(le));
      rn::copy_l2g<FPWR * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, le);
      rn::wave_lds_sync();
    }}
  }} else {{
    // ================================ scalar wavefront ================================
    const int zf = lane / {zmax}, zc = lane % {zmax};      // observation entry this lane carries between HBM and the slots
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
      const int64_t base = tile * FPWR;
      const int cnt = (n - base) < FPWR ? (int)(n - base) : FPWR;
      const int gg = g < cnt ? g : 0;
      const bool live = g < cnt;
      const bool zlive = zf < cnt;
      double* sl = s_sl + gg * SLOT_R2;
      double* slz = s_sl + (zlive ? zf : 0) * SLOT_R2 + {lay.OFF_Y} + zc;
      double xr[{D}];
#pragma unroll
      for (int i = 0; i < {D}; i++) xr[i] = gx[(base + gg) * {D} + i];
      if (zlive) *slz = gz[base * {zmax} + lane];
      rn::wave_lds_sync();
      int bad = 0;
      bool obs_done = false;
      if (T > 0) {{
        const double dt0 = dts[0];
        const double dtn = dt0;
        const bool p0 = {id0_next};
        if (c == 0 && live) {{
          if (p0) scal_predict_r2(xr, dt0, sl, norm_quats);
          else scal_keep_r2(xr, sl, norm_quats);
        }}
        rn::wave_lds_sync();
      }}
      rn::wg_barrier();                                   // B1 of step 0
      for (int64_t t = 0; t < T; t++) {{
        double zn = 0.0;                                  // next step's observation, in flight during this step
        if (t + 1 < T && zlive) zn = gz[((t + 1) * n + base) * {zmax} + lane];
        const int kind = kinds[t];
        if (!obs_done) {{
          bad = 0;
          if (c == 0 && live) {{
            switch (kind) {{
{scal_cases}
              default: bad = 8; break;      // unknown kind
            }}
          }}
          bad = __builtin_amdgcn_readfirstlane(__any(bad) ? 8 : 0);
          if (lane == 0) s_bad = bad;
        }}
        rn::wave_lds_sync();
        rn::wg_barrier();                                 // B2
        rn::wg_barrier();                                 // B3: dx and the gate flag are in the slot
        if (c == 0 && live) {{
          int fl = bad;
          if (!bad) fl = scal_inject_r2(sl, xr, norm_quats) | (int)sl[{lay.OFF_FL}];
          if (flags != nullptr) flags[t * n + base + g] = (uint8_t)fl;
        }}
        if (zlive) gz[(t * n + base) * {zmax} + lane] = *slz;          // y (the observation itself after an unknown kind)
        rn::wave_lds_sync();
        if (tx != nullptr) {{
          {stage_x}
          int lz = lane;
          asm volatile("" : "+v"(lz));
          for (int i = lz; i < cnt * {D}; i += 64) tx[(t * n + base) * {D} + i] = s_sl[(i / {D}) * SLOT_R2 + {lay.OFF_XS} + i % {D}];
          rn::wave_lds_sync();
        }}
        if (zlive) *slz = zn;
        rn::wave_lds_sync();
        obs_done = false;
        if (t + 1 < T) {{
          const double dtn = dts[t + 1];
          const bool pn = {id0_next};
          if (c == 0 && live) {{
            if (pn) scal_predict_r2(xr, dtn, sl, norm_quats);
            else scal_keep_r2(xr, sl, norm_quats);
          }}
          rn::wave_lds_sync();
          if (!pn) {{
            rn::wg_barrier();                             // B4: He of step t is dead
            const int kn = kinds[t + 1];
            bad = 0;
            if (c == 0 && live) {{
              switch (kn) {{
{scal_cases}
                default: bad = 8; break;
              }}
            }}
            bad = __builtin_amdgcn_readfirstlane(__any(bad) ? 8 : 0);
            if (lane == 0) s_bad = bad;
            rn::wave_lds_sync();
            obs_done = true;
          }}
        }}
        rn::wg_barrier();                                 // B1 of step t + 1
      }}
      {stage_x}
      int le = lane;
      asm volatile("" : "+v"(le));
      for (int i = le; i < cnt * {D}; i += 64) gx[base * {D} + i] = s_sl[(i / {D}) * SLOT_R2 + {lay.OFF_XS} + i % {D}];
      rn::wave_lds_sync();
    }}
  }}
}}
"""


def launch_run():
  return """  const int64_t tiles = (n + FPWR - 1) / FPWR;
  hipLaunchKernelGGL(k_run2, dim3(rn::grid_for_tiles(tiles)), dim3(128), 0, (hipStream_t)stream,
                     x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, trace_x, trace_P, ea, augment);"""
