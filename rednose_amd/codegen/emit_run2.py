"""Kernel family W, fused multi-step run with a SCALAR WAVEFRONT beside the matrix wavefront: `k_run2`.

`k_run` (emit_wide3) alternates, inside one wavefront, phases that use all 64 lanes (the covariance algebra on register rows) with
phases in which one lane per filter walks a chain of dependent fp64 instructions (f, F, h, He, the error injection): 2.6-3.1 us of
a 10-12 us traced step of live, with the LDS, the memory pipes and 56 of 64 lanes idle, and with the matrix phases' 132 row
registers parked in AGPRs around them (profiles/tuning_notes.md).  LDS (an E x E image per filter) allows four such wavefronts per
CU, one per SIMD: nothing else can issue while a chain waits.

Here a workgroup is TWO wavefronts on the same eight filters and the same LDS block:
  * wavefront 0 (matrix) keeps the rows of P in registers and runs the covariance algebra (predict_rows_r2 / update_rows_r2 below:
    emit_wide3's sums in emit_wide3's order, ONE body each for all kinds of process noise / observation), nothing else;
  * wavefront 1 (scalar) runs emit_wide2's scalar phase functions (one lane per filter) against the filters' slots, the
    observation / residual / flag / state-trace traffic, and the error injection.
They meet at workgroup barriers (s_waitcnt lgkmcnt(0) + s_barrier: LDS only -- a full fence would drain the matrix wavefront's
32 KB of trace stores at every barrier), per step t:

  B1  scalar: F(t), dt in the slot            | matrix: step t - 1 complete (image free)
      scalar: h, He, y of step t   (*)        | matrix: P <- F P F^T + dt Q
  B2  scalar: He, y, `bad` in the slot / LDS  | matrix: predict done
      scalar: waits                           | matrix: G, S, gate, K, dx -> slot
  B3  dx, flags in the slot
      scalar: x <- x (+) dx, flags / y / x trace out, next z in, f / F of step t + 1
                                              | matrix: P -= K G, Joseph coefficients
  F4  (a flag in LDS, only if step t + 1 has no predict: the matrix wavefront sets it and goes on, the scalar one waits for it) He is dead
      scalar: h, He, y of step t + 1  (*)     | matrix: P += D K^T, rows -> image -> trace
  (*) once per step: before B1 when the step has no predict (the matrix wavefront is still in the tail of the step before),
      after it otherwise (under the predict).

The slot cannot overlay F and He any more (both are live between B1 and B2); the room comes from the G / K^T buffer, which is the
first rows of the filter's covariance image here (the update never touches the image): eight filters' slots and images stay under
40 KB -- four workgroups = eight wavefronts per CU, two per SIMD, at <= 256 registers (what that took from the generator: DESIGN.md
section 10).  Served: 13 .. 22 error states without feature-track kinds (applicable()); `k_run` stays the fused run of every other
lane-group model and the fallback (`no_run2`).  Measured: profiles/tuning_notes.md, profiles/r5_run2_*.txt.
Arithmetic, order of operations and results are k_run's (same scalar functions against another slot layout): the parity tests
of the fused run apply unchanged (tests/test_gpu_run.py, test_gpu_random.py, test_gpu_fullsize.py, test_gpu_asymmetric.py) and
tests/test_emit_host.py runs the kernel on the host with a thread per lane of both wavefronts.
Reference: EKF_sym.predict_and_update_batch's loop body, ekf_sym.py:473-538, over a schedule; ekf_c.c:8-121.
"""
from rednose_amd.codegen import emit_wide3 as w3
from rednose_amd.codegen.emit_common import term, sum_terms

LDS_BUDGET = 40960      # bytes per workgroup for four workgroups per CU (160 KB)
MIN_E = 13              # smallest number of error states served (measured, see applicable())


FPG = 8           # filters per workgroup (tile): the scalar wavefront serves them with one lane each, lanes 0, 8, .. 56


def layout2(spec):
  """-> (GL lanes per filter, R rows of P per lane, FPW filters per matrix wavefront, ND matrix wavefronts per workgroup): emit_wide3's
  8 lanes x R rows, one matrix wavefront for the tile's 8 filters.  (Two matrix wavefronts of 4 filters each, 16 lanes x 2 rows, three
  wavefronts per SIMD at 167 registers, were built and measured: 32.5 ms per config-4 chunk against 20.5 -- the SIMD's fp64 pipe is shared,
  two matrix wavefronts on it take twice as long each.  profiles/tuning_notes.md; the kernel text keeps the wavefront index general.)"""
  E = spec.dim_err
  return 8, -(-E // 8), 8, 1


class Run2Layout:
  """Per-filter slot (doubles): [x] [F | dx] [He] [z / y] [dt] [flags].  F and He cannot share a region as in emit_wide3.RunLayout
  (both are live between B1 and B2); the room comes from the G / K^T buffer, which lives in the filter's covariance image here."""

  def __init__(self, spec, f_vars, he_vars_by_kind):
    D, E = spec.dim_x, spec.dim_err
    self.zmax = max(k.zdim for k in spec.kinds)
    self.nf = len(f_vars)
    self.nh = max([len(v) for v in he_vars_by_kind.values()] + [0])
    self.OFF_X = 0
    self.OFF_F = self.OFF_DX = D
    self.OFF_HE = D + max(self.nf, E)
    self.OFF_Y = self.OFF_HE + self.nh
    self.OFF_DT = self.OFF_Y + self.zmax
    self.OFF_FL = self.OFF_DT + 1
    self.OFF_RF = self.OFF_RP = self.OFF_YP = -(1 << 20)      # feature-track kinds stay with k_run (applicable())
    n = self.OFF_FL + 1
    self.SLOT = n + 1 - (n & 1)


def lds_bytes(spec):
  E = spec.dim_err
  FPW = FPG
  lay, _, _ = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  zmax = max(k.zdim for k in spec.kinds)
  return 8 * (FPW * E * E + 2 + FPW * lay.SLOT + E + (E & 1)) + 16


def applicable(spec):
  """Models of the 8-lanes-per-filter layout with 13 .. 22 error states, without feature-track kinds, extra arguments or a window shift,
  whose workgroup fits a quarter of a CU's LDS."""
  GL, _, FPW = w3.layout(spec)
  FPW = FPG
  zmax = max(k.zdim for k in spec.kinds)
  plain = all(k.He_sym is None and k.ea_sym is None for k in spec.kinds)
  from rednose_amd.codegen import tuning
  # below 13 error states k_run already has several wavefronts per SIMD and the second wavefront only adds barriers: kinematic9 (E = 9) 7.02 ms
  # with k_run against 8.04 with k_run2, rand13 21.3 against 18.2, rand17 34.9 against 31.0 (tools/ab_run, 32 768 filters x 1 000 steps;
  # profiles/r5_run2_small_models_ab.txt)
  return (GL == 8 and spec.dim_err >= MIN_E and plain and spec.N == 0 and FPW * zmax <= 64 and zmax <= spec.dim_err
          and lds_bytes(spec) <= LDS_BUDGET)


JB = 4            # columns per block of the rank-Z passes (2: 19.0 ms per config-4 chunk, 4: 18.7; emit_wide3 also runs 4)
CH = 8            # entries of a row of A = P F^T formed per block (and row slot) of predict (4: 19.0 ms, 8: 18.8)


def _ind(lines, n=2):
  pad = " " * n
  return [pad + x for x in lines]


def _tl(ph, base="tlb"):
  """Debug stamp (tuning knob wide_timeline; tools/timeline.py run2): slot `base` + ph of the workgroup's timeline, lane 0 of the calling wavefront."""
  from rednose_amd.codegen import tuning
  if not tuning.current().wide_timeline:
    return []
  return [f"if ((threadIdx.x == 0 || threadIdx.x == blockDim.x - 64) && blockIdx.x < 256) {{ const int ti_ = {base} + {ph}; g_tl[(blockIdx.x * 64 + ti_) * 2] = __builtin_readcyclecounter(); "
          "g_tl[(blockIdx.x * 64 + ti_) * 2 + 1] = wall_clock64(); }"]


def _tl_on():
  from rednose_amd.codegen import tuning
  return bool(tuning.current().wide_timeline)


def predict_fn(spec):
  """emit_wide3.predict_fn's algebra (one transposition through the LDS image, P = P^T), ONE body for both kinds of process
  noise -- the lane's diagonal entries of Q are register operands, the rows of a non-diagonal Q are added from HBM / L2 afterwards
  (`qdiag` false: rare) -- and in blocks of CH entries closed by a fence, so that the broadcast reads of F's entries stay
  inside their block: two inlined alternatives of this function, or one whose coefficient reads hipcc may hoist, do not fit 256
  registers beside the 132 of the rows."""
  E = spec.dim_err
  GL, R, _, _ = layout2(spec)
  lay, Fs, _ = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  b = [f"const double dt = sl[{lay.OFF_DT}];"]
  def shared(rows_of_f, blk):
    """Declarations of the slot-resident entries of F that the rows `rows_of_f` touch (one broadcast read each, used by every row slot),
    and the map entry text -> register name."""
    names, decl = {}, []
    for i in rows_of_f:
      for _, cf in Fs.row_nz(i):
        if cf is not None and cf[0] != 'one' and str(cf[1]).startswith("sl[") and cf[1] not in names:
          names[cf[1]] = f"f{blk}_{len(names)}"
          decl.append(f"  const double {names[cf[1]]} = {cf[1]};")
    return names, decl

  def tm(cf, operand, names):
    if cf is not None and cf[0] != 'one' and cf[1] in names:
      cf = (cf[0], names[cf[1]])
    return term(cf, operand)
  # first half: rows of A = P F^T, CH entries of all R row slots per block (one read of each entry of F per block, R * CH independent sums)
  for c0 in range(0, E, CH):
    cols = list(range(c0, min(c0 + CH, E)))
    names, decl = shared(cols, f"a{c0}")
    b.append("{")
    b += decl
    for s_ in range(R):
      b.append(f"  double a{s_}[{len(cols)}];")
    for i in cols:
      for s_ in range(R):
        b.append(f"  a{s_}[{i - c0}] = {sum_terms(tm(cf, f'row{s_}[{k}]', names) for k, cf in Fs.row_nz(i))};")
    for s_ in range(R):
      b += [f"  if (ok{s_}) {{", "#pragma unroll", f"    for (int i = 0; i < {len(cols)}; i++) sP[rr{s_} * {E} + {c0} + i] = a{s_}[i];", "  }"]
    b.append("}")
    b.append("rn::wave_lds_sync();")
  for s_ in range(R):
    b.append(f"int rd{s_} = rc{s_};")
    b.append(f'asm volatile("" : "+v"(rd{s_}));      // the diagonal selects below are computed here: as loop invariants they are {min(GL, E - GL * s_)} SGPR pairs per slot, across the whole step loop')
    b.append(f"const double dq{s_} = dt * sQd[rd{s_}];      // diag(Q) is the same for every filter: {E} doubles of LDS instead of {R} registers per lane across the step loop")
  # second half: column of A = row of B, fetched per block of outputs (only the entries that block's rows of F touch: the whole column at
  # once is 44 registers per slot), again all row slots per block
  for c0 in range(0, E, CH):
    outs = list(range(c0, min(c0 + CH, E)))
    need = sorted({m for j in outs for m, _ in Fs.row_nz(j)})
    names, decl = shared(outs, f"b{c0}")
    b.append("{")
    b += decl
    for s_ in range(R):
      for m in need:
        b.append(f"  const double a{s_}_{m} = sP[{m} * {E} + rc{s_}];")
    for j in outs:
      for s_ in range(R):
        diag = f" + (rd{s_} == {j} ? dq{s_} : 0.0)" if GL * s_ <= j < GL * (s_ + 1) else ""
        b.append(f"  row{s_}[{j}] = {sum_terms(tm(cf, f'a{s_}_{m}', names) for m, cf in Fs.row_nz(j))}{diag};")
    b.append("}")
    b.append("rn::wave_lds_sync();")
  b.append("rn::wave_lds_sync();      // the image is free again")
  b.append("if (!qdiag) {")
  for s_ in range(R):
    for c0 in range(0, E, CH):      # (in blocks: 66 loads in flight at once are 132 registers)
      for j in range(c0, min(c0 + CH, E)):
        b.append(f"  {{ const double q_ = gQ[rc{s_} * {E} + {j}]; row{s_}[{j}] += dt * ((rc{s_} == {j}) ? 0.0 : q_); }}")
      b.append("  rn::wave_lds_sync();")
  b.append("}")
  rows = ", ".join(f"double (&row{s_})[{E}]" for s_ in range(R))
  idx = ", ".join(f"const int rr{s_}, const int rc{s_}, const bool ok{s_}" for s_ in range(R))
  head = f"__device__ __forceinline__ void predict_rows_r2({rows}, double* sP, const double* sQd, const double* __restrict__ gQ, const bool qdiag, const double* sl, {idx}) {{"
  return "\n".join([head] + _ind(b) + ["}"])


def update_fn(spec):
  """emit_wide3.update_fn's Joseph-form update for ALL kinds in one body: the kind-specific parts -- the three sparse products with
  He = H H_mod -- sit in three small switches, everything else (S, its factor, the gate, the gain, both rank-Z passes) exists once,
  at the model's largest observation dimension ZM: a kind with fewer rows is padded with zero rows of He and an identity block of
  R, which adds exact zeros to the sums of the unpadded form.  (Eight inlined update bodies in one switch, as in k_run, cost hipcc
  ~200 spilled registers at a 256-register budget although each body alone fits: the row set is live across the switch.)
  dx and the flags leave for the slot as soon as the gain exists, followed by the workgroup barrier the scalar wavefront waits at;
  a flag (`he_release`) after the Joseph coefficients tells it that He and y are dead."""
  E = spec.dim_err
  _, R, _, _ = layout2(spec)
  lay, _, Hss = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  ZM = lay.zmax
  b = ["(void)sP;"]
  for s_ in range(R):
    b.append(f"double kk{s_}[{ZM}] = {{{', '.join('0.0' for _ in range(ZM))}}};")
  b += ["int zk = 0;", "double thr = 0.0;", "bool gate_on = false;"]
  # G^T = P He^T, row-local (P = P^T): the lane that owns row j has column j of G
  b.append("switch (kind) {")
  for k in spec.kinds:
    Hs, Z = Hss[k.kind], k.zdim
    ln = [f"kk{s_}[{zi}] = {sum_terms(term(cf, f'row{s_}[{c}]') for c, cf in Hs.row_nz(zi))};"
          for zi in range(Z) for s_ in range(R)]
    ln.append(f"zk = {Z};" + (f" thr = {k.maha_thresh!r}; gate_on = true;" if k.maha_test else ""))
    b.append(f"  case {k.kind}: {{ " + " ".join(ln) + " break; }")
  b += ["  default: break;", "}"]
  for s_ in range(R):
    b.append(f"if (ok{s_}) {{ " + " ".join(f"sG[{zi} * {E} + rr{s_}] = kk{s_}[{zi}];" for zi in range(ZM)) + " }")
  b.append("rn::wave_lds_sync();")
  b += _tl(3)
  ident = ", ".join("1.0" if i // ZM == i % ZM else "0.0" for i in range(ZM * ZM))
  b.append(f"double HPH[{ZM * ZM}] = {{{', '.join('0.0' for _ in range(ZM * ZM))}}}, Rl[{ZM * ZM}] = {{{ident}}}, S[{ZM * ZM}], L[{ZM * ZM}], iL[{ZM}];")
  b.append("switch (kind) {")
  for k in spec.kinds:
    Hs, Z = Hss[k.kind], k.zdim
    ln = [f"HPH[{zi * ZM + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w))};"
          for w in range(Z) for zi in range(Z)]
    ln += [f"Rl[{zi * ZM + w}] = gR[{zi * Z + w}];" for zi in range(Z) for w in range(Z)]
    b.append(f"  case {k.kind}: {{ " + " ".join(ln) + " break; }")
  b += ["  default: break;", "}"]
  b += ["#pragma unroll", f"for (int i = 0; i < {ZM * ZM}; i++) S[i] = HPH[i] + Rl[i];", f"rn::spd_factor<{ZM}>(S, L, iL);", "int gated = 0;",
        f"const double yv[{ZM}] = {{{', '.join(f'zk > {i} ? sl[{lay.OFF_Y + i}] : 0.0' for i in range(ZM))}}};"]
  if any(k.maha_test for k in spec.kinds):
    b += ["if (gate_on) {", f"  double v[{ZM}] = {{{', '.join(f'yv[{i}]' for i in range(ZM))}}};", f"  rn::spd_forward<{ZM}>(L, iL, v);",
          "  const double d2 = " + " + ".join(f"v[{i}]*v[{i}]*iL[{i}]" for i in range(ZM)) + ";", "  if (d2 > thr) {", "    gated = 1;",
          "#pragma unroll", f"    for (int i = 0; i < {ZM * ZM}; i++) {{ Rl[i] = 1.0e16 * Rl[i]; S[i] = HPH[i] + Rl[i]; }}",
          f"    rn::spd_factor<{ZM}>(S, L, iL);", "  }", "}"]
  else:
    b.append("(void)thr; (void)gate_on;")
  for s_ in range(R):
    b.append(f"rn::spd_solve<{ZM}>(L, iL, kk{s_});                       // K[row][:]")
    b.append(f"const double dx{s_} = " + " + ".join(f"kk{s_}[{zi}]*yv[{zi}]" for zi in range(ZM)) + ";")
  for s_ in range(R):
    b.append(f"if (ok{s_}) {{ sw[{lay.OFF_DX} + rr{s_}] = dx{s_};" + (f" if (rr{s_} == 0) sw[{lay.OFF_FL}] = (double)gated;" if s_ == 0 else "") + " }")
  b += _tl(4)
  b.append("rn::wg_barrier();      // B3: dx, flags -> the scalar wavefront")
  b += _tl(5)
  b += w3._rank_pass(E, ZM, R, "sG", "-=", "kk", JB=JB)      # pylint: disable=protected-access
  b += _tl(6)
  for s_ in range(R):
    b.append(f"double cc{s_}[{ZM}] = {{{', '.join('0.0' for _ in range(ZM))}}};")
  b.append("switch (kind) {")
  for k in spec.kinds:
    Hs, Z = Hss[k.kind], k.zdim
    ln = [f"cc{s_}[{zi}] = {sum_terms(term(cf, f'row{s_}[{j}]') for j, cf in Hs.row_nz(zi))};"
          for zi in range(Z) for s_ in range(R)]
    b.append(f"  case {k.kind}: {{ " + " ".join(ln) + " break; }")
  b += ["  default: break;", "}"]
  for s_ in range(R):
    b.append(f"double Dm{s_}[{ZM}];")
    for zi in range(ZM):
      kr = " + ".join(f"kk{s_}[{w}]*Rl[{w * ZM + zi}]" for w in range(ZM))
      b.append(f"Dm{s_}[{zi}] = ({kr}) - (cc{s_}[{zi}]);")
  b.append("rn::wave_lds_sync();")
  b.append("if (he_release > 0) rn::flag_set(he_flag, he_release);      // He, y are dead: the scalar wavefront may evaluate the next step's observation.  A FLAG, not a barrier --")
  b.append("                                                             // this wavefront needs nothing from the other one here and would wait ~1 us for its injection / trace stores")
  b += _tl(7)
  b.append("rn::wave_lds_sync();      // every lane has taken G: the buffer takes K^T")
  for s_ in range(R):
    b.append(f"if (ok{s_}) {{ " + " ".join(f"sG[{zi} * {E} + rr{s_}] = kk{s_}[{zi}];" for zi in range(ZM)) + " }")
  b.append("rn::wave_lds_sync();")
  b += w3._rank_pass(E, ZM, R, "sG", "+=", "Dm", JB=JB)      # pylint: disable=protected-access
  b.append("rn::wave_lds_sync();      // the broadcast buffer is free again")
  rows = ", ".join(f"double (&row{s_})[{E}]" for s_ in range(R))
  idx = ", ".join(f"const int rr{s_}, const int rc{s_}, const bool ok{s_}" for s_ in range(R))
  head = (f"__device__ __forceinline__ void update_rows_r2(const int kind, {rows}, const double* __restrict__ gR, double* sP, "
          f"double* sG, const double* sl, double* sw, {idx}, const int he_release, int* he_flag{', const int tlb' if _tl_on() else ''}) {{")
  return "\n".join([head] + _ind(b) + ["}"])


def kernels(spec, tri=False):
  """Scalar phase functions against Run2Layout (suffix _r2; the state lives in the filter's slot), the matrix functions, k_run2
  (tri: and k_run2_tri, the same kernel writing its covariance trace as packed lower triangles)."""
  from rednose_amd.codegen import emit_wide2 as w2
  scal_text, lay = w2.device_functions(spec, lay_cls=Run2Layout, sfx="_r2")
  return "\n".join([f"constexpr int SLOT_R2 = {lay.SLOT};   // two-wavefront fused run: doubles per scalar slot", "", scal_text, "",
                    predict_fn(spec), update_fn(spec), run_kernel(spec)] + ([run_kernel(spec, tri=True)] if tri else []))


def tri_trace(spec):
  """The packed covariance trace (k_run2_tri / k_rts4_tri: the lower triangle of every filtered covariance, E (E + 1) / 2 doubles instead of
  E^2, between batch_run_tri and batch_rts_tri) is generated for the models that have BOTH kernels."""
  from rednose_amd.codegen import emit_rts4, tuning
  return bool(tuning.current().tri_trace) and applicable(spec) and emit_rts4.tri_applicable(spec)


TRI_MACRO = r"""
// Row r of a lower triangle packed row-major starts at r (r + 1) / 2 and has r + 1 entries.  A lane stores ALL E entries of its row at that
// offset, in DESCENDING order of the column: entry j > r lands on entry j - (r' (r' + 1) - r (r + 1)) / 2 < j of a later row r' -- which that
// row's own lane writes afterwards (same instruction stream, lower column = later instruction; a later row slot = a later loop), so every
// valid entry is written last and nothing is predicated.  The last valid index of row r is the triangle's last for r = E - 1: nothing leaves
// the filter's E (E + 1) / 2 doubles.  Needs the wavefront's lockstep; the host emulation (a thread per lane) defines the guarded form.
// The compiler fence after every store is part of the scheme: to ONE lane its stores go to distinct addresses, so hipcc may reorder or pair
// them (ds_write2) -- the order only matters across lanes, which it cannot see.  (The first build without the fence wrote wrong triangles.)
#ifndef RN_TRI_ST
#define RN_TRI_ST(p, j, r, v) do { (p)[j] = (v); asm volatile("" ::: "memory"); } while (0)
#endif
"""


def run_kernel(spec, tri=False):
  """tri=True: `k_run2_tri`, the same kernel whose covariance trace is the packed lower triangle (trace_P: (T, n, E (E + 1) / 2))."""
  from rednose_amd.codegen import tuning
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  TRI = E * (E + 1) // 2
  GL, R, FPW, ND = layout2(spec)
  lay, _, _ = w3._tables(spec, Run2Layout)      # pylint: disable=protected-access
  zmax = max(k.zdim for k in spec.kinds)
  rows = ", ".join(f"row{s}" for s in range(R))
  idx = ", ".join(f"rr{s}, rc{s}, ok{s}" for s in range(R))
  nlc = chr(10)
  scal_cases = nlc.join(f"          case {k.kind}: scal_obs_{k.kind}_r2(sl, sl + {lay.OFF_Y}); break;" for k in spec.kinds)
  known = " || ".join(f"kind == {k.kind}" for k in spec.kinds)
  img = nlc.join(f"        if (ok{s}) {{\n#pragma unroll\n          for (int j = 0; j < {E}; j++) sP[rr{s} * {E} + j] = row{s}[j];\n        }}" for s in range(R))
  id0_guard = "true" if not spec.identity_at_dt0() else "dt != 0.0"
  id0_next = "true" if not spec.identity_at_dt0() else "dtn != 0.0"
  nt_trace = "true" if tuning.current().nt_trace else "false"
  prio = f"    __builtin_amdgcn_s_setprio({int(tuning.current().run2_prio)});      // a chain of dependent instructions: whenever one is ready it goes first\n" if tuning.current().run2_prio else ""
  s_lanes = f"    const int g = lane / {64 // FPG};\n    const int c = lane % {64 // FPG};"
  decl_rows = nlc.join(f"      double row{s}[{E}];" for s in range(R))
  # row indices from an opaque copy of the lane's position, per scope: as loop invariants they (and every LDS address derived from them)
  # would hold a dozen registers across the step loop
  def decl_idx_at(ind):
    return nlc.join([ind + "int cq = c;", ind + 'asm volatile("" : "+v"(cq));'] +
                    [ind + f"const int rr{s} = cq + {GL * s}; const bool ok{s} = live && rr{s} < {E}; const int rc{s} = rr{s} < {E} ? rr{s} : 0;" for s in range(R)])
  load_rows = nlc.join(f"#pragma unroll\n      for (int j = 0; j < {E}; j++) row{s}[j] = 0.5 * (sP[rc{s} * {E} + j] + sP[j * {E} + rc{s}]);" for s in range(R))
  def TL(ph, ind="        "):
    return "".join(ind + x + nlc for x in _tl(ph))
  tlb_decl = "        const int tlb = (int)(t % 3) * 20;\n" if _tl_on() else ""
  tl_arg = ", tlb" if _tl_on() else ""
  kname = "k_run2_tri" if tri else "k_run2"
  if tri:
    # rows -> the filter's packed triangle (at s_P + g * TRI: the tile's triangles are contiguous, like its records in the trace) -> one flat copy
    pk = []
    for s_ in range(R):
      pk.append(f"          if (ok{s_}) {{\n            double* pr_ = s_P + gg * {TRI} + (rr{s_} * (rr{s_} + 1)) / 2;")
      for j in range(E - 1, -1, -1):
        pk.append(f"            RN_TRI_ST(pr_, {j}, rr{s_}, row{s_}[{j}]);")
      pk.append("          }")
      pk.append("          rn::wave_lds_sync();      // (the next row slot's stores come after this one's)")
    trace_block = (nlc.join(pk) + nlc +
                   f"          rn::copy_l2g<R2_FPW * {TRI}, {nt_trace}>(tP + (t * n + base + wave * R2_FPW) * {TRI}, cntw * {TRI}, s_P + wave * (R2_FPW * {TRI}), lz);")
  else:
    trace_block = (f"{img}" + nlc + "          rn::wave_lds_sync();" + nlc +
                   f"          rn::copy_l2g<R2_FPW * {EE}, {nt_trace}>(tP + (t * n + base + wave * R2_FPW) * {EE}, cntw * {EE}, sPw, lz);")
  head = (TRI_MACRO if tri else "") + f"""
// ---- fused multi-step run, {ND} matrix wavefront(s) + a scalar wavefront per tile of {FPG} filters (emit_run2.py): same interface as k_run{' -- covariance trace as packed lower triangles' if tri else ''} ----"""
  consts = "" if tri else f"""
constexpr int R2_FPG = {FPG};      // filters per workgroup
constexpr int R2_FPW = {FPW};      // filters per matrix wavefront
constexpr int R2_GL = {GL};       // lanes per filter in a matrix wavefront
constexpr int R2_THREADS = {64 * (ND + 1)};"""
  return head + consts + f"""
__global__ __launch_bounds__({64 * (ND + 1)}, {ND + 1}) void {kname}(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* __restrict__ gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    double* __restrict__ tx, double* __restrict__ tP, const double* __restrict__ gea, const int32_t* __restrict__ augs) {{
  (void)gea; (void)augs;
  __shared__ __attribute__((aligned(16))) double s_P[R2_FPG * {EE} + 2];      // one image of P per filter; its first {zmax} rows are the G / K^T buffer of the update (which never touches the image)
  __shared__ __attribute__((aligned(16))) double s_sl[R2_FPG * SLOT_R2];
  __shared__ double s_qd[{E + (E & 1)}];                                     // diag(Q)
  __shared__ int s_bad;
  __shared__ int s_he;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t tiles = (n + R2_FPG - 1) / R2_FPG;
  if (wave < {ND}) {{
    // ================================ matrix wavefront(s): filters wave * R2_FPW .. of the tile ================================
    const int g = wave * R2_FPW + lane / R2_GL;
    const int c = lane % R2_GL;
    int qoff = 0;
    for (int i = lane; i < {EE}; i += 64) qoff |= (i / {E} != i % {E}) && (gQ[i] != 0.0);
    const bool qdiag = !__any(qoff);
    if (wave == 0 && lane < {E}) s_qd[lane] = gQ[lane * {E + 1}];
    rn::wave_lds_sync();
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
      const int64_t base = tile * R2_FPG;
      const int cnt = (n - base) < R2_FPG ? (int)(n - base) : R2_FPG;
      const int gg = g < cnt ? g : 0;
      const bool live = g < cnt;
      const int cntw = cnt - wave * R2_FPW < 0 ? 0 : (cnt - wave * R2_FPW > R2_FPW ? R2_FPW : cnt - wave * R2_FPW);      // this wavefront's filters
      double* sP = s_P + gg * {EE};
      double* sl = s_sl + gg * SLOT_R2;
      double* sPw = s_P + wave * (R2_FPW * {EE});
      int lb = lane;
      asm volatile("" : "+v"(lb));
      rn::copy_g2l<R2_FPW * {EE}>(gP + (base + wave * R2_FPW) * {EE}, cntw * {EE}, sPw, lb);
      rn::wave_lds_sync();
{decl_rows}
      {{
{decl_idx_at("        ")}
{load_rows}
      }}
      rn::wave_lds_sync();
      rn::wg_barrier();                                   // B1 of step 0
      for (int64_t t = 0; t < T; t++) {{
        const int kind = kinds[t];
        const double dt = dts[t];
        const bool do_pred = {id0_guard};
{decl_idx_at("        ")}
{tlb_decl}{TL(0)}        int he_release = 0;                               // > 0: the next step has no predict, its h / He go under this step's tail -- the value the flag takes when He is dead
        if (t + 1 < T) {{ const double dtn = dts[t + 1]; if (!({id0_next})) he_release = (int)t + 1; }}
        if (do_pred) predict_rows_r2({rows}, sP, s_qd, gQ, qdiag, sl, {idx});
{TL(1)}        rn::wg_barrier();                                 // B2
{TL(2)}        const int bad = __builtin_amdgcn_readfirstlane(s_bad);
        if (!bad && ({known})) {{
          update_rows_r2(kind, {rows}, gR + t * {zmax * zmax}, sP, sP, sl, sl, {idx}, he_release, &s_he{tl_arg});
        }} else {{
          rn::wg_barrier();                               // B3
          if (he_release > 0) rn::flag_set(&s_he, he_release);
        }}
{TL(8)}        if (tP != nullptr) {{
          int lz = lane;
          asm volatile("" : "+v"(lz));
{trace_block}
          rn::wave_lds_sync();
        }}
{TL(9)}        rn::wg_barrier();                                 // B1 of step t + 1
      }}
      {{
{decl_idx_at("        ")}
{img}
      }}
      rn::wave_lds_sync();
      int le = lane;
      asm volatile("" : "+v"(le));
      rn::copy_l2g<R2_FPW * {EE}>(gP + (base + wave * R2_FPW) * {EE}, cntw * {EE}, sPw, le);
      rn::wave_lds_sync();
    }}
  }} else {{
    // ================================ scalar wavefront: one lane per filter ================================
{prio}{s_lanes}
    const int zf = lane / {zmax}, zc = lane % {zmax};      // observation entry this lane carries between HBM and the slots
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
      const int64_t base = tile * R2_FPG;
      const int cnt = (n - base) < R2_FPG ? (int)(n - base) : R2_FPG;
      const int gg = g < cnt ? g : 0;
      const bool live = g < cnt;
      const bool zlive = zf < cnt;
      double* sl = s_sl + gg * SLOT_R2;
      double* slz = s_sl + (zlive ? zf : 0) * SLOT_R2 + {lay.OFF_Y} + zc;
      int l0 = lane;
      asm volatile("" : "+v"(l0));
      for (int i = l0; i < cnt * {D}; i += 64) s_sl[(i / {D}) * SLOT_R2 + {lay.OFF_X} + i % {D}] = gx[base * {D} + i];
      if (zlive) *slz = gz[base * {zmax} + lane];
      rn::wave_lds_sync();
      if (lane == 0) s_he = 0;
      int bad = 0;
      bool obs_done = false;
      if (T > 0) {{
        const double dt0 = dts[0];
        const double dtn = dt0;
        const bool p0 = {id0_next};
        if (c == 0 && live) {{
          if (p0) scal_predict_r2(sl + {lay.OFF_X}, dt0, sl, norm_quats);
          else scal_keep_r2(sl + {lay.OFF_X}, sl, norm_quats);
        }}
        rn::wave_lds_sync();
      }}
      rn::wg_barrier();                                   // B1 of step 0
      for (int64_t t = 0; t < T; t++) {{
{tlb_decl}{TL(10)}        double zn = 0.0;                                  // next step's observation, in flight during this step
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
{TL(11)}        rn::wg_barrier();                                 // B2
{TL(12)}        rn::wg_barrier();                                 // B3: dx and the gate flag are in the slot
{TL(13)}        if (c == 0 && live) {{
          int fl = bad;
          if (!bad) fl = scal_inject_r2(sl, sl + {lay.OFF_X}, norm_quats) | (int)sl[{lay.OFF_FL}];
          if (flags != nullptr) flags[t * n + base + g] = (uint8_t)fl;
        }}
{TL(14)}        if (zlive) gz[(t * n + base) * {zmax} + lane] = *slz;          // y (the observation itself after an unknown kind)
        rn::wave_lds_sync();
        if (tx != nullptr) {{
          int lz = lane;
          asm volatile("" : "+v"(lz));
          for (int i = lz; i < cnt * {D}; i += 64) tx[(t * n + base) * {D} + i] = s_sl[(i / {D}) * SLOT_R2 + {lay.OFF_X} + i % {D}];
          rn::wave_lds_sync();
        }}
        if (zlive) *slz = zn;
        rn::wave_lds_sync();
{TL(15)}        obs_done = false;
        if (t + 1 < T) {{
          const double dtn = dts[t + 1];
          const bool pn = {id0_next};
          if (c == 0 && live) {{
            if (pn) scal_predict_r2(sl + {lay.OFF_X}, dtn, sl, norm_quats);
            else scal_keep_r2(sl + {lay.OFF_X}, sl, norm_quats);
          }}
          rn::wave_lds_sync();
{TL(16, "          ")}          if (!pn) {{
            rn::flag_wait(&s_he, (int)t + 1);             // He of step t is dead (set by the matrix wavefront after its Joseph coefficients)
{TL(17, "            ")}
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
{TL(19)}        rn::wg_barrier();                                 // B1 of step t + 1
      }}
      int le = lane;
      asm volatile("" : "+v"(le));
      for (int i = le; i < cnt * {D}; i += 64) gx[base * {D} + i] = s_sl[(i / {D}) * SLOT_R2 + {lay.OFF_X} + i % {D}];
      rn::wave_lds_sync();
    }}
  }}
}}
"""


def launch_run(tri=False):
  return f"""  const int64_t tiles = (n + R2_FPG - 1) / R2_FPG;
  hipLaunchKernelGGL({'k_run2_tri' if tri else 'k_run2'}, dim3(rn::grid_for_tiles(tiles)), dim3(R2_THREADS), 0, (hipStream_t)stream,
                     x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, trace_x, trace_P, ea, augment);"""
