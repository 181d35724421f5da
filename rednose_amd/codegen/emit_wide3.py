"""Kernel family W, fused multi-step run `k_run`: state resident in registers, SEVERAL ROWS PER LANE.

The first version of this kernel (round 1) gave every filter a 32-lane group, one row of P per lane, and evaluated the
x-dependent scalars (f, F, h, H.H_mod, err_fun: ~1 600 instructions for live's accelerometer kind) in every lane of the
group -- per wavefront that is the same instruction count whether 2 or 8 filters share it, so with 2 filters per
wavefront the scalars alone cost 3 us per step, the rank-Z passes ran at 22 busy lanes of 32, each FMA fetched one
broadcast operand from LDS, and the kernel needed 256 + 166 registers with a thousand AGPR moves: 343 M steps/s on
live's config-4 forward pass, SLOWER per step than one launch per step.

Here a filter gets GL = 8 lanes (16 above 22 error states) and lane c owns rows c, c + GL, c + 2 GL ... of P
(R = ceil(E / GL) rows per lane), so a wavefront carries 8 (4) filters:
  * the scalar phases run ONCE per filter (one lane per filter, through an LDS slot -- the step kernels' generated functions,
    instantiated against RunLayout), amortised over 4x as many filters per wavefront;
  * every broadcast operand of the rank-Z passes (a row of G or K^T from LDS) feeds R FMAs instead of one;
  * live: 22 of 24 row slots busy instead of 22 of 32 lanes.
P lives in registers as rows only (R x E doubles per lane: 132 VGPRs for live).  LDS holds per filter one E x E image -- the
scratch of predict's single transposition and the staging buffer of the trace / MSCKF window shift, written only then --, one
Z x E buffer shared by G and K^T, and the scalar slot.  P = P^T (up to the Joseph form's rounding, as in the reference) is used
twice: columns of A = P F^T are rows of F P (predict_fn), and column j of G = He P is row j of P against He (update_fn).

  predict (ekf_c.c:8-33)   rows of A = P F^T (row-local) -> image -> columns of A -> P'[r] = (F P)[r] F^T + dt Q[r] (row-local)
  update  (ekf_c.c:37-121) Joseph form with its rank-Z structure, per row slot; S is factored redundantly per lane;
                           feature-track kinds project on the null space of the extra-argument Jacobian with Householder
                           reflectors (ekf_c.c:66-76)
Results agree with the step-granular path to rounding (different association of the same sums); tests/test_gpu_run.py,
test_gpu_random.py, test_gpu_msckf.py, test_gpu_fullsize.py bound it.  Measurements and what hipcc needed: DESIGN.md section 3.

(The same layout for the step-granular entry points was built in round 2 and measured slower than the three-phase kernels of
emit_wide2 -- one wavefront per SIMD cannot overlap its own HBM round trip; numbers in profiles/tuning_notes.md.)
"""
import sympy as sp

from rednose_amd.codegen.emit_common import term, sum_terms

EADIM = 3        # extra-argument dimension of feature-track kinds, hard-coded in the reference (ekf_sym.py:151)


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def ea_dim(k):
  return 0 if k.ea_sym is None else int(sp.Matrix(k.ea_sym).shape[0])


def ea_max(spec):
  return max([ea_dim(k) for k in spec.kinds] + [0])


def layout(spec):
  """-> (GL lanes per filter, R rows per lane, FPW filters per wavefront).  8 lanes while 8 covariance images fit the LDS
  budget of a wavefront (E <= 22), 16 up to 32 error states; above that the row sets of several rows per lane no longer fit the
  register file (feature36 at 16 lanes x 3 rows: 120 spilled registers) and a filter takes the whole wavefront, a row per lane."""
  E = spec.dim_err
  GL = 8 if E <= 22 else (16 if E <= 32 else 64)       # above 32 error states: one filter per wavefront, one row per lane
  return GL, -(-E // GL), 64 // GL


def _tl(ph):
  """Debug stamp inside the matrix phases (tuning knob wide_timeline): slot 20 * (t % 3) + ph of the workgroup's timeline."""
  from rednose_amd.codegen import tuning
  if not tuning.current().wide_timeline:
    return []
  return [f"if (threadIdx.x == 0 && blockIdx.x < 256) {{ const int ti_ = tl_t * 20 + {ph}; g_tl[(blockIdx.x * 64 + ti_) * 2] = "
          "__builtin_readcyclecounter(); g_tl[(blockIdx.x * 64 + ti_) * 2 + 1] = wall_clock64(); }"]


def _tl_arg(call=False):
  from rednose_amd.codegen import tuning
  if not tuning.current().wide_timeline:
    return ""
  return ", (int)(t % 3)" if call else ", const int tl_t"


def _rank_pass(E, Z, R, src, op, coef, JB=None):
  """Straight-line rank-Z pass over the register rows: row_s[j] op= sum_z coef_s[z] * src[z][j] for all j, in blocks of JB
  columns.  Two things hipcc does not do by itself here:
    * the broadcast operands of block b + 1 are loaded before the FMAs of block b and a compiler fence closes every block, so
      one block of loads is in flight under the arithmetic and no more (left alone it issues all Z*E LDS loads first: 2*Z*E
      registers on top of the rows -> hundreds of spills);
    * inside a block the FMAs are emitted term by term ACROSS the block's JB * R entries, so consecutive instructions belong to
      different accumulation chains: a dependent fp64 FMA issues ~40 cycles after its predecessor with one wavefront per SIMD
      (tools/fp64_ilp.hip), entry-by-entry order made every FMA wait for the one before it."""
  from rednose_amd.codegen import tuning
  JB = JB or 4      # columns per block (6 and 8 measured in round 5: 21.21 / 20.90 ms per config-4 chunk against 21.04 / 21.24 -- noise; profiles/tuning_notes.md)
  out = []
  blocks = [list(range(j, min(j + JB, E))) for j in range(0, E, JB)]
  sg = "-=" if op == "-=" else "+="

  def loads(bl):
    return [f"const double q_{zi}_{j} = {src}[{zi} * {E} + {j}];" for j in bl for zi in range(Z)]
  out += loads(blocks[0])
  for bi, bl in enumerate(blocks):
    if bi + 1 < len(blocks):
      out += loads(blocks[bi + 1])
    for zi in range(Z):
      for j in bl:
        for s in range(R):
          out.append(f"row{s}[{j}] {sg} {coef}{s}[{zi}]*q_{zi}_{j};")
    out.append(" ".join(f"rn::pin(row{s}[{j}]);" for j in bl for s in range(R)))
    out.append("rn::wave_lds_sync();")
  return ["{"] + _ind(out) + ["}"]


class RunLayout:
  """Per-filter scalar slot of the fused run (doubles).  Same fields as emit_wide2.Layout, packed by lifetime so that eight
  filters of a 22-state model, their covariance images and the G / K^T buffers stay under 40 KB (4 wavefronts per CU):
  the non-trivial entries of F (dead once predict's matrix phase is through), those of He = H H_mod (written after that, dead
  once the Joseph-form coefficients exist) and dx (written after that) share one region; x lives ONLY here (no separate
  copy), and z comes in / y goes out through the Y field."""

  def __init__(self, spec, f_vars, he_vars_by_kind):
    D, E = spec.dim_x, spec.dim_err
    self.zmax = max(k.zdim for k in spec.kinds)
    self.nf = len(f_vars)
    self.nh = max([len(v) for v in he_vars_by_kind.values()] + [0])
    self.OFF_X = 0
    self.OFF_F = self.OFF_HE = self.OFF_DX = D
    self.OFF_Y = D + max(self.nf, self.nh, E)
    self.OFF_DT = self.OFF_Y + self.zmax
    self.OFF_FL = self.OFF_DT + 1
    feat = [k for k in spec.kinds if k.He_sym is not None]
    self.zf = max([k.zdim for k in feat] + [0])
    self.OFF_RF = self.OFF_FL + 1
    self.OFF_RP = self.OFF_RF + (EADIM * self.zf + EADIM if feat else 0)
    self.OFF_YP = self.OFF_RP + ((self.zf - EADIM) ** 2 if feat else 0)      # (see emit_wide2.Layout: the residual in the reflectors' basis, the elimination's work space)
    n = self.OFF_YP + ((self.zf - EADIM) if feat else 0)
    self.SLOT = n + 1 - (n & 1)      # odd stride, as in the step kernels


def _tables(spec, lay_cls=None):
  """Slot layout and slot-addressed coefficient matrices; the phase-1 / phase-3 functions are emit_wide2's, instantiated
  against RunLayout under the suffix _r (emit_run2 passes its own layout class)."""
  from rednose_amd.codegen import emit_wide2 as w2
  _, _, F, f_vars = w2._lowered_predict(spec)                  # pylint: disable=protected-access
  obs = {k.kind: w2._lowered_obs(spec, k) for k in spec.kinds}  # pylint: disable=protected-access
  lay = (lay_cls or RunLayout)(spec, f_vars, {kk: v[3] for kk, v in obs.items()})
  Fs = w2._slotted(F, f_vars, lay.OFF_F)                        # pylint: disable=protected-access
  Hs = {kk: w2._slotted(v[2], v[3], lay.OFF_HE) for kk, v in obs.items()}   # pylint: disable=protected-access
  return lay, Fs, Hs


def predict_fn(spec, qdiag=False):
  """Matrix part of predict on register rows; F's non-trivial entries are broadcast reads of the filter's slot.

  P' = F P F^T + dt Q with ONE transposition through LDS: rows of A = P F^T are row-local; under P = P^T the columns of A are the
  rows of B = F P (B[r][m] = sum_k F[r][k] P[k][m] = sum_k F[r][k] P[m][k] = A[m][r]), and P'[r][:] = B[r][:] F^T + dt Q[r][:] is
  row-local again, so the new rows land in the registers that hold them for the next step.  (P is symmetric up to the rounding
  of the Joseph form, in the reference as here; the reference multiplies F P F^T out entry by entry, ekf_c.c:8-33.)

  qdiag=True emits the variant for a DIAGONAL process noise (detected once per launch by k_run): the lane's diagonal entries of Q
  arrive as register operands and no global memory is read.  That matters beyond the 3 R E loads saved: a load's s_waitcnt
  vmcnt also waits for every EARLIER store of the wavefront (the counter retires in order), so with the trace enabled the rows
  of Q fetched here drained the previous step's 32 KB of trace stores on every step -- the only vector load consumed in the
  middle of a step (profiles/r3a: 11.6 us per step with the trace against 8.7 without)."""
  E = spec.dim_err
  GL, R, _ = layout(spec)
  lay, Fs, _ = _tables(spec)
  b = [f"const double dt = sl[{lay.OFF_DT}];"]
  # rows of A, whole rows at a time: 16-byte LDS stores of a lane's contiguous row (entry-wise 8-byte stores at a row stride
  # collide on banks)
  for s in range(R):
    b.append("{")
    b.append(f"  double a[{E}];")
    for i in range(E):
      b.append(f"  a[{i}] = {sum_terms(term(cf, f'row{s}[{k}]') for k, cf in Fs.row_nz(i))};")
    b += [f"  if (ok{s}) {{", "#pragma unroll", f"    for (int i = 0; i < {E}; i++) sP[rr{s} * {E} + i] = a[i];", "  }"]
    b.append("}")
  b.append("rn::wave_lds_sync();")
  b += _tl(8)
  if qdiag:
    for s in range(R):
      b.append(f"const double dq{s} = dt * qd{s};")
      b.append("{")
      b.append(f"  double a[{E}];")
      b += ["#pragma unroll", f"  for (int m = 0; m < {E}; m++) a[m] = sP[m * {E} + rc{s}];      // column of A = row of B"]
      for j in range(E):
        diag = f" + (rc{s} == {j} ? dq{s} : 0.0)" if GL * s <= j < GL * (s + 1) else ""      # the lane's own row index is c + {GL} s
        b.append(f"  row{s}[{j}] = {sum_terms(term(cf, f'a[{m}]') for m, cf in Fs.row_nz(j))}{diag};")
      b.append("}")
  else:
    # Q is read from HBM / L2 (no LDS left for it): slot s's row of Q is requested one slot ahead of its use
    b.append(f"double q0[{E}];")
    b += ["#pragma unroll", f"for (int j = 0; j < {E}; j++) q0[j] = gQ[rc0 * {E} + j];"]
    for s in range(R):
      if s + 1 < R:
        b.append(f"double q{s + 1}[{E}];")
      b.append("{")
      b.append(f"  double a[{E}];")
      b += ["#pragma unroll", f"  for (int m = 0; m < {E}; m++) a[m] = sP[m * {E} + rc{s}];      // column of A = row of B"]
      if s + 1 < R:
        b += ["#pragma unroll", f"  for (int j = 0; j < {E}; j++) q{s + 1}[j] = gQ[rc{s + 1} * {E} + j];"]
      for j in range(E):
        b.append(f"  row{s}[{j}] = {sum_terms(term(cf, f'a[{m}]') for m, cf in Fs.row_nz(j))} + dt*q{s}[{j}];")
      b.append("}")
  b.append("rn::wave_lds_sync();      // the image is free again")
  b += _tl(9)
  rows = ", ".join(f"double (&row{s})[{E}]" for s in range(R))
  idx = ", ".join(f"const int rr{s}, const int rc{s}, const bool ok{s}" for s in range(R))
  if qdiag:
    qarg = ", ".join(f"const double qd{s}" for s in range(R))
    head = (f"__device__ __forceinline__ void predict_rows_qd({rows}, double* sP, {qarg}, const double* sl, {idx}{_tl_arg()}) {{")
  else:
    head = (f"__device__ __forceinline__ void predict_rows({rows}, double* sP, const double* __restrict__ gQ, const double* sl, {idx}{_tl_arg()}) {{")
  return "\n".join([head] + _ind(b) + ["}"])


def update_fn(spec, k):
  """Matrix part of the update of kind k on register rows.  y, the non-trivial entries of He = H H_mod and, for feature-track
  kinds, the Householder reflectors and the projected noise are read from the filter's slot (phase 1 put them there); dx and
  the gate / rank flags go back to it."""
  E, Zf = spec.dim_err, k.zdim
  _, R, _ = layout(spec)
  lay, _, Hss = _tables(spec)
  Hs = Hss[k.kind]
  feat = k.He_sym is not None
  Z = Zf - EADIM if feat else Zf
  RF, RB = lay.OFF_RF, lay.OFF_RF + EADIM * Zf
  b = ["(void)sP;", f"double R[{Z * Z}];"]
  if feat:
    b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) R[i] = sl[{lay.OFF_RP} + i];      // A^T R A (phase 1)", "(void)gR;",
          f"const double rank_deficient = sl[{lay.OFF_FL}];      // 4.0 when phase 1 found Hea rank deficient"]
  else:
    b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) R[i] = gR[i];"]
  # G = He P: G[z][j] = sum_k He[z][k] P[k][j]; with P = P^T (up to the Joseph form's rounding) that is row j of P against
  # He[z][:] -- row-local, so the lane that owns row j has column j of G in registers; the same numbers are the row view
  # P He^T that is solved into K below.  Nothing of the update reads the LDS image of P.
  for s in range(R):
    if feat:
      b.append(f"double t0_{s}[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'row{s}[{kk}]') for kk, cf in Hs.row_nz(zi)) for zi in range(Zf)) + "};")
      b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(sl + {RF}, sl + {RB}, t0_{s});")
      b.append(f"double kk{s}[{Z}] = {{{', '.join(f't0_{s}[{EADIM + zi}]' for zi in range(Z))}}};")
    else:
      b.append(f"double kk{s}[{Z}] = {{" + ", ".join(sum_terms(term(cf, f'row{s}[{kk}]') for kk, cf in Hs.row_nz(zi)) for zi in range(Z)) + "};")
    b.append(f"if (ok{s}) {{ " + " ".join(f"sG[{zi} * {E} + rr{s}] = kk{s}[{zi}];" for zi in range(Z)) + " }")
  b.append("rn::wave_lds_sync();")
  b += _tl(10)
  b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  if feat:
    for zi in range(Z):
      b.append("{")
      b.append(f"  double m[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w)) for w in range(Zf)) + "};")
      b.append(f"  rn::apply_reflectors<{Zf}, {EADIM}>(sl + {RF}, sl + {RB}, m);")
      b += ["#pragma unroll", f"  for (int w = 0; w < {Z}; w++) HPH[{zi * Z} + w] = m[{EADIM} + w];", "}"]
  else:
    for zi in range(Z):
      for w in range(Z):
        b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w))};")
  b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}", f"rn::spd_factor<{Z}>(S, L, iL);",
        "int gated = 0;"]
  if k.maha_test:
    b += ["{", f"  double v[{Z}] = {{{', '.join(f'sl[{(lay.OFF_YP if feat else lay.OFF_Y) + i}]' for i in range(Z))}}};", f"  rn::spd_forward<{Z}>(L, iL, v);",
          "  const double d2 = " + " + ".join(f"v[{i}]*v[{i}]*iL[{i}]" for i in range(Z)) + ";", f"  if (d2 > {k.maha_thresh!r}) {{", "    gated = 1;",
          "#pragma unroll", f"    for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = 1.0e16 * Rl[i]; S[i] = HPH[i] + Rl[i]; }}",
          f"    rn::spd_factor<{Z}>(S, L, iL);", "  }", "}"]
  b += _tl(11)
  for s in range(R):
    b.append(f"rn::spd_solve<{Z}>(L, iL, kk{s});                       // K[row][:]")
    if feat:     # the reference's numpy path ignores a measurement whose null-space projection failed (ekf_sym.py:589-591)
      b += ["if (rank_deficient != 0.0) {", "#pragma unroll", f"  for (int i = 0; i < {Z}; i++) kk{s}[i] = 0.0;", "}"]
    b.append(f"const double dx{s} = " + " + ".join(f"kk{s}[{zi}]*sl[{(lay.OFF_YP if feat else lay.OFF_Y) + zi}]" for zi in range(Z)) + ";")
  # B = P - K G: every broadcast row of G feeds all R row slots
  b += _tl(12)
  b += _rank_pass(E, Z, R, "sG", "-=", "kk")
  b += _tl(13)
  for s in range(R):
    if feat:
      b.append(f"double Cf{s}[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'row{s}[{j}]') for j, cf in Hs.row_nz(w)) for w in range(Zf)) + "};")
      b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(sl + {RF}, sl + {RB}, Cf{s});")
    b.append(f"double Dm{s}[{Z}];")
    for zi in range(Z):
      c = f"Cf{s}[{EADIM + zi}]" if feat else sum_terms(term(cf, f"row{s}[{j}]") for j, cf in Hs.row_nz(zi))
      kr = " + ".join(f"kk{s}[{w}]*Rl[{w * Z + zi}]" for w in range(Z))
      b.append(f"Dm{s}[{zi}] = " + ("rank_deficient != 0.0 ? 0.0 : " if feat else "") + f"({kr}) - ({c});")
  b.append("rn::wave_lds_sync();      // every lane has taken G (and y): the buffer takes K^T, the slot takes dx and the flags")
  fl = "(double)gated + rank_deficient" if feat else "(double)gated"
  for s in range(R):
    b.append(f"if (ok{s}) {{ " + " ".join(f"sG[{zi} * {E} + rr{s}] = kk{s}[{zi}];" for zi in range(Z)) + f" sw[{lay.OFF_DX} + rr{s}] = dx{s};" +
             (f" if (rr{s} == 0) sw[{lay.OFF_FL}] = {fl};" if s == 0 else "") + " }")
  b.append("rn::wave_lds_sync();")
  b += _tl(14)
  b += _rank_pass(E, Z, R, "sG", "+=", "Dm")
  b += _tl(15)
  b.append("rn::wave_lds_sync();      // the broadcast buffer is free again")
  rows = ", ".join(f"double (&row{s})[{E}]" for s in range(R))
  idx = ", ".join(f"const int rr{s}, const int rc{s}, const bool ok{s}" for s in range(R))
  head = (f"__device__ __forceinline__ void update_{k.kind}_rows({rows}, const double* __restrict__ gR, double* sP, "
          f"double* sG, const double* sl, double* sw, {idx}{_tl_arg()}) {{")
  return "\n".join([head] + _ind(b) + ["}"])


def kernels(spec, with_run=True):
  """Matrix-phase device functions + the fused multi-step kernel (the phase-1 / phase-3 functions are emit_wide2's, which must
  precede this text in the generated file).  with_run=False: only the layout constants -- the model's fused run is emit_run2's k_run2
  (the smoother emitters use these constants and emit their own functions)."""
  from rednose_amd.codegen import emit_wide2 as w2
  GL, R, FPW = layout(spec)
  scal_text, lay = w2.device_functions(spec, lay_cls=RunLayout, sfx="_r")
  if not with_run:
    return "\n".join([f"constexpr int GLR = {GL};    // fused run: lanes per filter", f"constexpr int RPL = {R};    // rows of P per lane",
                      f"constexpr int FPWR = {FPW};   // filters per wavefront", f"constexpr int SLOT_R = {lay.SLOT};   // doubles per scalar slot of the single-wavefront layout", ""])
  out = [f"constexpr int GLR = {GL};    // fused run: lanes per filter", f"constexpr int RPL = {R};    // rows of P per lane",
         f"constexpr int FPWR = {FPW};   // filters per wavefront", f"constexpr int SLOT_R = {lay.SLOT};   // fused run: doubles per scalar slot",
         "", scal_text, "", predict_fn(spec), predict_fn(spec, qdiag=True)]
  for k in spec.kinds:
    out.append(update_fn(spec, k))
  out.append(run_kernel(spec))
  return "\n".join(out)


def run_kernel(spec):
  from rednose_amd.codegen import tuning
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  GL, R, FPW = layout(spec)
  lay, _, _ = _tables(spec)
  zmax = max(k.zdim for k in spec.kinds)
  # observation entries a lane carries between HBM and the slots: FPW * zmax values per tile and step, one per lane and pass
  # (a single pass up to 8-dimensional observations at 8 filters per wavefront; the reference puts no limit on ZDIM, ekf_c.c:37)
  NZ = -(-(FPW * zmax) // 64)
  if NZ == 1:
    z_decl = f"const int zf = lane / {zmax}, zc = lane % {zmax};                // observation entry this lane carries between HBM and the slots"
    z_tile_a = "    const bool zlive = zf < cnt;\n"
    z_tile_b = f"    double* slz = s_sl + (zlive ? zf : 0) * SLOT_R + {lay.OFF_Y} + zc;\n"
    z_first = f"    if (zlive) *slz = gz[base * {zmax} + lane];"
    z_next = ("      double zn = 0.0;                                 // next step's observation, in flight during this step\n"
              f"      if (t + 1 < T && zlive) zn = gz[((t + 1) * n + base) * {zmax} + lane];")
    z_out = f"      if (zlive) gz[(t * n + base) * {zmax} + lane] = *slz;          // y (the observation itself after an unknown kind)"
    z_commit = "      if (zlive) *slz = zn;"
  else:
    q_ = range(NZ)
    z_decl = " ".join(f"const int zf{q} = (lane + {64 * q}) / {zmax}, zc{q} = (lane + {64 * q}) % {zmax};" for q in q_) + \
             f"      // {NZ} observation entries per lane ({FPW} filters x {zmax})"
    z_tile_a = "".join(f"    const bool zlive{q} = zf{q} < cnt;\n" for q in q_)
    z_tile_b = "".join(f"    double* slz{q} = s_sl + (zlive{q} ? zf{q} : 0) * SLOT_R + {lay.OFF_Y} + zc{q};\n" for q in q_)
    z_first = "\n".join(f"    if (zlive{q}) *slz{q} = gz[base * {zmax} + lane + {64 * q}];" for q in q_)
    z_next = "\n".join([f"      double zn[{NZ}] = {{{', '.join('0.0' for _ in q_)}}};                 // next step's observations, in flight during this step"] +
                        [f"      if (t + 1 < T && zlive{q}) zn[{q}] = gz[((t + 1) * n + base) * {zmax} + lane + {64 * q}];" for q in q_])
    z_out = "\n".join(f"      if (zlive{q}) gz[(t * n + base) * {zmax} + lane + {64 * q}] = *slz{q};" for q in q_)
    z_commit = "\n".join(f"      if (zlive{q}) *slz{q} = zn[{q}];" for q in q_)
  EAM = ea_max(spec)
  rows = ", ".join(f"row{s}" for s in range(R))
  idx = ", ".join(f"rr{s}, rc{s}, ok{s}" for s in range(R))
  scal_cases, mat_cases = [], []
  for k in spec.kinds:
    EA = ea_dim(k)
    feat = k.He_sym is not None
    args = f"sl, sl + {lay.OFF_Y}"
    guard = ""
    if EA:
      args += f", gea + ((int64_t)t * n + base + g) * {EAM}"
      guard = "if (gea == nullptr) { bad = 8; break; } "
    if feat:
      args += f", gR + t * {zmax * zmax}"
    scal_cases.append(f"          case {k.kind}: {{ {guard}scal_obs_{k.kind}_r{'<true>' if feat else ''}({args}); break; }}")
    mat_cases.append(f"        case {k.kind}: update_{k.kind}_rows({rows}, gR + t * {zmax * zmax}, sP, s_G + gg * {zmax * E}, sl, sl, {idx}{_tl_arg(True)}); break;")
  # rows -> LDS image (only where something reads the image: trace, window shift, the final store)
  img = "\n".join(f"        if (ok{s}) {{\n#pragma unroll\n          for (int j = 0; j < {E}; j++) sP[rr{s} * {E} + j] = row{s}[j];\n        }}" for s in range(R))
  aug = ""
  if spec.N > 0:
    d1, d2, d3, d4 = spec.dim_main, spec.dim_main_err, spec.dim_augment, spec.dim_augment_err
    src_x = [i if i < d1 else (i + d3 if i < D - d3 else i - (D - d3)) for i in range(D)]

    def se(i):
      r = i if i < E - d4 else i - (E - d4)
      return r if r < d2 else r + d4
    shift = []
    for s in range(R):
      shift.append(f"        {{ const int sr = (rc{s} < {E - d4} ? rc{s} : rc{s} - {E - d4}); const int srow = sr < {d2} ? sr : sr + {d4};")
      shift += [f"          row{s}[{j}] = sP[srow * {E} + {se(j)}];" for j in range(E)]
      shift.append("        }")
    nl = chr(10)
    aug = f"""
      // MSCKF window shift after this step (EKF_sym.augment, ekf_sym.py:365-391; the schedule's augment[t]): a fixed permutation
      // of the state (one lane per filter) and of the rows / columns of P, read out of the LDS image (the trace above holds the
      // estimate BEFORE the shift, like the reference's Estimate)
      if (augs != nullptr && augs[t] != 0) {{
{img}
        rn::wave_lds_sync();
        if (c == 0 && live) {{
          double xo[{D}];
#pragma unroll
          for (int i = 0; i < {D}; i++) xo[i] = sl[{lay.OFF_X} + i];
{nl.join(f"          sl[{lay.OFF_X + i}] = xo[{src_x[i]}];" for i in range(D) if src_x[i] != i)}
        }}
{nl.join(shift)}
        rn::wave_lds_sync();
      }}"""
  # predict(dt = 0) is skipped only for models where it is symbolically the identity (FilterSpec.identity_at_dt0)
  id0_guard = "true" if not spec.identity_at_dt0() else "dt != 0.0"
  nt_trace = "true" if tuning.current().nt_trace else "false"
  qd_decl = "\n".join(f"  const double qd{s} = gQ[((c + {GL * s}) < {E} ? (c + {GL * s}) : 0) * {E + 1}];" for s in range(R))
  qd_args = ", ".join(f"qd{s}" for s in range(R))
  decl_rows = "\n".join(f"    double row{s}[{E}];" for s in range(R))
  decl_idx = "\n".join(f"    const int rr{s} = c + {GL * s}; const bool ok{s} = live && rr{s} < {E}; const int rc{s} = rr{s} < {E} ? rr{s} : 0;" for s in range(R))
  # the fused run's covariance is symmetric by contract (include/rednose_amd_filter.h): (P + P^T) / 2 of the caller's matrix, once, as
  # the rows enter the registers -- predict_fn / update_fn below use P = P^T, the reference's dense products use both halves
  load_rows = "\n".join(f"#pragma unroll\n    for (int j = 0; j < {E}; j++) row{s}[j] = 0.5 * (sP[rc{s} * {E} + j] + sP[j * {E} + rc{s}]);" for s in range(R))
  nlc = chr(10)

  def TL(ph):      # debug stamps (tuning knob wide_timeline; tools/timeline.py run): the last three steps, twenty stamps each
    if not tuning.current().wide_timeline:
      return ""
    return (f"if (lane == 0 && blockIdx.x < 256) {{ const int ti_ = (int)(t % 3) * 20 + {ph}; "
            "g_tl[(blockIdx.x * 64 + ti_) * 2] = __builtin_readcyclecounter(); g_tl[(blockIdx.x * 64 + ti_) * 2 + 1] = wall_clock64(); }")
  # (Measured and not kept, round 4: the trace straight from the register rows as 8-byte stores to the TRANSPOSED positions -- lane c of
  # a group holds P[c + GL s][j]; written to [j][c + GL s] the lanes of a group cover 64 contiguous bytes per instruction, no LDS image,
  # no read-back -- config 4 forward 70.6 ms per chunk against 22.6 ms through the image: 66 store instructions per lane and step, each
  # a scatter of eight 64-byte pieces.  Round 3 measured the row-wise 16-byte variant at 25.3 ms.  profiles/tuning_notes.md.)
  # (Measured and not kept, round 4, profiles/tuning_notes.md: the copy of a step's covariance trace DEFERRED into the next step and woven
  # between the statements of its scalar phase when that step has no predict -- LDS / store throughput under a chain of dependent fp64
  # instructions, the arithmetic run redundantly on all lanes so that no divergent region separates the two streams.  Host-verified,
  # parity-green on the device, and 8 % SLOWER: config 4 forward 23.6 ms per chunk against 21.8.)
  scal_phase = f"""      if (c == 0 && live) {{
        switch (kind) {{
{nlc.join(scal_cases)}
          default: bad = 8; break;      // unknown kind
        }}
      }}"""
  pend_decl = flush_before_predict = flush_after_loop = ""
  trace_store = f"""{img}
        rn::wave_lds_sync();
        rn::copy_l2g<FPWR * {EE}, {nt_trace}>(tP + (t * n + base) * {EE}, cnt * {EE}, s_P, lz);"""
  return f"""
// ---- fused multi-step run: kinds[t], dts[t] shared by all filters; z is (T, n, {zmax}) in: z, out: y -----------
// Per step, with P in registers: (1a) one lane per filter evaluates f and the non-trivial entries of F into the filter's LDS
// slot, (2a) all lanes run predict's covariance algebra on their rows, (1b) one lane per filter evaluates h and He for the
// observation kind, (2b) all lanes run the update's covariance algebra, (3) one lane per filter injects the error state.
// x lives in the slot between the phases.
__global__ __launch_bounds__(64) void k_run(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* __restrict__ gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    double* __restrict__ tx, double* __restrict__ tP, const double* __restrict__ gea, const int32_t* __restrict__ augs) {{
  (void)gea; (void)augs;
  __shared__ __attribute__((aligned(16))) double s_P[FPWR * {EE} + 2];      // one image of P per filter (see emit_wide3.py)
  __shared__ __attribute__((aligned(16))) double s_G[FPWR * {zmax * E}];     // G, then K^T
  __shared__ __attribute__((aligned(16))) double s_sl[FPWR * SLOT_R];
  const int lane = threadIdx.x;
  const int g = lane / GLR;
  const int c = lane % GLR;
  {z_decl}
  // a diagonal process noise (the usual case) lives in registers: see predict_fn(qdiag=True)
  int qoff = 0;
  for (int i = lane; i < {EE}; i += 64) qoff |= (i / {E} != i % {E}) && (gQ[i] != 0.0);
  const bool qdiag = !__any(qoff);
{qd_decl}
  const int64_t tiles = (n + FPWR - 1) / FPWR;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * FPWR;
    const int cnt = (n - base) < FPWR ? (int)(n - base) : FPWR;
    const int gg = g < cnt ? g : 0;
    const bool live = g < cnt;
{z_tile_a}    double* sP = s_P + gg * {EE};
    double* sl = s_sl + gg * SLOT_R;
{z_tile_b}{decl_idx}
    int lb = lane;
    asm volatile("" : "+v"(lb));         // opaque copy of the lane index: the copies' index arithmetic stays inside the tile
    rn::copy_g2l<FPWR * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lb);
    for (int i = lane; i < cnt * {D}; i += 64) s_sl[(i / {D}) * SLOT_R + {lay.OFF_X} + i % {D}] = gx[base * {D} + i];
{z_first}
    rn::wave_lds_sync();
{decl_rows}
{load_rows}
{pend_decl}    for (int64_t t = 0; t < T; t++) {{
{z_next}
      const int kind = kinds[t];
      const double dt = dts[t];
      const bool do_pred = {id0_guard};
      {TL(0)}
      // ---- phase 1a / 2a: predict ----
      if (c == 0 && live) {{
        if (do_pred) scal_predict_r(sl + {lay.OFF_X}, dt, sl, norm_quats);
        else scal_keep_r(sl + {lay.OFF_X}, sl, norm_quats);                 // predict(dt = 0) still renormalises
      }}
      rn::wave_lds_sync();
      {TL(1)}
      if (do_pred) {{
{flush_before_predict}        if (qdiag) {{
          predict_rows_qd({rows}, sP, {qd_args}, sl, {idx}{_tl_arg(True)});
        }} else {{
          int qz = 0;
          asm volatile("" : "+v"(qz));                 // Q behind an opaque zero: its addresses are not worth registers across the step loop
          predict_rows({rows}, sP, gQ + qz, sl, {idx}{_tl_arg(True)});
        }}
      }}
      {TL(2)}
      // ---- phase 1b / 2b: update ----
      int bad = 0;
{scal_phase}
      bad = __builtin_amdgcn_readfirstlane(__any(bad) ? 8 : 0);
      rn::wave_lds_sync();
      {TL(3)}
      if (!bad) {{
        switch (kind) {{
{nlc.join(mat_cases)}
          default: break;
        }}
      }}
      {TL(4)}
      // ---- phase 3: lane 0 of each group injects the error state ----
      if (c == 0 && live) {{
        int fl = bad;
        if (!bad) fl = scal_inject_r(sl, sl + {lay.OFF_X}, norm_quats) | (int)sl[{lay.OFF_FL}];
        if (flags != nullptr) flags[t * n + base + g] = (uint8_t)fl;
      }}
      rn::wave_lds_sync();
      {TL(5)}
{z_out}
      int lz = lane;
      asm volatile("" : "+v"(lz));       // the copies' per-iteration indices are not worth registers across the step loop
      if (tx != nullptr) {{
        for (int i = lz; i < cnt * {D}; i += 64) tx[(t * n + base) * {D} + i] = s_sl[(i / {D}) * SLOT_R + {lay.OFF_X} + i % {D}];
      }}
      if (tP != nullptr) {{
{trace_store}
      }}
      rn::wave_lds_sync();
      {TL(6)}
{z_commit}
      rn::wave_lds_sync();
      {TL(7)}{aug}
    }}
{flush_after_loop}{img}
    rn::wave_lds_sync();
    int le = lane;
    asm volatile("" : "+v"(le));         // (same: nothing of the first copy's index arithmetic is kept alive across the step loop)
    rn::copy_l2g<FPWR * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, le);
    for (int i = le; i < cnt * {D}; i += 64) gx[base * {D} + i] = s_sl[(i / {D}) * SLOT_R + {lay.OFF_X} + i % {D}];
    rn::wave_lds_sync();
  }}
}}
"""


def launch_run():
  return """  const int64_t tiles = (n + FPWR - 1) / FPWR;
  hipLaunchKernelGGL(k_run, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, trace_x, trace_P, ea, augment);"""
