"""Kernel family W, smoother `k_rts4`: the RTS backward pass with EVERY BROADCAST OPERAND TAKEN FROM REGISTERS (`row_newbcast`).

Reference: the Python-only `EKF_sym.rts_smooth` (/root/reference/rednose/helpers/ekf_sym.py:651-690); per backward step k
    Fk = F(xk_k, t[k+1] - t[k]);  Ck = solve(Pk1_k, Fk Pk_k^T)^T;  xk_n = err(xk_k, Ck inv_err(xk1_k, xk1_n));
    Pk_n = Pk_k + Ck (Pk1_n - Pk1_k) Ck^T
with the predicted pair (xk1_k, Pk1_k) recomputed from the filtered one (templates/ekf_hip_rts.h explains the recursion's quirks,
which are kept: start from the PREDICTED pair of the last step, in-place renormalisation of xk1_n).

Why this layout.  Its predecessor `k_rts3` (the fused run's layout: 8 lanes x 3 rows per filter; deleted in round 6) ran ONE wavefront per
SIMD (two row sets of 132 registers each + an LDS image per filter) and took every operand another lane owns through LDS: one 8-byte
broadcast read feeds three FMAs, four wavefronts share a CU's LDS, and a lone wavefront issues an fp64 instruction every ~10 cycles (a
dependent one every ~27-40): 0.25 of the HBM roofline for two rounds, whatever was moved inside it (profiles/tuning_notes.md).  CDNA3/4 have exactly one cross-lane form for fp64
arithmetic: `v_fmac_f64_dpp ... row_newbcast:L` -- lane L of every 16-lane row feeds all 16 lanes of that row, at the plain FMA's
issue rate (tools/dpp_probe.hip: 5.4-5.8 cycles per instruction with two wavefronts per SIMD, 6.5 alone; semantics checked there).
So here a filter is ONE 16-LANE ROW (4 filters per wavefront), row r of every matrix lives in slot r // RS of lane r % RS (RS = rows per
slot: 22 error states = 2 slots x 11 lanes), and
  * the right-looking L D L^T factorisation, both substitutions and both E^3 products read the other lanes' rows straight out of
    their registers: no LDS broadcast, no LDS round trip inside a dependent chain, 22 independent accumulation chains per pass;
  * the smoothed covariance of step k + 1 is CARRIED in registers (k_rts3 stored it and read it back: +11 % traffic, a wait for
    the previous step's stores in the middle of every step);
  * at most two row sets (2 x 88 registers for 22 error states) + one half set are live at any point, the third matrix of each
    phase waits in the filter's single E x E image of LDS -- 19.5 KB per wavefront for live: EIGHT wavefronts per CU, two per
    SIMD, which is what hides the dependent chains (pivot reciprocals, the one-lane scalar phase) that a lone wavefront exposes.
The price: 22 of 32 row slots busy (k_rts3: 22 of 24), i.e. ~1.15 x the vector instructions per filter-step.

  step k (image I per filter; register row sets in capitals):
    Pk_k HBM -> I (one coalesced asynchronous burst); lead lanes evaluate f / F non-zeros (scal_predict) meanwhile
    rows of Pk_k <- I (lower triangle mirrored: the batch_rts contract);  rows of A = Pk_k Fk^T (row-local, sparse) -> I
    Pp <- rows of Pk1_k = (columns of A) Fk^T + dt Q;   D <- PS - Pp  (PS: smoothed covariance of step k + 1, carried)
    Pp <- its L D L^T factor, right-looking: column j scaled by the broadcast pivot's reciprocal, trailing columns updated with
          v_fmac_f64_dpp (unscaled entry of row m from lane m, own scaled entry as the other factor)
    per row slot: Y <- row of A from I; forward substitution, scaling, backward substitution (axpy form, 21 .. 1 independent
          FMAs per pivot); Y = row of Ck -> I
    state: delta = Ck inv_err(xk1_k, xk1_n), xk_n = err(xk_k, delta)  (lead lanes + one E-term dot per row)
    T <- Ck D per row slot (coefficients: the lane's own row of Ck from I; operands: rows of D by row_newbcast)
    CK <- I;  U = T Ck^T per row slot (operands: rows of Ck by row_newbcast); symmetric: slot 0 forms columns 0 .. 15 only
    I <- U (mirrored);  Pk_n = Pk_k + U: one coalesced read-add-write over the tile's records, the sum also goes back into I
    PS <- rows of I
  step k with t[k + 1] == t[k] (models whose predict(0) is the identity; not the recursion's first step): Ck = I, see dt0_path() / kernel():
    Pk_k HBM -> I;  xk_n = err(xk_k, inv_err(xk_k, xk1_n));  D <- PS - lower(Pk_k);  Pk_n = Pk_k + D row by row in place;  I -> HBM;  PS <- rows of I
  `k_rts4_tri` (kernel(tri=True)): the same on a trace of PACKED lower triangles -- elements are scattered to / gathered from the images' lower
  positions through a byte table of row indices in LDS.

Generated for ordinary (non-MSCKF) lane-group models whose 4 images + vectors fit 20 KB of LDS (8 .. 22 error states, odd counts
included: live, kinematic9); every other lane-group model -- MSCKF, more states, a dense F whose non-zeros do not fit the slot --
keeps rn::k_rts_group, which is also the fallback (`no_rts4`).  (k_rts3, the round-4 smoother in the fused run's layout, served the
odd counts until round 6 and is gone.)
"""
from rednose_amd.codegen.emit_common import term, sum_terms

GL = 16            # lanes per filter: one DPP row
FPW = 4            # filters per wavefront
LDS_BUDGET = 20480  # bytes per wavefront: eight wavefronts on a CU's 160 KB


MACROS = r"""
// ---- fp64 cross-lane operands without LDS: lane L of every 16-lane row feeds the row (the only DPP form double-precision ALU
// instructions have on gfx90a / gfx94x / gfx950).  tools/dpp_probe.hip: semantics, and the same issue rate as a plain v_fma_f64.
// EXEC must be full where these execute (a disabled source lane is not a valid source): the kernel keeps them out of divergent code.
#ifndef RN4_FMAC
#define RN4_FMAC(acc, src, coef, L)  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef))
#define RN4_FNMAC(acc, src, coef, L) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef))
// (s_nop 1: a DPP read of a register the previous vector instruction wrote wants two wait states; inline asm is opaque to hipcc's hazard pass)
#define RN4_BC(dst, src, L)          asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
// RN4_FMAC / RN4_FNMAC carry no wait states (two per instruction would cost the products a third of their issue rate).  Their DPP sources are
// register rows that ordinary C statements wrote (the rows of Pk1_k, the factor's scaled columns): hipcc may sink such a statement to right in
// front of its first reader.  So at the head of every phase that reads a row set through DPP the rows are pinned (rn::pin: the writers cannot sink
// below) and RN4_SETTLE() puts the two wait states between them and the phase -- volatile asm statements keep their order.  What remains possible is
// a register copy hipcc's allocator places in front of a reader: rednose_amd.build.dpp_hazards checks every built k_rts4 for that.
#define RN4_SETTLE()                 asm volatile("s_nop 1")
#endif
"""


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def rows_per_lane(spec):
  return -(-spec.dim_err // GL)


def lds_bytes(spec, slot, tri=False):
  D, E = spec.dim_x, spec.dim_err
  base = 8 * (FPW * E * E + 2 + FPW * slot + 2 * (FPW * D + 2) + FPW * E + 2 + E + 2 * FPW * (GL - -(-E // rows_per_lane(spec))) + 2)
  return base + (8 * -(-(E * (E + 1) // 2) // 8) if tri else 0)      # k_rts4_tri: the row of every packed index, one byte each


def tri_applicable(spec):
  """k_rts4_tri (the packed-triangle trace, see kernel()) needs E (E + 1) / 2 more bytes of LDS under the same budget."""
  if not applicable(spec):
    return False
  lay, _ = _tables(spec)
  return lds_bytes(spec, lay.SLOT, tri=True) <= LDS_BUDGET


class RtsLayout:
  """Scalar slot of the smoother (doubles per filter): x' = f(x) [normalised], the non-trivial entries of F, dt."""

  def __init__(self, spec, f_vars, he_vars_by_kind):        # same constructor as emit_wide2.Layout / emit_wide3.RunLayout
    D = spec.dim_x
    self.zmax = max(k.zdim for k in spec.kinds)
    self.nf = len(f_vars)
    self.nh = 0
    self.OFF_X = 0
    self.OFF_F = D
    self.OFF_DT = D + self.nf
    n = self.OFF_DT + 1
    # fields the shared scalar functions of emit_wide2.device_functions address but the smoother never calls
    self.OFF_HE = self.OFF_DX = self.OFF_Y = self.OFF_FL = self.OFF_RF = self.OFF_RP = self.OFF_YP = n
    self.zf = 0
    self.SLOT = n + 1 - (n & 1)


def _tables(spec):
  from rednose_amd.codegen import emit_wide2 as w2
  _, _, F, f_vars = w2._lowered_predict(spec)                  # pylint: disable=protected-access
  lay = RtsLayout(spec, f_vars, {})
  return lay, w2._slotted(F, f_vars, lay.OFF_F)                # pylint: disable=protected-access


def _scal_text(spec):
  """scal_predict against RtsLayout under the suffix _s (only that function of emit_wide2.device_functions is used)."""
  from rednose_amd.codegen import emit_wide2 as w2
  text, lay = w2.device_functions(spec, lay_cls=RtsLayout, sfx="_s")
  # keep scal_predict_s only: the observation / injection functions address slot fields the smoother does not have
  keep = []
  for fn in text.split("\n\n"):
    if "void scal_predict_s(" in fn:
      keep.append(fn)
  assert len(keep) == 1
  return keep[0], lay


def dt0_path(spec):
  """Steps with dt == 0 take the identity-gain path (see kernel()): only for models whose predict(dt = 0) is the identity symbolically."""
  from rednose_amd.codegen import tuning
  return bool(tuning.current().rts_dt0) and spec.identity_at_dt0()


def applicable(spec):
  """Ordinary (non-MSCKF) lane-group models with 8 .. 32 error states whose four images + vectors fit LDS_BUDGET."""
  msckf = any(k.He_sym is not None for k in spec.kinds) or spec.N > 0
  if msckf or spec.dim_main_err != spec.dim_err:
    return False
  E = spec.dim_err
  if E < 8 or E > 2 * GL:
    return False
  lay, _ = _tables(spec)
  return lds_bytes(spec, lay.SLOT) <= LDS_BUDGET


def kernel(spec, tri=False):
  """tri=True: `k_rts4_tri` -- the same recursion on a trace whose covariances are PACKED lower triangles (row-major, E (E + 1) / 2 doubles:
  what batch_run_tri writes): Pf, Ps and Pl are (.., E (E + 1) / 2) arrays.  The kernel's matrix image stays E x E; a packed element goes to /
  comes from the image's LOWER position (row of a packed index: a byte table in LDS, filled once per launch), which is all the kernel reads
  of a covariance anyway (the contract of batch_rts) -- the upper-right exchange of the identity-gain step and the mirroring of U drop out."""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  TRI = E * (E + 1) // 2
  REC = TRI if tri else EE          # doubles per covariance record in memory
  ITT = -(-(FPW * TRI) // 64)       # packed elements per lane of a tile
  kname = "k_rts4_tri" if tri else "k_rts4"
  R = rows_per_lane(spec)
  S = range(R)
  RS = -(-E // R)        # rows per slot: row r lives in slot r // RS of lane r % RS (22 states: 2 x 11 -- balanced slots keep the block lower
                         # triangles of the symmetric matrices at 11 + 22 columns per lane instead of 16 + 22, and lanes RS .. 15 idle)
  scal, lay = _scal_text(spec)
  _, Fs = _tables(spec)
  scal = scal.replace("scal_predict_s(", "scal_predict_s4(")
  quat = "".join(f" rn::normalize_quat<{D}>(xv, {q});" for q in spec.quaternion_idxs)
  b = []
  # Issue priority: the two E^3 products (stamps 7 .. 9) are straight FMA streams that can run any time; everything else in a step is a chain of
  # dependent latencies (LDS round trips, the pivots' reciprocals, the one-lane scalar phase, memory).  With the products at priority 0 and the
  # rest at 3 the co-resident wavefront's chains issue ahead of this one's FMA stream: 8.31 -> 7.73 ms on 8 192 x 300 steps, same call
  # (other windows measured: products + substitutions low 8.07, substitutions only 8.27, levels 1 / 2 instead of 3: 7.85; tools/rts4_time.py).
  PRIO = (7, 9, 3)

  def A(line):
    b.append(line)
    if PRIO and line.strip().startswith("RN_RTS_STAMP("):
      st = int(line.strip()[len("RN_RTS_STAMP("):].split(")")[0])
      if st == PRIO[0]:
        b.append("      __builtin_amdgcn_s_setprio(0);")
      elif st == PRIO[1]:
        b.append(f"      __builtin_amdgcn_s_setprio({PRIO[2]});")

  def slot_of(r):
    return r // RS

  def lane_of(r):
    return r % RS

  def last_row(s):       # last row index a slot can hold
    return min(E, RS * s + RS) - 1

  def ncol(s):           # columns a row slot keeps of a SYMMETRIC matrix: up to its last row (block lower triangle)
    return last_row(s) + 1

  def settle(name, ind="      "):
    """the block-lower rows `name` are about to be read through DPP: see RN4_SETTLE"""
    for s_ in S:
      A("#pragma unroll")
      A(f"{ind}for (int j = 0; j < {ncol(s_)}; j++) rn::pin({name}{s_}[j]);")
    A(f"{ind}RN4_SETTLE();")

  def rows_decl(name, sym_=False):
    return " ".join(f"double {name}{s}[{ncol(s) if sym_ else E}];" for s in S)

  def lower_rows(name, ind="      ", src="sI", rc="rq", full=True):
    """rows of a symmetric matrix from the LOWER triangle of the image: entry j of row r is M[r][j] for j <= r, M[j][r] above
    (full=False: only the columns up to the slot's last row)."""
    for s in S:
      lo, hi = RS * s, min(E, RS * s + RS)
      if lo:
        A("#pragma unroll")
        A(f"{ind}for (int j = 0; j < {lo}; j++) {name}{s}[j] = {src}[{rc}{s} * {E} + j];")
      A("#pragma unroll")
      A(f"{ind}for (int j = {lo}; j < {hi}; j++) {name}{s}[j] = {src}[{E} * max({rc}{s}, j) + min({rc}{s}, j)];      // (max / min / mad: no compare-select pair per entry)")
      if hi < E and full:
        A("#pragma unroll")
        A(f"{ind}for (int j = {hi}; j < {E}; j++) {name}{s}[j] = {src}[j * {E} + {rc}{s}];")

  A(f"// ---- smoother, operands by row_newbcast: {GL} lanes x {R} rows per filter, {FPW} filters per wavefront (emit_rts4.py){' -- covariances as packed lower triangles' if tri else ''} ----")
  if not tri:
    A(MACROS)
    A(f"constexpr int RTS4_SLOT = {lay.SLOT};")
    A(scal)

  def tri_off(idx, ind):
    """C lines: image offset `o_` (filter f_, lower position of packed index p_) of the tile's packed element `idx`"""
    return [f"{ind}const int f_ = ({idx}) / {TRI}; const int p_ = ({idx}) - f_ * {TRI}; const int r_ = s_tr[p_]; const int o_ = f_ * {EE} + r_ * {E} + p_ - (r_ * (r_ + 1)) / 2;"]
  tri_table = (f"""  __shared__ unsigned char s_tr[{-(-TRI // 8) * 8}];      // row of every packed index (p = r (r + 1) / 2 + j, j <= r)
  for (int p = lane; p < {TRI}; p += 64) {{ int r = 0; while ((r + 1) * (r + 2) / 2 <= p) r++; s_tr[p] = (unsigned char)r; }}
  rn::wave_lds_sync();
""" if tri else "")
  qd_decl = "\n".join(f"  const double qd{s} = gQ[((c < {RS} && (c + {RS * s}) < {E}) ? (c + {RS * s}) : 0) * {E + 1}];" for s in S)
  A(f"""
__global__ __launch_bounds__(64, 2) void {kname}(const double* __restrict__ xf, const double* __restrict__ Pf, const double* __restrict__ ts,
    const int64_t T, const double* __restrict__ gQ, const int64_t n, const int norm_quats, double* __restrict__ xs,
    double* __restrict__ Ps, const double* __restrict__ xl, const double* __restrict__ Pl) {{
  __shared__ __attribute__((aligned(16))) double s_I[{FPW} * {EE} + 2];          // the one matrix image per filter (see emit_rts4.py)
  __shared__ __attribute__((aligned(16))) double s_sl[{FPW} * RTS4_SLOT];        // x' = f(xk_k), F non-zeros, dt
  __shared__ __attribute__((aligned(16))) double s_xk[{FPW} * {D} + 2];          // xk_k
  __shared__ __attribute__((aligned(16))) double s_xn[{FPW} * {D} + 2];          // xk1_n, then xk_n
  __shared__ __attribute__((aligned(16))) double s_dv[{FPW} * {E} + 2];          // inv_err(xk1_k, xk1_n), then Ck delta
  __shared__ __attribute__((aligned(16))) double s_trash[{E} + 2 * {FPW * (GL - RS)} + 2];      // where the idle lanes' row stores go (no predicated regions around LDS stores): overlapping rows, 16 bytes apart
  const int lane = threadIdx.x;
{tri_table}  int qoff = 0;
  for (int i = lane; i < {EE}; i += 64) qoff |= (i / {E} != i % {E}) && (gQ[i] != 0.0);
  const bool qdiag = !__any(qoff);      // a diagonal process noise (the usual case): its row entry is requested at the head of every step
  const int64_t tiles = (n + {FPW} - 1) / {FPW};
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * {FPW};
    const int cnt = (n - base) < {FPW} ? (int)(n - base) : {FPW};
    int lt = lane;      // opaque per tile: index arithmetic derived from it is not hoisted to the kernel's entry (where it was ~80 spilled registers)
    asm volatile("" : "+v"(lt));
    const int g = lt / {GL};
    const int c = lt % {GL};
    const bool live = g < cnt;
    const int gg = live ? g : 0;
    const bool lead = live && c == 0;
    double* sI = s_I + gg * {EE};
    double* sl = s_sl + gg * RTS4_SLOT;
    double* sxk = s_xk + gg * {D};
    double* sxn = s_xn + gg * {D};
    double* sdv = s_dv + gg * {E};
    int ct = c;
    asm volatile("" : "+v"(ct));""")
  for s in S:
    A(f"    const int rr{s} = ct + {RS * s}; const bool ok{s} = live && ct < {RS} && rr{s} < {E}; const int rc{s} = (ct < {RS} && rr{s} < {E}) ? rr{s} : 0;")
  A("    if (T == 1) {      // nothing to smooth, the single estimate's predicted pair is not available: the filtered pair passes through")
  A(f"      if (Ps != Pf) {{ for (int i = lt; i < cnt * {REC}; i += 64) Ps[base * {REC} + i] = Pf[base * {REC} + i]; }}")
  A(f"      if (xs != xf) {{ for (int i = lt; i < cnt * {D}; i += 64) xs[base * {D} + i] = xf[base * {D} + i]; }}")
  A("      continue;")
  A("    }")
  XT = -(-(FPW * D) // 64)
  IT = -(-(FPW * EE // 2) // 64)
  ITF = (FPW * EE // 2) // 64          # iterations of the coalesced passes that are whole for a full tile
  A("    // xk_k of the first step to process; every later one is committed to LDS by the step before it (nothing read from memory crosses the")
  A("    // loop's back edge in registers: hipcc waits for such a value with vmcnt(0) at the loop header -- behind the previous step's stores)")
  A(f"    for (int i = lt; i < cnt * {D}; i += 64) s_xk[i] = xf[((T - 2) * n + base) * {D} + i];")
  A(f"    {rows_decl('ps', True)}      // rows of the smoothed covariance of step k + 1 (block lower triangle), carried from step to step; D = Pk1_n - Pk1_k inside a step")
  for s in S:
    A("#pragma unroll")
    A(f"    for (int j = 0; j < {ncol(s)}; j++) ps{s}[j] = 0.0;")
  A("    if (Pl != nullptr) {      // newest smoothed covariance := the predicted one of the last step as passed in (ekf_sym.py:658-659): through the image,")
  A("      // like every later step's (lower triangle mirrored), and out to Ps[T - 1]")
  if tri:
    A(f"      for (int i = lt; i < cnt * {TRI}; i += 64) {{")
    A(f"        const double v_ = Pl[base * {TRI} + i];")
    A(f"        Ps[((T - 1) * n + base) * {TRI} + i] = v_;")
    b.extend(tri_off("i", "        "))
    A("        s_I[o_] = v_;")
    A("      }")
  else:
    A(f"      rn::async_copy_g2l<{FPW} * {EE}>(Pl + base * {EE}, cnt * {EE}, s_I, lt);")
    A(f"      for (int i = lt; i < cnt * {EE}; i += 64) Ps[((T - 1) * n + base) * {EE} + i] = Pl[base * {EE} + i];")
    A("      rn::async_wait();")
  A("      rn::wave_lds_sync();")
  lower_rows("ps", ind="      ", rc="rc", full=False)
  A("      rn::wave_lds_sync();")
  A("    }")
  A("    double dtc = ts[T - 1] - ts[T - 2];      // the step's time difference, read one step ahead (a scalar load's latency at the head of every step otherwise)")
  if PRIO:
    A(f"    __builtin_amdgcn_s_setprio({PRIO[2]});")
  A("    for (int64_t k = T - 2; k >= 0; k--) {")
  A("      const bool first = (k == T - 2);")
  A("      int lb = lt;")
  A('      asm volatile("" : "+v"(lb));         // opaque copy of the lane index: the copies\' index arithmetic stays inside the step')
  A("      int " + ", ".join(f"rq{s} = rc{s}" for s in S) + ";      // (same for the row indices: hoisted out of the step loop, the LDS addresses they feed were ~150 spilled registers)")
  A('      asm volatile("" : ' + ", ".join(f'"+v"(rq{s})' for s in S) + ");")
  A(f"      double* strash = s_trash + 2 * ((g * {GL - RS} + (ct >= {RS} ? ct - {RS} : ct)) % {max(1, FPW * (GL - RS))});      // (an idle lane's own 16 bytes of every row store: stores of many lanes to ONE address serialise)")
  for s in S:
    A(f"      double* sw{s} = ok{s} ? sI + rq{s} * {E} : strash;      // the lane's row of the image as a store target (idle lanes: a trash row)")
  for s in S:
    A(f"      const double qd{s} = gQ[rq{s} * {E + 1}];      // (not kept across steps: four registers of a kernel that has none to spare)")
  A("      RN_RTS_STAMP(0);")
  A("      // ---- A. filtered pair of step k: Pk_k -> image in one coalesced burst, xk_k -> LDS ----")
  if tri:
    A(f"      double wt[{ITT}];      // the tile's packed triangles, requested now, scattered to the lower positions of the images after the scalar phase")
    A(f"      const double* __restrict__ gpt = Pf + (k * n + base) * {TRI};")
    A("#pragma unroll")
    A(f"      for (int it = 0; it < {ITT}; it++) {{ const int idx = lb + 64 * it; wt[it] = gpt[idx < cnt * {TRI} ? idx : cnt * {TRI} - 1]; }}")
    A("      if (false) {")
  elif EE % 2:
    # Records of an odd number of doubles: the tile of step k starts at (k n + base) E^2 doubles, 16-byte aligned only when k n is even.  The
    # direct HBM -> LDS transfer wants 16-byte aligned global addresses; ordinary 16-byte vector loads do not (dword alignment is enough on the
    # device), so these models stage the tile through registers -- the copy is not hidden under the scalar phase (the test models and kinematic9).
    A(f"      rn::copy_g2l<{FPW} * {EE}>(Pf + (k * n + base) * {EE}, cnt * {EE}, s_I, lb);")
    A("      if (false) {")
  else:
    A(f"      if (cnt == {FPW}) {{      // (full tile: unguarded transfers -- sixteen predicated regions otherwise)")
  A(f"        const double* __restrict__ gp = Pf + (k * n + base) * {EE};")
  A("#pragma unroll")
  A(f"        for (int it = 0; it < {IT}; it++) {{")
  A(f"          if (it < {ITF} || lb + 64 * it < {FPW * EE // 2})")
  A("            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 2 * (lb + 64 * it)), rn::lds_offset_ptr(s_I + 2 * 64 * it), 16, 0, 0);")
  A("        }")
  A("      } else {")
  A(f"        rn::async_copy_g2l<{FPW} * {EE}>(Pf + (k * n + base) * {EE}, cnt * {EE}, s_I, lb);      // no register staging: lands under the scalar phase")
  A("      }")
  if EE % 2 or tri:      # (the two branches above are dead text for these models; dropped from the output)
    i_ = max(i for i, ln in enumerate(b) if ln.strip() == "if (false) {")
    del b[i_:]

  def tri_land(ind):
    """the packed tile requested at A lands in the images (lower positions)"""
    A("#pragma unroll")
    A(f"{ind}for (int it = 0; it < {ITT}; it++) {{")
    A(f"{ind}  const int idx = lb + 64 * it;")
    A(f"{ind}  if (idx < cnt * {TRI}) {{")
    b.extend(tri_off("idx", ind + "    "))
    A(f"{ind}    s_I[o_] = wt[it];")
    A(f"{ind}  }}")
    A(f"{ind}}}")
  A("      const double dt = dtc;")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(1);")
  if dt0_path(spec):
    # ---- steps with dt == 0 of a model whose predict(dt = 0) is the identity (FilterSpec.identity_at_dt0: f(x, 0) == x, F(x, 0) == I symbolically, and
    # dt Q = 0): the predicted pair of step k IS the filtered one, bit for bit -- what the full path computes there is Pk1_k = sym(Pk_k), A = Pk_k and
    # Ck = Pk1_k^-1-solve of itself = I up to the rounding of the factorisation (|Ck - I| <= 2.3e-9 in the reference's own golden, np.linalg.solve).
    # With Ck = I (ekf_sym.py:672-686):  xk_n = err(xk_k, inv_err(xk1_k, xk1_n))  -- NOT xk1_n: err o inv_err is not the identity for finite
    # rotations --,  Pk_n = Pk_k + (Pk1_n - Pk1_k)  evaluated as written.  No factorisation, no substitutions, no products: the step is one read and
    # one write of the covariance.  dt is a scalar of the launch (ts is shared by the batch): the branch is uniform.  The recursion's first step always
    # takes the full path (it also produces the newest pair).  Tuning knob rts_dt0 = 0 keeps the full solve on every step.
    A("      if (dt == 0.0 && !first) {")
    A("        if (lead && (norm_quats & 2)) {")
    A(f"          double xv[{D}];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) xv[i] = sxn[i];")
    A(f"         {quat}")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) sxn[i] = xv[i];")
    A("        }")
    A("        rn::wave_lds_sync();")
    A("        int lo = lb;")
    A('        asm volatile("" : "+v"(lo));')
    A(f"        for (int i = lo; i < cnt * {D}; i += 64) xs[((k + 1) * n + base) * {D} + i] = s_xn[i];      // smoothed state of step k + 1 (after its renormalisation)")
    A(f"        double xnext[{XT}];      // filtered state of the next (older) step")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {XT}; it++) {{ const int i = lo + 64 * it; xnext[it] = xf[((k > 0 ? k - 1 : 0) * n + base) * {D} + (i < cnt * {D} ? i : 0)]; }}")
    A("        dtc = k > 0 ? ts[k] - ts[k - 1] : 0.0;")
    A("        rn::wave_lds_sync();      // every lane has read xk1_n: the buffer takes xk_n")
    A("        if (lead) {      // xk1_k = f(xk_k, 0) = xk_k [renormalised like the forward pass]; delta = inv_err(xk1_k, xk1_n); xk_n = err(xk_k, delta)")
    A(f"          double xa[{D}], xv[{D}], x1n[{D}], xnew[{D}], delta[{E}];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) {{ xa[i] = sxk[i]; xv[i] = xa[i]; x1n[i] = sxn[i]; }}")
    A(f"          if (norm_quats & 1) {{{quat} }}")
    A("          inv_err_fun(xv, x1n, delta);")
    A("          err_fun(xa, delta, xnew);")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) sxn[i] = xnew[i];       // xk_n: becomes xk1_n of the next (older) step")
    A("        }")
    if tri:
      tri_land("        ")
    else:
      A("        rn::async_wait();")
    A("        rn::wave_lds_sync();      // Pk_k has landed; the lead lanes have read xk_k: the buffer takes the next step's")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {XT}; it++) {{ const int i = lo + 64 * it; if (i < cnt * {D}) s_xk[i] = xnext[it]; }}")
    if tri:
      # Packed trace: only lower triangles exist.  Row by row on the block lower triangle: pk = the entry of Pk_k (mirrored inside the diagonal
      # block), Pk_n = pk + (Pk1_n - pk) as written -- which IS the carried row of the next step, so nothing is read back.  The rows go to the
      # image (lower positions) for the coalesced packed store; every read of another row's entry precedes every write (fence).
      for s in S:
        lo_, hi_ = RS * s, min(E, RS * s + RS)
        A("        {")
        A(f"          double pk_[{ncol(s)}];")
        if lo_:
          A("#pragma unroll")
          A(f"          for (int j = 0; j < {lo_}; j++) pk_[j] = sI[rq{s} * {E} + j];")
        A("#pragma unroll")
        A(f"          for (int j = {lo_}; j < {hi_}; j++) pk_[j] = sI[{E} * max(rq{s}, j) + min(rq{s}, j)];")
        A("#pragma unroll")
        A(f"          for (int j = 0; j < {ncol(s)}; j++) {{ ps{s}[j] -= pk_[j]; ps{s}[j] = pk_[j] + ps{s}[j]; }}      // D = Pk1_n - Pk1_k, then Pk_n = Pk_k + D (Ck = I)")
        A("        }")
        A("        __builtin_amdgcn_sched_barrier(0);")
      A("        rn::wave_lds_sync();")
      for s in S:
        A("#pragma unroll")
        A(f"        for (int j = 0; j < {ncol(s)}; j++) sw{s}[j] = ps{s}[j];")
      A("        rn::wave_lds_sync();")
      A(f"        double* __restrict__ gpo = Ps + (k * n + base) * {TRI};")
      A("#pragma unroll")
      A(f"        for (int it = 0; it < {ITT}; it++) {{")
      A("          const int idx = lo + 64 * it;")
      A(f"          if (idx < cnt * {TRI}) {{")
      b.extend(tri_off("idx", "            "))
      A("            gpo[idx] = s_I[o_];")
      A("          }")
      A("        }")
      A("        rn::wave_lds_sync();      // the image is free for the next step's tile")
      A("        RN_RTS_STAMP(10);")
      A("        continue;")
      A("      }")
      A("      if (false) {      // (the full-matrix form of the identity-gain step follows in the emitter; dropped from this kernel's text)")
      tri_cut = len(b) - 1
    else:
      tri_cut = None
    # Covariance.  U = D = Pk1_n - Pk1_k with Pk1_k = the LOWER triangle of Pk_k mirrored (the contract of batch_rts); a lane holds D on the block
    # lower triangle of its rows (in place of the carried rows); what a row needs of D beyond its slot's last row belongs to the lanes of the later
    # slots, which put it where the row's own entries of that block were (the upper-right block of the image, read into registers first).  The sum
    # Pk_k + U is then formed row by row in place, with the entries of Pk_k as stored (above the diagonal too).
    for s in S:
      lo_, hi_ = RS * s, min(E, RS * s + RS)
      if lo_:
        A("#pragma unroll")
        A(f"        for (int j = 0; j < {lo_}; j++) ps{s}[j] -= sI[rq{s} * {E} + j];")
      A("#pragma unroll")
      A(f"        for (int j = {lo_}; j < {hi_}; j++) ps{s}[j] -= sI[{E} * max(rq{s}, j) + min(rq{s}, j)];")
      A("        __builtin_amdgcn_sched_barrier(0);")
    for s in S:
      if ncol(s) < E:
        A(f"        double pu{s}[{E - ncol(s)}];")
        A("#pragma unroll")
        A(f"        for (int j = {ncol(s)}; j < {E}; j++) pu{s}[j - {ncol(s)}] = sI[rq{s} * {E} + j];")
    if R > 1:
      A("        rn::wave_lds_sync();      // every lane has read what the exchange overwrites")
      for s in S:
        if RS * s:
          A(f"        {{ double* swc = ok{s} ? sI + rq{s} : strash; const int st = ok{s} ? {E} : 0;      // column rq{s} of the rows above the slot (idle lanes: one trash entry)")
          A("#pragma unroll")
          A(f"          for (int j = 0; j < {RS * s}; j++) swc[j * st] = ps{s}[j]; }}")
      A("        rn::wave_lds_sync();")
    else:
      A("        rn::wave_lds_sync();      // every lane has read what it mirrors from other rows: the rows take the sums")
    for s in S:
      A("        {")
      A(f"          double o_[{E}];")
      A("#pragma unroll")
      A(f"          for (int j = 0; j < {E}; j++) o_[j] = sI[rq{s} * {E} + j];")
      A("#pragma unroll")
      A(f"          for (int j = 0; j < {ncol(s)}; j++) o_[j] += ps{s}[j];      // Pk_n = Pk_k + U, U = D (Ck = I)")
      if ncol(s) < E:
        A("#pragma unroll")
        A(f"          for (int j = {ncol(s)}; j < {E}; j++) o_[j] += pu{s}[j - {ncol(s)}];      // (the sum commutes: U arrived in the image, Pk_k waits in pu)")
      A("#pragma unroll")
      A(f"          for (int j = 0; j < {E}; j++) sw{s}[j] = o_[j];")
      A("        }")
      A("        __builtin_amdgcn_sched_barrier(0);")
    A("        rn::wave_lds_sync();")
    A("        {      // Pk_n leaves: one coalesced pass over the tile's records")
    A("          typedef double rts4_d2 __attribute__((ext_vector_type(2)));")
    A(f"          rts4_d2* __restrict__ out2 = reinterpret_cast<rts4_d2*>(Ps + (k * n + base) * {EE});")
    A(f"          const int nv = (cnt * {EE}) / 2;")
    A("#pragma unroll")
    A(f"          for (int it = 0; it < {IT}; it++) {{")
    A("            const int idx = lo + 64 * it;")
    A(f"            if ((cnt == {FPW} && it < {ITF}) || idx < nv) out2[idx] = *reinterpret_cast<const rts4_d2*>(s_I + 2 * idx);")
    A("          }")
    if EE % 2:
      A(f"          if (((cnt * {EE}) & 1) && lo == 0) Ps[(k * n + base) * {EE} + cnt * {EE} - 1] = s_I[cnt * {EE} - 1];      // (odd record length, ragged tile: the last double)")
    A("        }")
    A('        asm volatile("" : ' + ", ".join(f'"+v"(rq{s})' for s in S) + ");")
    lower_rows("ps", ind="        ", full=False)
    A("        rn::wave_lds_sync();      // every lane has its rows: the image is free for the next step's burst")
    A("        RN_RTS_STAMP(10);")
    A("        continue;")
    A("      }")
    if tri_cut is not None:
      del b[tri_cut:]
  A("      // ---- B. f(xk_k) [renormalised like the forward pass], non-zeros of Fk: once per filter -> slot ----")
  A("      if (lead) scal_predict_s4(sxk, dt, sl, norm_quats & 1);")
  if tri:
    tri_land("      ")
  A("      rn::async_wait();")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(2);")
  A("      {")
  A(f"        {rows_decl('pf')}      // rows of Pk_k (lower triangle mirrored: the contract of batch_rts, include/rednose_amd_filter.h)")
  lower_rows("pf", ind="        ")
  A("        rn::wave_lds_sync();      // every lane has its rows: the image takes A")
  A("        RN_RTS_STAMP(3);")
  A("        // ---- C. rows of A = Pk_k Fk^T (row-local, F's structural zeros cost nothing): the right-hand sides; they wait in the image,")
  A("        // whose columns are the rows of Fk Pk_k (P = P^T up to rounding, as in the fused run's predict) ----")
  # Both passes of the predict touch only the rows of Fk that have entries off the diagonal; their coefficients are read from the slot
  # in GROUPS, the next group requested before the current group's FMAs (read where it is used, every pair of FMAs waited for its own
  # LDS round trip: 3.5 us of a 24 us step; all 33 at once cost 66 registers under two row sets).
  busy = [j for j in range(E) if any(m != j or cf[0] != 'one' for m, cf in Fs.row_nz(j))]
  groups, cur, ncoef = [], [], 0
  for j in busy:
    nv_ = sum(1 for _, cf in Fs.row_nz(j) if cf[0] == 'var')
    if cur and ncoef + nv_ > 10:
      groups.append(cur)
      cur, ncoef = [], 0
    cur.append(j)
    ncoef += nv_
  if cur:
    groups.append(cur)

  def grouped(tag, ind, emit_row):
    """emit_row(j, reg_term) -> C lines for row j of Fk, with the coefficients of the groups software-pipelined one group ahead."""
    regname = {}

    def gl(grp):
      out = []
      for j in grp:
        for _, cf in Fs.row_nz(j):
          if cf[0] == 'var' and cf[1] not in regname:
            regname[cf[1]] = f"{tag}{len(regname)}"
            out.append(f"{ind}const double {regname[cf[1]]} = {cf[1]};")
      return out

    def reg_term(cf, operand):
      return f"{regname[cf[1]]}*{operand}" if cf[0] == 'var' else term(cf, operand)
    b.extend(gl(groups[0]) if groups else [])
    for gi, grp in enumerate(groups):
      if gi + 1 < len(groups):
        b.extend(gl(groups[gi + 1]))
      A(f"{ind}__builtin_amdgcn_sched_barrier(0);")
      for j in grp:
        b.extend(ind + ln for ln in emit_row(j, reg_term))
    A(f"{ind}__builtin_amdgcn_sched_barrier(0);")

  for s in S:      # the columns Fk leaves alone first: most of the row set is dead before the sums are formed
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {E}; j++) {{ if (" + (" && ".join(f"j != {jb}" for jb in busy) or "true") + f") sw{s}[j] = pf{s}[j]; }}")
  A("        __builtin_amdgcn_sched_barrier(0);")
  grouped("fp", "        ", lambda j, rt: [f"sw{s}[{j}] = {sum_terms(rt(cf, f'pf{s}[{kk}]') for kk, cf in Fs.row_nz(j))};" for s in S])
  A("      }")
  A("      rn::wave_lds_sync();")
  A(f"      {rows_decl('a', True)}      // rows of Pk1_k up to each slot's last row (symmetric: the block lower triangle is all anything reads), then its L D L^T factor (unit lower triangle, reciprocal pivots on the diagonal)")
  for s in S:      # second pass, one row slot at a time: rows of Pk1_k = (columns of A) Fk^T
    A("      {")
    A(f"        double col[{E}];")
    A("#pragma unroll")
    A(f"        for (int m = 0; m < {E}; m++) col[m] = sI[m * {E} + rq{s}];      // column of A = row of Fk Pk_k")
    grouped(f"fq{s}_", "        ", lambda j, rt, s=s: ([f"a{s}[{j}] = {sum_terms(rt(cf, f'col[{m}]') for m, cf in Fs.row_nz(j))};"] if j < ncol(s) else []))
    for j in range(ncol(s)):
      if j not in busy:
        A(f"        a{s}[{j}] = col[{j}];")
    A("      }")
  A("      if (qdiag) {")
  for s in S:
    A(f"        const double dq{s} = dt * qd{s};")
    for j in range(RS * s, min(E, RS * (s + 1))):
      A(f"        a{s}[{j}] += (rq{s} == {j} ? dq{s} : 0.0);")
  A("      } else {")
  for s in S:
    A("        {")
    A(f"          double q[{(ncol(s) + 1) // 2}];")
    H2 = (ncol(s) + 1) // 2
    for lo_, hi_ in ((0, H2), (H2, ncol(s))):
      A("#pragma unroll")
      A(f"          for (int j = {lo_}; j < {hi_}; j++) q[j - {lo_}] = gQ[rq{s} * {E} + j];")
      A("#pragma unroll")
      A(f"          for (int j = {lo_}; j < {hi_}; j++) a{s}[j] = fma(dt, q[j - {lo_}], a{s}[j]);")
    A("        }")
  A("      }")
  A("      RN_RTS_STAMP(4);")
  A("      // ---- D. recursion start / difference matrix (the image keeps A: the substitutions take their right-hand sides from it) ----")
  A("      if (first) {")
  A("        // newest estimate := the predicted pair of the last step (passed in, or recomputed just now): ekf_sym.py:658-659")
  A("        if (live) {       // (two loops, not one select between a global and an LDS source: hipcc 7.2 trips over the generic pointer)")
  A(f"          const int cf = lb % {GL};      // (from the step's opaque lane index: addresses formed from c are hoisted to the kernel's entry and held in registers for this one step)")
  A(f"          if (xl != nullptr) {{ for (int i = cf; i < {D}; i += {GL}) sxn[i] = xl[(base + lb / {GL}) * {D} + i]; }}")
  A(f"          else {{ for (int i = cf; i < {D}; i += {GL}) sxn[i] = sl[{lay.OFF_X} + i]; }}")
  A("        }")
  A("        if (Pl == nullptr) {      // (a covariance that was passed in went into ps* before the loop)")
  if tri:
    A(f"          double* __restrict__ po = Ps + ((k + 1) * n + base + gg) * {TRI};      // the recomputed Pk1_k leaves as its packed lower triangle (once per tile: predicated stores)")
    for s in S:
      A("#pragma unroll")
      A(f"          for (int j = 0; j < {ncol(s)}; j++) {{ if (ok{s} && j <= rr{s}) po[(rr{s} * (rr{s} + 1)) / 2 + j] = a{s}[j]; }}")
  else:
    A(f"          double* __restrict__ po = Ps + ((k + 1) * n + base + gg) * {EE};      // the recomputed Pk1_k leaves mirrored from the block lower triangle the lanes hold")
    for s in S:
      A("#pragma unroll")
      A(f"          for (int j = 0; j < {ncol(s)}; j++) {{ if (ok{s}) po[rr{s} * {E} + j] = a{s}[j]; }}")
      if RS * s:
        A("#pragma unroll")
        A(f"          for (int j = 0; j < {RS * s}; j++) {{ if (ok{s}) po[j * {E} + rr{s}] = a{s}[j]; }}")
  A("        }")
  A("      }")
  A("      rn::wave_lds_sync();")
  A("      if (lead && (norm_quats & 2)) {")
  A(f"        double xv[{D}];")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) xv[i] = sxn[i];")
  A(f"       {quat}")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) sxn[i] = xv[i];")
  A("      }")
  A("      rn::wave_lds_sync();")
  A("      {")
  A("        int lo = lb;")
  A('        asm volatile("" : "+v"(lo));')
  A(f"        for (int i = lo; i < cnt * {D}; i += 64) xs[((k + 1) * n + base) * {D} + i] = s_xn[i];      // smoothed state of step k + 1 (after its renormalisation)")
  A("      }")
  A("      // D = Pk1_n - Pk1_k as fma(-1, Pk1_k, Pk1_n) -- the subtraction, bit for bit.  When the recursion starts from the predicted pair it has just")
  A("      // recomputed, Pk1_n IS Pk1_k: the carried rows are still zero and the weight is 0, so D = 0 without a copy of the row set")
  A("      const double dw = (first && Pl == nullptr) ? 0.0 : -1.0;")
  A("      // state, first half: delta = inv_err(xk1_k, xk1_n) by the lead lanes (its chain runs under the factorisation of the other wavefront)")
  A(f"      if (lead) inv_err_fun(sl + {lay.OFF_X}, sxn, sdv);      // (straight from / to LDS: staged through arrays, the two states were 4 D registers next to two row sets)")
  for s in S:
    A("#pragma unroll")
    A(f"      for (int j = 0; j < {ncol(s)}; j++) {{ ps{s}[j] = fma(dw, a{s}[j], ps{s}[j]); rn::pin(ps{s}[j]); }}      // D = Pk1_n - Pk1_k, block lower triangle (the product takes D[kk][j] = D[j][kk] from whichever lane holds it).  (pinned: hipcc sinks the subtraction to its first use after the substitutions and keeps BOTH operands -- a copy of Pk1_k beside its factor -- alive until then)")
  A("      RN_RTS_STAMP(5);")
  A("      // ---- E. L D L^T of Pk1_k, right-looking, lower triangle.  Column j: the pivot comes from lane j by row_newbcast, every lane")
  A("      // forms its reciprocal (v_rcp_f64 + two Newton steps) and its scaled entries; the trailing update of column m reads the")
  A("      // UNSCALED entry (m, j) from lane m's register and multiplies with the lane's own scaled entry.  Column j + 1 is")
  A("      // completed first and its pivot's reciprocal chain started before the rest of column j's update is issued. ----")

  def upd_slots(m):          # slots that hold a row >= m
    return [s for s in S if last_row(s) >= m]

  settle("a")
  A("      double dj_0;")
  A(f"      RN4_BC(dj_0, a{slot_of(0)}[0], {lane_of(0)});")
  A("      double id_0 = rn::fast_recip(dj_0);")
  for s in S:
    if last_row(s) > 0:
      A(f"      double l{s}_0 = a{s}[0] * id_0;")
  for j in range(E):
    if j + 1 < E:
      m = j + 1
      for s in upd_slots(m):
        A(f"      RN4_FNMAC(a{s}[{m}], a{slot_of(m)}[{j}], l{s}_{j}, {lane_of(m)});")
      A(f"      double dj_{m};")
      A(f"      RN4_BC(dj_{m}, a{slot_of(m)}[{m}], {lane_of(m)});")
      A(f"      const double id_{m} = rn::fast_recip(dj_{m});")
      for m2 in range(j + 2, E):
        for s in upd_slots(m2):
          A(f"      RN4_FNMAC(a{s}[{m2}], a{slot_of(m2)}[{j}], l{s}_{j}, {lane_of(m2)});")
    for s in S:
      if last_row(s) >= j:
        if last_row(s) > j:
          A(f"      a{s}[{j}] = (rr{s} == {j}) ? id_{j} : l{s}_{j};")
        else:
          A(f"      a{s}[{j}] = id_{j};")
    if j + 1 < E:
      for s in S:
        if last_row(s) > j + 1:
          A(f"      const double l{s}_{j + 1} = a{s}[{j + 1}] * id_{j + 1};")
  settle("a")
  A("      RN_RTS_STAMP(6);")
  A("      // ---- F. Ck^T = Pk1_k^-1 M, one row slot at a time (its right-hand side = the lane's row of A, waiting in the image):")
  A("      // forward substitution with the unit factor, scaling by the reciprocal pivots, backward substitution, all in axpy form --")
  A("      // the factor's entry comes from its owner's register by row_newbcast, 21 .. 1 independent FMAs per pivot. ----")
  for s in S:
    A("      {")
    A(f"        double y[{E}];")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {E}; j++) y[j] = sI[rq{s} * {E} + j];")
    for m in range(E):
      for i in range(m + 1, E):
        A(f"        RN4_FNMAC(y[{i}], a{slot_of(i)}[{m}], y[{m}], {lane_of(i)});")
    for m in range(E):      # y[m] *= 1 / d_m as ONE multiply-add with the reciprocal pivot broadcast by the instruction itself (0 + (1 / d_m) y[m]: the product, bit for bit)
      A(f"        {{ double t_ = 0.0; RN4_FMAC(t_, a{slot_of(m)}[{m}], y[{m}], {lane_of(m)}); y[{m}] = t_; }}")
    for m in range(E - 1, 0, -1):
      for i in range(m - 1, -1, -1):
        A(f"        RN4_FNMAC(y[{i}], a{slot_of(m)}[{i}], y[{m}], {lane_of(m)});")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {E}; j++) sw{s}[j] = y[j];      // row of Ck (each lane overwrites the row it read; idle lanes: the trash row)")
    A("      }")
  A("      rn::wave_lds_sync();")
  A("      RN4_SETTLE();      // (the rows of D were pinned where they were formed)")
  A("      RN_RTS_STAMP(7);")
  A("      // ---- H. T = Ck D, one row slot at a time: coefficients = the lane's row of Ck, operands = rows of D from their owners ----")
  A(f"      {rows_decl('t')}")
  A("      double " + ", ".join(f"dx{s}" for s in S) + ";      // state, second half: Ck delta (one E-term dot per row), formed while the row of Ck is in registers anyway")
  for s in S:
    A("      {")
    A(f"        double ck[{E}];")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {E}; j++) ck[j] = sI[rq{s} * {E} + j];")
    A(f"        double dxa{s} = 0.0, dxb{s} = 0.0;")
    for c0_ in range(0, E, 8):      # (delta in pieces of eight: all E at once are 2 E registers next to three row sets)
      c1_ = min(E, c0_ + 8)
      A("        {")
      A(f"          double de[{c1_ - c0_}];")
      A("#pragma unroll")
      A(f"          for (int j = {c0_}; j < {c1_}; j++) de[j - {c0_}] = sdv[j];")
      for j in range(c0_, c1_):
        A(f"          dx{'ab'[j & 1]}{s} = fma(ck[{j}], de[{j - c0_}], dx{'ab'[j & 1]}{s});")
      A(f"          rn::pin(dxa{s}); rn::pin(dxb{s});")
      A("        }")
    A(f"        dx{s} = dxa{s} + dxb{s};")
    A(f"        rn::pin(dx{s});")
    A("        rn::wave_lds_sync();      // (compiler fence: delta is re-read by the next slot instead of staying in 2 E registers under this slot's product)")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {E}; j++) t{s}[j] = 0.0;")
    for kk in range(E):
      for j in range(E):
        if j < ncol(slot_of(kk)):
          A(f"        RN4_FMAC(t{s}[{j}], ps{slot_of(kk)}[{j}], ck[{kk}], {lane_of(kk)});")
        else:            # D[kk][j] above the block triangle of row kk: D[j][kk] from row j's lane
          A(f"        RN4_FMAC(t{s}[{j}], ps{slot_of(j)}[{kk}], ck[{kk}], {lane_of(j)});")
    A("      }")
  A("      RN_RTS_STAMP(8);")
  A("      // ---- I. U = T Ck^T: operands = rows of Ck from their owners' registers.  U is symmetric (D is, up to rounding): a row slot")
  A("      // forms the columns up to its last row, the upper-right block is mirrored inside the image ----")
  A(f"      {rows_decl('ck')}")
  for s in S:
    A("#pragma unroll")
    A(f"      for (int j = 0; j < {E}; j++) ck{s}[j] = sI[rq{s} * {E} + j];")
  A("      rn::wave_lds_sync();      // every lane has read delta and has its rows of Ck: the buffer takes Ck delta, the image U")
  for s in S:
    A(f"      *(ok{s} ? sdv + rq{s} : strash) = dx{s};")
  A("      typedef double rts4_d2 __attribute__((ext_vector_type(2)));")
  A("      int le = lb;")
  A('      asm volatile("" : "+v"(le));')
  if tri:
    A(f"      const double* __restrict__ in1 = Pf + (k * n + base) * {TRI};")
    A(f"      double* __restrict__ out1 = Ps + (k * n + base) * {TRI};")
    A(f"      double v[{ITT}];")
  else:
    A(f"      const rts4_d2* __restrict__ in2 = reinterpret_cast<const rts4_d2*>(Pf + (k * n + base) * {EE});")
    A(f"      rts4_d2* __restrict__ out2 = reinterpret_cast<rts4_d2*>(Ps + (k * n + base) * {EE});")
    A(f"      const int nv = (cnt * {EE}) / 2;")
    A(f"      rts4_d2 v[{IT}];")
  A(f"      double xnext[{XT}];      // filtered state of the next (older) step")
  # U in column blocks of one slot's rows each, ordered so that row sets die early: a block of columns [RS q, RS q + RS) broadcasts only
  # slot q's rows of Ck.  Last slot first, its own (diagonal) block first: after it slot R - 1's rows of Ck are dead; the last block of
  # all is slot 0's, under which the tile's filtered records for the final read-add-write are requested (below).
  blocks = [(s_, q) for s_ in reversed(S) for q in reversed(range(s_ + 1))]
  for bi, (s, q) in enumerate(blocks):
    c0, c1 = RS * q, min(E, RS * q + RS)
    if bi == len(blocks) - 1:
      # The tile's filtered records for the final read-add-write are requested HERE, in front of the last product block: the other
      # slots' rows of T and of Ck are dead, their registers take the loads, and the HBM / Infinity Cache round trip passes under the
      # block's FMAs instead of being waited for after them.
      if tri:
        A("#pragma unroll")
        A(f"      for (int it = 0; it < {ITT}; it++) {{ const int idx = le + 64 * it; v[it] = in1[idx < cnt * {TRI} ? idx : cnt * {TRI} - 1]; }}")
      else:
        A(f"      if (cnt == {FPW}) {{")
        A("#pragma unroll")
        A(f"        for (int it = 0; it < {IT}; it++) {{ const int idx = le + 64 * it; v[it] = in2[(it < {ITF} || idx < {FPW * EE // 2}) ? idx : {FPW * EE // 2 - 1}]; }}")
        A("      } else {")
        A("#pragma unroll")
        A(f"        for (int it = 0; it < {IT}; it++) {{ const int idx = le + 64 * it; v[it] = in2[idx < nv ? idx : nv - 1]; }}")
        A("      }")
      A("#pragma unroll")
      A(f"      for (int it = 0; it < {XT}; it++) {{ const int i = le + 64 * it; xnext[it] = xf[((k > 0 ? k - 1 : 0) * n + base) * {D} + (i < cnt * {D} ? i : 0)]; }}")
      A("      __builtin_amdgcn_sched_barrier(0);")
    A("      {")
    A(f"        double u[{c1 - c0}];")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {c1 - c0}; j++) u[j] = 0.0;")
    for kk in range(E):
      for j in range(c0, c1):
        A(f"        RN4_FMAC(u[{j - c0}], ck{q}[{kk}], t{s}[{kk}], {lane_of(j)});")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {c1 - c0}; j++) sw{s}[{c0} + j] = u[j];")
    A("      }")
    A("      __builtin_amdgcn_sched_barrier(0);")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(9);")
  A("      // ---- J. Pk_n = Pk_k + U leaves: one coalesced read-add-write over the tile's records; the sum also returns to the image,")
  A("      // from which every lane takes its rows of the smoothed covariance for the next (older) step ----")
  if any(last_row(s) + 1 < E for s in S) and not tri:      # (packed output: only lower positions are read)
    A("      {      // upper-right blocks: U[r][j] = U[j][r] for the columns beyond a slot's last row")
    for s in S:
      nc_ = last_row(s) + 1
      if nc_ < E:
        A("        {")
        A(f"          double m_[{E - nc_}];")
        A("#pragma unroll")
        A(f"          for (int j = {nc_}; j < {E}; j++) m_[j - {nc_}] = sI[j * {E} + rq{s}];")
        A("#pragma unroll")
        A(f"          for (int j = {nc_}; j < {E}; j++) sw{s}[j] = m_[j - {nc_}];")
        A("        }")
    A("      }")
    A("      rn::wave_lds_sync();")
  A("      if (lead) {      // state update, last part: one lane per filter, while the loads above are in flight")
  A(f"        double xa[{D}], xnew[{D}], delta[{E}];")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) xa[i] = sxk[i];")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {E}; i++) delta[i] = sdv[i];")
  A("        err_fun(xa, delta, xnew);")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) sxn[i] = xnew[i];       // xk_n: becomes xk1_n of the next (older) step")
  A("      }")
  A("      dtc = k > 0 ? ts[k] - ts[k - 1] : 0.0;")
  A("      rn::wave_lds_sync();      // the lead lanes have read xk_k: the buffer takes the next step's")
  A("#pragma unroll")
  A(f"      for (int it = 0; it < {XT}; it++) {{ const int i = le + 64 * it; if (i < cnt * {D}) s_xk[i] = xnext[it]; }}")
  if tri:
    A("#pragma unroll")
    A(f"      for (int it = 0; it < {ITT}; it++) {{")
    A("        const int idx = le + 64 * it;")
    A(f"        if (idx < cnt * {TRI}) {{")
    b.extend(tri_off("idx", "          "))
    A("          const double w_ = v[it] + s_I[o_];")
    A("          out1[idx] = w_;")
    A("          s_I[o_] = w_;")
    A("        }")
    A("      }")
    A("      if (false) {")
    tri_cut2 = len(b) - 1
  else:
    tri_cut2 = None
  A(f"      if (cnt == {FPW}) {{      // full tile: the first {ITF} passes of the wavefront are whole")
  A("#pragma unroll")
  A(f"        for (int it = 0; it < {IT}; it++) {{")
  A("          const int idx = le + 64 * it;")
  A(f"          if (it < {ITF} || idx < {FPW * EE // 2}) {{")
  A("            rts4_d2* im = reinterpret_cast<rts4_d2*>(s_I + 2 * idx);")
  A("            const rts4_d2 w_ = v[it] + *im;")
  A("            out2[idx] = w_;")
  A("            *im = w_;")
  A("          }")
  A("        }")
  A("      } else {")
  A("#pragma unroll")
  A(f"        for (int it = 0; it < {IT}; it++) {{")
  A("          const int idx = le + 64 * it;")
  A("          if (idx < nv) {")
  A("            rts4_d2* im = reinterpret_cast<rts4_d2*>(s_I + 2 * idx);")
  A("            const rts4_d2 w_ = v[it] + *im;")
  A("            out2[idx] = w_;")
  A("            *im = w_;")
  A("          }")
  A("        }")
  if EE % 2:
    A(f"        if (((cnt * {EE}) & 1) && le == 0) {{      // odd record length, ragged tile: the last double")
    A(f"          const int e_ = cnt * {EE} - 1;")
    A(f"          const double w_ = Pf[(k * n + base) * {EE} + e_] + s_I[e_];")
    A(f"          Ps[(k * n + base) * {EE} + e_] = w_;")
    A("          s_I[e_] = w_;")
    A("        }")
  A("      }")
  if tri_cut2 is not None:      # (the full-matrix read-add-write above is dead text for the packed kernel)
    del b[tri_cut2:]
  A("      rn::wave_lds_sync();")
  A('      asm volatile("" : ' + ", ".join(f'"+v"(rq{s})' for s in S) + ");      // (fresh addresses: those of the step's first row read are not worth registers across the step)")
  lower_rows("ps", full=False)
  A("      rn::wave_lds_sync();      // every lane has its rows: the image is free for the next step's burst")
  A("      RN_RTS_STAMP(10);")
  A("    }")
  A("    // ---- the oldest smoothed state goes out un-normalised (ekf_sym.py:665-667 never reaches it); its covariance left above ----")
  A("    {")
  A("      int lz = lane;")
  A('      asm volatile("" : "+v"(lz));')
  A(f"      for (int i = lz; i < cnt * {D}; i += 64) xs[base * {D} + i] = s_xn[i];")
  A("    }")
  A("    rn::wave_lds_sync();")
  A("  }")
  A("}")
  return "\n".join(b)


def launch(spec, tri=False):
  return f"""  const int64_t tiles = (n + {FPW - 1}) / {FPW};
  hipLaunchKernelGGL({'k_rts4_tri' if tri else 'k_rts4'}, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     xf, Pf, ts, T, Q, n, norm_quats, xs, Ps, x_last, P_last);"""
