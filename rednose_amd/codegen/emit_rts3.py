"""Kernel family W, smoother `k_rts3`: the RTS backward pass in the fused run's layout -- SEVERAL ROWS PER LANE, 8 FILTERS PER WAVEFRONT.

Reference: the Python-only `EKF_sym.rts_smooth` (/root/reference/rednose/helpers/ekf_sym.py:651-690); per backward step k
    Fk = F(xk_k, t[k+1] - t[k]);  Ck = solve(Pk1_k, Fk Pk_k^T)^T;  xk_n = err(xk_k, Ck inv_err(xk1_k, xk1_n));
    Pk_n = Pk_k + Ck (Pk1_n - Pk1_k) Ck^T
with the predicted pair (xk1_k, Pk1_k) recomputed from the filtered one (templates/ekf_hip_rts.h explains the memory plan and
the recursion's quirks, which are kept: start from the PREDICTED pair of the last step, in-place renormalisation of xk1_n).

Why a third smoother.  `rn::k_rts_group` (round 2) gives a filter a 32-lane group -- 2 filters per wavefront for live's 22 error
states -- and one row of every matrix per lane.  Its backward step is a chain of dependent LDS round trips and fp64
instructions (a 22-column factorisation, 44 substitution pivots) that TWO filters share: 25 us per wavefront-step, 31 % of the
wave cycles issuing (profiles/r2_sq_counters_smoother.txt), 160 M steps/s = 16 % of the HBM roofline; the two E^3 products ran on
v_mfma_f64_16x16x4 tiles that are 57 % padding for E = 22.  Here the layout is the fused run's (emit_wide3): a filter gets
GL = 8 lanes (16 above 22 error states), lane c owns rows c, c + GL, c + 2 GL ... of every matrix (R = ceil(E / GL) row slots),
so EIGHT filters share each dependent chain, every broadcast operand read from LDS feeds R FMAs, 22 of 24 row slots are busy,
and all arithmetic is vector fp64 (the products need 2 R E^2 FMAs per lane and product -- cheaper than the padded MFMA tiles).
The price is the fused run's: one wavefront per SIMD and ONE matrix image of LDS per filter (8 images + vectors = 41 KB per
wavefront, 3-4 per CU), and general fp64 arithmetic only sees the 256 architectural registers (the other half of the file is
parking space): so only TWO row sets live in registers (a, y: 4 R E registers), every matrix that has to be broadcast takes its
turn in the image -- which holds either one full E x E matrix or the packed lower triangles of TWO symmetric ones -- and the
smoothed covariance is not carried from step to step in registers but read back from where the previous step stored it:

  step k (image I; row sets a, y):
    Pk_k HBM -> I (one coalesced asynchronous burst), xk_k -> LDS; lead lanes evaluate f / F non-zeros (scal_predict)
    a <- rows of Pk_k;  y <- rows of A = Pk_k Fk^T (= columns of M = Fk Pk_k^T: the right-hand sides, row-local)
    I <- A;  a <- rows of Pk1_k = (columns of A) Fk^T + dt Q   (P = P^T as in the fused run's predict: one transposition)
    rows of Pk1_n <- Ps[k + 1] (L2; the previous step's output);  I <- tril(Pk1_k) | tril(D = Pk1_n - Pk1_k)
    Cholesky of Pk1_k, left-looking: pivot row broadcast from I, every lane forms its R entries and the pivot redundantly
    y <- Pk1_k^-1 y: forward substitution in dot form, backward in axpy form -- both read ROWS of the packed factor
    state: delta = Ck inv_err(xk1_k, xk1_n), xk_n = err(xk_k, delta)  (lead lanes + one R x E dot per lane)
    a <- T = Ck D, dot form against the rows of the symmetric D (packed triangle read both ways)
    I <- Ck;  y <- U = T Ck^T, dot form against the rows of Ck
    I <- U;  Ps[k] <- Pf[k] + U: one coalesced read-add-write

Generated for ordinary (non-MSCKF) lane-group models up to 32 error states; MSCKF models and larger ones keep rn::k_rts_group.
"""
from rednose_amd.codegen.emit_common import term, sum_terms


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def applicable(spec):
  msckf = any(k.He_sym is not None for k in spec.kinds) or spec.N > 0
  return (not msckf) and spec.dim_main_err == spec.dim_err and spec.dim_err <= 32


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


def _region(b, head, body, pins):
  """One scheduling region: operand loads of the NEXT piece of work first (`head`, held there by a scheduling barrier), then this
  piece's arithmetic, then a compiler fence with the freshly written values pinned in front of it (see emit_wide3._rank_pass for
  why hipcc needs both)."""
  b.extend(head)
  if head:
    # without this hipcc sinks the requests to the end of the region, right in front of their first use in the next one, and
    # every region then starts with a full LDS round trip (seen in the ISA of the first build: ds_read x 11, s_waitcnt, FMAs)
    b.append("      __builtin_amdgcn_sched_barrier(0);")
  b.extend(body)
  if pins:
    b.append("      " + " ".join(f"rn::pin({v});" for v in pins))
  b.append("      rn::wave_lds_sync();")


def tri(i, j):
  """Index of entry (i, j), j <= i, in row-major packed lower-triangular storage."""
  assert j <= i
  return i * (i + 1) // 2 + j


def sym(i, j):
  return tri(i, j) if j <= i else tri(j, i)


def layout(spec):
  """The fused run's layout (emit_wide3.layout).  Wider lane groups at two wavefronts per SIMD (16 lanes x 2 rows: 114 spilled registers
  under the 256 budget; 32 lanes x 1 row: 7.58 ms against 4.48 ms on 16 384 x 60 steps) were measured in round 3 and removed in round 4:
  profiles/tuning_notes.md."""
  from rednose_amd.codegen import emit_wide3 as w3
  return w3.layout(spec)


def kernel(spec):
  from rednose_amd.codegen import emit_wide3 as w3      # noqa: F401
  from rednose_amd.codegen import tuning
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  TRI = E * (E + 1) // 2
  IMG = max(EE, 2 * TRI)
  IMG += IMG & 1
  GL, R, FPW = layout(spec)
  bounds = "__launch_bounds__(64)"
  scal, lay = _scal_text(spec)
  _, Fs = _tables(spec)
  quat = "".join(f" rn::normalize_quat<{D}>(xv, {q});" for q in spec.quaternion_idxs)
  S = range(R)
  aligned = EE % 2 == 0          # every tile's records start 16-byte aligned: direct HBM -> LDS copies
  b = []
  A = b.append

  def rows_decl(name):
    return " ".join(f"double {name}{s}[{E}];" for s in S)

  def rows_to_image(name, ind="      "):
    for s in S:
      A(f"{ind}if (ok{s}) {{")
      A("#pragma unroll")
      A(f"{ind}  for (int j = 0; j < {E}; j++) sI[rr{s} * {E} + j] = {name}{s}[j];")
      A(f"{ind}}}")

  A(f"// ---- smoother in the fused run's layout: {GL} lanes x {R} rows per filter, {FPW} filters per wavefront (emit_rts3.py) ----")
  A(f"constexpr int RTS3_SLOT = {lay.SLOT};")
  A(f"constexpr int RTS3_IMG = {IMG};      // doubles of LDS image per filter: a full E x E matrix, or two packed triangles")
  A(scal)
  qd_decl = "\n".join(f"  const double qd{s} = gQ[((c + {GL * s}) < {E} ? (c + {GL * s}) : 0) * {E + 1}];" for s in S)
  A(f"""
__global__ {bounds} void k_rts3(const double* __restrict__ xf, const double* __restrict__ Pf, const double* __restrict__ ts,
    const int64_t T, const double* __restrict__ gQ, const int64_t n, const int norm_quats, double* __restrict__ xs,
    double* __restrict__ Ps, const double* __restrict__ xl, const double* __restrict__ Pl) {{
  __shared__ __attribute__((aligned(16))) double s_I[{FPW} * RTS3_IMG + 2];      // the one matrix image per filter (see emit_rts3.py)
  __shared__ __attribute__((aligned(16))) double s_sl[{FPW} * RTS3_SLOT];        // x' = f(xk_k), F non-zeros, dt
  __shared__ __attribute__((aligned(16))) double s_xk[{FPW} * {D} + 2];          // xk_k
  __shared__ __attribute__((aligned(16))) double s_xn[{FPW} * {D} + 2];          // xk1_n, then xk_n
  __shared__ __attribute__((aligned(16))) double s_dv[{FPW} * {E} + 2];          // inv_err(xk1_k, xk1_n), then Ck delta
  const int lane = threadIdx.x;
  const int g = lane / {GL};
  const int c = lane % {GL};
  // a diagonal process noise (the usual case) lives in registers, as in the fused run (emit_wide3.predict_fn): the rows of Q were 3 E
  // eight-byte loads per lane and step, each instruction a gather of eight rows
  int qoff = 0;
  for (int i = lane; i < {EE}; i += 64) qoff |= (i / {E} != i % {E}) && (gQ[i] != 0.0);
  const bool qdiag = !__any(qoff);
{qd_decl}
  const int64_t tiles = (n + {FPW} - 1) / {FPW};
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * {FPW};
    const int cnt = (n - base) < {FPW} ? (int)(n - base) : {FPW};
    const bool live = g < cnt;
    const int gg = live ? g : 0;
    const bool lead = live && c == 0;
    double* sI = s_I + gg * RTS3_IMG;       // full-matrix view: sI[r * {E} + j]
    double* sL = sI;                        // packed lower triangle of Pk1_k / its factor: sL[r (r + 1) / 2 + j], j <= r
    double* sD = sI + {TRI};                // packed lower triangle of D = Pk1_n - Pk1_k (symmetric up to rounding)
    double* sl = s_sl + gg * RTS3_SLOT;
    double* sxk = s_xk + gg * {D};
    double* sxn = s_xn + gg * {D};
    double* sdv = s_dv + gg * {E};""")
  for s in S:
    A(f"    const int rr{s} = c + {GL * s}; const bool ok{s} = live && rr{s} < {E}; const int rc{s} = rr{s} < {E} ? rr{s} : 0;"
      f" const int tb{s} = rc{s} * (rc{s} + 1) / 2;")
  A("    if (T == 1) {      // nothing to smooth, the single estimate's predicted pair is not available: the filtered pair passes through")
  A(f"      if (Ps != Pf) {{ for (int i = lane; i < cnt * {EE}; i += 64) Ps[base * {EE} + i] = Pf[base * {EE} + i]; }}")
  A(f"      if (xs != xf) {{ for (int i = lane; i < cnt * {D}; i += 64) xs[base * {D} + i] = xf[base * {D} + i]; }}")
  A("      continue;")
  A("    }")
  XT = -(-(FPW * D) // 64)
  A(f"    double xnext[{XT}];      // filtered state of the next step to process, in flight across the loop's back edge")
  A("#pragma unroll")
  A(f"    for (int it = 0; it < {XT}; it++) {{ const int i = lane + 64 * it; xnext[it] = xf[((T - 2) * n + base) * {D} + (i < cnt * {D} ? i : 0)]; }}")
  A("    for (int64_t k = T - 2; k >= 0; k--) {")
  A("      const bool first = (k == T - 2);")
  A("      int lb = lane;")
  A('      asm volatile("" : "+v"(lb));         // opaque copy of the lane index: the copies\' index arithmetic stays inside the step')
  A("      RN_RTS_STAMP(0);")
  A("      // ---- A. filtered pair of step k: Pk_k -> image in one coalesced burst, xk_k -> LDS ----")
  if aligned and IMG == EE:
    A(f"      rn::async_copy_g2l<{FPW} * {EE}>(Pf + (k * n + base) * {EE}, cnt * {EE}, s_I, lb);      // no register staging: lands under the scalar phase")
    pre_wait = "      rn::async_wait();"
  elif aligned:
    A("      // (a filter's image is larger than its record: one asynchronous copy per filter)")
    A(f"      for (int f_ = 0; f_ < cnt; f_++) rn::async_copy_g2l<{EE}>(Pf + (k * n + base + f_) * {EE}, {EE}, s_I + f_ * RTS3_IMG, lb);")
    pre_wait = "      rn::async_wait();"
  else:
    A(f"      for (int i = lb; i < cnt * {EE}; i += 64) s_I[(i / {EE}) * RTS3_IMG + i % {EE}] = Pf[(k * n + base) * {EE} + i];      // (odd record size)")
    pre_wait = None
  A("      {      // xk_k was requested at the end of the previous (newer) step (before the loop for the first).  (Committing it BEFORE the copy is")
  A("             // issued -- its wait is a vmcnt(0), the value crosses the loop's back edge, and behind the copy it waits for the burst to land --")
  A("             // measured slower, 67.3 against 66.0 ms per config-4 chunk: the wait then delays the ISSUE of the copy by the store drain.)")
  A("#pragma unroll")
  A(f"        for (int it = 0; it < {XT}; it++) {{ const int i = lb + 64 * it; if (i < cnt * {D}) s_xk[i] = xnext[it]; }}")
  A("      }")
  A("      const double dt = ts[k + 1] - ts[k];")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(1);")
  A("      // ---- B. f(xk_k) [renormalised like the forward pass], non-zeros of Fk: once per filter -> slot ----")
  A("      if (lead) scal_predict_s(sxk, dt, sl, norm_quats & 1);")
  if pre_wait:
    A(pre_wait)
  A("      rn::wave_lds_sync();")
  A(f"      {rows_decl('a')}      // rows of Pk_k, then Pk1_k / its factor, then T")
  A(f"      {rows_decl('y')}      // right-hand sides (rows of A = Pk_k Fk^T), then rows of Ck")
  # Contract of batch_rts (include/rednose_amd_filter.h): the LOWER triangle of every covariance it is given is read, mirrored.  The
  # recursion below uses P = P^T throughout (one transposition in the predict, Cholesky factor and D as packed lower triangles), so
  # the rows are taken from the lower triangle here: entry j of row r is P[r][j] for j <= r and P[j][r] above the diagonal.  A slot's
  # columns left of its diagonal block are plain row reads, right of it column reads (the group's lanes read consecutive doubles),
  # the diagonal block selects per lane.  Same number of LDS reads as taking the rows as they are.
  for s in S:
    lo, hi = GL * s, min(E, GL * s + GL)
    if lo:
      A("#pragma unroll")
      A(f"      for (int j = 0; j < {lo}; j++) a{s}[j] = sI[rc{s} * {E} + j];")
    A("#pragma unroll")
    A(f"      for (int j = {lo}; j < {hi}; j++) a{s}[j] = sI[(j <= rc{s}) ? rc{s} * {E} + j : j * {E} + rc{s}];")
    if hi < E:
      A("#pragma unroll")
      A(f"      for (int j = {hi}; j < {E}; j++) a{s}[j] = sI[j * {E} + rc{s}];")
  A("      rn::wave_lds_sync();      // every lane has its rows: the image takes A")
  A("      RN_RTS_STAMP(2);")
  A("      // ---- C. rows of A = Pk_k Fk^T (row-local, F's structural zeros cost nothing): the right-hand sides, and through the")
  A("      // image the columns of A = rows of Fk Pk_k (P = P^T up to rounding, as in the fused run's predict) ----")
  for s in S:
    for i in range(E):
      A(f"      y{s}[{i}] = {sum_terms(term(cf, f'a{s}[{kk}]') for kk, cf in Fs.row_nz(i))};")
  rows_to_image("y")
  A("      rn::wave_lds_sync();")
  A("      if (qdiag) {")
  for s in S:
    A("        {")
    A(f"          double col[{E}];")
    A(f"          const double dq = dt * qd{s};")
    A("#pragma unroll")
    A(f"          for (int m = 0; m < {E}; m++) col[m] = sI[m * {E} + rc{s}];      // column of A = row of Fk Pk_k")
    for j in range(E):
      diag = f" + (rc{s} == {j} ? dq : 0.0)" if GL * s <= j < GL * (s + 1) else ""      # the lane's own row index is c + GL s
      A(f"          a{s}[{j}] = {sum_terms(term(cf, f'col[{m}]') for m, cf in Fs.row_nz(j))}{diag};")
    A("        }")
  A("      } else {")
  A("        int qz = 0;")
  A('        asm volatile("" : "+v"(qz));       // Q behind an opaque zero: its addresses are not worth registers across the step loop')
  A("        const double* __restrict__ gq = gQ + qz;")
  for s in S:
    A("        {")
    A(f"          double col[{E}], q[{E}];")
    A("#pragma unroll")
    A(f"          for (int m = 0; m < {E}; m++) col[m] = sI[m * {E} + rc{s}];      // column of A = row of Fk Pk_k")
    A("#pragma unroll")
    A(f"          for (int j = 0; j < {E}; j++) q[j] = gq[rc{s} * {E} + j];")
    for j in range(E):
      A(f"          a{s}[{j}] = {sum_terms(term(cf, f'col[{m}]') for m, cf in Fs.row_nz(j))} + dt*q[{j}];")
    A("        }")
  A("      }")
  A("      rn::wave_lds_sync();      // the image is free: rows of Pk1_k are in a*")
  A("      RN_RTS_STAMP(3);")
  A("      // ---- D. recursion start / difference matrix.  The smoothed covariance of step k + 1 is NOT carried in registers: the")
  A("      // previous step left it in Ps[k + 1] (through the L2), its rows are read back here, the lower triangle of")
  A("      // D = Pk1_n - Pk1_k goes straight to LDS and the lower triangle of Pk1_k beside it. ----")
  A("      if (first) {")
  A("        // newest estimate := the predicted pair of the last step (passed in, or recomputed just now): ekf_sym.py:658-659")
  A("        if (live) {       // (two loops, not one select between a global and an LDS source: hipcc 7.2 trips over the generic pointer)")
  A(f"          if (xl != nullptr) {{ for (int i = c; i < {D}; i += {GL}) sxn[i] = xl[(base + gg) * {D} + i]; }}")
  A(f"          else {{ for (int i = c; i < {D}; i += {GL}) sxn[i] = sl[{lay.OFF_X} + i]; }}")
  A("        }")
  A("        if (Pl != nullptr) {")
  A(f"          for (int i = lane; i < cnt * {EE}; i += 64) Ps[((k + 1) * n + base) * {EE} + i] = Pl[base * {EE} + i];")
  A("        } else {")
  rows_to_image("a", ind="          ")
  A("          rn::wave_lds_sync();")
  A(f"          for (int i = lane; i < cnt * {EE}; i += 64) Ps[((k + 1) * n + base) * {EE} + i] = s_I[(i / {EE}) * RTS3_IMG + i % {EE}];")
  A("          rn::wave_lds_sync();")
  A("        }")
  A('        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
  A("        rn::wave_lds_sync();")
  A("      }")
  A("      if (lead && (norm_quats & 2)) {")
  A(f"        double xv[{D}];")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) xv[i] = sxn[i];")
  A(f"       {quat}")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) sxn[i] = xv[i];")
  A("      }")
  A("      rn::wave_lds_sync();")
  A('      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the previous step\'s output stores have reached the L2')
  if aligned:
    # Round 4: the smoothed covariance of step k + 1 comes back THROUGH THE IMAGE -- one coalesced asynchronous burst (1 KiB per
    # wave-instruction, as for Pk_k in phase A), then conflict-free LDS row reads.  Until then every lane fetched its rows with 16-byte
    # loads at a 176-byte stride: 33 instructions per lane and step, each a gather of 64 pieces that the texture-address path takes
    # one cache line at a time (64 cycles an instruction, four wavefronts of a CU behind one such path): the 4 us "read back + D" of
    # the phase timeline.  The image is free here (rows of Pk1_k are in a*), and the L1 is invalidated first: in place (Ps == Pf) this
    # CU read the same addresses as Pf[k + 1] one step ago.
    A('      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");')
    if IMG == EE:
      A(f"      rn::async_copy_g2l<{FPW} * {EE}>(Ps + ((k + 1) * n + base) * {EE}, cnt * {EE}, s_I, lb);")
    else:
      A(f"      for (int f_ = 0; f_ < cnt; f_++) rn::async_copy_g2l<{EE}>(Ps + ((k + 1) * n + base + f_) * {EE}, {EE}, s_I + f_ * RTS3_IMG, lb);")
    A("      {")
    A("        int lo = lane;")
    A('        asm volatile("" : "+v"(lo));')
    A(f"        for (int i = lo; i < cnt * {D}; i += 64) xs[((k + 1) * n + base) * {D} + i] = s_xn[i];      // smoothed state of step k + 1 (after its renormalisation)")
    A("      }")
    A("      rn::async_wait();")
    A("      rn::wave_lds_sync();")
    A(f"      {rows_decl('pn')}      // rows of Pk1_n up to each slot's last row (its lower block triangle is all D needs)")
    for s in S:
      A("#pragma unroll")
      A(f"      for (int j = 0; j < {min(E, GL * s + GL)}; j++) pn{s}[j] = sI[rc{s} * {E} + j];")
    A("      rn::wave_lds_sync();      // every lane has its rows: the image takes the two packed triangles")
  else:
    A(f"      {rows_decl('pn')}      // rows of Pk1_n, read back from Ps[k + 1] (L2-served: nontemporal, not this CU's L1 -- another lane stored them)")
    for s in S:
      A("#pragma unroll")
      A(f"      for (int j = 0; j < {E}; j++) pn{s}[j] = __builtin_nontemporal_load(Ps + (((k + 1) * n + base + gg) * {EE} + rc{s} * {E}) + j);")
    A("      {      // issued AFTER those loads: the wait for them then leaves these stores in flight (vmcnt retires in order)")
    A("        int lo = lane;")
    A('        asm volatile("" : "+v"(lo));')
    A(f"        for (int i = lo; i < cnt * {D}; i += 64) xs[((k + 1) * n + base) * {D} + i] = s_xn[i];      // smoothed state of step k + 1 (after its renormalisation)")
    A("      }")
  for s in S:
    A("#pragma unroll")
    A(f"      for (int j = 0; j < {min(E, GL * s + GL)}; j++) {{      // (columns beyond the slot's last row are above the diagonal for every lane)")
    A(f"        if (ok{s} && j <= rr{s}) {{ sD[tb{s} + j] = pn{s}[j] - a{s}[j]; sL[tb{s} + j] = a{s}[j]; }}")
    A("      }")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(4);")
  A("      // ---- E. Cholesky of Pk1_k, left-looking.  Rows in a*, finished rows of the factor in LDS (packed): the pivot row is")
  A("      // broadcast and every lane forms its entries AND the pivot redundantly (no publish / wait per column).  Column j + 1's")
  A("      // sums over the columns before j depend on nothing column j produces: they are emitted inside column j's region, so they")
  A("      // issue while column j's reciprocal square root (a chain of dependent fp64 operations) is in flight.  The diagonal of")
  A("      // the packed factor ends up holding the RECIPROCAL pivots (the only thing the substitutions need of it). ----")
  def below(j):      # row slots that hold a row >= j (slot s holds rows GL s .. GL s + GL - 1): the others' entries of column j are
    return [s for s in S if GL * s + GL - 1 >= j]      # upper-triangle junk that nothing reads -- no instruction is emitted for them
  A(f"      double pv_0 = sL[{tri(0, 0)}], pw_0 = 0.0;")
  for s in S:
    A(f"      double t{s}_0 = a{s}[0], u{s}_0 = 0.0;")
  for j in range(E):
    body = []
    Sj = below(j)
    if j >= 1:        # the last term: entry j - 1 of the pivot row exists since the previous region's store
      m = j - 1
      tg = "w" if m & 1 else "v"
      body.append(f"      {{ const double ql = sL[{tri(j, m)}]; p{tg}_{j} = fma(-ql, ql, p{tg}_{j});" +
                  "".join(f" {'u' if m & 1 else 't'}{s}_{j} = fma(-a{s}[{m}], ql, {'u' if m & 1 else 't'}{s}_{j});" for s in Sj) + " }")
    body.append(f"      const double il_{j} = rn::fast_rsqrt(pv_{j} + pw_{j});")
    if j + 1 < E:     # early part of column j + 1: entries 0 .. j - 1 of its pivot row are final
      jn = j + 1
      Sn = below(jn)
      body.append(f"      double pv_{jn} = sL[{tri(jn, jn)}], pw_{jn} = 0.0;")
      for s in Sn:
        body.append(f"      double t{s}_{jn} = a{s}[{jn}], u{s}_{jn} = 0.0;")
      if j >= 1:
        body.append(f"      {{ double q[{j}];")
        body.append("#pragma unroll")
        body.append(f"        for (int m = 0; m < {j}; m++) q[m] = sL[{tri(jn, 0)} + m];")
        for m in range(j):
          tg = "w" if m & 1 else "v"
          body.append(f"        p{tg}_{jn} = fma(-q[{m}], q[{m}], p{tg}_{jn});" +
                      "".join(f" {'u' if m & 1 else 't'}{s}_{jn} = fma(-a{s}[{m}], q[{m}], {'u' if m & 1 else 't'}{s}_{jn});" for s in Sn))
        body.append("      }")
    for s in Sj:
      body.append(f"      a{s}[{j}] = (rr{s} == {j}) ? il_{j} : (t{s}_{j} + u{s}_{j}) * il_{j};")
      body.append(f"      if (ok{s} && rr{s} >= {j}) sL[tb{s} + {j}] = a{s}[{j}];")
    pins = [f"a{s}[{j}]" for s in Sj]
    if j + 1 < E:
      pins += [f"t{s}_{j + 1}" for s in below(j + 1)] + [f"u{s}_{j + 1}" for s in below(j + 1)] + [f"pv_{j + 1}", f"pw_{j + 1}"]
    _region(b, [], body, pins)
  A("      RN_RTS_STAMP(5);")
  A("      // ---- F. Ck^T = Pk1_k^-1 M, i.e. every row slot solves with its own right-hand side (row of A).  Forward substitution in")
  A("      // dot form (row i of the factor against the solved part), backward in axpy form (row m of the factor scaled into the")
  A("      // unsolved part): both read ROWS of the packed factor, and in both the next row is requested before the current one is")
  A("      // used. ----")

  def lrow(i, name):          # entries 0 .. i - 1 of row i of the factor and its reciprocal pivot
    out = []
    if i:
      out += [f"      double {name}[{i}];", "#pragma unroll", f"      for (int m = 0; m < {i}; m++) {name}[m] = sL[{tri(i, 0)} + m];"]
    out.append(f"      const double {name}_il = sL[{tri(i, i)}];")
    return out
  b.extend(lrow(0, "f0"))
  for i in range(E):
    head = lrow(i + 1, f"f{i + 1}") if i + 1 < E else []
    body = []
    for s in S:
      body.append(f"      double ft{s}_{i} = y{s}[{i}], fu{s}_{i} = 0.0, fv{s}_{i} = 0.0, fw{s}_{i} = 0.0;")
    for m in range(i):
      body.append("      " + " ".join(f"f{'tuvw'[m & 3]}{s}_{i} = fma(-y{s}[{m}], f{i}[{m}], f{'tuvw'[m & 3]}{s}_{i});" for s in S))
    body.append("      " + " ".join(f"y{s}[{i}] = ((ft{s}_{i} + fu{s}_{i}) + (fv{s}_{i} + fw{s}_{i})) * f{i}_il;" for s in S))
    _region(b, head, body, [f"y{s}[{i}]" for s in S])
  b.extend(lrow(E - 1, f"g{E - 1}"))
  for m in range(E - 1, -1, -1):
    head = lrow(m - 1, f"g{m - 1}") if m >= 1 else []
    body = ["      " + " ".join(f"y{s}[{m}] *= g{m}_il;" for s in S)]
    for i in range(m - 1, -1, -1):        # the entry the next pivot needs first
      body.append("      " + " ".join(f"y{s}[{i}] = fma(-g{m}[{i}], y{s}[{m}], y{s}[{i}]);" for s in S))
    _region(b, head, body, [f"y{s}[{i}]" for s in S for i in range(m + 1)])
  A("      // y* = rows of Ck")
  A("      RN_RTS_STAMP(6);")
  A("      // ---- G. state: delta = Ck inv_err(xk1_k, xk1_n); xk_n = err(xk_k, delta) ----")
  A("      if (lead) {")
  A(f"        double xb[{D}], xn1[{D}], delta[{E}];")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {D}; i++) {{ xb[i] = sl[{lay.OFF_X} + i]; xn1[i] = sxn[i]; }}")
  A("        inv_err_fun(xb, xn1, delta);")
  A("#pragma unroll")
  A(f"        for (int i = 0; i < {E}; i++) sdv[i] = delta[i];")
  A("      }")
  A("      rn::wave_lds_sync();")
  A("      {")
  A(f"        double de[{E}];")
  A("#pragma unroll")
  A(f"        for (int j = 0; j < {E}; j++) de[j] = sdv[j];")
  for s in S:
    A(f"        const double dx{s} = (" + " + ".join(f"y{s}[{j}]*de[{j}]" for j in range(0, E, 2)) + ") + (" +
      (" + ".join(f"y{s}[{j}]*de[{j}]" for j in range(1, E, 2)) or "0.0") + ");")
  A("        rn::wave_lds_sync();      // every lane has delta: the buffer takes Ck delta")
  for s in S:
    A(f"        if (ok{s}) sdv[rr{s}] = dx{s};")
  A("      }")
  A("      rn::wave_lds_sync();      // (xk_n = err(xk_k, Ck delta) is formed by the lead lanes in phase J, under the latency of its loads)")
  A("      RN_RTS_STAMP(7);")
  A("      // ---- H. T = Ck D in dot form: entry j of a row of T is that row of Ck against row j of the symmetric D (its packed")
  A("      // triangle is read both ways); each finished column of T is final, only Ck's rows stay live as coefficients ----")

  # Both products in dot form, a region = one output column against HALF a broadcast row: two half rows of operands in flight
  # (44 registers) next to the coefficient row set (4 R E = 132 for live) fit the 256 architectural registers; with whole rows
  # hipcc parked the coefficients in AGPRs and read each one back in front of its FMA (two extra instructions per FMA).
  H2 = (E + 1) // 2
  halves = [list(range(0, H2)), list(range(H2, E))]

  def product(coef, out, src_index, tmp, opname, slots=lambda j: list(S)):
    """out_s[j] = sum_kk coef_s[kk] * src(j, kk), FOUR partial sums per row slot (12 independent chains for 3 slots: a dependent
    fp64 FMA issues ~40 cycles after its predecessor when the wavefront is alone on its SIMD, two sums per slot left the chains
    24 cycles apart)."""
    pieces = [(j, h) for j in range(E) for h in (0, 1)]
    NP = 4      # partial sums per slot (6 / 8 measured in round 5: 74.1 / 75.7 ms per config-4 chunk against 68.5 -- the pairwise tree's extra adds cost more than the longer chains hide)

    def tree(ts):        # pairwise sum of the partial sums: (s0 + s1) + (s2 + s3) for four
      while len(ts) > 1:
        ts = [f"({ts[i]} + {ts[i + 1]})" if i + 1 < len(ts) else ts[i] for i in range(0, len(ts), 2)]
      return ts[0][1:-1] if ts[0].startswith("(") else ts[0]

    def acc(s_, j, kk):
      return f"{out}{s_}[{j}]" if kk % NP == 0 else f"{tmp}{kk % NP}_{s_}_{j}"

    def loads(pc):
      j, h = pc
      return [f"      const double {opname}{j}_{kk} = {src_index(j, kk)};" for kk in halves[h]]
    b.extend(loads(pieces[0]))
    for pi, (j, h) in enumerate(pieces):
      head = loads(pieces[pi + 1]) if pi + 1 < len(pieces) else []
      body = []
      Sj = slots(j)
      if h == 0:
        body.append("      " + " ".join(f"double {tmp}{q}_{s}_{j} = 0.0;" for s in Sj for q in range(1, NP)))
      for kk in halves[h]:
        body.append("      " + " ".join(
          (f"{out}{s}[{j}] = {coef}{s}[{kk}]*{opname}{j}_{kk};" if kk == 0 else
           f"{acc(s, j, kk)} = fma({coef}{s}[{kk}], {opname}{j}_{kk}, {acc(s, j, kk)});") for s in Sj))
      pins = [f"{out}{s}[{j}]" for s in Sj]
      if h == 1:
        body.append("      " + " ".join(f"{out}{s}[{j}] = {tree([f'{out}{s}[{j}]'] + [f'{tmp}{q}_{s}_{j}' for q in range(1, NP)])};" for s in Sj))
      else:
        pins += [f"{tmp}{q}_{s}_{j}" for s in Sj for q in range(1, NP)]
      _region(b, head, body, pins)

  product("y", "a", lambda j, kk: f"sD[{sym(j, kk)}]", "h", "d")
  A("      RN_RTS_STAMP(8);")
  A("      // ---- I. U = T Ck^T: rows of Ck are broadcast from the image (full layout again: factor and D are dead), dot form ----")
  rows_to_image("y")
  A("      rn::wave_lds_sync();")
  # U = Ck D Ck^T is symmetric (D is, up to rounding): a row slot forms only the columns up to its last row -- whole slots drop out
  # of the later columns' regions -- and the missing upper-right blocks are mirrored inside the image in phase J
  product("a", "y", lambda j, kk: f"sI[{j * E + kk}]", "e", "c", slots=below)
  A("      RN_RTS_STAMP(9);")
  A("      // ---- J. Pk_n = Pk_k + U leaves: U's rows through the image, then one coalesced read-add-write over the tile's records ----")
  early = EE % 2 == 0          # (config 4 backward, same call: 66.35 ms per chunk with the early request, 66.90 ms without)
  if early:
    # The tile's filtered records are requested HERE, before U goes through the image: the coefficient row set of the last product
    # (a*) is dead, its registers take the loads, and the HBM / Infinity Cache round trip passes under the image writes, the
    # mirroring and the lead lanes' state update instead of being waited for after them.
    IT = -(-(FPW * EE // 2) // 64)
    A("      typedef double rts3_d2 __attribute__((ext_vector_type(2)));")
    A("      int le = lane;")
    A('      asm volatile("" : "+v"(le));')
    A(f"      const rts3_d2* __restrict__ in2 = reinterpret_cast<const rts3_d2*>(Pf + (k * n + base) * {EE});")
    A(f"      const int nv = cnt * {EE // 2};")
    A(f"      rts3_d2 v[{IT}];")
    A("#pragma unroll")
    A(f"      for (int it = 0; it < {IT}; it++) {{ const int idx = le + 64 * it; v[it] = in2[idx < nv ? idx : nv - 1]; }}")
    A("      if (k > 0) {")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {XT}; it++) {{ const int i = lane + 64 * it; xnext[it] = xf[((k - 1) * n + base) * {D} + (i < cnt * {D} ? i : 0)]; }}")
    A("      }")
    A("      __builtin_amdgcn_sched_barrier(0);")
  for s in S:
    ncol = min(E, GL * s + GL)        # columns this slot formed
    A(f"      if (ok{s}) {{")
    A("#pragma unroll")
    A(f"        for (int j = 0; j < {ncol}; j++) sI[rr{s} * {E} + j] = y{s}[j];")
    A("      }")
  A("      rn::wave_lds_sync();")
  if any(min(E, GL * s + GL) < E for s in S):
    A("      {      // upper-right blocks: U[r][j] = U[j][r] for the columns beyond a slot's last row")
    for s in S:
      ncol = min(E, GL * s + GL)
      if ncol < E:
        A(f"        if (ok{s}) {{")
        A(f"          double m_[{E - ncol}];")
        A("#pragma unroll")
        A(f"          for (int j = {ncol}; j < {E}; j++) m_[j - {ncol}] = sI[j * {E} + rr{s}];")
        A("#pragma unroll")
        A(f"          for (int j = {ncol}; j < {E}; j++) sI[rr{s} * {E} + j] = m_[j - {ncol}];")
        A("        }")
    A("      }")
    A("      rn::wave_lds_sync();")
  if not early:
    A("      if (k > 0) {")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {XT}; it++) {{ const int i = lane + 64 * it; xnext[it] = xf[((k - 1) * n + base) * {D} + (i < cnt * {D} ? i : 0)]; }}")
    A("      }")
  A("      {")
  if not early:
    A("        int le = lane;")
    A('        asm volatile("" : "+v"(le));')
  # every load of the tile's records is issued before the first is used (a rolled loop of load -> add -> store pays one
  # memory round trip per iteration: 61 of them per step in the first build, half of the kernel's time in s_waitcnt)
  if EE % 2 == 0:
    IT = -(-(FPW * EE // 2) // 64)
    if not early:
      A("        typedef double rts3_d2 __attribute__((ext_vector_type(2)));")
      A(f"        const rts3_d2* __restrict__ in2 = reinterpret_cast<const rts3_d2*>(Pf + (k * n + base) * {EE});")
      A(f"        const int nv = cnt * {EE // 2};")
      A(f"        rts3_d2 v[{IT}];")
      A("#pragma unroll")
      A(f"        for (int it = 0; it < {IT}; it++) {{ const int idx = le + 64 * it; v[it] = in2[idx < nv ? idx : nv - 1]; }}")
    A(f"        rts3_d2* __restrict__ out2 = reinterpret_cast<rts3_d2*>(Ps + (k * n + base) * {EE});")
    A("        if (lead) {      // state update, second half (see phase G): one lane per filter, while the loads above are in flight")
    A(f"          double xa[{D}], xnew[{D}], delta[{E}];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) xa[i] = sxk[i];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {E}; i++) delta[i] = sdv[i];")
    A("          err_fun(xa, delta, xnew);")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) sxn[i] = xnew[i];       // xk_n: becomes xk1_n of the next (older) step")
    A("        }")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {IT}; it++) {{")
    A("          const int idx = le + 64 * it;")
    A("          if (idx < nv) {")
    A(f"            const int f_ = idx / {EE // 2}, r_ = idx - f_ * {EE // 2};")
    A("            const rts3_d2 u_ = *reinterpret_cast<const rts3_d2*>(s_I + f_ * RTS3_IMG + 2 * r_);")
    A("            out2[idx] = v[it] + u_;")
    A("          }")
    A("        }")
  else:
    IT = -(-(FPW * EE) // 64)
    A(f"        const double* __restrict__ in1 = Pf + (k * n + base) * {EE};")
    A(f"        double* __restrict__ out1 = Ps + (k * n + base) * {EE};")
    A(f"        const int nv = cnt * {EE};")
    A(f"        double v[{IT}];")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {IT}; it++) {{ const int idx = le + 64 * it; v[it] = in1[idx < nv ? idx : nv - 1]; }}")
    A("        if (lead) {      // state update, second half (see phase G): one lane per filter, while the loads above are in flight")
    A(f"          double xa[{D}], xnew[{D}], delta[{E}];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) xa[i] = sxk[i];")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {E}; i++) delta[i] = sdv[i];")
    A("          err_fun(xa, delta, xnew);")
    A("#pragma unroll")
    A(f"          for (int i = 0; i < {D}; i++) sxn[i] = xnew[i];       // xk_n: becomes xk1_n of the next (older) step")
    A("        }")
    A("#pragma unroll")
    A(f"        for (int it = 0; it < {IT}; it++) {{")
    A("          const int idx = le + 64 * it;")
    A(f"          if (idx < nv) out1[idx] = v[it] + s_I[(idx / {EE}) * RTS3_IMG + idx % {EE}];")
    A("        }")
  A("      }")
  A("      rn::wave_lds_sync();")
  A("      RN_RTS_STAMP(10);")
  A("    }")
  A("    // ---- the oldest smoothed state goes out un-normalised (ekf_sym.py:665-667 never reaches it); its covariance left above ----")
  A(f"    for (int i = lane; i < cnt * {D}; i += 64) xs[base * {D} + i] = s_xn[i];")
  A("    rn::wave_lds_sync();")
  A("  }")
  A("}")
  return "\n".join(b)


def launch(spec):
  fpw = layout(spec)[2]
  return f"""  const int64_t tiles = (n + {fpw - 1}) / {fpw};
  hipLaunchKernelGGL(k_rts3, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     xf, Pf, ts, T, Q, n, norm_quats, xs, Ps, x_last, P_last);"""
