"""Kernel family W ("lane group per filter"), state-resident structure: one 32-lane half-wavefront per filter,
lane c owns row c of P, the nominal state x replicated in every lane of the group.

Used when the covariance does not fit one lane's registers (live: D=23, E=22 -> P is 484 doubles).  This structure
is what the FUSED MULTI-STEP kernel `k_run` uses: P rows and x stay in VGPRs for T steps, only z / y and the optional
trace cross HBM.  (The step-granular kernels are the three-phase ones of emit_wide2.py; step kernels in this structure
were the first version, 2.6x slower, and are gone.)

  predict (ekf_c.c:8-33)    A = P F^T   row-local sparse mat-vec (F has ~33 non-trivial entries of 484)
                            --LDS transpose-->  column c of A;  P' = F A + dt Q  column-local
                            --LDS transpose-->  row c of P'
  update  (ekf_c.c:37-121)  G[:,c] = He P[:,c] (column-local), Gt[:,c] = He P[c,:]^T (row-local);
                            G is broadcast through LDS so every lane forms S = G He^T + R redundantly and
                            factors the Z x Z matrix in registers (no cross-lane reduction);
                            K[c,:] = S^-1 Gt[:,c];  B[c,:] = P[c,:] - K[c,:] G;  C[c,:] = B[c,:] He^T;
                            D[c,:] = K[c,:] R - C[c,:];  K broadcast;  P'[c,:] = B[c,:] + D[c,:] K^T
                            (the Joseph form of :115 with its rank-Z structure, see emit_small.py);
                            dx broadcast; x' = err_fun(x, dx) redundantly per lane.
The x-dependent scalars (f, F, h, H.H_mod, err_fun) are CSE'd straight-line code evaluated by every lane of the
group; with the state resident there is no cheaper place to evaluate them.
"""
import sympy as sp

from rednose_amd.codegen.lower import Block, vector_names
from rednose_amd.codegen.emit_common import SMat, term, sum_terms

G_LANES = 32
FPW = 2          # filters per wavefront


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def _even(n):
  return n + (n & 1)


def predict_fn(spec):
  D, E, M = spec.dim_x, spec.dim_err, spec.dim_main_err
  names = {**vector_names(spec.x_sym, 'x'), spec.dt_sym: 'dt'}
  blk = Block(names, tmp_prefix="pt")
  for i in range(D):
    blk.add(f"xn_{i}", spec.f_sym[i])
  fmtF = lambda i, j: f"F_{i}_{j}"  # noqa: E731
  for i in range(M):
    for j in range(M):
      blk.add(fmtF(i, j), spec.F_sym[i, j])
  stmts, st = blk.lower()
  F = SMat.identity_padded(SMat.from_structure(M, M, st, fmtF), E)
  b = list(stmts)
  # a = F row  (row c of P F^T)
  b.append(f"double a[{E}];")
  for i in range(E):
    b.append(f"a[{i}] = {sum_terms(term(cf, f'row[{k}]') for k, cf in F.row_nz(i))};")
  b.append("if (act) {")
  b.append("#pragma unroll")
  b.append(f"  for (int i = 0; i < {E}; i++) sP[cc * {E} + i] = a[i];")
  b.append("}")
  b.append("rn::wave_lds_sync();")
  b.append("#pragma unroll")
  b.append(f"for (int k = 0; k < {E}; k++) a[k] = sP[k * {E} + cc];        // column c of P F^T")
  for i in range(E):
    b.append(f"col[{i}] = {sum_terms(term(cf, f'a[{k}]') for k, cf in F.row_nz(i))} + dt*sQ[{i} * {E} + cc];")
  b.append("rn::wave_lds_sync();")
  b.append("if (act) {")
  b.append("#pragma unroll")
  b.append(f"  for (int k = 0; k < {E}; k++) sP[k * {E} + cc] = col[k];")
  b.append("}")
  b.append("rn::wave_lds_sync();")
  b.append("if (WANT_ROW) {")
  b.append("#pragma unroll")
  b.append(f"  for (int j = 0; j < {E}; j++) row[j] = sP[cc * {E} + j];")
  b.append("}")
  for i in range(D):
    kind, val = st[f"xn_{i}"]
    b.append(f"x[{i}] = xn_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  head = (f"template <bool WANT_ROW>\n__device__ __forceinline__ void predict_wide(double (&x)[{D}], double (&row)[{E}], double (&col)[{E}], "
          "double* sP, const double* sQ, const double dt, const int cc, const bool act) {")
  return "\n".join([head] + _ind(b) + ["}"])


EADIM = 3        # extra-argument dimension of feature-track kinds, hard-coded in the reference (ekf_sym.py:151)


def ea_dim(k):
  return 0 if k.ea_sym is None else int(sp.Matrix(k.ea_sym).shape[0])


def update_fn(spec, k):
  """update_{kind}_wide.  Feature-track kinds of MSCKF models (k.He_sym): residual, H and R go to the left null space of the
  extra-argument Jacobian (ekf_c.c:66-76) exactly as in the step kernels (emit_wide2._lean_update): Householder reflectors of
  Hea, applied to y, to the columns of G / Gt and to the rows of He P He^T; everything after that is the ordinary update with
  Z - EADIM rows.  Here every lane of the group evaluates the scalars (state replicated), so it builds the reflectors itself."""
  D, E, Zf = spec.dim_x, spec.dim_err, k.zdim
  feat = k.He_sym is not None
  Z = Zf - EADIM if feat else Zf
  EA = ea_dim(k)
  names = dict(vector_names(spec.x_sym, 'x'))
  names.update(vector_names(k.ea_sym, 'ea'))
  Herr = sp.Matrix(k.H_sym) * sp.Matrix(spec.H_mod_sym)
  blk = Block(names, tmp_prefix="ut")
  for i in range(Zf):
    blk.add(f"hx_{i}", k.h_sym[i])
  fmtH = lambda i, j: f"He_{i}_{j}"  # noqa: E731
  for i in range(Zf):
    for j in range(E):
      blk.add(fmtH(i, j), Herr[i, j])
  if feat:
    assert tuple(k.He_sym.shape) == (Zf, EADIM), "feature-track kinds take EADIM = 3 extra arguments (ekf_sym.py:151)"
    for i in range(Zf):
      for j in range(EADIM):
        blk.add(f"Hea_{i}_{j}", k.He_sym[i, j])
  stmts, st = blk.lower()
  He = SMat.from_structure(Zf, E, st, fmtH)
  val = lambda nm: nm if st[nm][0] == 'expr' else repr(float(st[nm][1] or 0.0))  # noqa: E731
  b = list(stmts)
  if feat:
    b.append(f"double yf[{Zf}] = {{{', '.join(f'z[{i}] - ' + val(f'hx_{i}') for i in range(Zf))}}};")
    hea = ", ".join(("0.0" if st[f"Hea_{i}_{j}"][0] == 'zero' else ("1.0" if st[f"Hea_{i}_{j}"][0] == 'one' else val(f"Hea_{i}_{j}")))
                    for i in range(Zf) for j in range(EADIM))
    b += [f"double Hea[{Zf * EADIM}] = {{{hea}}};", f"double u[{EADIM * Zf}], beta[{EADIM}];",
          f"const bool ok = rn::householder_qr<{Zf}, {EADIM}>(Hea, u, beta);",
          f"rn::apply_reflectors<{Zf}, {EADIM}>(u, beta, yf);"]
    for i in range(Z):
      b.append(f"const double y_{i} = ok ? yf[{EADIM + i}] : 0.0;")
    b += [f"double Rm[{Zf * Zf}];", "#pragma unroll", f"for (int i = 0; i < {Zf * Zf}; i++) Rm[i] = Rin[i];",
          f"rn::project_noise<{Zf}, {EADIM}>(u, beta, Rm);", f"double R[{Z * Z}];"]
    for a in range(Z):
      for c in range(Z):
        b.append(f"R[{a * Z + c}] = Rm[{(EADIM + a) * Zf + EADIM + c}];")
    b.append(f"double G0[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'col[{kk}]') for kk, cf in He.row_nz(zi)) for zi in range(Zf)) + "};")
    b.append(f"double Gt0[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'row[{kk}]') for kk, cf in He.row_nz(zi)) for zi in range(Zf)) + "};")
    b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(u, beta, G0);")
    b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(u, beta, Gt0);")
    for zi in range(Z):
      b.append(f"const double G_{zi} = G0[{EADIM + zi}], Gt_{zi} = Gt0[{EADIM + zi}];")
  else:
    for i in range(Z):
      b.append(f"const double y_{i} = z[{i}] - {val(f'hx_{i}')};")
    for zi in range(Z):
      nz = He.row_nz(zi)
      b.append(f"const double G_{zi} = {sum_terms(term(cf, f'col[{kk}]') for kk, cf in nz)};")
      b.append(f"const double Gt_{zi} = {sum_terms(term(cf, f'row[{kk}]') for kk, cf in nz)};")
  b.append("if (act) { " + " ".join(f"sG[{zi} * {E} + cc] = G_{zi};" for zi in range(Z)) + " }")
  b.append("rn::wave_lds_sync();")
  b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  if feat:
    for zi in range(Z):
      b.append("{")
      b.append(f"  double m[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in He.row_nz(w)) for w in range(Zf)) + "};")
      b.append(f"  rn::apply_reflectors<{Zf}, {EADIM}>(u, beta, m);")
      b += ["#pragma unroll", f"  for (int w = 0; w < {Z}; w++) HPH[{zi * Z} + w] = m[{EADIM} + w];", "}"]
  else:
    for zi in range(Z):
      for w in range(Z):
        b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in He.row_nz(w))};")
  b.append("#pragma unroll")
  b.append(f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}")
  b.append(f"rn::spd_factor<{Z}>(S, L, iL);")
  b.append("int gated = 0;")
  if k.maha_test:
    b.append("{")
    b.append(f"  double v[{Z}] = {{{', '.join(f'y_{i}' for i in range(Z))}}};")
    b.append(f"  rn::spd_forward<{Z}>(L, iL, v);")
    b.append("  const double d2 = " + " + ".join(f"v[{i}]*v[{i}]*iL[{i}]" for i in range(Z)) + ";")
    b.append(f"  if (d2 > {k.maha_thresh!r}) {{")
    b.append("    gated = 1;")
    b.append("#pragma unroll")
    b.append(f"    for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = 1.0e16 * Rl[i]; S[i] = HPH[i] + Rl[i]; }}")
    b.append(f"    rn::spd_factor<{Z}>(S, L, iL);")
    b.append("  }")
    b.append("}")
  b.append(f"double kk[{Z}] = {{{', '.join(f'Gt_{zi}' for zi in range(Z))}}};")
  b.append(f"rn::spd_solve<{Z}>(L, iL, kk);                       // K[c][:]")
  if feat:     # the reference's numpy path ignores a measurement whose null-space projection failed (ekf_sym.py:589-591)
    b += ["if (!ok) {", "#pragma unroll", f"  for (int i = 0; i < {Z}; i++) kk[i] = 0.0;", "}"]
  b.append("const double dxc = " + " + ".join(f"kk[{zi}]*y_{zi}" for zi in range(Z)) + ";")
  # B row
  b.append("#pragma unroll")
  b.append(f"for (int j = 0; j < {E}; j++) row[j] -= " + " + ".join(f"kk[{zi}]*sG[{zi} * {E} + j]" for zi in range(Z)) + ";")
  if feat:      # C[c][:] = (B He^T) projected: reflect the Zf-vector of row-local dot products, keep the last Z entries
    b.append(f"double Cf[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'row[{j}]') for j, cf in He.row_nz(w)) for w in range(Zf)) + "};")
    b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(u, beta, Cf);")
  for zi in range(Z):
    c = f"Cf[{EADIM + zi}]" if feat else sum_terms(term(cf, f"row[{j}]") for j, cf in He.row_nz(zi))
    kr = " + ".join(f"kk[{w}]*Rl[{w * Z + zi}]" for w in range(Z))
    b.append(f"const double Dm_{zi} = " + ("!ok ? 0.0 : " if feat else "") + f"({kr}) - ({c});")
  b.append("if (act) { " + " ".join(f"sK[{zi} * {E} + cc] = kk[{zi}];" for zi in range(Z)) + " sdx[cc] = dxc; }")
  b.append("rn::wave_lds_sync();")
  b.append("#pragma unroll")
  b.append(f"for (int j = 0; j < {E}; j++) row[j] += " + " + ".join(f"Dm_{zi}*sK[{zi} * {E} + j]" for zi in range(Z)) + ";")
  # error injection, replicated
  b.append(f"double dxa[{E}];")
  b.append("#pragma unroll")
  b.append(f"for (int j = 0; j < {E}; j++) dxa[j] = sdx[j];")
  nom, delta = spec.err_eqs[1], spec.err_eqs[2]
  enames = dict(vector_names(nom, 'x'))
  enames.update({(delta, i, 0): f"dxa[{i}]" for i in range(E)})
  eblk = Block(enames, tmp_prefix="et")
  for i in range(D):
    eblk.add(f"xi_{i}", sp.Matrix(spec.err_eqs[0])[i])
  estmts, est = eblk.lower()
  b += estmts
  for i in range(D):
    kind, v_ = est[f"xi_{i}"]
    b.append(f"x[{i}] = xi_{i};" if kind == 'expr' else f"x[{i}] = {float(v_)!r};")
  for i in range(Z):
    b.append(f"z[{i}] = y_{i};")        # feature kinds: Z - EADIM residual rows, the tail of z is left alone (ekf_c.c:120)
  b.append("rn::wave_lds_sync();      // broadcast buffers are free again")
  b.append("return gated" + (" | (ok ? 0 : 4);" if feat else ";"))
  ea_arg = f", const double (&ea)[{EA}]" if EA else ""
  r_arg = f"const double (&Rin)[{Zf * Zf}]" if feat else f"const double (&R)[{Z * Z}]"
  head = (f"__device__ __forceinline__ int update_{k.kind}_wide(double (&x)[{D}], double (&row)[{E}], const double (&col)[{E}], "
          f"double (&z)[{Zf}], {r_arg}{ea_arg}, double* sG, double* sK, double* sdx, const int cc, const bool act) {{")
  return "\n".join([head] + _ind(b) + ["}"]), He


def run_kinds(spec):
  """Kinds the fused multi-step run serves: all of them.  The schedule (kinds[t], dts[t], R[t], augment[t]) is shared by all
  filters; kinds that take extra arguments (MSCKF feature tracks: the landmark) read them per filter and step from the
  (T, n, EA) array `ea` of the entry point (flag bit 8 when that array is missing)."""
  return list(spec.kinds)


def ea_max(spec):
  return max([ea_dim(k) for k in spec.kinds] + [0])


def kernels(spec):
  """Device functions + the fused multi-step kernel of the state-resident structure."""
  D, E = spec.dim_x, spec.dim_err
  out = [f"constexpr int GL = {G_LANES};    // lanes per filter", f"constexpr int FPW = {FPW};   // filters per wavefront", ""]
  out.append(predict_fn(spec))
  for k in run_kinds(spec):
    utxt, _ = update_fn(spec, k)
    out.append(utxt)
  quat = "".join(f" rn::normalize_quat<{D}>(x, {q});" for q in spec.quaternion_idxs)
  norm = f"if (norm_quats) {{{quat} }}" if spec.quaternion_idxs else "(void)norm_quats;"
  out.append(run_kernel(spec, norm))
  return "\n".join(out)


def run_kernel(spec, norm):
  """T steps per launch: rows of P stay in VGPRs (x replicated), only z/y and the optional trace touch HBM."""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  DP = _even(D)
  zmax = max(k.zdim for k in spec.kinds)
  EAM = ea_max(spec)
  cases = []
  for k in run_kinds(spec):
    Zf = k.zdim
    EA = ea_dim(k)
    ea_load = ""
    ea_arg = ""
    if EA:
      ea_load = f"""          if (gea == nullptr) {{ fl = 8; break; }}
          double eak[{EA}];
#pragma unroll
          for (int i = 0; i < {EA}; i++) eak[i] = gea[((int64_t)t * n + base + gg) * {EAM} + i];
"""
      ea_arg = ", eak"
    cases.append(f"""        case {k.kind}: {{
{ea_load}          double zk[{Zf}], Rk[{Zf * Zf}];
#pragma unroll
          for (int i = 0; i < {Zf}; i++) zk[i] = z[i];
#pragma unroll
          for (int i = 0; i < {Zf * Zf}; i++) Rk[i] = gR[t * {zmax * zmax} + i];
          fl = update_{k.kind}_wide(x, row, col, zk, Rk{ea_arg}, s_G + gg * {zmax * E}, s_K + gg * {zmax * E}, s_dx + gg * {E}, cc, on);
#pragma unroll
          for (int i = 0; i < {Zf}; i++) z[i] = zk[i];
          break;
        }}""")
  aug = ""
  if spec.N > 0:
    d1, d2, d3, d4 = spec.dim_main, spec.dim_main_err, spec.dim_augment, spec.dim_augment_err
    src_x = [i if i < d1 else (i + d3 if i < D - d3 else i - (D - d3)) for i in range(D)]

    def se(i):
      r = i if i < E - d4 else i - (E - d4)
      return r if r < d2 else r + d4
    aug = f"""
      // MSCKF window shift after this step (EKF_sym.augment, ekf_sym.py:365-391; the schedule's augment[t]): a fixed permutation
      // of the replicated state, and of the rows / columns of P through LDS (the trace above holds the estimate BEFORE the shift,
      // like the reference's Estimate)
      if (augs != nullptr && augs[t] != 0) {{
        double xo[{D}];
#pragma unroll
        for (int i = 0; i < {D}; i++) xo[i] = x[i];
{chr(10).join(f"        x[{i}] = xo[{src_x[i]}];" for i in range(D) if src_x[i] != i)}
        if (on) {{
#pragma unroll
          for (int j = 0; j < {E}; j++) s_P[g * {EE} + c * {E} + j] = row[j];
        }}
        rn::wave_lds_sync();
        const int sr = (cc < {E - d4} ? cc : cc - {E - d4});
        const int srow = sr < {d2} ? sr : sr + {d4};
{chr(10).join(f"        row[{j}] = s_P[gg * {EE} + srow * {E} + {se(j)}];" for j in range(E))}
        rn::wave_lds_sync();
      }}"""
  # the dt == 0 shortcut below is only emitted for models whose predict(dt = 0) is symbolically the identity
  id0_guard = "" if spec.identity_at_dt0() else "true || "
  return f"""
// ---- fused multi-step run: kinds[t], dts[t] shared by all filters; z is (T, n, {zmax}) in: z, out: y -----------
__global__ __launch_bounds__(64) void k_run(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* __restrict__ gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    double* __restrict__ tx, double* __restrict__ tP, const double* __restrict__ gea, const int32_t* __restrict__ augs) {{
  (void)gea; (void)augs;
  __shared__ __attribute__((aligned(16))) double s_P[FPW * {EE}];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  __shared__ __attribute__((aligned(16))) double s_x[FPW * {DP}];
  __shared__ __attribute__((aligned(16))) double s_z[FPW * {zmax} + 2];
  __shared__ __attribute__((aligned(16))) double s_G[FPW * {zmax * E}];
  __shared__ __attribute__((aligned(16))) double s_K[FPW * {zmax * E}];
  __shared__ __attribute__((aligned(16))) double s_dx[FPW * {E}];
  const int lane = threadIdx.x;
  const int g = lane / GL;
  const int c = lane % GL;
  const bool act = c < {E};
  const int cc = act ? c : 0;
  rn::copy_g2l<{EE}>(gQ, {EE}, s_Q, lane);
  const int64_t tiles = (n + FPW - 1) / FPW;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * FPW;
    const int cnt = (n - base) < FPW ? (int)(n - base) : FPW;
    const int gg = g < cnt ? g : 0;
    const bool on = act && g < cnt;
    rn::copy_g2l<FPW * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    rn::copy_g2l<FPW * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    if (lane < cnt * {zmax}) s_z[lane] = gz[base * {zmax} + lane];
    rn::wave_lds_sync();
    double x[{D}], row[{E}], col[{E}], z[{zmax}];
#pragma unroll
    for (int i = 0; i < {D}; i++) x[i] = s_x[gg * {D} + i];
#pragma unroll
    for (int j = 0; j < {E}; j++) row[j] = s_P[gg * {EE} + cc * {E} + j];
    for (int64_t t = 0; t < T; t++) {{
#pragma unroll
      for (int i = 0; i < {zmax}; i++) z[i] = s_z[gg * {zmax} + i];
      rn::wave_lds_sync();
      double zn = 0.0;                                 // next step's observation, in flight during this step
      if (t + 1 < T && lane < cnt * {zmax}) zn = gz[((t + 1) * n + base) * {zmax} + lane];
      const int kind = kinds[t];
      const double dt = dts[t];
      if ({id0_guard}dt != 0.0) {{
        predict_wide<true>(x, row, col, s_P + gg * {EE}, s_Q, dt, cc, on);
      }} else {{
        // predict(dt = 0) is the identity on (x, P) for finite states (F = I, dt Q = 0): only the column view of P
        // that the update needs is rebuilt, through one LDS transpose
        if (on) {{
#pragma unroll
          for (int j = 0; j < {E}; j++) s_P[g * {EE} + c * {E} + j] = row[j];
        }}
        rn::wave_lds_sync();
#pragma unroll
        for (int k = 0; k < {E}; k++) col[k] = s_P[gg * {EE} + k * {E} + cc];
        rn::wave_lds_sync();
      }}
      {norm}
      int fl = 0;
      switch (kind) {{
{chr(10).join(cases)}
        default: fl = 8; break;      // kind not available in the fused run (unknown, or it takes extra arguments)
      }}
      {norm}
      if (c == 0 && g < cnt) {{
#pragma unroll
        for (int i = 0; i < {zmax}; i++) s_z[g * {zmax} + i] = z[i];
        if (tx != nullptr) {{
#pragma unroll
          for (int i = 0; i < {D}; i++) s_x[g * {D} + i] = x[i];
        }}
        if (flags != nullptr) flags[t * n + base + g] = (uint8_t)fl;
      }}
      if (tP != nullptr && on) {{
#pragma unroll
        for (int j = 0; j < {E}; j++) s_P[g * {EE} + c * {E} + j] = row[j];
      }}
      rn::wave_lds_sync();
      if (lane < cnt * {zmax}) gz[(t * n + base) * {zmax} + lane] = s_z[lane];
      if (tx != nullptr) rn::copy_l2g<FPW * {D}>(tx + (t * n + base) * {D}, cnt * {D}, s_x, lane);
      if (tP != nullptr) rn::copy_l2g<FPW * {EE}>(tP + (t * n + base) * {EE}, cnt * {EE}, s_P, lane);
      rn::wave_lds_sync();
      if (lane < cnt * {zmax}) s_z[lane] = zn;
      rn::wave_lds_sync();{aug}
    }}
    if (on) {{
#pragma unroll
      for (int j = 0; j < {E}; j++) s_P[g * {EE} + c * {E} + j] = row[j];
    }}
    if (c == 0 && g < cnt) {{
#pragma unroll
      for (int i = 0; i < {D}; i++) s_x[g * {D} + i] = x[i];
    }}
    rn::wave_lds_sync();
    rn::copy_l2g<FPW * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    rn::copy_l2g<FPW * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::wave_lds_sync();
  }}
}}
"""


def launch_run():
  return """  const int64_t tiles = (n + 1) / 2;
  hipLaunchKernelGGL(k_run, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, trace_x, trace_P, ea, augment);"""
