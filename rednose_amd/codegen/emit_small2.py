"""Kernel family S, step-granular kernels, lane-PAIR per filter: two lanes share one filter, each owns half the rows of P.

STATUS: experiment, NOT the default (tuning knob small_lpf=2).  Parity-green on MI355X, but measured 9.8-10.0 us per
launch on kinematic6 / 65 536 filters against 9.4 us for the lane-per-filter kernels of emit_small.py.

Why: with one lane per filter (emit_small.py) a batch of 65 536 filters is exactly 1 024 wavefronts for 1 024
SIMDs -- one wave per SIMD, running load -> compute -> store serially.  The round-1 PMC profile of that kernel shows
31 % of the wave's cycles in issue stalls of dependent fp64 chains and 45 % waiting on memory, with nothing else on
the SIMD to fill either.  Splitting each filter over an (even, odd) lane pair halves the per-wave instruction
stream and doubles the number of wavefronts (2 048 tiles of 32 filters), so every SIMD hosts two waves that cover
each other's stalls and memory phases.  The price is cross-lane traffic: the partner's rows of P for F.P, the
partial sums of G = He.P, dx and the partner's rows of K -- all moved with DPP quad_perm swaps (rn::pair_xchg),
no LDS.  The algebra and its order of operations are those of emit_small.py (same docstring, same reference
lines); only the partial sums of G are associated differently (own rows + partner rows).

Rows are split in two contiguous halves: lane h of the pair owns rows [h*E/2, (h+1)*E/2).  Coefficients that depend
on the owned row index (rows of F, columns of He) are selected per lane with v_cndmask from the two structural
candidates; a coefficient that is structurally zero for both lanes still emits nothing.
"""
import sympy as sp

from rednose_amd.codegen.lower import Block, vector_names
from rednose_amd.codegen.emit_common import SMat, term, sum_terms, coef_text

TF = 32      # filters per wavefront tile


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def _pair_coef(e0, e1, name, body):
  """Coefficient that is e0 on lane h=0 and e1 on lane h=1 (SMat entries).  Returns an SMat-style entry usable with
  term(); emits a select into `body` only when the two candidates differ."""
  if e0 is None and e1 is None:
    return None
  if e0 == e1:
    return e0
  body.append(f"const double {name} = h ? {coef_text(e1)} : {coef_text(e0)};")
  return ('var', name)


def predict_pair(spec):
  D, E, M = spec.dim_x, spec.dim_err, spec.dim_main_err
  HR = E // 2
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
  # full P from own + partner rows
  b.append(f"double Ptop[{HR * E}], Pbot[{HR * E}];")
  b.append("#pragma unroll")
  b.append(f"for (int i = 0; i < {HR * E}; i++) {{ const double pp = rn::pair_xchg(Po[i]); Ptop[i] = h ? pp : Po[i]; Pbot[i] = h ? Po[i] : pp; }}")
  PF = lambda k, j: f"Ptop[{k * E + j}]" if k < HR else f"Pbot[{(k - HR) * E + j}]"  # noqa: E731
  # T = F P, own rows
  for r in range(HR):
    coefs = []
    for k in range(E):
      c = _pair_coef(F.e[r][k], F.e[HR + r][k], f"cf_{r}_{k}", b)
      if c is not None:
        coefs.append((k, c))
    for j in range(E):
      b.append(f"const double T_{r}_{j} = {sum_terms(term(c, PF(k, j)) for k, c in coefs)};")
  # P' = T F^T + dt Q, own rows (rows of F indexed by the COLUMN j of the result: same for both lanes)
  b.append(f"const double* qrow = sQ + h * {HR * E};")
  for r in range(HR):
    for j in range(E):
      s = sum_terms(term(c, f"T_{r}_{k}") for k, c in F.row_nz(j))
      b.append(f"const double Pn_{r}_{j} = {s} + dt*qrow[{r * E + j}];")
  for r in range(HR):
    for j in range(E):
      b.append(f"Po[{r * E + j}] = Pn_{r}_{j};")
  for i in range(D):
    kind, val = st[f"xn_{i}"]
    b.append(f"x[{i}] = xn_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  head = (f"__device__ __forceinline__ void predict_pair(double (&x)[{D}], double (&Po)[{HR * E}], const double* sQ, "
          "const double dt, const int h) {")
  return "\n".join([head] + _ind(b) + ["}"])


def update_pair(spec, k):
  D, E, Z = spec.dim_x, spec.dim_err, k.zdim
  HR = E // 2
  names = dict(vector_names(spec.x_sym, 'x'))
  Herr = sp.Matrix(k.H_sym) * sp.Matrix(spec.H_mod_sym)
  blk = Block(names, tmp_prefix="ut")
  for i in range(Z):
    blk.add(f"hx_{i}", k.h_sym[i])
  fmtH = lambda i, j: f"He_{i}_{j}"  # noqa: E731
  for i in range(Z):
    for j in range(E):
      blk.add(fmtH(i, j), Herr[i, j])
  stmts, st = blk.lower()
  He = SMat.from_structure(Z, E, st, fmtH)
  b = list(stmts)
  for i in range(Z):
    kind, val = st[f"hx_{i}"]
    hx = f"hx_{i}" if kind == 'expr' else repr(float(val))
    b.append(f"const double y_{i} = z[{i}] - {hx};")
  # G = He P: partial sums over the rows this lane owns, completed with the partner's partial sums
  for zi in range(Z):
    coefs = []
    for r in range(HR):
      c = _pair_coef(He.e[zi][r], He.e[zi][HR + r], f"hc_{zi}_{r}", b)
      if c is not None:
        coefs.append((r, c))
    for j in range(E):
      b.append(f"const double Gp_{zi}_{j} = {sum_terms(term(c, f'Po[{r * E + j}]') for r, c in coefs)};")
      b.append(f"const double G_{zi}_{j} = Gp_{zi}_{j} + rn::pair_xchg(Gp_{zi}_{j});")
  # Gt = He P^T, own rows
  for zi in range(Z):
    nz = He.row_nz(zi)
    for r in range(HR):
      b.append(f"const double Gt_{zi}_{r} = {sum_terms(term(c, f'Po[{r * E + kk}]') for kk, c in nz)};")
  b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  for zi in range(Z):
    for w in range(Z):
      b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(c, f'G_{zi}_{j}') for j, c in He.row_nz(w))};")
  b.append("#pragma unroll")
  b.append(f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}")
  b.append(f"rn::spd_factor<{Z}>(S, L, iL);")
  b.append("int gated = 0;")
  if k.maha_test:
    b += ["{", f"  double v[{Z}] = {{{', '.join(f'y_{i}' for i in range(Z))}}};", f"  rn::spd_forward<{Z}>(L, iL, v);",
          "  const double d2 = " + " + ".join(f"v[{i}]*v[{i}]*iL[{i}]" for i in range(Z)) + ";", f"  if (d2 > {k.maha_thresh!r}) {{", "    gated = 1;",
          "#pragma unroll", f"    for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = 1.0e16 * Rl[i]; S[i] = HPH[i] + Rl[i]; }}",
          f"    rn::spd_factor<{Z}>(S, L, iL);", "  }", "}"]
  # K, own rows
  for r in range(HR):
    b.append(f"double k_{r}[{Z}] = {{{', '.join(f'Gt_{zi}_{r}' for zi in range(Z))}}};")
    b.append(f"rn::spd_solve<{Z}>(L, iL, k_{r});")
  # dx: own rows, then the full vector (both lanes inject the error into their copy of x)
  b.append(f"double dxt[{HR}], dxb[{HR}];")
  for r in range(HR):
    b.append(f"{{ const double dxo = " + " + ".join(f"k_{r}[{zi}]*y_{zi}" for zi in range(Z)) + f"; const double dxp = rn::pair_xchg(dxo); dxt[{r}] = h ? dxp : dxo; dxb[{r}] = h ? dxo : dxp; }}")
  DX = lambda i: f"dxt[{i}]" if i < HR else f"dxb[{i - HR}]"  # noqa: E731
  nom, delta = spec.err_eqs[1], spec.err_eqs[2]
  enames = dict(vector_names(nom, 'x'))
  enames.update({(delta, i, 0): DX(i) for i in range(E)})
  eblk = Block(enames, tmp_prefix="et")
  for i in range(D):
    eblk.add(f"xi_{i}", sp.Matrix(spec.err_eqs[0])[i])
  estmts, est = eblk.lower()
  b += estmts
  # B = P - K G, own rows
  for r in range(HR):
    for j in range(E):
      b.append(f"Po[{r * E + j}] -= " + " + ".join(f"k_{r}[{zi}]*G_{zi}_{j}" for zi in range(Z)) + ";")
  # C = B He^T, D = K R - C, own rows
  for r in range(HR):
    for zi in range(Z):
      c = sum_terms(term(cf, f"Po[{r * E + j}]") for j, cf in He.row_nz(zi))
      kr = " + ".join(f"k_{r}[{w}]*Rl[{w * Z + zi}]" for w in range(Z))
      b.append(f"const double Dm_{r}_{zi} = ({kr}) - ({c});")
  # all rows of K: own + partner
  b.append(f"double Kt[{HR * Z}], Kb[{HR * Z}];")
  for r in range(HR):
    for zi in range(Z):
      b.append(f"{{ const double kp = rn::pair_xchg(k_{r}[{zi}]); Kt[{r * Z + zi}] = h ? kp : k_{r}[{zi}]; Kb[{r * Z + zi}] = h ? k_{r}[{zi}] : kp; }}")
  KF = lambda j, zi: f"Kt[{j * Z + zi}]" if j < HR else f"Kb[{(j - HR) * Z + zi}]"  # noqa: E731
  for r in range(HR):
    for j in range(E):
      b.append(f"Po[{r * E + j}] += " + " + ".join(f"Dm_{r}_{zi}*{KF(j, zi)}" for zi in range(Z)) + ";")
  for i in range(D):
    kind, val = est[f"xi_{i}"]
    b.append(f"x[{i}] = xi_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  for i in range(Z):
    b.append(f"z[{i}] = y_{i};")
  b.append("return gated;")
  head = (f"__device__ __forceinline__ int update_{k.kind}_pair(double (&x)[{D}], double (&Po)[{HR * E}], double (&z)[{Z}], "
          f"const double (&R)[{Z * Z}], const int h) {{")
  return "\n".join([head] + _ind(b) + ["}"])


def kernels(spec):
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  HR = E // 2
  assert E % 2 == 0
  out = [f"// ---- family S, lane-pair step kernels: {TF} filters per wavefront, lane h of a pair owns rows [h*{HR}, h*{HR}+{HR}) ----",
         f"constexpr int TF2 = {TF};", predict_pair(spec)]
  for k in spec.kinds:
    out.append(update_pair(spec, k))
  quat = "".join(f" rn::normalize_quat<{D}>(x, {q});" for q in spec.quaternion_idxs)
  norm = f"if (norm_quats) {{{quat} }}" if spec.quaternion_idxs else "(void)norm_quats;"

  out.append(f"""
__global__ __launch_bounds__(64) void k_predict(double* __restrict__ gx, double* __restrict__ gP,
    const double* __restrict__ gQ, const double* __restrict__ gdt, const double dt_scalar, const int64_t n,
    const int norm_quats) {{
  __shared__ __attribute__((aligned(16))) double s_x[TF2 * {D} + 2];
  __shared__ __attribute__((aligned(16))) double s_P[TF2 * {EE} + 2];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  const int lane = threadIdx.x;
  const int f = lane >> 1, h = lane & 1;
  for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];
  const int64_t tiles = (n + TF2 - 1) / TF2;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * TF2;
    const int cnt = (n - base) < TF2 ? (int)(n - base) : TF2;
    rn::async_copy_g2l<TF2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::async_copy_g2l<TF2 * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    const double dt = (gdt != nullptr && f < cnt) ? gdt[base + f] : dt_scalar;
    rn::async_wait();
    rn::wave_lds_sync();
    double x[{D}], Po[{HR * E}];
#pragma unroll
    for (int i = 0; i < {D}; i++) x[i] = s_x[f * {D} + i];
#pragma unroll
    for (int i = 0; i < {HR * E}; i++) Po[i] = s_P[f * {EE} + h * {HR * E} + i];
    predict_pair(x, Po, s_Q, dt, h);
    {norm}
    rn::wave_lds_sync();
#pragma unroll
    for (int i = 0; i < {HR * E}; i++) s_P[f * {EE} + h * {HR * E} + i] = Po[i];
    if (h == 0) {{
#pragma unroll
      for (int i = 0; i < {D}; i++) s_x[f * {D} + i] = x[i];
    }}
    rn::wave_lds_sync();
    rn::copy_l2g<TF2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::copy_l2g<TF2 * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    rn::wave_lds_sync();
  }}
}}
""")
  for k in spec.kinds:
    Z = k.zdim
    ZZ = Z * Z
    out.append(f"""
template <bool DO_PREDICT>
__global__ __launch_bounds__(64) void k_step_{k.kind}(double* __restrict__ gx, double* __restrict__ gP,
    double* __restrict__ gz, const double* __restrict__ gR, const int r_per_filter, const double* __restrict__ gea,
    const double* __restrict__ gQ, const double* __restrict__ gdt, const double dt_scalar, const int64_t n,
    const int norm_quats, uint8_t* __restrict__ flags) {{
  __shared__ __attribute__((aligned(16))) double s_x[TF2 * {D} + 2];
  __shared__ __attribute__((aligned(16))) double s_P[TF2 * {EE} + 2];
  __shared__ __attribute__((aligned(16))) double s_z[TF2 * {Z} + 2];
  __shared__ __attribute__((aligned(16))) double s_R[TF2 * {ZZ} + 2];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  (void)gea;
  const int lane = threadIdx.x;
  const int f = lane >> 1, h = lane & 1;
  if (DO_PREDICT) {{
    for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];
  }}
  const int64_t tiles = (n + TF2 - 1) / TF2;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * TF2;
    const int cnt = (n - base) < TF2 ? (int)(n - base) : TF2;
    rn::async_copy_g2l<TF2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::async_copy_g2l<TF2 * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    rn::async_copy_g2l<TF2 * {Z}>(gz + base * {Z}, cnt * {Z}, s_z, lane);
    if (r_per_filter) rn::async_copy_g2l<TF2 * {ZZ}>(gR + base * {ZZ}, cnt * {ZZ}, s_R, lane);
    double dt = dt_scalar;
    if (DO_PREDICT && gdt != nullptr && f < cnt) dt = gdt[base + f];
    rn::async_wait();
    rn::wave_lds_sync();
    double x[{D}], Po[{HR * E}], z[{Z}], R[{ZZ}];
#pragma unroll
    for (int i = 0; i < {D}; i++) x[i] = s_x[f * {D} + i];
#pragma unroll
    for (int i = 0; i < {HR * E}; i++) Po[i] = s_P[f * {EE} + h * {HR * E} + i];
#pragma unroll
    for (int i = 0; i < {Z}; i++) z[i] = s_z[f * {Z} + i];
#pragma unroll
    for (int i = 0; i < {ZZ}; i++) R[i] = r_per_filter ? s_R[f * {ZZ} + i] : gR[i];
    if (DO_PREDICT) {{
      predict_pair(x, Po, s_Q, dt, h);
      {norm}
    }}
    int fl = update_{k.kind}_pair(x, Po, z, R, h);
    {norm}
    rn::wave_lds_sync();
#pragma unroll
    for (int i = 0; i < {HR * E}; i++) s_P[f * {EE} + h * {HR * E} + i] = Po[i];
    if (h == 0) {{
#pragma unroll
      for (int i = 0; i < {D}; i++) s_x[f * {D} + i] = x[i];
#pragma unroll
      for (int i = 0; i < {Z}; i++) s_z[f * {Z} + i] = z[i];
      if (flags != nullptr && f < cnt) {{
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < {D}; i++) acc += x[i];
        if (!(acc - acc == 0.0)) fl |= 2;
        flags[base + f] = (uint8_t)fl;
      }}
    }}
    rn::wave_lds_sync();
    rn::copy_l2g<TF2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::copy_l2g<TF2 * {EE}>(gP + base * {EE}, cnt * {EE}, s_P, lane);
    rn::copy_l2g<TF2 * {Z}>(gz + base * {Z}, cnt * {Z}, s_z, lane);
    rn::wave_lds_sync();
  }}
}}
""")
  return "\n".join(out)


def launch_predict():
  return """  const int64_t tiles = (n + TF2 - 1) / TF2;
  hipLaunchKernelGGL(k_predict, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, dt_vec, dt, n, norm_quats);"""


def launch_step(kind, do_predict):
  tf = "true" if do_predict else "false"
  if do_predict:
    args = "x, P, z, R, r_per_filter, ea, Q, dt_vec, dt, n, norm_quats, flags"
  else:
    args = "x, P, z, R, r_per_filter, ea, nullptr, nullptr, 0.0, n, norm_quats, flags"
  return f"""  const int64_t tiles = (n + TF2 - 1) / TF2;
  hipLaunchKernelGGL(k_step_{kind}<{tf}>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     {args});"""
