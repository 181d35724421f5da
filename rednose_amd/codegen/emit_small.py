"""Kernel family S ("lane per filter"): the whole filter state lives in one lane's VGPRs.

Used when E is small (kinematic: D=E=2; kinematic6: D=E=6 -> 6 + 36 doubles = 84 VGPRs).  A
wavefront owns 64 consecutive filters: it pulls their x / P / z records out of HBM as one
contiguous, 16-byte-vectorised burst per array, transposes through its private LDS slice so that
lane l ends up holding filter l, runs the fully unrolled predict / update algebra below in
registers, and writes back the same way.  All sparsity of F and H.H_mod is resolved here, at
generation time: structural zeros emit no instruction, unit entries no multiply.

Algebra emitted (reference lines in /root/reference/rednose/templates/ekf_c.c):
  predict :8-33   x' = f(x,dt); F = F(x,dt) at the PRE-propagation state; P' = (F P) F^T + dt Q
                  (MEDIM < EDIM handled as F_full = blockdiag(F_main, I), identical to :23-26)
  update  :37-121 y = z - h(x); He = H(x) H_mod(x); G = He P; Gt = He P^T; S = G He^T + R;
                  optional gate d2 = y^T S^-1 y > thresh => R *= 1e16 (:88-94);
                  K^T = S^-1 Gt via Cholesky (:101 uses fullPivLu; S is SPD);
                  dx = K y; x = err_fun(x, dx);
                  Joseph form (:115) evaluated with its rank-Z structure:
                     B = P - K G            (= I_KH P)
                     C = B He^T             (E x Z)
                     P' = B + (K R - C) K^T (= B I_KH^T + K R K^T)
                  which is the same polynomial in the same inputs -- including the property that the
                  rounding error of B is cancelled to first order by the correction term.
"""
import sympy as sp

from rednose_amd.codegen import tuning
from rednose_amd.codegen.lower import Block, vector_names
from rednose_amd.codegen.emit_common import SMat, term, sum_terms, innovation_solver


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def predict_regs(spec, sym=False):
  """-> text of `predict_regs(x, P, Q, dt)` operating on registers.

  sym=True emits `predict_regs_sym` for the fused multi-step kernels (k_run*), whose contract is a SYMMETRIC covariance: they read
  (P + P^T) / 2 of the caller's matrix once, when the state enters the registers (include/rednose_amd_filter.h), and from then on only
  the upper triangle is read and formed, mirrored at the end of every function -- register renames the compiler drops wherever the
  lower triangle is not consumed.  kinematic6: 548 -> 446 fp64 instructions per step, 42 -> 56 G steps/s in the blocked run (MI355X,
  same call).  The step-granular kernels keep the full product: they use both halves of P exactly as the reference does
  (ekf_c.c:24)."""
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

  U = (lambda i, j: f"P[{min(i, j) * E + max(i, j)}]") if sym else (lambda i, j: f"P[{i * E + j}]")      # noqa: E731
  body = list(stmts)
  # T = F P   (rows of F that are a bare unit diagonal alias the row of P)
  T = [[None] * E for _ in range(E)]
  for i in range(E):
    nz = F.row_nz(i)
    if len(nz) == 1 and nz[0][0] == i and nz[0][1][0] == 'one':
      for j in range(E):
        T[i][j] = U(i, j)
      continue
    for j in range(E):
      body.append(f"const double T_{i}_{j} = {sum_terms(term(c, U(k, j)) for k, c in nz)};")
      T[i][j] = f"T_{i}_{j}"
  # P' = T F^T + dt Q
  newP = []
  for i in range(E):
    for j in range(i if sym else 0, E):
      s = sum_terms(term(c, T[i][k]) for k, c in F.row_nz(j))
      newP.append(f"const double Pn_{i}_{j} = {s} + dt*Q[{i * E + j}];")
  body += newP
  for i in range(E):
    for j in range(i if sym else 0, E):
      body.append(f"P[{i * E + j}] = Pn_{i}_{j};")
  if sym:
    body += [f"P[{j * E + i}] = P[{i * E + j}];" for i in range(E) for j in range(i + 1, E)]
  for i in range(D):
    kind, val = st[f"xn_{i}"]
    body.append(f"x[{i}] = xn_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  head = (f"__device__ __forceinline__ void predict_regs{'_sym' if sym else ''}(double (&x)[{D}], double (&P)[{E * E}], "
          "const double* Q, const double dt) {")
  return "\n".join([head] + _ind(body) + ["}"]), F


def update_regs(spec, k, sym=False):
  """-> text of `update_<kind>_regs(x, P, z, R)`; returns the gate flag.  sym=True: `update_<kind>_regs_sym`, see predict_regs."""
  D, E, Z = spec.dim_x, spec.dim_err, k.zdim
  names = dict(vector_names(spec.x_sym, 'x'))
  if k.ea_sym is not None:
    names.update(vector_names(k.ea_sym, 'ea'))
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
  # G = He P ; Gt = He P^T   (sym: P = P^T, upper triangle only -- see predict_regs -- and Gt IS G)
  U = (lambda i, j: f"P[{min(i, j) * E + max(i, j)}]") if sym else (lambda i, j: f"P[{i * E + j}]")      # noqa: E731
  for zi in range(Z):
    nz = He.row_nz(zi)
    for j in range(E):
      b.append(f"const double G_{zi}_{j} = {sum_terms(term(c, U(kk, j)) for kk, c in nz)};")
      b.append(f"const double Gt_{zi}_{j} = " + (f"G_{zi}_{j};" if sym else f"{sum_terms(term(c, f'P[{j * E + kk}]') for kk, c in nz)};"))
  # HPHt, S, Cholesky, optional gate
  b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  for zi in range(Z):
    for w in range(Z):
      b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(c, f'G_{zi}_{j}') for j, c in He.row_nz(w))};")
  b.append("#pragma unroll")
  b.append(f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}")
  factor, gate, solve = innovation_solver(Z, not sym, [f"y_{i}" for i in range(Z)], k.maha_thresh if k.maha_test else None)
  b.append(factor)
  b.append("int gated = 0;")
  b += gate
  # K (E x Z): column j of Gt solved against S
  for j in range(E):
    b.append(f"double k_{j}[{Z}] = {{{', '.join(f'Gt_{zi}_{j}' for zi in range(Z))}}};")
    b.append(solve(f"k_{j}"))
  K = lambda i, zi: f"k_{i}[{zi}]"  # noqa: E731
  for j in range(E):
    b.append(f"const double dx_{j} = " + " + ".join(f"{K(j, zi)}*y_{zi}" for zi in range(Z)) + ";")
  # error injection
  nom, delta = spec.err_eqs[1], spec.err_eqs[2]
  enames = dict(vector_names(nom, 'x'))
  enames.update({(delta, i, 0): f"dx_{i}" for i in range(E)})
  eblk = Block(enames, tmp_prefix="et")
  for i in range(D):
    eblk.add(f"xi_{i}", sp.Matrix(spec.err_eqs[0])[i])
  estmts, est = eblk.lower()
  b += estmts
  # B = P - K G (in place).  sym: B = (I - K He) P is NOT symmetric -- its upper triangle is needed for the result and the
  # columns He touches for C below, all rows of those; an entry below the diagonal starts from its mirror image and lives in the
  # (otherwise unused) lower half of the array until the final mirroring overwrites it.
  hcols = sorted({j for zi in range(Z) for j, _ in He.row_nz(zi)})
  # The rank-Z passes are printed as nested multiply-adds, acc -+ a0 b0 -+ a1 b1 .. = fma(-+a0, b0, fma(-+a1, b1, .. acc)): Z instructions per
  # entry.  Printed as `acc -= a0*b0 + a1*b1 + ..` hipcc contracts the sum (1 multiply + Z - 1 multiply-adds) but does not reassociate it into
  # the accumulator: one more add per entry, 20-25 % of this issue-bound kernel's fp64 instructions (ekf_c.c:105,115: same products, the sum
  # taken in another order).
  def chain(acc, pairs, neg=False):
    out = acc
    for a_, b_ in reversed(list(pairs)):
      out = f"fma({'-' if neg else ''}{a_}, {b_}, {out})"
    return out
  if sym:       # the entries below the diagonal first: they start from upper-triangle values the in-place pass below overwrites
    for i in range(E):
      for j in (c_ for c_ in hcols if c_ < i):
        b.append(f"P[{i * E + j}] = {chain(U(i, j), ((K(i, zi), f'G_{zi}_{j}') for zi in range(Z)), neg=True)};")
  for i in range(E):
    for j in range(i if sym else 0, E):
      b.append(f"P[{i * E + j}] = {chain(f'P[{i * E + j}]', ((K(i, zi), f'G_{zi}_{j}') for zi in range(Z)), neg=True)};")
  # C = B He^T, D = K R - C
  for i in range(E):
    for zi in range(Z):
      c = sum_terms(term(cf, f"P[{i * E + j}]") for j, cf in He.row_nz(zi))
      b.append(f"const double Dm_{i}_{zi} = {chain(f'-({c})', ((K(i, w), f'Rl[{w * Z + zi}]') for w in range(Z)))};")
  for i in range(E):
    for j in range(i if sym else 0, E):
      b.append(f"P[{i * E + j}] = {chain(f'P[{i * E + j}]', ((f'Dm_{i}_{zi}', K(j, zi)) for zi in range(Z)))};")
  if sym:
    b += [f"P[{j * E + i}] = P[{i * E + j}];" for i in range(E) for j in range(i + 1, E)]
  for i in range(D):
    kind, val = est[f"xi_{i}"]
    b.append(f"x[{i}] = xi_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  for i in range(Z):
    b.append(f"z[{i}] = y_{i};")
  b.append("return gated;")
  ea_arg = ", const double* __restrict__ ea" if k.ea_sym is not None else ""
  head = (f"__device__ __forceinline__ int update_{k.kind}_regs{'_sym' if sym else ''}(double (&x)[{D}], double (&P)[{E * E}], "
          f"double (&z)[{Z}], const double (&R)[{Z * Z}]{ea_arg}) {{")
  return "\n".join([head] + _ind(b) + ["}"]), He


def maha_regs(spec, k):
  """d2 = y^T (He P He^T + R)^-1 y for one observation, state untouched (reference: EKF_sym.maha_test, ekf_sym.py:626-649)."""
  D, E, Z = spec.dim_x, spec.dim_err, k.zdim
  names = dict(vector_names(spec.x_sym, 'x'))
  Herr = sp.Matrix(k.H_sym) * sp.Matrix(spec.H_mod_sym)
  blk = Block(names, tmp_prefix="mt")
  for i in range(Z):
    blk.add(f"hx_{i}", k.h_sym[i])
  fmtH = lambda i, j: f"He_{i}_{j}"  # noqa: E731
  for i in range(Z):
    for j in range(E):
      blk.add(fmtH(i, j), Herr[i, j])
  stmts, st = blk.lower()
  He = SMat.from_structure(Z, E, st, fmtH)
  b = list(stmts)
  b.append(f"double v[{Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  for i in range(Z):
    kind, val = st[f"hx_{i}"]
    hx = f"hx_{i}" if kind == 'expr' else repr(float(val))
    b.append(f"v[{i}] = z[{i}] - {hx};")
  for zi in range(Z):
    nz = He.row_nz(zi)
    for j in range(E):
      b.append(f"const double G_{zi}_{j} = {sum_terms(term(c, f'P[{kk * E + j}]') for kk, c in nz)};")
  for zi in range(Z):
    for w in range(Z):
      b.append(f"S[{zi * Z + w}] = {sum_terms(term(c, f'G_{zi}_{j}') for j, c in He.row_nz(w))} + R[{zi * Z + w}];")
  b.append(f"double w[{Z}];")
  b += ["#pragma unroll", f"for (int i = 0; i < {Z}; i++) w[i] = v[i];"]
  b.append(f"rn::ldu_factor<{Z}>(S, L, iL);")
  b.append(f"rn::ldu_forward<{Z}>(L, iL, v);")
  b.append(f"rn::ldu_forward_t<{Z}>(L, iL, w);")
  b.append("return " + " + ".join(f"v[{i}]*w[{i}]*iL[{i}]" for i in range(Z)) + ";")
  head = (f"__device__ __forceinline__ double maha_{k.kind}_regs(const double (&x)[{D}], const double (&P)[{E * E}], "
          f"const double (&z)[{Z}], const double (&R)[{Z * Z}]) {{")
  return "\n".join([head] + _ind(b) + ["}"])


def maha_kernels(spec):
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  out = []
  for k in spec.kinds:
    Z = k.zdim
    ZZ = Z * Z
    out.append(maha_regs(spec, k))
    out.append(f"""
__global__ __launch_bounds__(64) void k_maha_{k.kind}(const double* __restrict__ gx, const double* __restrict__ gP,
    const double* __restrict__ gz, const double* __restrict__ gR, const int r_per_filter, const int64_t n,
    double* __restrict__ d2) {{
  __shared__ __attribute__((aligned(16))) double s_x[64 * {D | 1}];
  __shared__ __attribute__((aligned(16))) double s_P[64 * {EE | 1}];
  __shared__ __attribute__((aligned(16))) double s_z[64 * {Z | 1}];
  __shared__ __attribute__((aligned(16))) double s_R[64 * {ZZ | 1}];
  const int lane = threadIdx.x;
  const int64_t tiles = (n + 63) >> 6;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile << 6;
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    rn::tile_g2l_async<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_g2l_async<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::tile_g2l_async<{Z}>(gz + base * {Z}, cnt, s_z, lane);
    if (r_per_filter) rn::tile_g2l_async<{ZZ}>(gR + base * {ZZ}, cnt, s_R, lane);
    rn::async_wait();
    rn::wave_lds_sync();
    double x[{D}], P[{EE}], z[{Z}], R[{ZZ}];
    rn::lds_to_regs<{D}>(s_x, lane, x);
    rn::lds_to_regs<{EE}>(s_P, lane, P);
    rn::lds_to_regs<{Z}>(s_z, lane, z);
    if (r_per_filter) {{
      rn::lds_to_regs<{ZZ}>(s_R, lane, R);
    }} else {{
#pragma unroll
      for (int i = 0; i < {ZZ}; i++) R[i] = gR[i];
    }}
    const double d = maha_{k.kind}_regs(x, P, z, R);
    if (lane < cnt) d2[base + lane] = d;
    rn::wave_lds_sync();
  }}
}}
""")
  return "\n".join(out)


def launch_maha(kind):
  return f"""  const int64_t tiles = (n + 63) >> 6;
  hipLaunchKernelGGL(k_maha_{kind}, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, z, R, r_per_filter, n, d2);"""


def norm_text(spec):
  """Quaternion renormalisation after predict / update (EKFSym::normalize_quaternions, ekf_sym.cc:69-77,207,213)."""
  quat = "".join(f" rn::normalize_quat<{spec.dim_x}>(x, {q});" for q in spec.quaternion_idxs)
  return f"if (norm_quats) {{{quat} }}" if spec.quaternion_idxs else "(void)norm_quats;"


def kernels(spec):
  """Device functions + __global__ kernels of family S for every kind."""
  waves = tuning.current().small_waves
  kattr = f" __attribute__((amdgpu_waves_per_eu({waves}, {waves})))" if waves else ""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  out = []
  for sym in (False, True):          # full products for the step-granular kernels, symmetric arithmetic for the fused runs
    ptxt, _ = predict_regs(spec, sym)
    out.append(ptxt)
    for k in spec.kinds:
      utxt, _ = update_regs(spec, k, sym)
      out.append(utxt)
  out.append(f"""
// the fused multi-step kernels read (P + P^T) / 2 of the caller's covariance, once, as the state enters the registers
__device__ __forceinline__ void symmetrize_regs(double (&P)[{EE}]) {{
#pragma unroll
  for (int i = 0; i < {E}; i++) {{
#pragma unroll
    for (int j = i + 1; j < {E}; j++) {{ P[i * {E} + j] = 0.5 * (P[i * {E} + j] + P[j * {E} + i]); P[j * {E} + i] = P[i * {E} + j]; }}
  }}
}}
""")
  norm = norm_text(spec)

  out.append(f"""
// ---- predict only: one launch propagates n filters by dt -------------------------------------------
__global__ __launch_bounds__(64) void k_predict(double* __restrict__ gx, double* __restrict__ gP,
    const double* __restrict__ gQ, const double* __restrict__ gdt, const double dt_scalar, const int64_t n,
    const int norm_quats, const uint8_t* __restrict__ active) {{
  __shared__ __attribute__((aligned(16))) double s_x[64 * {D | 1}];
  __shared__ __attribute__((aligned(16))) double s_P[64 * {EE | 1}];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  const int lane = threadIdx.x;
  for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];       // Q as LDS broadcast operands (36 SGPR pairs spilled otherwise)
  const int64_t tiles = (n + 63) >> 6;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile << 6;
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    rn::tile_g2l_async<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_g2l_async<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    const double dt = (gdt != nullptr && lane < cnt) ? gdt[base + lane] : dt_scalar;
    rn::async_wait();
    rn::wave_lds_sync();
    double x[{D}], P[{EE}];
    rn::lds_to_regs<{D}>(s_x, lane, x);
    rn::lds_to_regs<{EE}>(s_P, lane, P);
    predict_regs(x, P, s_Q, dt);
    {norm}
    rn::wave_lds_sync();
    // a masked-out filter (active[i] == 0: no observation for it in this call) keeps the record it came with: its lane
    // does not overwrite the LDS image, so the coalesced write-back returns the loaded bytes
    if (active == nullptr || (lane < cnt && active[base + lane] != 0)) {{
      rn::regs_to_lds<{D}>(s_x, lane, x);
      rn::regs_to_lds<{EE}>(s_P, lane, P);
    }}
    rn::wave_lds_sync();
    rn::tile_l2g<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_l2g<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::wave_lds_sync();
  }}
}}
""")
  for k in spec.kinds:
    Z = k.zdim
    ZZ = Z * Z
    # extra arguments are per observation, i.e. per filter of the batch: (n, len(ea)) row-major
    ea = f", gea + (base + (lane < cnt ? lane : 0)) * {int(sp.Matrix(k.ea_sym).shape[0])}" if k.ea_sym is not None else ""
    # k_stepc_{kind}: the same kernel writing a CHECKPOINT on its way -- the observations as they came (cz) and the filtered pair (cx, cP): what the
    # orchestrators' rewind rings keep of every call (ekf_sym.cc:142-156, 191).  A kernel of its own, so that k_step_{kind} stays as it is.
    for ckpt in (False, True):
      kn = f"k_stepc_{k.kind}" if ckpt else f"k_step_{k.kind}"
      cargs = ", double* __restrict__ cx, double* __restrict__ cP, double* __restrict__ cz" if ckpt else ""
      cz_store = f"\n    rn::tile_l2g<{Z}>(cz + base * {Z}, cnt, s_z, lane);      // the observations, before the residuals take their place" if ckpt else ""
      c_store = f"\n    rn::tile_l2g<{D}>(cx + base * {D}, cnt, s_x, lane);\n    rn::tile_l2g<{EE}>(cP + base * {EE}, cnt, s_P, lane);" if ckpt else ""
      out.append(f"""
// ---- kind {k.kind}: [predict +] update{" + checkpoint" if ckpt else ""}, state round-trips HBM once per launch --------------------------
template <bool DO_PREDICT>
__global__ __launch_bounds__(64){kattr} void {kn}(double* __restrict__ gx, double* __restrict__ gP,
    double* __restrict__ gz, const double* __restrict__ gR, const int r_per_filter, const double* __restrict__ gea,
    const double* __restrict__ gQ, const double* __restrict__ gdt, const double dt_scalar, const int64_t n,
    const int norm_quats, uint8_t* __restrict__ flags, const uint8_t* __restrict__ active{cargs}) {{
  __shared__ __attribute__((aligned(16))) double s_x[64 * {D | 1}];
  __shared__ __attribute__((aligned(16))) double s_P[64 * {EE | 1}];
  __shared__ __attribute__((aligned(16))) double s_z[64 * {Z | 1}];
  __shared__ __attribute__((aligned(16))) double s_R[64 * {ZZ | 1}];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  const int lane = threadIdx.x;
  if (DO_PREDICT) {{
    for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];
  }}
  const int64_t tiles = (n + 63) >> 6;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile << 6;
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    // the observations first: in a stream every launch reads a z buffer nothing has touched since it was written (HBM), while x and P
    // were written by the previous launch and sit in the L2 / Infinity Cache -- the tile waits for its slowest load
    rn::tile_g2l_async<{Z}>(gz + base * {Z}, cnt, s_z, lane);
    if (r_per_filter) rn::tile_g2l_async<{ZZ}>(gR + base * {ZZ}, cnt, s_R, lane);
    rn::tile_g2l_async<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_g2l_async<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    double dt = dt_scalar;
    if (DO_PREDICT && gdt != nullptr && lane < cnt) dt = gdt[base + lane];
    rn::async_wait();
    rn::wave_lds_sync();{cz_store}
    double x[{D}], P[{EE}], z[{Z}], R[{ZZ}];
    rn::lds_to_regs<{D}>(s_x, lane, x);
    rn::lds_to_regs<{EE}>(s_P, lane, P);
    rn::lds_to_regs<{Z}>(s_z, lane, z);
    if (r_per_filter) {{
      rn::lds_to_regs<{ZZ}>(s_R, lane, R);
    }} else {{
#pragma unroll
      for (int i = 0; i < {ZZ}; i++) R[i] = gR[i];
    }}
    if (DO_PREDICT) {{
      predict_regs(x, P, s_Q, dt);
      {norm}
    }}
    int fl = update_{k.kind}_regs(x, P, z, R{ea});
    {norm}
    rn::wave_lds_sync();
    // masked-out filters (active[i] == 0) pass through untouched: x, P and z leave as they came, flag bit 4 is set
    const bool on = active == nullptr || (lane < cnt && active[base + lane] != 0);
    if (on) {{
      rn::regs_to_lds<{D}>(s_x, lane, x);
      rn::regs_to_lds<{EE}>(s_P, lane, P);
      rn::regs_to_lds<{Z}>(s_z, lane, z);
    }}
    rn::wave_lds_sync();
    rn::tile_l2g<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_l2g<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::tile_l2g<{Z}>(gz + base * {Z}, cnt, s_z, lane);{c_store}
    if (flags != nullptr && lane < cnt) {{
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < {D}; i++) acc += x[i];
      if (!(acc - acc == 0.0)) fl |= 2;          // non-finite state
      flags[base + lane] = (uint8_t)(on ? fl : 16);
    }}
    rn::wave_lds_sync();
  }}
}}
""")
  if run_block(spec) > 0:          # blocked fused runs: k_run_blk (no trace) and k_run_blk_tr (filtered trace)
    out.append(run_kernel_blk(spec, norm))
    out.append(run_kernel_blk(spec, norm, trace=True))
  else:                            # fallback "no_run_blk" (the blocked kernels spilled): the step-at-a-time k_run serves both
    out.append(run_kernel(spec, norm))
  return "\n".join(out)


def run_unroll(spec):
  """Steps of the schedule per iteration of the traced fused run's loop (= depth of its observation prefetch ring)."""
  zmax = max(k.zdim for k in spec.kinds)
  return 8 if zmax <= 2 else 4          # observation rows in flight per wavefront (zmax doubles of staging registers per lane each)


def run_block(spec):
  """Steps per block of the untraced fused run (k_run_blk): observation rows of one block are in flight while the previous block is
  computed, so a block has to outlast one HBM round trip (~2 us); a step of the 2-state model takes ~0.1 us, of a 6-state model
  ~1 us.  Bounded by the staging registers (2 x K x zmax doubles per lane) and the code size (the K steps are unrolled)."""
  from rednose_amd.codegen import emit
  forced = -1 if "no_run_blk" in emit._active else tuning.current().run_block      # pylint: disable=protected-access
  if forced:
    return max(0, forced)
  zmax = max(k.zdim for k in spec.kinds)
  # measured (tools/ab_run.cpp, 65 536 filters, one wavefront per SIMD): 2-state model 8 / 16 / 32 / 64 steps per block ->
  # 322 / 331 / 273 / 233 G steps/s (the traced kernel's structure: 138); 6-state model 2 / 4 / 8 -> 42.1 / 40.5 / 42.2 (31.5)
  K = 16 if spec.dim_err <= 2 else 8
  while K > 2 and (K * zmax > 32 or K * len(spec.kinds) > 16):      # staging registers; code size (K steps x every kind, unrolled)
    K //= 2
  return K


def run_kernel(spec, norm):
  """T steps per launch: x and P stay in VGPRs, only z (in) / y (out) and the optional trace touch HBM."""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  zmax = max(k.zdim for k in spec.kinds)
  KP = run_unroll(spec)
  cases = []
  EAM = max([int(sp.Matrix(k.ea_sym).shape[0]) for k in spec.kinds if k.ea_sym is not None] + [0])
  for k in spec.kinds:
    Z = k.zdim
    ea, guard = "", ""
    if k.ea_sym is not None:      # per-filter, per-step extra arguments: the (T, n, EA) array of the entry point (flag 8 without it)
      ea = f", gea + ((int64_t)t * n + base + (lane < cnt ? lane : 0)) * {EAM}"
      guard = "          if (gea == nullptr) { fl = 8; break; }\n"
    cases.append(f"""        case {k.kind}: {{
{guard}          double zk[{Z}], Rk[{Z * Z}];
#pragma unroll
          for (int i = 0; i < {Z}; i++) zk[i] = z[i];
#pragma unroll
          for (int i = 0; i < {Z * Z}; i++) Rk[i] = gR[t * {zmax * zmax} + i];
          fl = update_{k.kind}_regs_sym(x, P, zk, Rk{ea});
#pragma unroll
          for (int i = 0; i < {Z}; i++) z[i] = zk[i];
          break;
        }}""")
  return f"""
// ---- fused multi-step run: kinds[t], dts[t] shared by all filters; z is (T, n, {zmax}) in: z, out: y -----------
__global__ __launch_bounds__(64) void k_run(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* __restrict__ gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    double* __restrict__ tx, double* __restrict__ tP, const double* __restrict__ gea, const int32_t* __restrict__ augs) {{
  (void)gea; (void)augs;      // lane-per-filter models are never MSCKF models (those use the lane-group family): no window shift here
  __shared__ __attribute__((aligned(16))) double s_x[64 * {D | 1}];
  __shared__ __attribute__((aligned(16))) double s_P[64 * {EE | 1}];
  __shared__ __attribute__((aligned(16))) double s_z[64 * {zmax | 1}];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  const int lane = threadIdx.x;
  for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];
  const int64_t tiles = (n + 63) >> 6;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile << 6;
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    rn::tile_g2l<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_g2l<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::tile_g2l<{zmax}>(gz + base * {zmax}, cnt, s_z, lane);
    rn::wave_lds_sync();
    double x[{D}], P[{EE}], z[{zmax}];
    rn::lds_to_regs<{D}>(s_x, lane, x);
    rn::lds_to_regs<{EE}>(s_P, lane, P);
    symmetrize_regs(P);
    // Observation prefetch, {KP} steps deep: a step of a small model takes a fraction of a microsecond, less than one HBM round
    // trip, so with the next step's row alone in flight every step waited for its observation (r3a counters of the 2-state
    // model: 59 % of the wave cycles in s_waitcnt, 13 % issuing).  ring[j] carries the row of the step t = j (mod {KP}); the
    // step loop is unrolled {KP} times so that the ring index is a compile-time constant (a runtime index would put the staging
    // registers into scratch memory).  Loads are issued unconditionally on a clamped row (see TilePrefetch).
    rn::TilePrefetch<{zmax}> ring[{KP}];
#pragma unroll
    for (int u = 1; u < {KP}; u++) ring[u].issue(gz + ((u < T ? u : T - 1) * n + base) * {zmax}, cnt, lane);
    for (int64_t tb = 0; tb < T; tb += {KP}) {{
#pragma unroll
    for (int u = 0; u < {KP}; u++) {{
      const int64_t t = tb + u;
      if (t < T) {{
      rn::lds_to_regs<{zmax}>(s_z, lane, z);
      rn::wave_lds_sync();
      ring[u].issue(gz + ((t + {KP} < T ? t + {KP} : T - 1) * n + base) * {zmax}, cnt, lane);
      const int kind = kinds[t];
      const double dt = dts[t];
      predict_regs_sym(x, P, s_Q, dt);
      {norm}
      int fl = 0;
      switch (kind) {{
{chr(10).join(cases)}
        default: fl = 8; break;      // kind not available in the fused run (unknown, or it takes extra arguments)
      }}
      {norm}
      // y(t) and the optional trace go out through LDS as coalesced 16-byte stores
      rn::regs_to_lds<{zmax}>(s_z, lane, z);
      if (tx != nullptr) rn::regs_to_lds<{D}>(s_x, lane, x);
      if (tP != nullptr) rn::regs_to_lds<{EE}>(s_P, lane, P);
      rn::wave_lds_sync();
      rn::tile_l2g<{zmax}>(gz + (t * n + base) * {zmax}, cnt, s_z, lane);
      if (tx != nullptr) rn::tile_l2g<{D}>(tx + (t * n + base) * {D}, cnt, s_x, lane);
      if (tP != nullptr) rn::tile_l2g<{EE}>(tP + (t * n + base) * {EE}, cnt, s_P, lane);
      if (flags != nullptr && lane < cnt) flags[t * n + base + lane] = (uint8_t)fl;
      rn::wave_lds_sync();
      if (t + 1 < T) ring[(u + 1) % {KP}].commit(s_z, cnt, lane);
      rn::wave_lds_sync();
      }}
    }}
    }}
    rn::regs_to_lds<{D}>(s_x, lane, x);
    rn::regs_to_lds<{EE}>(s_P, lane, P);
    rn::wave_lds_sync();
    rn::tile_l2g<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_l2g<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::wave_lds_sync();
  }}
}}
"""


def run_kernel_blk(spec, norm, trace=False):
  """The fused run without trace (tx == tP == nullptr), restructured around what the counters of k_run show for small models: the
  arithmetic of a step is tens of fp64 instructions, the step took thousands of cycles, because every step (a) waited for its own
  y store to retire -- vmcnt counts loads and stores in issue order, and the wait for the prefetched observation row behind the
  (conditional) stores degrades to vmcnt(0) --, (b) waited twice for scalar loads (kind / dt, then R), and (c) crossed the LDS
  three times (row in, y out, Q).  Here a wavefront works in blocks of K steps entirely in registers:
    * lane l reads / writes its own filter's observation row directly (zmax doubles per step, contiguous per lane);
    * the K rows of block b + 1, and the block's schedule (dt, kind: lane u of a schedule register holds step u of the block; R: the block's
      K x zmax^2 doubles spread over the lanes; all broadcast with v_readlane when the step runs), are loaded while block b is computed -- ONE vmcnt(0) per block;
    * y (which replaces z in its registers) and the flags of block b are stored at the start of block b + 1, AFTER that wait, so no
      load the wavefront is waiting for ever queues behind a store it has just issued.
  Same arithmetic as k_run (predict_regs / update_*_regs); results agree to the last bits (FMA contraction may differ per kernel).

  trace=True: the same structure writing the filtered trace -- every step's x / P leave through the LDS image as coalesced stores that
  nothing waits for until the next block starts (k_run_blk_tr; the step-at-a-time k_run paid a store wait per step: MI355X, same call,
  8 192 x 200 kinematic6 3.04 -> 3.38 G steps/s, 65 536 x 200 kinematic 59 -> 75 G steps/s, results bit-identical)."""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  zmax = max(k.zdim for k in spec.kinds)
  ZZ = zmax * zmax
  K = run_block(spec)
  NR = (K * ZZ + 63) // 64
  EAM = max([int(sp.Matrix(k.ea_sym).shape[0]) for k in spec.kinds if k.ea_sym is not None] + [0])
  cases = []
  for k in spec.kinds:
    Z = k.zdim
    ea, guard = "", ""
    if k.ea_sym is not None:
      ea = f", gea + ((int64_t)t * n + base + lc) * {EAM}"
      guard = "            if (gea == nullptr) { fl = 8; break; }\n"
    cases.append(f"""          case {k.kind}: {{
{guard}            double zk[{Z}], Rk[{Z * Z}];
#pragma unroll
            for (int i = 0; i < {Z}; i++) zk[i] = cur[u][i];
#pragma unroll
            for (int i = 0; i < {Z * Z}; i++) Rk[i] = lane_bcast(Rv[(u * {ZZ} + i) >> 6], (u * {ZZ} + i) & 63);
            fl = update_{k.kind}_regs_sym(x, P, zk, Rk{ea});
#pragma unroll
            for (int i = 0; i < {Z}; i++) cur[u][i] = zk[i];
            break;
          }}""")
  # rows are addressed by pointer increments (one 64-bit multiply per block, none per row): a row past the end of the schedule
  # re-reads row T - 1 (the increment is zero there), so the loads stay unconditional
  issue = f"""{{
      const int64_t tb0_ = TB_ < T ? TB_ : T - 1;
      const int64_t ts_ = TB_ + (lane < {K} ? lane : {K - 1});
      const int64_t tsc_ = ts_ < T ? ts_ : T - 1;
      dtn = dts[tsc_];
      kn = kinds[tsc_];
#pragma unroll
      for (int r = 0; r < {NR}; r++) {{           // R of the block's steps, flat: lane l of register r holds double r * 64 + l
        const int64_t ri_ = tb0_ * {ZZ} + r * 64 + lane;
        Rn[r] = gR[ri_ < T * {ZZ} ? ri_ : T * {ZZ} - 1];
      }}
      const double* zp_ = zrow + tb0_ * rowstride;
      const int64_t left_ = T - 1 - tb0_;          // rows of the schedule after row tb0_
#pragma unroll
      for (int u = 0; u < {K}; u++) {{
#pragma unroll
        for (int i = 0; i < {zmax}; i++) nxt[u][i] = zp_[i];
        zp_ += (u < left_) ? rowstride : 0;
      }}
    }}"""
  store = f"""#pragma unroll
          for (int i = 0; i < {zmax}; i++) yp_[i] = cur[u][i];
          if (flags != nullptr) fp_[0] = (uint8_t)flb[u];
          yp_ += rowstride;
          fp_ += n;"""
  helpers = "" if trace else """
__device__ __forceinline__ double lane_bcast(const double v, const int l) {      // lane l's value in every lane (l uniform): two v_readlane_b32
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void pin_i(int& v) { asm volatile("" : "+v"(v)); }
"""
  title = (f"// ---- fused multi-step run with the filtered trace: blocks of {K} steps, no store is waited for inside a block -----------------"
           if trace else
           f"// ---- fused multi-step run without trace: blocks of {K} steps in registers (see emit_small.run_kernel_blk) -------------------")
  kname = "k_run_blk_tr" if trace else "k_run_blk"
  targs = ",\n    double* __restrict__ tx, double* __restrict__ tP" if trace else ""
  tstore = f"""
          // filtered pair of step t: through the LDS image, coalesced; the stores drain under the following steps
          rn::regs_to_lds<{D}>(s_x, lane, x);
          rn::regs_to_lds<{EE}>(s_P, lane, P);
          rn::wave_lds_sync();
          if (tx != nullptr) rn::tile_l2g<{D}>(tx + (t * n + base) * {D}, cnt, s_x, lane);
          if (tP != nullptr) rn::tile_l2g<{EE}>(tP + (t * n + base) * {EE}, cnt, s_P, lane);
          rn::wave_lds_sync();""" if trace else ""
  return f"""{helpers}
{title}
__global__ __launch_bounds__(64) void {kname}(double* __restrict__ gx, double* __restrict__ gP, const double* __restrict__ gQ,
    const int32_t* __restrict__ kinds, const double* __restrict__ dts, const int64_t T, double* gz,
    const double* __restrict__ gR, const int64_t n, const int norm_quats, uint8_t* __restrict__ flags,
    const double* __restrict__ gea{targs}) {{
  (void)gea;
  __shared__ __attribute__((aligned(16))) double s_x[64 * {D | 1}];
  __shared__ __attribute__((aligned(16))) double s_P[64 * {EE | 1}];
  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];
  const int lane = threadIdx.x;
  for (int i = lane; i < {EE}; i += 64) s_Q[i] = gQ[i];
  const int64_t tiles = (n + 63) >> 6;
  const int64_t rowstride = n * {zmax};               // doubles between the rows of one filter in consecutive steps
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile << 6;
    const int cnt = (n - base) < 64 ? (int)(n - base) : 64;
    rn::tile_g2l<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_g2l<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::wave_lds_sync();
    double x[{D}], P[{EE}];
    rn::lds_to_regs<{D}>(s_x, lane, x);
    rn::lds_to_regs<{EE}>(s_P, lane, P);
    symmetrize_regs(P);
    // lanes past the end of a ragged tile compute on a copy of the last filter's rows and store nothing
    const int lc = lane < cnt ? lane : cnt - 1;
    const bool live = lane < cnt;
    double* zrow = gz + (base + lc) * {zmax};
    double cur[{K}][{zmax}], nxt[{K}][{zmax}], Rv[{NR}], Rn[{NR}], dtv, dtn;
    int flb[{K}], kv, kn;
    {issue.replace("TB_", "((int64_t)0)")}
    for (int64_t tb = 0; tb < T; tb += {K}) {{
      // everything in flight lands: this block's rows and schedule (and the stores issued one block ago)
#pragma unroll
      for (int u = 0; u < {K}; u++) {{
#pragma unroll
        for (int i = 0; i < {zmax}; i++) rn::pin(nxt[u][i]);
      }}
      rn::pin(dtn);
      pin_i(kn);
#pragma unroll
      for (int r = 0; r < {NR}; r++) rn::pin(Rn[r]);
      if (tb > 0 && live) {{
        double* yp_ = zrow + (tb - {K}) * rowstride;
        uint8_t* fp_ = flags + (tb - {K}) * n + base + lane;
#pragma unroll
        for (int u = 0; u < {K}; u++) {{
{store}
        }}
      }}
#pragma unroll
      for (int u = 0; u < {K}; u++) {{
#pragma unroll
        for (int i = 0; i < {zmax}; i++) cur[u][i] = nxt[u][i];
      }}
      dtv = dtn;
      kv = kn;
#pragma unroll
      for (int r = 0; r < {NR}; r++) Rv[r] = Rn[r];
      {issue.replace("TB_", f"(tb + {K})")}
#pragma unroll
      for (int u = 0; u < {K}; u++) {{
        const int64_t t = tb + u;
        flb[u] = 0;
        if (t < T) {{
          const double dt = lane_bcast(dtv, u);
          const int kind = __builtin_amdgcn_readlane(kv, u);
          predict_regs_sym(x, P, s_Q, dt);
          {norm}
          int fl = 0;
          switch (kind) {{
{chr(10).join(cases)}
            default: fl = 8; break;      // kind not available in the fused run (unknown, or it takes extra arguments)
          }}
          {norm}
          flb[u] = fl;{tstore}
        }}
      }}
    }}
    if (live) {{
      const int64_t tl = ((T - 1) / {K}) * {K};          // first step of the last block
      double* yp_ = zrow + tl * rowstride;
      uint8_t* fp_ = flags + tl * n + base + lane;
#pragma unroll
      for (int u = 0; u < {K}; u++) {{
        if (tl + u < T) {{
{store}
        }}
      }}
    }}
    rn::regs_to_lds<{D}>(s_x, lane, x);
    rn::regs_to_lds<{EE}>(s_P, lane, P);
    rn::wave_lds_sync();
    rn::tile_l2g<{D}>(gx + base * {D}, cnt, s_x, lane);
    rn::tile_l2g<{EE}>(gP + base * {EE}, cnt, s_P, lane);
    rn::wave_lds_sync();
  }}
}}
"""


def launch_run(spec=None):
  blk = spec is not None and run_block(spec) > 0
  if not blk:
    return """  const int64_t tiles = (n + 63) >> 6;
  hipLaunchKernelGGL(k_run, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, trace_x, trace_P, ea, augment);"""
  return """  (void)augment;
  const int64_t tiles = (n + 63) >> 6;
  if (trace_x == nullptr && trace_P == nullptr) {
    hipLaunchKernelGGL(k_run_blk, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                       x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, ea);
  } else {
    hipLaunchKernelGGL(k_run_blk_tr, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                       x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, ea, trace_x, trace_P);
  }"""


def launch_predict():
  return """  const int64_t tiles = (n + 63) >> 6;
  hipLaunchKernelGGL(k_predict, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, dt_vec, dt, n, norm_quats, active);"""


def launch_step_ckpt(kind):
  return f"""  const int64_t tiles = (n + 63) >> 6;
  hipLaunchKernelGGL(k_stepc_{kind}<true>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, z, R, r_per_filter, ea, Q, dt_vec, dt, n, norm_quats, flags, active, ckpt_x, ckpt_P, ckpt_z);"""


def launch_step(kind, do_predict):
  tf = "true" if do_predict else "false"
  if do_predict:
    args = "x, P, z, R, r_per_filter, ea, Q, dt_vec, dt, n, norm_quats, flags, active"
  else:
    args = "x, P, z, R, r_per_filter, ea, nullptr, nullptr, 0.0, n, norm_quats, flags, active"
  return f"""  const int64_t tiles = (n + 63) >> 6;
  hipLaunchKernelGGL(k_step_{kind}<{tf}>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     {args});"""
