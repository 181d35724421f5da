"""Kernel family W, step-granular kernels, second structure: scalar phase lane-per-filter, matrix phase lane-group.

emit_wide.py evaluates the x-dependent scalars (f, F, h, H.H_mod, err_fun: ~2 000 instructions with the
accelerometer kind's gravity/rotation terms) redundantly in all 32 lanes of a filter's lane group, twice per
wavefront, and keeps them in VGPRs next to the covariance rows: 256 VGPRs + AGPR spills, 1 wave per SIMD, 13.7 % of
the HBM roofline on live (round-1 profile).  Here a wavefront owns a tile of FT filters and works in three phases:

  phase 1  lane l = filter l of the tile (FT lanes active): x, z -> f, F non-zeros, normalise, h, He = H.H_mod
           non-zeros, y = z - h; results go to the filter's scalar SLOT in LDS.  One evaluation per filter.
  phase 2  for each GROUP of filters (as many as fit a wavefront with one lane per row of P: 64 // dim_err, see
           filters_per_wave): the lane-group-per-filter covariance algebra of emit_wide.py, but every x-dependent
           coefficient is an LDS broadcast read from the slot; the next group's P records stream HBM -> LDS
           asynchronously (global_load_lds_dwordx4, double buffer) while the current group computes -- or, for small
           records, a single buffer and several wavefronts per SIMD (double_buffered).  dx goes back to the slot.
           Feature-track kinds of MSCKF models project G, Gt and He P He^T on the left null space of the
           extra-argument Jacobian with the reflectors phase 1 left in the slot (_lean_update).
  phase 3  lane l = filter l again: x' = err_fun(x, dx), renormalise, x / y / flags leave through LDS, coalesced.

The algebra and its order are unchanged (see emit_wide.py / emit_small.py docstrings); only who evaluates the
scalars changed.  Results agree with the first structure to rounding (different instruction streams contract
FMAs differently), which tests/test_gpu_run.py bounds.
"""
import os
import re

import sympy as sp

from rednose_amd.codegen import tuning
from rednose_amd.codegen.lower import Block, vector_names
from rednose_amd.codegen.emit_common import SMat, term, sum_terms, innovation_solver


def group_lanes(spec):
  """Lanes per filter in the matrix phase: one lane per row/column of P, groups packed back to back."""
  fpw = filters_per_wave(spec)
  if fpw == 1:
    return 64              # 33 .. 64 error states: the whole wavefront works on one filter
  return 32 if fpw == 2 else spec.dim_err     # two groups: one per 32-lane half (fewest LDS conflicts)


def filters_per_wave(spec):
  """Filters whose covariance algebra one wavefront does at a time (64 // dim_err, e.g. 3 for 21 error states:
  63 of 64 lanes busy instead of 42 with two 32-lane groups)."""
  if spec.dim_err > 32:
    return 1
  fpw = tuning.current().wide_fpw
  return fpw if fpw else max(2, 64 // spec.dim_err)


EADIM = 3        # extra-argument dimension of feature-track kinds, hard-coded in the reference (ekf_sym.py:151)


def ea_dim(k):
  return 0 if k.ea_sym is None else int(sp.Matrix(k.ea_sym).shape[0])


def _ind(lines, n=2):
  pad = " " * n
  return [pad + s for s in lines]


def _odd(n):
  return n if n & 1 else n + 1


def tile_filters(spec):
  """Filters per wavefront tile: the tuning value rounded up to whole groups."""
  fpw = filters_per_wave(spec)
  ft = tuning.current().wide_ft
  if ft == 0:      # auto: small records (<= 16 error states) -> two groups per tile, several wavefronts per SIMD; else 16;
    #                above 40 error states 8, to keep the block's LDS (P buffer + per-filter slots) under 64 KB
    ft = 2 * fpw if spec.dim_err <= 16 else (16 if spec.dim_err <= 40 else 8)
  return -(-ft // fpw) * fpw


def double_buffered(spec):
  """Asynchronous double-buffered prefetch of the next group's covariance records, or a single buffer.  With small records
  several wavefronts fit per SIMD and hide each other's HBM latency, and the second buffer only costs occupancy
  (kinematic9: 25.5-26.3 us per launch with a single buffer and tiles of 14 filters, 30.5 us with the live-tuned 16 / double)."""
  db = tuning.current().wide_db
  return bool(db) if db >= 0 else 16 < spec.dim_err <= 40       # above 40 error states a second E x E buffer does not fit 64 KB


class Layout:
  """Per-filter scalar slot in LDS (doubles)."""

  def __init__(self, spec, f_vars, he_vars_by_kind):
    D, E = spec.dim_x, spec.dim_err
    self.zmax = max(k.zdim for k in spec.kinds)
    self.nf = len(f_vars)
    self.nh = max([len(v) for v in he_vars_by_kind.values()] + [0])
    self.OFF_X = 0
    self.OFF_F = self.OFF_X + D
    self.OFF_Y = self.OFF_F + self.nf
    self.OFF_HE = self.OFF_Y + self.zmax
    self.OFF_DX = self.OFF_HE + self.nh
    self.OFF_DT = self.OFF_DX + E
    self.OFF_FL = self.OFF_DT + 1
    # feature-track kinds (MSCKF): EADIM Householder reflectors of the extra-argument Jacobian (EADIM x Z entries +
    # EADIM betas) and the projected noise (Z - EADIM)^2
    feat = [k for k in spec.kinds if k.He_sym is not None]
    self.zf = max([k.zdim for k in feat] + [0])
    self.OFF_RF = self.OFF_FL + 1
    self.OFF_RP = self.OFF_RF + (EADIM * self.zf + EADIM if feat else 0)
    # the residual of a feature-track kind exists twice: in the orthonormal basis of the reflectors (what the update consumes: OFF_YP)
    # and in the reference's fullPivLu basis (what goes back into z: the Y field)
    self.OFF_YP = self.OFF_RP + ((self.zf - EADIM) ** 2 if feat else 0)
    self.SLOT = _odd(self.OFF_YP + ((self.zf - EADIM) if feat else 0))     # odd stride: lane-per-filter ds_*_b64 accesses hit 32 distinct bank pairs


def _lowered_predict(spec):
  D, M = spec.dim_x, spec.dim_main_err
  names = {**vector_names(spec.x_sym, 'x'), spec.dt_sym: 'dt'}
  blk = Block(names, tmp_prefix="pt")
  for i in range(D):
    blk.add(f"xn_{i}", spec.f_sym[i])
  fmtF = lambda i, j: f"F_{i}_{j}"  # noqa: E731
  for i in range(M):
    for j in range(M):
      blk.add(fmtF(i, j), spec.F_sym[i, j])
  stmts, st = blk.lower()
  F = SMat.identity_padded(SMat.from_structure(M, M, st, fmtF), spec.dim_err)
  f_vars = [c[1] for row in F.e for c in row if c is not None and c[0] == 'var']
  return stmts, st, F, f_vars


def _lowered_obs(spec, k):
  E, Z = spec.dim_err, k.zdim
  names = dict(vector_names(spec.x_sym, 'x'))
  names.update(vector_names(k.ea_sym, 'ea'))
  Herr = sp.Matrix(k.H_sym) * sp.Matrix(spec.H_mod_sym)
  blk = Block(names, tmp_prefix="ut")
  for i in range(Z):
    blk.add(f"hx_{i}", k.h_sym[i])
  fmtH = lambda i, j: f"He_{i}_{j}"  # noqa: E731
  for i in range(Z):
    for j in range(E):
      blk.add(fmtH(i, j), Herr[i, j])
  if k.He_sym is not None:      # d h / d extra args (the reference's He_{kind}), Z x EADIM
    assert tuple(k.He_sym.shape) == (Z, EADIM), "feature-track kinds take EADIM = 3 extra arguments (ekf_sym.py:151)"
    for i in range(Z):
      for j in range(EADIM):
        blk.add(f"Hea_{i}_{j}", k.He_sym[i, j])
  stmts, st = blk.lower()
  He = SMat.from_structure(Z, E, st, fmtH)
  he_vars = [c[1] for row in He.e for c in row if c is not None and c[0] == 'var']
  return stmts, st, He, he_vars


def _obs_call(k, project):
  """Phase-1 call of kind k's scalar function for filter `lane` of the tile (extra args / R are per filter)."""
  Z = k.zdim
  feat = k.He_sym is not None
  args = f"sl, s_z + lane * {Z}"
  if ea_dim(k):
    args += f", gea + (base + lane) * {ea_dim(k)}"
  if feat:
    args += f", r_per_filter ? gR + (base + lane) * {Z * Z} : gR"
  return f"scal_obs_{k.kind}{'<' + project + '>' if feat else ''}({args})"


def _slotted(smat, var_list, off):
  """Copy of `smat` whose named entries read the filter's LDS slot instead of a register."""
  m = SMat(smat.rows, smat.cols)
  index = {v: i for i, v in enumerate(var_list)}
  for i in range(smat.rows):
    for j in range(smat.cols):
      e = smat.e[i][j]
      if e is not None and e[0] == 'var':
        m.e[i][j] = ('var', f"sl[{off + index[e[1]]}]")
      else:
        m.e[i][j] = e
  return m


COEF_BATCH_MAX = 40      # coefficients held in registers at once by _in_registers (live: 33 of F, 27 of He; a denser model -- the 24-state
                         # random test model has 86 -- would spill under the 256-register budget of two wavefronts per SIMD: it keeps
                         # the slot reads where they are used)


def _in_registers(smat, name):
  """-> (copy of `smat` whose named entries are `name[i]`, C statements loading them from where the original read them): every
  coefficient of the register-lean matrix phase is read from the slot ONCE, all reads issued before the first use and held there by a
  scheduling barrier -- left alone hipcc sinks each LDS broadcast read next to its use, and every output of the phase becomes a
  read -> wait -> FMA -> write chain of its own.  live, dt > 0 launch at 16 384 filters, same call: 40.5 -> 37.8 us, results bit-identical."""
  m = SMat(smat.rows, smat.cols)
  src = []
  for i in range(smat.rows):
    for j in range(smat.cols):
      e = smat.e[i][j]
      if e is not None and e[0] == 'var':
        if e[1] not in src:
          src.append(e[1])
        m.e[i][j] = ('var', f"{name}[{src.index(e[1])}]")
      else:
        m.e[i][j] = e
  if not src or len(src) > COEF_BATCH_MAX:
    return (m if not src else smat), []
  loads = [f"double {name}[{len(src)}];"] + [f"{name}[{i}] = {v};" for i, v in enumerate(src)] + ["__builtin_amdgcn_sched_barrier(0);"]
  return m, loads


WIDE_Z_LDS = 7      # observation dimension from which the general update keeps the innovation covariance in LDS (below: in registers)


def wide_z(k, E):
  """The LDS path of the innovation covariance (_wide_obs_update: lane z forms row z of S, so Z <= E lanes of the filter's group are needed);
  a kind with more rows than the model has error states (Z > E) keeps the in-register general update whatever its size."""
  return WIDE_Z_LDS <= k.zdim <= E


def wide_s_doubles(k):
  """LDS doubles per filter of a wide kind's innovation covariance: S / its L D U factor in place, the reciprocal pivots, and a copy of
  S for the gated second factorisation."""
  Z = k.zdim
  return Z * Z + Z + (Z * Z if k.maha_test else 0)


def _wide_obs_update(k, Hs, lay, E):
  """The general (reference-on-asymmetric-P) update for a WIDE observation kind: Z >= 7.  In registers the five Z x Z matrices of the
  ordinary path (R, He P He^T, the gated copy, S, its factor) are 405 doubles for Z = 9 -- the 10-state test model with a 9-dimensional
  kind shipped with 516 B of scratch.  Here S lives in the filter's LDS buffer sS: lane z < Z forms row z of S = G He^T + R (the
  generation-time sparsity is in He's rows, the lane only picks its row of G), the group factors it IN PLACE as L D U without pivoting
  -- column by column, lane i eliminating its own row against the broadcast pivot row (two fences per column) -- and every lane solves
  its own right-hand side against the factor with broadcast reads.  R is read from memory where it is used (the gate's 1e16 as a scalar)."""
  Z = k.zdim
  ZZ = Z * Z
  b = [f"double row[{E}];", "#pragma unroll", f"for (int j = 0; j < {E}; j++) row[j] = sP[cc * {E} + j];"]
  b += [f"double col[{E}];", "#pragma unroll", f"for (int kq = 0; kq < {E}; kq++) col[kq] = sP[kq * {E} + cc];", "rn::wave_lds_sync();"]
  b.append(f"double kk[{Z}];")
  for zi in range(Z):
    nz = Hs.row_nz(zi)
    b.append(f"const double G_{zi} = {sum_terms(term(cf, f'col[{kk}]') for kk, cf in nz)};")
    b.append(f"kk[{zi}] = {sum_terms(term(cf, f'row[{kk}]') for kk, cf in nz)};      // Gt")
  b.append("if (act) { " + " ".join(f"sG[{zi} * {E} + cc] = G_{zi};" for zi in range(Z)) + " }")
  b.append("rn::wave_lds_sync();")
  b.append(f"static_assert({Z} <= {E}, \"a lane per row of the innovation covariance\");")
  b.append(f"const bool zrow = act && cc < {Z};")
  b.append(f"const int zc = cc < {Z} ? cc : 0;")
  b.append(f"const double* gz_ = sG + zc * {E};")
  b.append("{")
  for w in range(Z):
    b.append(f"  const double s_{w} = {sum_terms(term(cf, f'gz_[{j}]') for j, cf in Hs.row_nz(w))} + gR[zc * {Z} + {w}];")
  b.append("  if (zrow) { " + " ".join(f"sS[cc * {Z} + {w}] = s_{w};" + (f" sS[{ZZ + Z} + cc * {Z} + {w}] = s_{w};" if k.maha_test else "") for w in range(Z)) + " }")
  b.append("}")
  b.append("rn::wave_lds_sync();")

  def factor():
    f = []
    for j in range(Z):
      f.append("{")
      f.append(f"  const double id_ = rn::fast_recip(sS[{j * Z + j}]);")
      if j + 1 < Z:
        f.append(f"  double pj[{Z - j - 1}], rw[{Z - j - 1}];")
        f += ["#pragma unroll", f"  for (int m = {j + 1}; m < {Z}; m++) {{ pj[m - {j + 1}] = sS[{j * Z} + m]; rw[m - {j + 1}] = sS[zc * {Z} + m]; }}"]
        f.append(f"  const double l_ = sS[zc * {Z} + {j}] * id_;")
      f.append("  rn::wave_lds_sync();")
      f.append("  if (zrow) {")
      if j + 1 < Z:
        f.append(f"    if (cc > {j}) {{")
        f.append(f"      sS[cc * {Z} + {j}] = l_;")
        f += ["#pragma unroll", f"      for (int m = {j + 1}; m < {Z}; m++) sS[cc * {Z} + m] = rw[m - {j + 1}] - l_ * pj[m - {j + 1}];", "    }"]
        f.append(f"    if (cc == {j}) {{")
        f += ["#pragma unroll", f"      for (int m = {j + 1}; m < {Z}; m++) sS[{j * Z} + m] = pj[m - {j + 1}] * id_;", "    }"]
      f.append(f"    if (cc == {j}) sS[{ZZ + j}] = id_;")
      f.append("  }")
      f.append("  rn::wave_lds_sync();")
      f.append("}")
    return f
  b += factor()
  b.append("int gated = 0;")
  b.append("double rs = 1.0;")
  if k.maha_test:
    ys = [f"sl[{lay.OFF_Y + i}]" for i in range(Z)]
    b += ["{", f"  double v[{Z}] = {{{', '.join(ys)}}}, w[{Z}] = {{{', '.join(ys)}}};"]
    for i in range(Z):
      if i:
        b.append(f"  v[{i}] -= " + " + ".join(f"sS[{i * Z + kq}]*v[{kq}]" for kq in range(i)) + ";")
        b.append(f"  w[{i}] -= " + " + ".join(f"sS[{kq * Z + i}]*w[{kq}]" for kq in range(i)) + ";")
    b.append("  const double d2 = " + " + ".join(f"v[{i}]*w[{i}]*sS[{ZZ + i}]" for i in range(Z)) + ";")
    b.append(f"  gated = d2 > {k.maha_thresh!r} ? 1 : 0;      // (uniform over the filter's lanes: every lane evaluates the same numbers)")
    b.append("}")
    b.append("rn::wave_lds_sync();")
    b.append("if (gated) {      // R *= 1e16 (ekf_c.c:88-94) and a second factorisation, from the kept copy of S = He P He^T + R")
    b.append("  rs = 1.0e16;")
    b.append("  if (zrow) { " + " ".join(f"sS[cc * {Z} + {w}] = sS[{ZZ + Z} + cc * {Z} + {w}] + (1.0e16 - 1.0) * gR[cc * {Z} + {w}];" for w in range(Z)) + " }")
    b.append("  rn::wave_lds_sync();")
    b += ["  " + ln for ln in factor()]
    b.append("}")
  # kk <- (L D U)^-1 Gt: the factor by broadcast reads
  for i in range(1, Z):
    b.append(f"kk[{i}] -= " + " + ".join(f"sS[{i * Z + kq}]*kk[{kq}]" for kq in range(i)) + ";")
  for i in range(Z - 1, -1, -1):
    tail = "".join(f" - sS[{i * Z + kq}]*kk[{kq}]" for kq in range(i + 1, Z))
    b.append(f"kk[{i}] = kk[{i}]*sS[{ZZ + i}]{tail};")
  b.append("const double dxc = " + " + ".join(f"kk[{zi}]*sl[{lay.OFF_Y + zi}]" for zi in range(Z)) + ";")
  for j in range(E):
    b.append(f"row[{j}] -= " + " + ".join(f"kk[{zi}]*sG[{zi * E + j}]" for zi in range(Z)) + ";")
  for zi in range(Z):
    c = sum_terms(term(cf, f"row[{j}]") for j, cf in Hs.row_nz(zi))
    kr = " + ".join(f"kk[{w}]*gR[{w * Z + zi}]" for w in range(Z))
    b.append(f"const double Dm_{zi} = rs*({kr}) - ({c});")
  b.append("if (act) { " + " ".join(f"sK[{zi} * {E} + cc] = kk[{zi}];" for zi in range(Z)) + f" sw[{lay.OFF_DX} + cc] = dxc; if (cc == 0) sw[{lay.OFF_FL}] = (double)gated; }}")
  b.append("rn::wave_lds_sync();")
  for j in range(E):
    b.append(f"row[{j}] += " + " + ".join(f"Dm_{zi}*sK[{zi * E + j}]" for zi in range(Z)) + ";")
  b += ["if (act) {", "#pragma unroll", f"  for (int j = 0; j < {E}; j++) sP[cc * {E} + j] = row[j];", "}", "rn::wave_lds_sync();"]
  return b


def _lean_update(k, Hs, lay, E, rows_in_regs=False):
  """Update with the Joseph correction Dm = K R - B He^T formed from Gt - K (He P He^T) (the same quantity, B = P - K G
  never materialised), only the columns of P that He touches read, and ONE pass over the lane's row:
  P'[cc, j] = (P[cc, j] - sum_z K_z G[z, j]) + sum_z Dm_z K[j, z].  rows_in_regs=False leaves the row in LDS (rolled
  in-place pass, a few dozen live registers: two or more wavefronts per SIMD).

  Feature-track kinds (k.He_sym, MSCKF): G, Gt and the rows of He P He^T are taken to the left null space of the
  extra-argument Jacobian by the reflectors phase 1 left in the slot (ekf_c.c:66-76: H <- A^T H); everything after
  that is the ordinary update with Z - EADIM rows and the projected y / R of the slot."""
  Zf = k.zdim
  feat = k.He_sym is not None
  Z = Zf - EADIM if feat else Zf
  used = sorted({kk for zi in range(Zf) for kk, _ in Hs.row_nz(zi)})
  b = [f"double R[{Z * Z}];", f"double* pr = sP + cc * {E};"]
  if not feat:      # the non-zeros of He (27 for live) are read from the slot once, up front, behind a scheduling barrier (_in_registers)
    Hs, coef_loads = _in_registers(Hs, "hc")
    b += coef_loads
  if rows_in_regs:
    b += [f"double row[{E}];", "#pragma unroll", f"for (int j = 0; j < {E}; j++) row[j] = pr[j];"]
    b += [f"const double col_{kk} = sP[{kk} * {E} + cc], row_{kk} = row[{kk}];" for kk in used]
  else:
    b += [f"const double col_{kk} = sP[{kk} * {E} + cc], row_{kk} = pr[{kk}];" for kk in used]
  if feat:
    b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) R[i] = sl[{lay.OFF_RP} + i];      // A^T R A (phase 1)", "(void)gR;"]
    b.append(f"double G0[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'col_{kk}') for kk, cf in Hs.row_nz(zi)) for zi in range(Zf)) + "};")
    b.append(f"double Gt0[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'row_{kk}') for kk, cf in Hs.row_nz(zi)) for zi in range(Zf)) + "};")
    b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(sl + {lay.OFF_RF}, sl + {lay.OFF_RF + EADIM * Zf}, G0);")
    b.append(f"rn::apply_reflectors<{Zf}, {EADIM}>(sl + {lay.OFF_RF}, sl + {lay.OFF_RF + EADIM * Zf}, Gt0);")
    for zi in range(Z):
      b.append(f"const double G_{zi} = G0[{EADIM + zi}], Gt_{zi} = Gt0[{EADIM + zi}];")
    b.append(f"const double rank_deficient = sl[{lay.OFF_FL}];      // 4.0 when phase 1 found Hea rank deficient")
  else:
    b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) R[i] = gR[i];"]
    for zi in range(Z):
      nz = Hs.row_nz(zi)
      b.append(f"const double G_{zi} = {sum_terms(term(cf, f'col_{kk}') for kk, cf in nz)};")
      b.append(f"const double Gt_{zi} = {sum_terms(term(cf, f'row_{kk}') for kk, cf in nz)};")
  b.append("if (act) { " + " ".join(f"sG[{zi} * {E} + cc] = G_{zi};" for zi in range(Z)) + " }")
  b.append("rn::wave_lds_sync();")
  b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
  if feat:
    for zi in range(Z):
      b.append("{")
      b.append(f"  double m[{Zf}] = {{" + ", ".join(sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w)) for w in range(Zf)) + "};")
      b.append(f"  rn::apply_reflectors<{Zf}, {EADIM}>(sl + {lay.OFF_RF}, sl + {lay.OFF_RF + EADIM * Zf}, m);")
      b += ["#pragma unroll", f"  for (int w = 0; w < {Z}; w++) HPH[{zi * Z} + w] = m[{EADIM} + w];", "}"]
  else:
    for zi in range(Z):
      for w in range(Z):
        b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w))};")
  # S is solved as a GENERAL matrix (L D U): the step-granular kernels follow the reference on asymmetric covariances (ekf_c.c:100-101)
  YO = lay.OFF_YP if feat else lay.OFF_Y      # the residual the update consumes (feature-track kinds: the one in the reflectors' basis)
  factor, gate, solve = innovation_solver(Z, True, [f"sl[{YO + i}]" for i in range(Z)], k.maha_thresh if k.maha_test else None)
  b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}", factor, "int gated = 0;"]
  b += gate
  b.append(f"double kk[{Z}] = {{{', '.join(f'Gt_{zi}' for zi in range(Z))}}};")
  b.append(solve("kk"))
  if feat:     # the reference's numpy path ignores a measurement whose null-space projection failed (ekf_sym.py:589-591)
    b += ["if (rank_deficient != 0.0) {", "#pragma unroll", f"  for (int i = 0; i < {Z}; i++) kk[i] = 0.0;", "}"]
  b.append("const double dxc = " + " + ".join(f"kk[{zi}]*sl[{YO + zi}]" for zi in range(Z)) + ";")
  for zi in range(Z):
    c = f"Gt_{zi} - (" + " + ".join(f"kk[{w}]*HPH[{w * Z + zi}]" for w in range(Z)) + ")"
    kr = " + ".join(f"kk[{w}]*Rl[{w * Z + zi}]" for w in range(Z))
    b.append(f"const double Dm_{zi} = " + ("rank_deficient != 0.0 ? 0.0 : " if feat else "") + f"({kr}) - ({c});")
  fl = "(double)gated + rank_deficient" if feat else "(double)gated"
  b.append("if (act) { " + " ".join(f"sK[{zi} * {E} + cc] = kk[{zi}];" for zi in range(Z)) + f" sw[{lay.OFF_DX} + cc] = dxc; if (cc == 0) sw[{lay.OFF_FL}] = {fl}; }}")
  b.append("rn::wave_lds_sync();")
  src = "row[j]" if rows_in_regs else "pr[j]"
  b += ["if (act) {", "#pragma unroll" if rows_in_regs else "#pragma unroll 2", f"  for (int j = 0; j < {E}; j++) {{",
        f"    const double bj = {src} - (" + " + ".join(f"kk[{zi}]*sG[{zi * E} + j]" for zi in range(Z)) + ");",
        "    pr[j] = bj + (" + " + ".join(f"Dm_{zi}*sK[{zi * E} + j]" for zi in range(Z)) + ");", "  }", "}", "rn::wave_lds_sync();"]
  return b


def device_functions(spec, lay_cls=None, sfx=""):
  """Phase functions of the three-phase kernels -> (text, slot layout).  With `lay_cls` / `sfx` only the scalar phases are
  emitted, against another slot layout and under suffixed names (the fused runs keep more compact slots: emit_wide3, emit_run2)."""
  D, E = spec.dim_x, spec.dim_err
  INL = "__forceinline__" if tuning.current().wide_inline else "__noinline__"
  pst, pstruct, F, f_vars = _lowered_predict(spec)
  obs = {k.kind: _lowered_obs(spec, k) for k in spec.kinds}
  lay = (lay_cls or Layout)(spec, f_vars, {kk: v[3] for kk, v in obs.items()})
  out = []

  # ---- phase 1: scalars of predict ---------------------------------------------------------------
  # Phase functions have pointer-only interfaces (state lives in LDS between them) and are inlined by default: inlined
  # into the group loop hipcc keeps ~460 registers live (1 wave/SIMD); __noinline__ (tuning knob wide_inline=0) keeps each
  # under 256 but pays scratch frames and is 5x slower.
  quat = "".join(f" rn::normalize_quat<{D}>(x, {q});" for q in spec.quaternion_idxs)
  normq = f"if (norm_quats) {{{quat} }}" if spec.quaternion_idxs else "(void)norm_quats;"
  b = [f"double x[{D}];", "#pragma unroll", f"for (int i = 0; i < {D}; i++) x[i] = xin[i];"]
  # every non-trivial entry of F goes to the slot as soon as it exists (short live ranges: a dense F is hundreds of values)
  f_at = {v: i for i, v in enumerate(f_vars)}
  stored = set()
  for st_ in pst:
    b.append(st_)
    mm = re.match(r"\s*const double (\w+) =", st_)
    if mm and mm.group(1) in f_at:
      b.append(f"sl[{lay.OFF_F + f_at[mm.group(1)]}] = {mm.group(1)};")
      stored.add(mm.group(1))
      if len(stored) % 32 == 0:
        b.append("rn::wave_lds_sync();      // (a scheduling boundary: keeps hipcc from computing everything before storing anything)")
  for i, v in enumerate(f_vars):
    if v not in stored:
      b.append(f"sl[{lay.OFF_F + i}] = {v};")
  for i in range(D):
    kind, val = pstruct[f"xn_{i}"]
    b.append(f"x[{i}] = xn_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  b.append(normq)
  b.append(f"sl[{lay.OFF_DT}] = dt;")
  xw = f"sl[{lay.OFF_X} + i]"      # where the new state goes
  xarg = "const double* xin"
  b += ["#pragma unroll", f"for (int i = 0; i < {D}; i++) {xw} = x[i];"]
  out.append("\n".join(["__device__ {INL} void scal_predict" + sfx + f"({xarg}, const double dt, double* sl, const int norm_quats) {{"] + _ind(b) + ["}"]))

  b = [f"double x[{D}];", "#pragma unroll", f"for (int i = 0; i < {D}; i++) x[i] = xin[i];", normq,
       "#pragma unroll", f"for (int i = 0; i < {D}; i++) {xw} = x[i];"]
  out.append("\n".join(["__device__ {INL} void scal_keep" + sfx + f"({xarg}, double* sl, const int norm_quats) {{"] + _ind(b) + ["}"]))

  # ---- phase 1: scalars of each observation kind ---------------------------------------------------
  for k in spec.kinds:
    stmts, st, He, he_vars = obs[k.kind]
    Z = k.zdim
    EA = ea_dim(k)
    feat = k.He_sym is not None
    xr_ = f"sl[{lay.OFF_X} + i]"
    b = [f"double x[{D}], z[{Z}];", "#pragma unroll", f"for (int i = 0; i < {D}; i++) x[i] = {xr_};",
         "#pragma unroll", f"for (int i = 0; i < {Z}; i++) z[i] = zin[i];"]
    if EA:
      b += [f"double ea[{EA}];", "#pragma unroll", f"for (int i = 0; i < {EA}; i++) ea[i] = eain[i];"]
    b += list(stmts)
    val = lambda nm: nm if st[nm][0] == 'expr' else repr(float(st[nm][1] or 0.0))  # noqa: E731
    for i, v in enumerate(he_vars):
      b.append(f"sl[{lay.OFF_HE + i}] = {v};")
    if not feat:
      for i in range(Z):
        b.append(f"sl[{lay.OFF_Y + i}] = z[{i}] - {val(f'hx_{i}')};")
    else:
      # ekf_c.c:66-76: residual and R go to the left null space of Hea; the reflectors stay in the slot for phase 2
      Zp = Z - EADIM
      b.append(f"double y[{Z}] = {{{', '.join(f'z[{i}] - ' + val(f'hx_{i}') for i in range(Z))}}};")
      b.append("if (PROJECT) {")
      hea = ", ".join(("0.0" if st[f"Hea_{i}_{j}"][0] == 'zero' else ("1.0" if st[f"Hea_{i}_{j}"][0] == 'one' else val(f"Hea_{i}_{j}")))
                      for i in range(Z) for j in range(EADIM))
      b += [f"  double Hea[{Z * EADIM}] = {{{hea}}};", f"  double u[{EADIM * Z}], beta[{EADIM}], yref[{Zp}];",
            f"  const bool ok2 = rn::nullspace_residual<{Z}, {EADIM}>(Hea, y, yref);      // y in the reference's basis (ekf_c.c:71-73); Hea and y are modified below",
            f"  const bool ok = rn::householder_qr<{Z}, {EADIM}>(Hea, u, beta) && ok2;",
            f"  rn::apply_reflectors<{Z}, {EADIM}>(u, beta, y);",
            "#pragma unroll", f"  for (int i = 0; i < {EADIM * Z}; i++) sl[{lay.OFF_RF} + i] = u[i];",
            "#pragma unroll", f"  for (int i = 0; i < {EADIM}; i++) sl[{lay.OFF_RF + EADIM * Z} + i] = beta[i];",
            "#pragma unroll", f"  for (int i = 0; i < {Zp}; i++) {{ sl[{lay.OFF_YP} + i] = ok ? y[{EADIM} + i] : 0.0; sl[{lay.OFF_Y} + i] = ok ? yref[i] : 0.0; }}",
            "#pragma unroll", f"  for (int i = {Zp}; i < {Z}; i++) sl[{lay.OFF_Y} + i] = z[i];      // y has Z - EADIM rows (ekf_c.c:120)",
            f"  sl[{lay.OFF_FL}] = ok ? 0.0 : 4.0;",
            f"  double Rm[{Z * Z}];", "#pragma unroll", f"  for (int i = 0; i < {Z * Z}; i++) Rm[i] = gRf[i];",
            f"  rn::project_noise<{Z}, {EADIM}>(u, beta, Rm);",
            "#pragma unroll", f"  for (int a = 0; a < {Zp}; a++) {{", "#pragma unroll",
            f"    for (int c = 0; c < {Zp}; c++) sl[{lay.OFF_RP} + a * {Zp} + c] = Rm[({EADIM} + a) * {Z} + {EADIM} + c];", "  }",
            "} else {", "#pragma unroll", f"  for (int i = 0; i < {Z}; i++) sl[{lay.OFF_Y} + i] = y[i];", "}"]
    tmpl = "template <bool PROJECT>\n" if feat else ""
    sig = "double* sl, const double* zin" + (", const double* eain" if EA else "") + (", const double* gRf" if feat else "")
    out.append("\n".join([f"{tmpl}__device__ {{INL}} void scal_obs_{k.kind}{sfx}({sig}) {{"] + _ind(b) + ["}"]))

  # ---- phase 3: error injection ----------------------------------------------------------------------
  nom, delta = spec.err_eqs[1], spec.err_eqs[2]
  enames = dict(vector_names(nom, 'x'))
  enames.update({(delta, i, 0): f"sl[{lay.OFF_DX + i}]" for i in range(E)})
  eblk = Block(enames, tmp_prefix="et")
  for i in range(D):
    eblk.add(f"xi_{i}", sp.Matrix(spec.err_eqs[0])[i])
  estmts, est = eblk.lower()
  b = [f"double x[{D}];", "#pragma unroll", f"for (int i = 0; i < {D}; i++) x[i] = sl[{lay.OFF_X} + i];"]
  b += list(estmts)
  for i in range(D):
    kind, val = est[f"xi_{i}"]
    b.append(f"x[{i}] = xi_{i};" if kind == 'expr' else f"x[{i}] = {float(val)!r};")
  b.append(normq)
  b += ["#pragma unroll", f"for (int i = 0; i < {D}; i++) xout[i] = x[i];", "double acc = 0.0;", "#pragma unroll",
        f"for (int i = 0; i < {D}; i++) acc += x[i];", "return (acc - acc == 0.0) ? 0 : 2;"]
  out.append("\n".join(["__device__ {INL} int scal_inject" + sfx + "(const double* sl, double* xout, const int norm_quats) {"] + _ind(b) + ["}"]))
  if lay_cls is not None:
    return "\n\n".join(out).replace("{INL}", INL), lay

  # ---- phase 2: predict, matrix part (P in sP -> P' in sP) ------------------------------------------------
  Fs = _slotted(F, f_vars, lay.OFF_F)
  lean = tuning.current().wide_lean
  lean_p = lean == 1      # predict through LDS only in the fully lean variant
  if lean_p:
    # rows, then columns, pass through ONE register array; every result goes straight back to LDS (in place: the lane's
    # own row / column is in registers, other lanes' are untouched), so nothing but the array stays live
    b = [f"const double dt = sl[{lay.OFF_DT}];", f"double v[{E}];"]
    Fp, coef_loads = _in_registers(Fs, "fc")      # the non-zeros of F (33 for live): one batch of slot reads, not one read-wait-FMA chain per output
    b += coef_loads
    b += ["if (act) {", "#pragma unroll", f"  for (int j = 0; j < {E}; j++) v[j] = sP[cc * {E} + j];"]
    for i in range(E):
      b.append(f"  sP[cc * {E} + {i}] = {sum_terms(term(cf, f'v[{kk}]') for kk, cf in Fp.row_nz(i))};")
    b += ["}", "rn::wave_lds_sync();", "if (act) {", "#pragma unroll", f"  for (int k = 0; k < {E}; k++) v[k] = sP[k * {E} + cc];"]
    for i in range(E):
      qv = f"qcol[{i}]" if tuning.current().wide_lean_q else f"sQ[{i} * {E} + cc]"
      b.append(f"  sP[{i} * {E} + cc] = {sum_terms(term(cf, f'v[{kk}]') for kk, cf in Fp.row_nz(i))} + dt*{qv};")
    b += ["}", "rn::wave_lds_sync();"]
  else:
    b = [f"const double dt = sl[{lay.OFF_DT}];", f"double row[{E}], a[{E}], col[{E}];", "#pragma unroll",
         f"for (int j = 0; j < {E}; j++) row[j] = sP[cc * {E} + j];"]
    for i in range(E):
      b.append(f"a[{i}] = {sum_terms(term(cf, f'row[{kk}]') for kk, cf in Fs.row_nz(i))};")
    b += ["if (act) {", "#pragma unroll", f"  for (int i = 0; i < {E}; i++) sP[cc * {E} + i] = a[i];", "}", "rn::wave_lds_sync();",
          "#pragma unroll", f"for (int k = 0; k < {E}; k++) a[k] = sP[k * {E} + cc];"]
    for i in range(E):
      b.append(f"col[{i}] = {sum_terms(term(cf, f'a[{kk}]') for kk, cf in Fs.row_nz(i))} + dt*qcol[{i}];")
    b += ["rn::wave_lds_sync();", "if (act) {", "#pragma unroll", f"  for (int k = 0; k < {E}; k++) sP[k * {E} + cc] = col[k];", "}",
          "rn::wave_lds_sync();"]
  qarg = "const double* sQ" if (lean_p and not tuning.current().wide_lean_q) else f"const double (&qcol)[{E}]"
  out.append("\n".join([f"__device__ {INL} void mat_predict(double* sP, {qarg}, const double* sl, const int cc, const bool act) {{"]
                        + _ind(b) + ["}"]))

  # ---- smoother (templates/ekf_hip_rts.h, k_rts_group): main block of the predicted pair from a row held in registers -----
  # row = row c of Pk_k[:M, :M]; y <- column c of F Pk_k^T; sB (M x M, stride M) <- F Pk_k F^T + dt Q[:M, :M].
  # Only two register vectors are live at a time (y + the column of P F^T): each entry of the result goes to LDS as soon as
  # it is formed -- the lane rewrites its OWN column of sB, which no other lane reads in this function.
  M = spec.dim_main_err
  b = [f"const double dt = sl[{lay.OFF_DT}];"]
  for i in range(M):
    b.append(f"y[{i}] = {sum_terms(term(cf, f'row[{kk}]') for kk, cf in Fs.row_nz(i) if kk < M)};")
  b += ["if (act) {", "#pragma unroll", f"  for (int i = 0; i < {M}; i++) sB[cc * {M} + i] = y[i];", "}", "rn::wave_lds_sync();",
        f"double a[{M}];", "#pragma unroll", f"for (int k = 0; k < {M}; k++) a[k] = sB[k * {M} + cc];", "rn::wave_lds_sync();"]
  for i in range(M):
    b.append(f"{{ const double v = {sum_terms(term(cf, f'a[{kk}]') for kk, cf in Fs.row_nz(i) if kk < M)} + dt*gQc[{i * E}]; if (act) sB[{i * M} + cc] = v; }}")
  b += ["rn::wave_lds_sync();"]
  out.append("\n".join([f"__device__ __forceinline__ void mat_predict_rts(const double (&row)[{M}], double* sB, const double* __restrict__ gQc, "
                        f"const double* sl, const int cc, const bool act, double (&y)[{M}]) {{"] + _ind(b) + ["}"]))

  # ---- phase 2: update, matrix part ----------------------------------------------------------------------
  for k in spec.kinds:
    _, _, He, he_vars = obs[k.kind]
    Hs = _slotted(He, he_vars, lay.OFF_HE)
    Z = k.zdim
    if lean == 1:
      b = _lean_update(k, Hs, lay, E)
    elif k.He_sym is not None:
      b = _lean_update(k, Hs, lay, E, rows_in_regs=True)
    elif wide_z(k, E):
      b = _wide_obs_update(k, Hs, lay, E)
    else:
      b = [f"double row[{E}], R[{Z * Z}];", "#pragma unroll", f"for (int j = 0; j < {E}; j++) row[j] = sP[cc * {E} + j];"]
      b += [f"double col[{E}];", "#pragma unroll", f"for (int kq = 0; kq < {E}; kq++) col[kq] = sP[kq * {E} + cc];"]
      b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) R[i] = gR[i];", "rn::wave_lds_sync();"]
      for zi in range(Z):
        nz = Hs.row_nz(zi)
        b.append(f"const double G_{zi} = {sum_terms(term(cf, f'col[{kk}]') for kk, cf in nz)};")
        b.append(f"const double Gt_{zi} = {sum_terms(term(cf, f'row[{kk}]') for kk, cf in nz)};")
      b.append("if (act) { " + " ".join(f"sG[{zi} * {E} + cc] = G_{zi};" for zi in range(Z)) + " }")
      b.append("rn::wave_lds_sync();")
      b.append(f"double HPH[{Z * Z}], Rl[{Z * Z}], S[{Z * Z}], L[{Z * Z}], iL[{Z}];")
      for zi in range(Z):
        for w in range(Z):
          b.append(f"HPH[{zi * Z + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w))};")
      factor, gate, solve = innovation_solver(Z, True, [f"sl[{lay.OFF_Y + i}]" for i in range(Z)], k.maha_thresh if k.maha_test else None)
      b += ["#pragma unroll", f"for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = R[i]; S[i] = HPH[i] + Rl[i]; }}", factor, "int gated = 0;"]
      b += gate
      b.append(f"double kk[{Z}] = {{{', '.join(f'Gt_{zi}' for zi in range(Z))}}};")
      b.append(solve("kk"))
      b.append("const double dxc = " + " + ".join(f"kk[{zi}]*sl[{lay.OFF_Y + zi}]" for zi in range(Z)) + ";")
      for j in range(E):
        b.append(f"row[{j}] -= " + " + ".join(f"kk[{zi}]*sG[{zi * E + j}]" for zi in range(Z)) + ";")
      for zi in range(Z):
        c = sum_terms(term(cf, f"row[{j}]") for j, cf in Hs.row_nz(zi))
        kr = " + ".join(f"kk[{w}]*Rl[{w * Z + zi}]" for w in range(Z))
        b.append(f"const double Dm_{zi} = ({kr}) - ({c});")
      b.append("if (act) { " + " ".join(f"sK[{zi} * {E} + cc] = kk[{zi}];" for zi in range(Z)) + f" sw[{lay.OFF_DX} + cc] = dxc; if (cc == 0) sw[{lay.OFF_FL}] = (double)gated; }}")
      b.append("rn::wave_lds_sync();")
      for j in range(E):
        b.append(f"row[{j}] += " + " + ".join(f"Dm_{zi}*sK[{zi * E + j}]" for zi in range(Z)) + ";")
      b += ["if (act) {", "#pragma unroll", f"  for (int j = 0; j < {E}; j++) sP[cc * {E} + j] = row[j];", "}", "rn::wave_lds_sync();"]
    ss_arg = ", double* sS" if wide_z(k, E) and lean != 1 and k.He_sym is None else ""
    out.append("\n".join([f"__device__ {INL} void mat_update_{k.kind}(double* sP, const double* __restrict__ gR, const double* sl, double* sw, "
                          f"double* sG, double* sK{ss_arg}, const int cc, const bool act) {{"] + _ind(b) + ["}"]))
  return "\n".join(out).replace("{INL}", INL), lay


def kernels(spec):
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  FT = tile_filters(spec)
  FPW = filters_per_wave(spec)
  GL = group_lanes(spec)
  fn_text, lay = device_functions(spec)
  out = [f"// ---- family W, three-phase step kernels (tile of {FT} filters per wavefront, slot = {lay.SLOT} doubles) ----",
         f"constexpr int FT2 = {FT};", f"constexpr int SLOT = {lay.SLOT};", f"constexpr int SLOT_OFF_X = {lay.OFF_X};",
         f"constexpr int SLOT_OFF_DT = {lay.OFF_DT};", fn_text]
  tune = tuning.current()
  lbs = f"__launch_bounds__(64, {tune.wide_lb})" if tune.wide_lb else "__launch_bounds__(64)"
  if tune.wide_timeline:
    out.append("__device__ unsigned long long g_tl[256 * 64 * 2];      // debug timeline (tuning knob wide_timeline)")
    out.append("__device__ unsigned long long g_tlb[4096 * 2];         // start / end of EVERY workgroup's first tile")

  def kernel(kname, k=None, ckpt=False):
    # ckpt: the kernel also writes a CHECKPOINT -- the observations as they came (cz), the filtered pair (cx, cP) --, what the orchestrators' rewind
    # rings keep of every call (ekf_sym.cc:142-156, 191); a kernel of its own (k_stepc_{kind}), k_step_{kind} stays as it is
    upd = k is not None
    Z = k.zdim if upd else 1
    ZZ = Z * Z
    wide_s = upd and wide_z(k, spec.dim_err) and tune.wide_lean != 1 and k.He_sym is None
    tmpl = "template <bool DO_PREDICT>\n" if upd else ""
    dop = "DO_PREDICT" if upd else "true"
    sig_obs = ("double* __restrict__ gz, const double* __restrict__ gR, const int r_per_filter, const double* __restrict__ gea,\n    "
               if upd else "")
    flags_arg = (", uint8_t* __restrict__ flags" if upd else "") + ", const uint8_t* __restrict__ active"
    if ckpt:
      flags_arg += ", double* __restrict__ cx, double* __restrict__ cP, double* __restrict__ cz"
    L = []
    A = L.append
    TLK = tune.wide_timeline

    def TL(idx):      # debug stamps (tuning knob wide_timeline): [block][slot][0] = shader cycles, [1] = 100 MHz wall clock
      if TLK:
        A(f"    if (lane == 0 && blockIdx.x < 256) {{ const int ti_ = {idx}; if (ti_ < 64) {{ g_tl[(blockIdx.x * 64 + ti_) * 2] = __builtin_readcyclecounter(); g_tl[(blockIdx.x * 64 + ti_) * 2 + 1] = wall_clock64(); }} }}")
    A(f"{tmpl}__global__ {lbs} void {kname}(double* __restrict__ gx, double* __restrict__ gP,")
    A(f"    {sig_obs}const double* __restrict__ gQ, const double* __restrict__ gdt, const double dt_scalar, const int64_t n,")
    A(f"    const int norm_quats{flags_arg}) {{")
    DB = 1 if double_buffered(spec) else 0
    ODD = (FPW * EE) % 2 == 1 or (FT * EE) % 2 == 1       # can a group's record start on an odd double?
    PBUF = (FPW * EE + (3 if ODD else 1)) // 2 * 2      # one slack double for the shifted image, whole 16-byte vectors
    CPIN = "rn::async_copy_g2l_any" if ODD else "rn::async_copy_g2l"
    A(f"  __shared__ __attribute__((aligned(16))) double s_P[{1 + DB}][{PBUF}];     // double buffer: group p computes, group p+1 lands")
    A(f"  __shared__ __attribute__((aligned(16))) double s_x[FT2 * {D} + 2];")
    if upd:
      A(f"  __shared__ __attribute__((aligned(16))) double s_z[FT2 * {Z} + 2];")
      A(f"  __shared__ __attribute__((aligned(16))) double s_G[{FPW} * {Z * E}];")
      A(f"  __shared__ __attribute__((aligned(16))) double s_K[{FPW} * {Z * E}];")
      if wide_s:
        A(f"  __shared__ __attribute__((aligned(16))) double s_S[{FPW} * {wide_s_doubles(k)}];      // the wide kind's innovation covariance, factored in place (_wide_obs_update)")
    A("  __shared__ __attribute__((aligned(16))) double s_sl[FT2 * SLOT];")
    A("  const int lane = threadIdx.x;")
    A(f"  const int g = lane / {GL};")
    A(f"  const int c = lane % {GL};")
    A(f"  const bool act = c < {E} && g < {FPW};")
    A("  const int cc = act ? c : 0;")
    if tune.wide_lean == 1 and not tune.wide_lean_q:
      A(f"  __shared__ __attribute__((aligned(16))) double s_Q[{EE}];      // process noise, staged once per wavefront")
      A(f"  for (int i = lane; i < {EE}; i += 64) s_Q[i] = ({dop} && gQ != nullptr) ? gQ[i] : 0.0;")
      A("  const double* qcol = s_Q;")
    elif tune.wide_lean == 1 and tune.wide_lean_q:
      A(f"  double qcol[{E}];                          // column cc of Q: loaded per tile AFTER the scalar phase (see below)")
    else:
      A(f"  double qcol[{E}];                          // column cc of Q, resident for the whole launch")
      A("#pragma unroll")
      A(f"  for (int i = 0; i < {E}; i++) qcol[i] = ({dop} && gQ != nullptr) ? gQ[i * {E} + cc] : 0.0;")
    if spec.identity_at_dt0():
      A("  // predict with a uniform dt == 0 (a second observation at the same timestamp) is the identity on (x, P) for finite")
      A("  // states: f(x, 0) == x and F(x, 0) == I were checked SYMBOLICALLY for this model at generation time")
      A("  // (FilterSpec.identity_at_dt0) and dt Q = 0, so the covariance phase is skipped; results are unchanged.")
      A(f"  const bool do_pred = {dop} && !(gdt == nullptr && dt_scalar == 0.0);")
    else:
      A("  // this model's f(x, 0) != x or F(x, 0) != I: predict runs on every call, dt == 0 included (ekf_c.c:15-28)")
      A(f"  const bool do_pred = {dop};")
    A("  const int64_t tiles = (n + FT2 - 1) / FT2;")
    A("  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {")
    A("    const int64_t base = tile * FT2;")
    A("    const int cnt = (n - base) < FT2 ? (int)(n - base) : FT2;")
    TL(0)
    if TLK:
      A("    if (lane == 0 && tile == blockIdx.x && blockIdx.x < 4096) g_tlb[blockIdx.x * 2] = wall_clock64();")
    A("    // ---------------- phase 1: lane l = filter l, x-dependent scalars -> LDS slot ----------------")
    A(f"    rn::copy_g2l<FT2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);")
    if upd:
      A(f"    rn::copy_g2l<FT2 * {Z}>(gz + base * {Z}, cnt * {Z}, s_z, lane);")
    if DB:
      A(f"    {CPIN}<{PBUF}>(gP + base * {EE}, (cnt < {FPW} ? cnt : {FPW}) * {EE}, s_P[0], lane);")
    A("    rn::wave_lds_sync();")
    if ckpt:
      A(f"    rn::copy_l2g<FT2 * {Z}>(cz + base * {Z}, cnt * {Z}, s_z, lane);      // the observations, before the residuals take their place")
    TL(1)
    A("    if (lane < cnt) {")
    A("      double* sl = s_sl + lane * SLOT;")
    A("      if (do_pred) {")
    A("        int ld_ = lane;")
    A('        asm volatile("" : "+v"(ld_));      // (the address gdt + lane is otherwise formed at the kernel\'s entry and kept -- in scratch, where registers are short -- for this one use)')
    A("        const double dt = gdt != nullptr ? gdt[base + ld_] : dt_scalar;")
    A(f"        scal_predict(s_x + lane * {D}, dt, sl, norm_quats);")
    A("      } else {")
    A(f"        scal_keep(s_x + lane * {D}, sl, {dop} ? norm_quats : 0);      // predict(dt = 0) still renormalises")
    A("      }")
    if upd:
      A("      __builtin_amdgcn_sched_barrier(0);       // f / F first, then h / H: interleaved for ILP they need the sum of both register sets")
      A(f"      {_obs_call(k, 'true')};")
    A("    }")
    A("    rn::wave_lds_sync();")
    if tune.wide_lean == 1 and tune.wide_lean_q:
      A("    {")
      A("      // Q's column is fetched here, behind an opaque zero, so that its 2 x dim_err registers are not live during the scalar")
      A("      // phase above (the widest point of the kernel: with them it spilled); an L2 hit per tile, hidden under the first P wait")
      A("      int qoff = 0;")
      A("      asm volatile(\"\" : \"+v\"(qoff) :: \"memory\");")
      A("#pragma unroll")
      A(f"      for (int i = 0; i < {E}; i++) qcol[i] = ({dop} && gQ != nullptr) ? gQ[i * {E} + cc + qoff] : 0.0;")
      A("    }")
    TL(2)
    A(f"    // ---------------- phase 2: {GL}-lane group per filter, {FPW} filters at a time, covariance algebra ----------")
    A(f"    const int ngroups = (cnt + {FPW - 1}) / {FPW};")
    A("    for (int p = 0; p < ngroups; p++) {")
    A(f"      const int pcnt = (cnt - {FPW} * p) < {FPW} ? (cnt - {FPW} * p) : {FPW};")
    A(f"      double* gPp = gP + (base + {FPW} * p) * {EE};")
    A("      // a group record may start on an odd double (odd dim_err^2 x odd group index): the LDS image is then shifted")
    A("      // by one double so that the 16-byte transfers stay aligned on both sides")
    A("      const int sh = rn::odd_start(gPp);" if ODD else "      constexpr int sh = 0;                    // records of this model always start 16-byte aligned")
    if DB:
      A("      double* sPb = s_P[p & 1];")
      A("      rn::async_wait();                        // group p has landed (issued one iteration ago)")
      A("      rn::wave_lds_sync();")
      A(f"      if (p + 1 < ngroups) {CPIN}<{PBUF}>(gP + (base + {FPW} * (p + 1)) * {EE}, ((cnt - {FPW} * (p + 1)) < {FPW} ? (cnt - {FPW} * (p + 1)) : {FPW}) * {EE}, s_P[(p + 1) & 1], lane);")
    else:
      A("      double* sPb = s_P[0];                   // single buffer: the co-resident wave hides the HBM latency")
      A(f"      {CPIN}<{PBUF}>(gPp, pcnt * {EE}, sPb, lane);")
      A("      rn::async_wait();")
      A("      rn::wave_lds_sync();")
    TL("4 + 4 * p")
    A("      double* sPc = sPb + sh;")
    A("      const int gg = g < pcnt ? g : 0;")
    A("      // a masked-out filter (active[i] == 0) is not `on`: every LDS store of the matrix phase is predicated, so its image")
    A("      // of P goes back to HBM as it came")
    A(f"      const bool on = act && g < pcnt && (active == nullptr || active[base + {FPW} * p + gg] != 0);")
    A(f"      double* sl = s_sl + ({FPW} * p + gg) * SLOT;")
    A(f"      if (do_pred) mat_predict(sPc + gg * {EE}, qcol, sl, cc, on);")
    TL("5 + 4 * p")
    if upd:
      ss_ = f", s_S + gg * {wide_s_doubles(k)}" if wide_s else ""
      A(f"      mat_update_{k.kind}(sPc + gg * {EE}, r_per_filter ? gR + (base + {FPW} * p + gg) * {ZZ} : gR, sl, sl, s_G + gg * {Z * E}, s_K + gg * {Z * E}{ss_}, cc, on);")
    TL("6 + 4 * p")
    A(f"      rn::copy_l2g_any<{PBUF}>(gPp, pcnt * {EE}, sPb, sh, lane);" if ODD else f"      rn::copy_l2g<{PBUF}>(gPp, pcnt * {EE}, sPb, lane);")
    if ckpt:      # (cP is 16-byte aligned like gP: the group's record starts on the same parity)
      A(f"      rn::copy_l2g_any<{PBUF}>(cP + (base + {FPW} * p) * {EE}, pcnt * {EE}, sPb, sh, lane);" if ODD else f"      rn::copy_l2g<{PBUF}>(cP + (base + {FPW} * p) * {EE}, pcnt * {EE}, sPb, lane);")
    TL("7 + 4 * p")
    A("      rn::wave_lds_sync();")
    A("    }")
    TL(3)
    A("    // ---------------- phase 3: lane l = filter l, inject the error state, write x / y / flags ---------")
    A("    if (lane < cnt && (active == nullptr || active[base + lane] != 0)) {")
    A("      const double* sl = s_sl + lane * SLOT;")
    if upd:
      A(f"      int fl = scal_inject(sl, s_x + lane * {D}, norm_quats);")
      A("#pragma unroll")
      A(f"      for (int i = 0; i < {Z}; i++) s_z[lane * {Z} + i] = sl[{lay.OFF_Y} + i];")
      A(f"      if (flags != nullptr) flags[base + lane] = (uint8_t)(fl | (int)sl[{lay.OFF_FL}]);     // 1 gated, 4 projection failed")
    else:
      A("#pragma unroll")
      A(f"      for (int i = 0; i < {D}; i++) s_x[lane * {D} + i] = sl[{lay.OFF_X} + i];")
    if upd:
      A("    } else if (lane < cnt && flags != nullptr) {")
      A("      flags[base + lane] = 16;       // masked out: x, P and z pass through untouched")
    A("    }")
    A("    rn::wave_lds_sync();")
    A(f"    rn::copy_l2g<FT2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);")
    if upd:
      A(f"    rn::copy_l2g<FT2 * {Z}>(gz + base * {Z}, cnt * {Z}, s_z, lane);")
    if ckpt:
      A(f"    rn::copy_l2g<FT2 * {D}>(cx + base * {D}, cnt * {D}, s_x, lane);")
    A("    rn::wave_lds_sync();")
    TL(63)
    if TLK:
      A("    if (lane == 0 && tile == blockIdx.x && blockIdx.x < 4096) g_tlb[blockIdx.x * 2 + 1] = wall_clock64();")
    A("  }")
    A("}")
    return "\n".join(L) + "\n"

  out.append(kernel("k_predict"))
  for k in spec.kinds:
    out.append(kernel(f"k_step_{k.kind}", k))
    out.append(kernel(f"k_stepc_{k.kind}", k, ckpt=True))
  return "\n".join(out)


def maha_kernels(spec):
  """Standalone Mahalanobis distance (reference: EKF_sym.maha_test, ekf_sym.py:626-649): d2 per filter, state untouched."""
  D, E = spec.dim_x, spec.dim_err
  EE = E * E
  FT = tile_filters(spec)
  FPW = filters_per_wave(spec)
  GL = group_lanes(spec)
  obs = {k.kind: _lowered_obs(spec, k) for k in spec.kinds}
  _, _, _, f_vars = _lowered_predict(spec)
  lay = Layout(spec, f_vars, {kk: v[3] for kk, v in obs.items()})
  out = []
  for k in spec.kinds:
    Z = k.zdim
    ZZ = Z * Z
    PBUF = (FPW * EE + 3) // 2 * 2
    _, _, He, he_vars = obs[k.kind]
    Hs = _slotted(He, he_vars, lay.OFF_HE)
    b = [f"double col[{E}], R[{ZZ}];", "#pragma unroll", f"for (int kq = 0; kq < {E}; kq++) col[kq] = sP[kq * {E} + cc];",
         "#pragma unroll", f"for (int i = 0; i < {ZZ}; i++) R[i] = gR[i];"]
    for zi in range(Z):
      b.append(f"const double G_{zi} = {sum_terms(term(cf, f'col[{kk}]') for kk, cf in Hs.row_nz(zi))};")
    b.append("if (act) { " + " ".join(f"sG[{zi} * {E} + cc] = G_{zi};" for zi in range(Z)) + " }")
    b.append("rn::wave_lds_sync();")
    b.append(f"double S[{ZZ}], L[{ZZ}], iL[{Z}], v[{Z}], w[{Z}];")
    for zi in range(Z):
      for w in range(Z):
        b.append(f"S[{zi * Z + w}] = {sum_terms(term(cf, f'sG[{zi} * {E} + {j}]') for j, cf in Hs.row_nz(w))} + R[{zi * Z + w}];")
    for i in range(Z):
      b.append(f"v[{i}] = w[{i}] = sl[{lay.OFF_Y + i}];")
    b += [f"rn::ldu_factor<{Z}>(S, L, iL);", f"rn::ldu_forward<{Z}>(L, iL, v);", f"rn::ldu_forward_t<{Z}>(L, iL, w);", "rn::wave_lds_sync();",
          "return " + " + ".join(f"v[{i}]*w[{i}]*iL[{i}]" for i in range(Z)) + ";"]
    out.append("\n".join([f"__device__ __forceinline__ double mat_maha_{k.kind}(const double* sP, const double* __restrict__ gR, const double* sl, "
                          "double* sG, const int cc, const bool act) {"] + _ind(b) + ["}"]))
    out.append(f"""
__global__ __launch_bounds__(64) void k_maha_{k.kind}(const double* __restrict__ gx, const double* __restrict__ gP,
    const double* __restrict__ gz, const double* __restrict__ gR, const int r_per_filter, const double* __restrict__ gea,
    const int64_t n, double* __restrict__ d2) {{
  (void)gea;
  __shared__ __attribute__((aligned(16))) double s_P[{PBUF}];
  __shared__ __attribute__((aligned(16))) double s_x[FT2 * {D} + 2];
  __shared__ __attribute__((aligned(16))) double s_z[FT2 * {Z} + 2];
  __shared__ __attribute__((aligned(16))) double s_G[{FPW} * {Z * E}];
  __shared__ __attribute__((aligned(16))) double s_sl[FT2 * SLOT];
  const int lane = threadIdx.x;
  const int g = lane / {GL};
  const int c = lane % {GL};
  const bool act = c < {E} && g < {FPW};
  const int cc = act ? c : 0;
  const int64_t tiles = (n + FT2 - 1) / FT2;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {{
    const int64_t base = tile * FT2;
    const int cnt = (n - base) < FT2 ? (int)(n - base) : FT2;
    rn::copy_g2l<FT2 * {D}>(gx + base * {D}, cnt * {D}, s_x, lane);
    rn::copy_g2l<FT2 * {Z}>(gz + base * {Z}, cnt * {Z}, s_z, lane);
    rn::wave_lds_sync();
    if (lane < cnt) {{
      double* sl = s_sl + lane * SLOT;
      scal_keep(s_x + lane * {D}, sl, 0);
      {_obs_call(k, 'false')};
    }}
    rn::wave_lds_sync();
    const int ngroups = (cnt + {FPW - 1}) / {FPW};
    for (int p = 0; p < ngroups; p++) {{
      const int pcnt = (cnt - {FPW} * p) < {FPW} ? (cnt - {FPW} * p) : {FPW};
      const double* gPp = gP + (base + {FPW} * p) * {EE};
      const int sh = {"rn::odd_start(gPp)" if (FPW * EE) % 2 or (FT * EE) % 2 else "0"};
      rn::async_copy_g2l_any<{PBUF}>(gPp, pcnt * {EE}, s_P, lane);
      rn::async_wait();
      rn::wave_lds_sync();
      const int gg = g < pcnt ? g : 0;
      const double d = mat_maha_{k.kind}(s_P + sh + gg * {EE}, r_per_filter ? gR + (base + {FPW} * p + gg) * {ZZ} : gR, s_sl + ({FPW} * p + gg) * SLOT,
                                  s_G + gg * {Z * E}, cc, act && g < pcnt);
      if (c == 0 && g < pcnt) d2[base + {FPW} * p + g] = d;
      rn::wave_lds_sync();
    }}
  }}
}}
""")
  return "\n".join(out)


def launch_maha(kind):
  return f"""  const int64_t tiles = (n + FT2 - 1) / FT2;
  hipLaunchKernelGGL(k_maha_{kind}, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, z, R, r_per_filter, ea, n, d2);"""


def launch_predict():
  return """  const int64_t tiles = (n + FT2 - 1) / FT2;
  hipLaunchKernelGGL(k_predict, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, Q, dt_vec, dt, n, norm_quats, active);"""


def launch_step(kind, do_predict):
  tf = "true" if do_predict else "false"
  if do_predict:
    args = "x, P, z, R, r_per_filter, ea, Q, dt_vec, dt, n, norm_quats, flags, active"
  else:
    args = "x, P, z, R, r_per_filter, ea, nullptr, nullptr, 0.0, n, norm_quats, flags, active"
  return f"""  const int64_t tiles = (n + FT2 - 1) / FT2;
  hipLaunchKernelGGL(k_step_{kind}<{tf}>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     {args});"""


def launch_step_ckpt(kind):
  return f"""  const int64_t tiles = (n + FT2 - 1) / FT2;
  hipLaunchKernelGGL(k_stepc_{kind}<true>, dim3(rn::grid_for_tiles(tiles)), dim3(64), 0, (hipStream_t)stream,
                     x, P, z, R, r_per_filter, ea, Q, dt_vec, dt, n, norm_quats, flags, active, ckpt_x, ckpt_P, ckpt_z);"""
