"""Expression lowering: sympy -> straight-line HIP device code (CSE + a device-friendly printer).

Replaces the reference's `sympy_into_c` (/root/reference/rednose/helpers/sympy_helpers.py:122-162),
which prints every matrix entry as an independent fully expanded C99 expression (no CSE; `x**2`
becomes `pow(x, 2)`; live.cpp contains 1 638 pow(), 252 sin(), 234 cos() calls -- SURVEY.md 3.5).
On a GPU that turns a memory-bound filter step into a transcendental-bound one, so here

  * all outputs of a fused block (e.g. f and F, or h and H.H_mod) go through ONE `sympy.cse` pass;
  * sin(a) and cos(a) of one argument come from ONE straight-line rn::sincos_fast(a, s, c) per distinct argument (hipcc does not merge the
    two ocml calls, ~170 instructions each behind a branch; live's IMU kinds made six of them per step);
  * small integer powers are printed as multiplications, reciprocals as `rn::safe_recip`, negative half-integer powers as odd
    powers of the reciprocal square root (`pow(r2, -1.5)` -> `rn::rsqrt_pow<3>(r2)`), positive ones through sqrt, everything else
    uses the C99 names which hipcc maps to the ocml double-precision device functions;
  * entries that are structurally 0 / 1 / numeric constants are reported as such so the kernel
    emitters can skip or fold them (sparsity is resolved at generation time, never at run time).

Rounding differs from the reference's expression order by a few ulp; tests bound it (tests/test_oracle.py::test_port_flavour_equals_ref_flavour on the CPU, tests/test_gpu_parity.py::test_scalar_sympy_routines_on_gpu on the device).
"""
import sympy as sp
from sympy.printing.c import C99CodePrinter


# Emitted once into every generated file that evaluates a sin or cos (emit.emit; libraries without trigonometric terms are not touched).
SINCOS_FAST = r"""namespace rn {
// sin and cos of one argument for the MODEL'S OWN expressions (rednose_amd/codegen/lower.py prints every sin(a) / cos(a) of a fused block
// through ONE call per distinct argument).  The ocml sin and cos are ~170 instructions each, mostly one dependent chain behind a branch
// (the large-argument reduction), and hipcc does not merge sin(a) with cos(a): live's gyro / accelerometer kinds called them six times
// for the three mounting angles -- ~500 of the ~630 instructions of the gyro kind's scalar phase, executed by the ONE lane per filter
// that evaluates the scalars.  Here: k = rint(a 2/pi); r = a - k pi/2 with pi/2 in three doubles and three FMAs -- the first difference
// is exact for every |a| >= 1 (both terms are multiples of 2^-52, the result is below 1), the second rounds at 5.5e-17 ABSOLUTE, the
// constants leave 6e-50 k --; minimax polynomials of degree 13 / 12 on [-pi/4, pi/4] (the classic fdlibm kernels, Horner, two
// independent chains); quadrant by select.  Straight-line and branch-free: the calls of a block are emitted next to each other, so the
// chains of different arguments interleave.  |error| < 2.5e-16 absolute for |a| <= 2^45 = 3.5e13 (tests/test_host_logic.py: 2e6 arguments
// against libm's long double routines; measured flat at 1.7e-16 up to 2^47, then the estimate of k is off by a visible fraction); beyond
// 2^45, where neighbouring doubles are 0.008 rad apart, and for NaN / inf the pair is NaN (libm returns the sine of the exact double there:
// a limit of this engine, README.md).  An inlined library fallback for that
// range was built and costs 25-35 registers in every kernel that evaluates a model with trigonometric terms (profiles/tuning_notes.md).
__device__ __forceinline__ void sincos_fast(const double a_in, double& s, double& c) {
#ifdef RN_EXACT_MATH      // tuning knob exact_math=1: the library's routines (full range; what the fast pair is measured against)
  s = sin(a_in);
  c = cos(a_in);
  return;
#endif
  const double a = (fabs(a_in) <= 35184372088832.0) ? a_in : __builtin_nan("");
  const double k = rint(a * 6.36619772367581382433e-01);
  double r = fma(-k, 1.5707963267948966e+00, a);
  r = fma(-k, 6.123233995736766e-17, r);
  r = fma(-k, -1.4973849048591698e-33, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double sr = fma(r * z, ps, r);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)fma(-4.0, floor(0.25 * k), k);      // k mod 4 (k is an integer below 2^45: exact)
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
}  // namespace rn
"""


# Emitted once into the generated file of a model with feature-track kinds (emit.emit).
NULLSPACE_RESIDUAL = r"""namespace rn {
// The residual of a feature-track observation in the REFERENCE'S basis of the left null space of the extra-argument Jacobian:
// /root/reference/rednose/templates/ekf_c.c:71-73 -- A = Hea^T.fullPivLu().kernel(); y = A^T y -- is what the reference writes back into z
// (:120) and returns in its Estimate (ekf_sym.cc:164-184).  x and P do not depend on the basis (the update runs in the orthonormal one the
// Householder reflectors give, templates/ekf_hip_rt.h), y does; so y alone is also formed here the way Eigen forms it: Gaussian
// elimination of M = Hea^T (A x Z) with FULL pivoting, P M Q = L U, U = [U1 U2], kernel vectors Q [-U1^-1 U2 ; I] (column c has its 1 at
// the (A + c)-th permuted position) -- the loops of oracle/ekf_oracle.c:fullpiv_kernel, Eigen's pivot choice on ties (its maxCoeff
// visitor walks the corner column by column and keeps the FIRST maximum) and Eigen's rank decision (every pivot against
// epsilon x min(A, Z) x the LARGEST pivot met, FullPivLU::threshold() / rank()).  The pivots' positions are run-time values; every array here is indexed STATICALLY and the exchanges are selects
// (a register array indexed at run time goes to scratch memory, and a first version that kept U in the filter's LDS slot spent 3 us per
// tile in ~20 dependent LDS round trips on the one lane per filter that runs this: feature36 launch 83.5 -> 89.8 us; this form: see
// profiles/tuning_notes.md).  One division per pivot.  Returns false (yout = 0) when Hea^T has rank < A by Eigen's threshold.
template <int Z, int A>
__device__ __forceinline__ bool nullspace_residual(const double (&Hea)[Z * A], const double (&y)[Z], double (&yout)[Z - A]) {
  double U[A * Z];
  int pm[Z];
#pragma unroll
  for (int i = 0; i < A; i++) {
#pragma unroll
    for (int j = 0; j < Z; j++) U[i * Z + j] = Hea[j * A + i];
  }
#pragma unroll
  for (int j = 0; j < Z; j++) pm[j] = j;
  double maxpiv = 0.0;
  double ipv[A], piv[A];
#pragma unroll
  for (int k = 0; k < A; k++) {
    int pr = k, pc = k;
    double best = 0.0;
#pragma unroll
    for (int j = k; j < Z; j++) {          // column by column, first maximum: Eigen's visitor order (a tie between equal entries picks the same one)
#pragma unroll
      for (int i = k; i < A; i++) {
        const double a = fabs(U[i * Z + j]);
        const bool gt = a > best;
        best = gt ? a : best; pr = gt ? i : pr; pc = gt ? j : pc;
      }
    }
    piv[k] = best;
    maxpiv = best > maxpiv ? best : maxpiv;          // the largest pivot so far (later pivots of a Schur complement can exceed the first)
    // rows k <-> pr (columns left of k hold nothing that is read again)
#pragma unroll
    for (int j = k; j < Z; j++) {
      const double rk = U[k * Z + j];
      double rp = rk;
#pragma unroll
      for (int i = k + 1; i < A; i++) { rp = (pr == i) ? U[i * Z + j] : rp; U[i * Z + j] = (pr == i) ? rk : U[i * Z + j]; }
      U[k * Z + j] = rp;
    }
    // columns k <-> pc, and the permutation
#pragma unroll
    for (int i = 0; i < A; i++) {
      const double ck = U[i * Z + k];
      double cp = ck;
#pragma unroll
      for (int j = k + 1; j < Z; j++) { cp = (pc == j) ? U[i * Z + j] : cp; U[i * Z + j] = (pc == j) ? ck : U[i * Z + j]; }
      U[i * Z + k] = cp;
    }
    {
      const int pk = pm[k];
      int pp = pk;
#pragma unroll
      for (int j = k + 1; j < Z; j++) { pp = (pc == j) ? pm[j] : pp; pm[j] = (pc == j) ? pk : pm[j]; }
      pm[k] = pp;
    }
    ipv[k] = 1.0 / U[k * Z + k];
#pragma unroll
    for (int i = k + 1; i < A; i++) {
      const double l = U[i * Z + k] * ipv[k];
#pragma unroll
      for (int j = k; j < Z; j++) U[i * Z + j] -= l * U[k * Z + j];
    }
  }
  bool full = true;      // rank == A: every pivot above epsilon x diagonalSize x largest pivot (an exact zero pivot fails it too)
#pragma unroll
  for (int k = 0; k < A; k++) full = full && (piv[k] > 2.220446049250313e-16 * (A < Z ? A : Z) * maxpiv);
  double yp[Z];      // y in pivot order
#pragma unroll
  for (int j = 0; j < Z; j++) {
    double v = y[0];
#pragma unroll
    for (int m = 1; m < Z; m++) v = (pm[j] == m) ? y[m] : v;
    yp[j] = v;
  }
#pragma unroll
  for (int c = 0; c < Z - A; c++) {
    double v[A];
#pragma unroll
    for (int i = A - 1; i >= 0; i--) {            // U1 v = -U2[:, c]
      double acc = -U[i * Z + A + c];
#pragma unroll
      for (int p = i + 1; p < A; p++) acc -= U[i * Z + p] * v[p];
      v[i] = acc * ipv[i];
    }
    double r = yp[A + c];
#pragma unroll
    for (int i = 0; i < A; i++) r += v[i] * yp[i];
    yout[c] = full ? r : 0.0;
  }
  return full;
}
}  // namespace rn
"""


class HipPrinter(C99CodePrinter):
  """C99 printer with power strength-reduction suitable for fp64 device code."""

  def __init__(self, symbol_names=None, trig=None):
    super().__init__(dict(precision=17, contract=False))
    self._names = symbol_names or {}
    self._trig = trig or {}      # argument -> name of the (name_s, name_c) pair Block.lower() computes once through rn::sincos_fast

  def _print_sin(self, expr):
    return self._trig[expr.args[0]] + "_s"

  def _print_cos(self, expr):
    return self._trig[expr.args[0]] + "_c"

  def _print_Symbol(self, expr):
    return self._names.get(expr, super()._print_Symbol(expr))

  def _print_MatrixElement(self, expr):
    key = (expr.parent, int(expr.i), int(expr.j))
    if key in self._names:
      return self._names[key]
    rows, cols = expr.parent.shape
    return f"{self._print(expr.parent)}[{int(expr.i) * int(cols) + int(expr.j)}]"

  def _mul_chain(self, base, n):
    b = self.parenthesize(base, 1000)
    if n == 1:
      return b
    return "(" + "*".join([b] * n) + ")"

  def _print_Pow(self, expr):
    base, exp = expr.base, expr.exp
    if exp.is_Integer:
      n = int(exp)
      if 1 <= n <= 4:
        return self._mul_chain(base, n)
      if -4 <= n <= -1:
        return f"rn::safe_recip({self._mul_chain(base, -n)})"      # v_rcp_f64 + two Newton steps, IEEE's answer kept for 0 / inf (templates/ekf_hip_rt.h): 6 dependent instructions, an IEEE division ~12
    if exp.is_Rational or exp.is_Float:
      two = sp.nsimplify(2 * exp)
      if two.is_Integer and abs(int(two)) <= 9 and int(two) % 2:
        k = (abs(int(two)) - 1) // 2          # |exp| = k + 1/2
        if two < 0:      # a^-(k + 1/2): odd power of the reciprocal square root (rn::rsqrt_pow), no IEEE sqrt / division chain
          return f"rn::rsqrt_pow<{2 * k + 1}>({self._print(base)})"
        root = f"sqrt({self._print(base)})"
        return root if k == 0 else f"({self._mul_chain(base, k)}*{root})"
    return super()._print_Pow(expr)

  def _print_Rational(self, expr):
    return f"({int(expr.p)}.0/{int(expr.q)}.0)"

  def _print_Integer(self, expr):
    # keep arithmetic in double (sympy would print bare ints, e.g. `1` for F's unit diagonal)
    return f"{int(expr)}.0" if int(expr) >= 0 else f"(-{-int(expr)}.0)"


def classify(expr):
  """-> ('zero'|'one'|'const'|'expr', value)."""
  e = sp.sympify(expr)
  if e.is_zero:
    return 'zero', 0.0
  if e.is_number:
    v = float(e)
    if v == 0.0:
      return 'zero', 0.0
    if v == 1.0:
      return 'one', 1.0
    return 'const', v
  return 'expr', e


class Block:
  """A fused group of symbolic outputs lowered together.

  outputs: list of (c_lvalue, sympy expr).  After `lower()`, `.statements` holds C statements
  (temporaries first) and `.structure[c_lvalue]` the ('zero'|'one'|'const'|'expr') classification.
  Structural zeros/ones/constants are NOT assigned by the statements unless `materialize_all`.
  """

  def __init__(self, names=None, tmp_prefix="t"):
    self.outputs = []
    self.names = dict(names or {})
    self.tmp_prefix = tmp_prefix

  def add(self, lvalue, expr):
    self.outputs.append((lvalue, sp.sympify(expr)))

  def lower(self, materialize_all=False, decl="const double "):
    structure = {}
    live = []
    for lv, e in self.outputs:
      kind, val = classify(e)
      structure[lv] = (kind, val)
      if kind == 'expr' or materialize_all:
        live.append((lv, e))
    exprs = [e for _, e in live]
    stmts = []
    if exprs:
      repl, reduced = sp.cse(exprs, symbols=sp.numbered_symbols(self.tmp_prefix), optimizations='basic', order='none')
      # sin / cos: ONE rn::sincos_fast per distinct argument (SINCOS_FAST above says why), placed where its argument becomes
      # available (the calls that land at the same place are emitted next to each other: straight-line code, their chains interleave)
      trig, where = {}, {}
      defined = {sym: i for i, (sym, _) in enumerate(repl)}
      for e in [sub for _, sub in repl] + list(reduced):
        # post-order: the argument of an outer call may itself contain a sin / cos that CSE left in place (sin(x + cos(y))); the
        # inner pair is then registered -- and, landing at the same place, emitted -- BEFORE the call that reads it.  (Encounter
        # order otherwise: the names must not depend on a set's iteration order.)
        for node in sp.postorder_traversal(e):
          if isinstance(node, (sp.sin, sp.cos)) and node.args[0] not in trig:
            arg = node.args[0]
            trig[arg] = f"{self.tmp_prefix}sc{len(trig)}"
            inner = [where[t.args[0]] for t in arg.atoms(sp.sin, sp.cos)]
            where[arg] = max([defined[q] + 1 for q in arg.free_symbols if q in defined] + inner + [0])
      pr = HipPrinter(self.names, trig)

      def trig_group(i):
        return [f"double {trig[a]}_s, {trig[a]}_c; rn::sincos_fast({pr.doprint(a)}, {trig[a]}_s, {trig[a]}_c);" for a in trig if where[a] == i]
      for i, (sym, sub) in enumerate(repl):
        stmts += trig_group(i)
        stmts.append(f"const double {sym} = {pr.doprint(sub)};")
      stmts += trig_group(len(repl))
      for (lv, _), red in zip(live, reduced):
        stmts.append(f"{decl}{lv} = {pr.doprint(red)};")
    self.statements = stmts
    self.structure = structure
    return stmts, structure


def matrix_entries(mat, fmt):
  """[(fmt(i,j), mat[i,j])] row-major."""
  m = sp.Matrix(mat)
  return [(fmt(i, j), m[i, j]) for i in range(m.shape[0]) for j in range(m.shape[1])]


def vector_names(sym, cname):
  """Map MatrixSymbol elements of a column vector `sym` to `cname[i]`."""
  if sym is None:
    return {}
  if isinstance(sym, sp.MatrixSymbol):
    return {(sym, i, 0): f"{cname}[{i}]" for i in range(sym.shape[0])}
  if isinstance(sym, sp.Symbol):
    return {sym: cname}
  out = {}
  for i, s in enumerate(sp.Matrix(sym)):
    if isinstance(s, sp.Symbol):
      out[s] = f"{cname}[{i}]"
  return out
