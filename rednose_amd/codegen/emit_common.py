"""Pieces shared by the kernel emitters: structured (sparse-at-generation-time) matrices and
the per-routine device functions / host wrappers of the reference's scalar C ABI."""
import sympy as sp

from rednose_amd.codegen.lower import Block, vector_names


class SMat:
  """Matrix whose entries are known at generation time to be zero, one, a constant or a named double."""

  def __init__(self, rows, cols):
    self.rows, self.cols = rows, cols
    self.e = [[None] * cols for _ in range(rows)]    # None == structural zero

  @classmethod
  def from_structure(cls, rows, cols, structure, fmt):
    m = cls(rows, cols)
    for i in range(rows):
      for j in range(cols):
        kind, val = structure[fmt(i, j)]
        if kind == 'zero':
          continue
        if kind == 'one':
          m.e[i][j] = ('one', None)
        elif kind == 'const':
          m.e[i][j] = ('const', repr(float(val)))
        else:
          m.e[i][j] = ('var', fmt(i, j))
    return m

  @classmethod
  def dense(cls, rows, cols, fmt):
    m = cls(rows, cols)
    for i in range(rows):
      for j in range(cols):
        m.e[i][j] = ('var', fmt(i, j))
    return m

  @classmethod
  def identity_padded(cls, inner, size):
    """blockdiag(inner, I): the MSCKF block predict (ekf_c.c:23-26) as one structured matrix."""
    m = cls(size, size)
    for i in range(size):
      for j in range(size):
        if i < inner.rows and j < inner.cols:
          m.e[i][j] = inner.e[i][j]
        elif i == j:
          m.e[i][j] = ('one', None)
    return m

  def nnz(self):
    return sum(1 for r in self.e for c in r if c is not None)

  def row_nz(self, i):
    return [(j, self.e[i][j]) for j in range(self.cols) if self.e[i][j] is not None]

  def col_nz(self, j):
    return [(i, self.e[i][j]) for i in range(self.rows) if self.e[i][j] is not None]


def term(coef, operand):
  """C text of coef*operand (coef from an SMat entry, operand a C expression), None for zero."""
  if coef is None:
    return None
  kind, val = coef
  if kind == 'one':
    return operand
  return f"{val}*{operand}"


def sum_terms(terms):
  terms = [t for t in terms if t is not None]
  return " + ".join(terms) if terms else "0.0"


def coef_text(coef):
  if coef is None:
    return "0.0"
  kind, val = coef
  return "1.0" if kind == 'one' else val


def routine_device_function(routine):
  """One `__device__` function per sympy routine with the reference's argument order (inputs, then the
  flat row-major output).  Dense: every output entry is written, zeros included, like the reference's
  generated C."""
  names = {}
  params = []
  for idx, a in enumerate(routine.args):
    if a is None:
      params.append(("ptr", f"unused{idx}", 1))
      continue
    if isinstance(a, sp.MatrixSymbol):
      cname = str(a.name)
      names.update(vector_names(a, cname))
      params.append(("ptr", cname, int(a.shape[0] * a.shape[1])))
    elif isinstance(a, sp.Symbol):
      names[a] = str(a.name)
      params.append(("scalar", str(a.name), 1))
    else:  # a sympy Matrix of plain symbols (e.g. extra args)
      cname = f"ea{idx}"
      names.update(vector_names(a, cname))
      params.append(("ptr", cname, len(sp.Matrix(a))))
  expr = sp.Matrix(routine.expr)
  blk = Block(names, tmp_prefix="t")
  n_out = int(expr.shape[0] * expr.shape[1])
  for i in range(expr.shape[0]):
    for j in range(expr.shape[1]):
      blk.add(f"out[{i * expr.shape[1] + j}]", expr[i, j])
  stmts, _ = blk.lower(materialize_all=True, decl="")
  sig = ", ".join((f"const double* __restrict__ {n}" if k == "ptr" else f"double {n}") for k, n, _ in params)
  text = [f"__device__ __forceinline__ void {routine.name}({sig}, double* __restrict__ out) {{"]
  text += ["  " + s for s in stmts]
  text.append("}")
  return "\n".join(text), params, n_out


def innovation_solver(Z, general, y_terms, thresh=None):
  """Text pieces for the Z x Z innovation covariance S held in `S` (with `HPH`, `Rl` beside it): -> (factor, gate, solve) where
  `factor` factors S into (L, iL), `gate` (when `thresh` is given) is the Mahalanobis test of ekf_c.c:88-94 on the residual `y_terms`
  -- R *= 1e16 and a second factorisation when it trips --, and `solve(v)` is the call that overwrites the array v with S^-1 v.
  general=True: S is taken as a general matrix (L D U, templates/ekf_hip_rt.h: ldu_*) -- the step-granular kernels, which follow the
  reference on asymmetric covariances; False: L D L^T from the lower triangle (spd_*) -- the fused multi-step kernels, whose covariance
  is symmetric by contract."""
  pre = "rn::ldu" if general else "rn::spd"
  factor = f"{pre}_factor<{Z}>(S, L, iL);"
  gate = []
  if thresh is not None:
    ys = ", ".join(y_terms)
    if general:
      gate = ["{", f"  double v[{Z}] = {{{ys}}}, w[{Z}] = {{{ys}}};", f"  rn::ldu_forward<{Z}>(L, iL, v);", f"  rn::ldu_forward_t<{Z}>(L, iL, w);",
              "  const double d2 = " + " + ".join(f"v[{i}]*w[{i}]*iL[{i}]" for i in range(Z)) + ";"]
    else:
      gate = ["{", f"  double v[{Z}] = {{{ys}}};", f"  rn::spd_forward<{Z}>(L, iL, v);",
              "  const double d2 = " + " + ".join(f"v[{i}]*v[{i}]*iL[{i}]" for i in range(Z)) + ";"]
    gate += [f"  if (d2 > {thresh!r}) {{", "    gated = 1;", "#pragma unroll",
             f"    for (int i = 0; i < {Z * Z}; i++) {{ Rl[i] = 1.0e16 * Rl[i]; S[i] = HPH[i] + Rl[i]; }}", f"    {factor}", "  }", "}"]
  return factor, gate, (lambda v: f"{pre}_solve<{Z}>(L, iL, {v});")
