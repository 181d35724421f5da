"""Model front end: sympy definitions -> `FilterSpec` (everything the HIP emitter needs).

Behavioural parity with the FRONT half of the reference's gen_code
(/root/reference/rednose/helpers/ekf_sym.py:36-116): ESKF parameter unpacking or identity defaults
(:36-53), MSCKF dimensions (:57-73), `F = d f_err / d x_err` with the error symbols substituted by
zero for an ESKF (:76-80), the `dt in F` assertion (:82), per-kind `H = dh/dx` and `He = dh/dea`
(:84-89) and the routine list err_fun, inv_err_fun, H_mod_fun, f_fun, F_fun, h_k, H_k, [He_k],
extra routines (:91-113).  What differs is what happens next: the reference prints every routine to
C99 with sympy's codegen (no CSE); here the routines stay symbolic so rednose_amd/codegen can apply
CSE, exploit sparsity and fuse them into HIP kernels.
"""
from dataclasses import dataclass, field
from typing import Any, Optional

import sympy as sp

from rednose_amd.helpers.chi2_lookup import chi2_ppf


@dataclass
class Routine:
  """One symbolic function: flat row-major outputs of `expr`, positional `args` (None = unused slot)."""
  name: str
  expr: sp.Matrix
  args: list


@dataclass
class ObsKind:
  kind: int
  zdim: int
  h_sym: sp.Matrix
  H_sym: sp.Matrix
  ea_sym: Any
  He_sym: Optional[sp.Matrix]
  maha_test: bool
  maha_thresh: float


@dataclass
class FilterSpec:
  name: str
  dim_x: int
  dim_err: int
  dim_main: int
  dim_main_err: int
  dim_augment: int
  dim_augment_err: int
  N: int
  x_sym: Any
  dt_sym: Any
  f_sym: sp.Matrix
  F_sym: sp.Matrix
  H_mod_sym: sp.Matrix
  err_eqs: list
  inv_err_eqs: list
  is_eskf: bool
  kinds: list = field(default_factory=list)           # list[ObsKind]
  feature_track_kinds: list = field(default_factory=list)
  quaternion_idxs: list = field(default_factory=list)
  global_vars: list = field(default_factory=list)
  extra_routines: list = field(default_factory=list)

  def routines(self):
    """The routine list in the reference's order (ekf_sym.py:91-113)."""
    out = [Routine(*r) for r in self.extra_routines]
    out.append(Routine('err_fun', self.err_eqs[0], [self.err_eqs[1], self.err_eqs[2]]))
    out.append(Routine('inv_err_fun', self.inv_err_eqs[0], [self.inv_err_eqs[1], self.inv_err_eqs[2]]))
    out.append(Routine('H_mod_fun', self.H_mod_sym, [self.x_sym]))
    out.append(Routine('f_fun', self.f_sym, [self.x_sym, self.dt_sym]))
    out.append(Routine('F_fun', self.F_sym, [self.x_sym, self.dt_sym]))
    for k in self.kinds:
      out.append(Routine(f'h_{k.kind}', k.h_sym, [self.x_sym, k.ea_sym]))
      out.append(Routine(f'H_{k.kind}', k.H_sym, [self.x_sym, k.ea_sym]))
      if k.He_sym is not None:
        out.append(Routine(f'He_{k.kind}', k.He_sym, [self.x_sym, k.ea_sym]))
    return out

  def identity_at_dt0(self):
    """True when predict(dt = 0) is the identity on (x, P) for every finite state and finite Q: f(x, 0) == x and
    F(x, 0)[:M, :M] == I symbolically.  The reference makes no such assumption -- it evaluates f and F on every call,
    including the first one of a filter, whose dt is always 0 (ekf_c.c:15-28, ekf_sym.cc:198-206) -- so the kernel
    emitters may skip the covariance phase of a dt == 0 predict ONLY when this holds (e.g. not for f = A x + dt g(x)
    with A != I, a clamp, or a reset term)."""
    if getattr(self, "_id0", None) is None:
      x = sp.Matrix(self.x_sym)
      f0 = sp.Matrix(self.f_sym).subs(self.dt_sym, 0)
      M = self.dim_main_err
      F0 = sp.Matrix(self.F_sym)[:M, :M].subs(self.dt_sym, 0)
      ok = all(sp.simplify(f0[i] - x[i]) == 0 for i in range(self.dim_x))
      ok = ok and all(sp.simplify(F0[i, j] - (1 if i == j else 0)) == 0 for i in range(M) for j in range(M))
      object.__setattr__(self, "_id0", bool(ok))
    return self._id0

  def kind(self, kind):
    for k in self.kinds:
      if k.kind == kind:
        return k
    raise KeyError(kind)


def build_spec(name, f_sym, dt_sym, x_sym, obs_eqs, dim_x, dim_err, eskf_params=None, msckf_params=None,
               maha_test_kinds=(), quaternion_idxs=(), global_vars=None, extra_routines=()):
  if eskf_params:
    err_eqs, inv_err_eqs, H_mod_sym, f_err_sym, x_err_sym = eskf_params[:5]
  else:
    nom_x = sp.MatrixSymbol('nom_x', dim_x, 1)
    true_x = sp.MatrixSymbol('true_x', dim_x, 1)
    delta_x = sp.MatrixSymbol('delta_x', dim_x, 1)
    err_eqs = [sp.Matrix(nom_x + delta_x), nom_x, delta_x]
    inv_err_eqs = [sp.Matrix(true_x - nom_x), nom_x, true_x]
    H_mod_sym = sp.eye(dim_x)
    f_err_sym = f_sym
    x_err_sym = x_sym

  if msckf_params:
    dim_main, dim_augment, dim_main_err, dim_augment_err, N, feature_track_kinds = msckf_params[:6]
    if dim_main + dim_augment * N != dim_x or dim_main_err + dim_augment_err * N != dim_err:
      raise AssertionError("msckf_params dimensions do not add up to dim_x / dim_err")
  else:
    dim_main, dim_augment, dim_main_err, dim_augment_err, N, feature_track_kinds = dim_x, 0, dim_err, 0, 0, []

  F_sym = sp.Matrix(f_err_sym).jacobian(sp.Matrix(x_err_sym))
  if eskf_params:
    F_sym = F_sym.subs({s: 0 for s in sp.Matrix(x_err_sym)})
  if dt_sym not in F_sym.free_symbols:
    raise AssertionError("the transition Jacobian does not depend on dt")

  x_vec = sp.Matrix(x_sym)
  kinds = []
  for eq in obs_eqs:
    h_sym, kind, ea_sym = eq[0], eq[1], eq[2]
    h_sym = sp.Matrix(h_sym)
    H_sym = h_sym.jacobian(x_vec)
    He_sym = None
    if msckf_params and kind in feature_track_kinds:
      He_sym = h_sym.jacobian(sp.Matrix(ea_sym))
    zdim = int(h_sym.shape[0])
    kinds.append(ObsKind(kind=int(kind), zdim=zdim, h_sym=h_sym, H_sym=H_sym, ea_sym=ea_sym, He_sym=He_sym,
                         maha_test=kind in list(maha_test_kinds),
                         maha_thresh=float(chi2_ppf(0.95, zdim))))

  return FilterSpec(name=name, dim_x=int(dim_x), dim_err=int(dim_err), dim_main=int(dim_main),
                    dim_main_err=int(dim_main_err), dim_augment=int(dim_augment),
                    dim_augment_err=int(dim_augment_err), N=int(N), x_sym=x_sym, dt_sym=dt_sym,
                    f_sym=sp.Matrix(f_sym), F_sym=F_sym, H_mod_sym=sp.Matrix(H_mod_sym),
                    err_eqs=list(err_eqs), inv_err_eqs=list(inv_err_eqs), is_eskf=bool(eskf_params),
                    kinds=kinds, feature_track_kinds=list(feature_track_kinds),
                    quaternion_idxs=list(quaternion_idxs),
                    global_vars=list(global_vars) if global_vars is not None else [],
                    extra_routines=list(extra_routines))
