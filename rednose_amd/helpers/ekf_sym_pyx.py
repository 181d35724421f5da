"""`EKF_sym_pyx` under the module path the reference's models import it from: a COMPILED binding of the C++ orchestrator, like the
reference's (/root/reference/rednose/helpers/ekf_sym_pyx.pyx:85-195 is a Cython class over the C++ `EKFSym`, ekf_sym.cc).

Here the C++ orchestrator is `rednose_amd::EKFSymBatch` (include/rednose_amd/ekf_sym_batch.hpp: N filters resident on the GPU, checkpoint
ring, n observations per call) and the binding is the pybind11 module `rednose_amd/helpers/_ekf_sym_batch*.so` built in-tree by
`__graft_entry__.build()` from rednose_amd/csrc/ekf_sym_batch_py.cpp.  This file is the thin Python face with the Cython class's
constructor signature (ekf_sym_pyx.pyx:87-90) and return shapes: with the default batch of ONE filter, state() is (D,), covs() (E, E) and
the Estimate 9-tuple holds 1-D / 2-D arrays and a list of (Z,) residuals, exactly like the reference's; `batch=N` (an extension) keeps the
leading filter axis.  The methods the reference's Cython class leaves unimplemented (augment, get_augment_times, rts_smooth, maha_test:
ekf_sym_pyx.pyx:181-192) raise NotImplementedError here too -- `rednose_amd.helpers.ekf_sym.EKF_sym` / `BatchedEKF` have them.
No compiled module, no class: constructing one raises ImportError (there is no Python fallback behind this name).
"""
import numpy as np


def _module():
  try:
    from rednose_amd.helpers import _ekf_sym_batch      # pylint: disable=import-outside-toplevel
  except ImportError as e:
    raise ImportError("rednose_amd.helpers._ekf_sym_batch (the compiled binding of EKFSymBatch) is not built: run __graft_entry__.build() "
                      f"or rednose_amd.build.build_python_binding() ({e})") from e
  return _ekf_sym_batch


class EKF_sym_pyx:  # pylint: disable=invalid-name
  def __init__(self, gen_dir, name, Q, x_initial, P_initial, dim_main, dim_main_err, N=0, dim_augment=0, dim_augment_err=0,  # pylint: disable=dangerous-default-value
               maha_test_kinds=[], quaternion_idxs=[], global_vars=[], max_rewind_age=1.0, logger=None, batch=1, rewind_to_keep=512):
    del dim_main, dim_main_err, N, dim_augment, dim_augment_err, maha_test_kinds, global_vars, logger      # (properties of the generated library)
    self.batch = int(batch)
    self._ekf = _module().EKFSymBatch(str(gen_dir), str(name), np.ascontiguousarray(Q, dtype=np.float64), np.ascontiguousarray(x_initial, dtype=np.float64).reshape(-1),
                                      np.ascontiguousarray(P_initial, dtype=np.float64), self.batch, len(quaternion_idxs) > 0, int(rewind_to_keep),
                                      float(max_rewind_age))

  def _out(self, a):
    return a[0] if self.batch == 1 else a

  def init_state(self, state, covs, filter_time):
    self._ekf.init_state(np.ascontiguousarray(state, dtype=np.float64), np.ascontiguousarray(covs, dtype=np.float64), filter_time)

  def state(self):
    return self._out(self._ekf.state())

  def covs(self):
    return self._out(self._ekf.covs())

  def set_filter_time(self, t):
    self._ekf.set_filter_time(float(t))

  def get_filter_time(self):
    return self._ekf.get_filter_time()

  def set_global(self, global_var, val):
    self._ekf.set_global(str(global_var), float(val))

  def reset_rewind(self):
    self._ekf.reset_rewind()

  def predict(self, t):
    self._ekf.predict(float(t))

  def predict_and_update_batch(self, t, kind, z, R, extra_args=[[]], augment=False, estimate=True):  # pylint: disable=dangerous-default-value
    """z: n observations (Z,) [or (N, Z) per filter, or device pointers / tensors (N, Z)], R: n matrices (Z, Z), extra_args: n lists
    (ekf_sym_pyx.pyx:143-166).  Returns the Estimate 9-tuple (xk_km1, xk_k, Pk_km1, Pk_k, t, kind, y, z, extra_args) or None for an
    observation that is too old (:168-169); estimate=False (an extension) returns True instead of the tuple and skips the host copies."""
    res = self._ekf.predict_and_update_batch(float(t), int(kind), list(z), [np.ascontiguousarray(r_, dtype=np.float64) for r_ in R],
                                             [list(np.ravel(e)) for e in extra_args], bool(augment), bool(estimate))
    if res is None or res is True:
      return res
    xk1, xk, Pk1, Pk, t_, kind_, ys, _, _ = res
    return self._out(xk1), self._out(xk), self._out(Pk1), self._out(Pk), t_, kind_, [self._out(y) for y in ys], z, extra_args

  def augment(self):
    raise NotImplementedError()      # as the reference's Cython class (ekf_sym_pyx.pyx:181-182); BatchedEKF.augment has it

  def get_augment_times(self):
    raise NotImplementedError()

  def rts_smooth(self, estimates, norm_quats=False):
    raise NotImplementedError()      # BatchedEKF.rts_smooth / EKF_sym.rts_smooth

  def maha_test(self, x, P, kind, z, R, extra_args=[], maha_thresh=0.95):  # pylint: disable=dangerous-default-value
    raise NotImplementedError()      # BatchedEKF.maha_test / EKF_sym.maha_test
