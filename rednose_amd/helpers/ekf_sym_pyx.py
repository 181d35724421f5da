"""`EKF_sym_pyx` under the module path the reference's models import it from (ekf_sym_pyx.pyx:85-196): the same orchestrator as
`EKF_sym`, with the Cython class's constructor signature (ekf_sym_pyx.pyx:87-90)."""
import logging

from rednose_amd.helpers.ekf_sym import EKF_sym


class EKF_sym_pyx(EKF_sym):  # pylint: disable=invalid-name
  def __init__(self, gen_dir, name, Q, x_initial, P_initial, dim_main, dim_main_err, N=0, dim_augment=0, dim_augment_err=0,  # pylint: disable=dangerous-default-value
               maha_test_kinds=[], quaternion_idxs=[], global_vars=[], max_rewind_age=1.0, logger=None):
    super().__init__(gen_dir, name, Q, x_initial, P_initial, dim_main, dim_main_err, N=N, dim_augment=dim_augment,
                     dim_augment_err=dim_augment_err, maha_test_kinds=maha_test_kinds, quaternion_idxs=quaternion_idxs,
                     global_vars=list(global_vars) or None, max_rewind_age=max_rewind_age, logger=logger or logging)
