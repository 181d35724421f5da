#!/usr/bin/env python3
"""Error-state (quaternion) localisation filter: 23 nominal states, 22 error states, 8 observation kinds.

Model parity: /root/reference/examples/live_kf.py -- state layout :73-91, x0 :97-104, P0 :107-114,
Q :117-124, nominal dynamics :154-168, error dynamics :170-184, H_mod :187-190, error injection and
its inverse :196-211, observation equations :219-244, observation noise :252-258.  The symbolic model
is re-stated here (not imported) and checked entry-by-entry against the reference's generated C by
tests/test_oracle.py::test_live_model_matches_reference.

Differences by design:
  * `generate_code(dir, maha_test_kinds=[...])` forwards a Mahalanobis-gated kind list to gen_code
    (the reference generates none); config 4 uses maha_test_kinds=[ECEF_POS].
  * the quaternion slice is renormalised inside the kernels (quaternion_idxs=[3]) after every predict
    and update, as EKFSym does when given the index (/root/reference/rednose/helpers/ekf_sym.cc:207,213).
"""
import os
import sys

if __name__ == "__main__":  # allow running as a script from anywhere (generator CLI contract)
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import sympy as sp

from rednose_amd.helpers import KalmanError
from rednose_amd.helpers.kalmanfilter import KalmanFilter
from rednose_amd.helpers.ekf_sym import EKF_sym, BatchedEKF, gen_code
from rednose_amd.helpers.sympy_helpers import euler_rotate, quat_matrix_r, quat_rotate

EARTH_GM = 3.986005e14  # m^3/s^2


class ObservationKind:
  """Kind numbers are part of the generated symbol names ({name}_update_{kind})."""
  UNKNOWN = 0
  NO_OBSERVATION = 1
  GPS_NED = 2
  ODOMETRIC_SPEED = 3
  PHONE_GYRO = 4
  GPS_VEL = 5
  PSEUDORANGE_GPS = 6
  PSEUDORANGE_RATE_GPS = 7
  SPEED = 8
  NO_ROT = 9
  PHONE_ACCEL = 10
  ORB_POINT = 11
  ECEF_POS = 12
  CAMERA_ODO_TRANSLATION = 13
  CAMERA_ODO_ROTATION = 14
  ORB_FEATURES = 15
  MSCKF_TEST = 16
  FEATURE_TRACK_TEST = 17
  LANE_PT = 18
  IMU_FRAME = 19
  PSEUDORANGE_GLONASS = 20
  PSEUDORANGE_RATE_GLONASS = 21
  PSEUDORANGE = 22
  PSEUDORANGE_RATE = 23

  @classmethod
  def to_string(cls, kind):
    for key, val in vars(cls).items():
      if key.isupper() and val == kind:
        return key.replace('_', ' ').lower()
    return 'unknown'


class States:
  # nominal state
  ECEF_POS = slice(0, 3)            # m
  ECEF_ORIENTATION = slice(3, 7)    # quaternion device -> ecef
  ECEF_VELOCITY = slice(7, 10)      # m/s
  ANGULAR_VELOCITY = slice(10, 13)  # rad/s, device frame
  GYRO_BIAS = slice(13, 16)
  ODO_SCALE = slice(16, 17)
  ACCELERATION = slice(17, 20)      # m/s^2, device frame
  IMU_OFFSET = slice(20, 23)        # rad

  # error state (orientation error is 3 small angles)
  ECEF_POS_ERR = slice(0, 3)
  ECEF_ORIENTATION_ERR = slice(3, 6)
  ECEF_VELOCITY_ERR = slice(6, 9)
  ANGULAR_VELOCITY_ERR = slice(9, 12)
  GYRO_BIAS_ERR = slice(12, 15)
  ODO_SCALE_ERR = slice(15, 16)
  ACCELERATION_ERR = slice(16, 19)
  IMU_OFFSET_ERR = slice(19, 22)


def _sq(vals):
  return np.asarray(vals, dtype=np.float64) ** 2


class LiveKalman(KalmanFilter):
  name = 'live'

  initial_x = np.array([-2.7e6, 4.2e6, 3.8e6] + [1, 0, 0, 0] + [0] * 9 + [1] + [0] * 6, dtype=np.float64)

  initial_P_diag = _sq([10000] * 3 + [10] * 3 + [10] * 3 + [1] * 3 + [0.05] * 3 + [0.02] + [1] * 3 + [0.01] * 3)

  Q = np.diag(_sq([0.03] * 3 + [0.0] * 3 + [0.0] * 3 + [0.1] * 3 + [0.005 / 100] * 3 + [0.02 / 100] + [3] * 3 + [0.05 / 60] * 3))

  obs_noise = {ObservationKind.ODOMETRIC_SPEED: np.atleast_2d(0.2**2),
               ObservationKind.PHONE_GYRO: np.diag(_sq([0.025] * 3)),
               ObservationKind.PHONE_ACCEL: np.diag(_sq([.5] * 3)),
               ObservationKind.CAMERA_ODO_ROTATION: np.diag(_sq([0.05] * 3)),
               ObservationKind.IMU_FRAME: np.diag(_sq([0.05] * 3)),
               ObservationKind.NO_ROT: np.diag(_sq([0.00025] * 3)),
               ObservationKind.ECEF_POS: np.diag(_sq([5] * 3))}

  quaternion_idxs = [States.ECEF_ORIENTATION.start]

  @classmethod
  def model(cls):
    D = cls.initial_x.shape[0]
    E = cls.initial_P_diag.shape[0]
    S = States

    state_sym = sp.MatrixSymbol('state', D, 1)
    state = sp.Matrix(state_sym)
    pos = state[S.ECEF_POS, :]
    quat = state[S.ECEF_ORIENTATION, :]
    vel = state[S.ECEF_VELOCITY, :]
    omega = state[S.ANGULAR_VELOCITY, :]
    gyro_bias = state[S.GYRO_BIAS, :]
    odo_scale = state[S.ODO_SCALE.start, 0]
    accel = state[S.ACCELERATION, :]
    imu_angles = state[S.IMU_OFFSET, :]
    dt = sp.Symbol('dt')

    device_to_ecef = quat_rotate(*quat)

    # --- nominal dynamics: first-order integration of [v, q_dot, R a]
    wr, wp, wy = omega
    half_omega = 0.5 * sp.Matrix([[0, -wr, -wp, -wy],
                                  [wr, 0, wy, -wp],
                                  [wp, -wy, 0, wr],
                                  [wy, wp, -wr, 0]])
    rate = sp.Matrix(np.zeros((D, 1)))
    rate[S.ECEF_POS, :] = vel
    rate[S.ECEF_ORIENTATION, :] = half_omega * quat
    rate[S.ECEF_VELOCITY, 0] = device_to_ecef * accel
    f_sym = state + dt * rate

    # --- error-state dynamics
    err_sym = sp.MatrixSymbol('state_err', E, 1)
    err = sp.Matrix(err_sym)
    att_err = err[S.ECEF_ORIENTATION_ERR, :]
    small_rot = euler_rotate(att_err[0], att_err[1], att_err[2])
    err_rate = sp.Matrix(np.zeros((E, 1)))
    err_rate[S.ECEF_POS_ERR, :] = err[S.ECEF_VELOCITY_ERR, :]
    err_rate[S.ECEF_ORIENTATION_ERR, :] = small_rot * device_to_ecef * (omega + err[S.ANGULAR_VELOCITY_ERR, :])
    err_rate[S.ECEF_VELOCITY_ERR, :] = small_rot * device_to_ecef * (accel + err[S.ACCELERATION_ERR, :])
    f_err_sym = err + dt * err_rate

    # --- d(nominal)/d(error) used to map observation Jacobians into error space
    H_mod_sym = sp.Matrix(np.zeros((D, E)))
    H_mod_sym[S.ECEF_POS, S.ECEF_POS_ERR] = np.eye(3)
    H_mod_sym[S.ECEF_ORIENTATION, S.ECEF_ORIENTATION_ERR] = 0.5 * quat_matrix_r(state[S.ECEF_ORIENTATION.start:S.ECEF_ORIENTATION.stop])[:, 1:]
    H_mod_sym[S.ECEF_ORIENTATION.stop:, S.ECEF_ORIENTATION_ERR.stop:] = np.eye(D - S.ECEF_ORIENTATION.stop)

    # --- true = err_function(nominal, delta), delta = inv_err_function(nominal, true)
    nom_x = sp.MatrixSymbol('nom_x', D, 1)
    true_x = sp.MatrixSymbol('true_x', D, 1)
    delta_x = sp.MatrixSymbol('delta_x', E, 1)

    inject = sp.Matrix(np.zeros((D, 1)))
    dq = sp.Matrix(np.ones(4))
    dq[1:, :] = sp.Matrix(0.5 * delta_x[S.ECEF_ORIENTATION_ERR, :])
    inject[S.ECEF_POS, :] = sp.Matrix(nom_x[S.ECEF_POS, :] + delta_x[S.ECEF_POS_ERR, :])
    inject[S.ECEF_ORIENTATION, 0] = quat_matrix_r(nom_x[S.ECEF_ORIENTATION, 0]) * dq
    inject[S.ECEF_ORIENTATION.stop:, :] = sp.Matrix(nom_x[S.ECEF_ORIENTATION.stop:, :] + delta_x[S.ECEF_ORIENTATION_ERR.stop:, :])

    extract = sp.Matrix(np.zeros((E, 1)))
    extract[S.ECEF_POS_ERR, 0] = sp.Matrix(-nom_x[S.ECEF_POS, 0] + true_x[S.ECEF_POS, 0])
    dq_back = quat_matrix_r(nom_x[S.ECEF_ORIENTATION, 0]).T * true_x[S.ECEF_ORIENTATION, 0]
    extract[S.ECEF_ORIENTATION_ERR, 0] = sp.Matrix(2 * dq_back[1:])
    extract[S.ECEF_ORIENTATION_ERR.stop:, 0] = sp.Matrix(-nom_x[S.ECEF_ORIENTATION.stop:, 0] + true_x[S.ECEF_ORIENTATION.stop:, 0])

    eskf_params = [[inject, nom_x, delta_x], [extract, nom_x, true_x], H_mod_sym, f_err_sym, err_sym]

    # --- observation models
    imu_rot = euler_rotate(*imu_angles)
    px, py, pz = pos
    ecef = sp.Matrix([px, py, pz])
    gravity = device_to_ecef.T * ((EARTH_GM / ((px**2 + py**2 + pz**2)**(3.0 / 2.0))) * ecef)
    vx, vy, vz = vel
    speed = sp.sqrt(vx**2 + vy**2 + vz**2)
    body_rates = sp.Matrix([wr, wp, wy])

    K = ObservationKind
    obs_eqs = [[sp.Matrix([speed * odo_scale]), K.ODOMETRIC_SPEED, None],
               [imu_rot * sp.Matrix([wr + gyro_bias[0], wp + gyro_bias[1], wy + gyro_bias[2]]), K.PHONE_GYRO, None],
               [body_rates, K.NO_ROT, None],
               [imu_rot * (gravity + accel), K.PHONE_ACCEL, None],
               [ecef, K.ECEF_POS, None],
               [sp.Matrix(device_to_ecef.T * vel), K.CAMERA_ODO_TRANSLATION, None],
               [body_rates, K.CAMERA_ODO_ROTATION, None],
               [sp.Matrix(imu_angles), K.IMU_FRAME, None]]

    return dict(name=cls.name, f_sym=f_sym, dt_sym=dt, x_sym=state_sym, obs_eqs=obs_eqs, dim_x=D, dim_err=E,
                eskf_params=eskf_params, quaternion_idxs=list(cls.quaternion_idxs))

  @classmethod
  def generate_code(cls, generated_dir, name=None, maha_test_kinds=(), **gen_kwargs):
    mdl = cls.model()
    if name is not None:
      mdl['name'] = name
    gen_code(generated_dir, maha_test_kinds=list(maha_test_kinds), **mdl, **gen_kwargs)

  def __init__(self, generated_dir, name=None, batch=None, device=None, maha_test_kinds=()):
    D = self.initial_x.shape[0]
    E = self.initial_P_diag.shape[0]
    P0 = np.diag(self.initial_P_diag)
    name = name or self.name
    common = dict(maha_test_kinds=list(maha_test_kinds), quaternion_idxs=list(self.quaternion_idxs))
    if batch is None:
      self.filter = EKF_sym(generated_dir, name, self.Q, self.initial_x, P0, D, E, **common)
    else:
      self.filter = BatchedEKF(generated_dir, name, self.Q, self.initial_x, P0, D, E, batch=batch, device=device, **common)

  def rts_smooth(self, estimates):
    return self.filter.rts_smooth(estimates, norm_quats=True)

  def predict_and_observe(self, t, kind, data, R=None):
    if len(data) > 0:
      data = np.atleast_2d(data)
    K = ObservationKind
    if R is None and kind in (K.CAMERA_ODO_TRANSLATION, K.CAMERA_ODO_ROTATION):
      # rows are [value(3), std(3)]
      z, std = data[:, :3], data[:, 3:]
      R = np.stack([np.diag(s**2) for s in std])
      data = z
    elif R is None:
      R = self.get_R(kind, len(data))
    r = self.filter.predict_and_update_batch(t, kind, data, R)

    # the kernels renormalise the quaternion; a wildly off norm before that means the filter diverged
    q = np.asarray(self.filter.state())[..., States.ECEF_ORIENTATION]
    qn = np.linalg.norm(q, axis=-1)
    if not np.all((0.1 < qn) & (qn < 10)):
      raise KalmanError("Kalman filter quaternions unstable")
    return r


if __name__ == "__main__":
  LiveKalman.generate_code(sys.argv[2])
