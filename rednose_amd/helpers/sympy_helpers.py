"""Symbolic and numeric rotation helpers used by model definitions.

Same function names and conventions as /root/reference/rednose/helpers/sympy_helpers.py:5-119
(`quat_rotate` :101, `euler_rotate` :87, `quat_matrix_l/r` :108/:115, `cross` :62, `quat2rot` :5,
`euler2quat` :30, `rot_matrix` :77) so a model written for the reference builds unchanged.
`sympy_into_c` (:122-162) is NOT reproduced: expression lowering lives in
rednose_amd/codegen/lower.py (CSE + a HIP device-code printer).
"""
import numpy as np
import sympy as sp


def _hamilton_matrix(p, sign):
  """4x4 matrix M with M @ q == p (x) q (sign=+1, left) or q (x) p (sign=-1, right)."""
  w, x, y, z = p[0], p[1], p[2], p[3]
  s = sign
  return sp.Matrix([[w, -x, -y, -z],
                    [x, w, -s * z, s * y],
                    [y, s * z, w, -s * x],
                    [z, -s * y, s * x, w]])


def quat_matrix_l(p):
  return _hamilton_matrix(p, 1)


def quat_matrix_r(p):
  return _hamilton_matrix(p, -1)


def cross(v):
  """Skew-symmetric matrix [v]x."""
  return sp.Matrix([[0, -v[2], v[1]],
                    [v[2], 0, -v[0]],
                    [-v[1], v[0], 0]])


def quat_rotate(q0, q1, q2, q3):
  """Rotation matrix of a (unit) quaternion, body -> reference frame."""
  diag = [q0**2 + q1**2 - q2**2 - q3**2,
          q0**2 - q1**2 + q2**2 - q3**2,
          q0**2 - q1**2 - q2**2 + q3**2]
  rot = sp.zeros(3, 3)
  for i in range(3):
    rot[i, i] = diag[i]
  # off-diagonals: 2 (qi qj -/+ q0 qk)
  rot[0, 1] = 2 * (q1 * q2 - q0 * q3)
  rot[1, 0] = 2 * (q1 * q2 + q0 * q3)
  rot[0, 2] = 2 * (q1 * q3 + q0 * q2)
  rot[2, 0] = 2 * (q1 * q3 - q0 * q2)
  rot[1, 2] = 2 * (q2 * q3 - q0 * q1)
  rot[2, 1] = 2 * (q2 * q3 + q0 * q1)
  return rot


def _axis_rotation(axis, angle, cos=sp.cos, sin=sp.sin, mat=sp.Matrix):
  c, s = cos(angle), sin(angle)
  if axis == 0:
    return mat([[1, 0, 0], [0, c, -s], [0, s, c]])
  if axis == 1:
    return mat([[c, 0, s], [0, 1, 0], [-s, 0, c]])
  return mat([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def euler_rotate(roll, pitch, yaw):
  """Symbolic yaw-pitch-roll rotation matrix Rz(yaw) Ry(pitch) Rx(roll)."""
  return _axis_rotation(2, yaw) * _axis_rotation(1, pitch) * _axis_rotation(0, roll)


def rot_matrix(roll, pitch, yaw):
  """Numeric twin of euler_rotate."""
  m = lambda rows: np.array(rows, dtype=np.float64)  # noqa: E731
  rz = _axis_rotation(2, yaw, np.cos, np.sin, m)
  ry = _axis_rotation(1, pitch, np.cos, np.sin, m)
  rx = _axis_rotation(0, roll, np.cos, np.sin, m)
  return rz @ ry @ rx


def rot_to_euler(R):
  return sp.Matrix([sp.atan2(R[2, 1], R[2, 2]), sp.asin(-R[2, 0]), sp.atan2(R[1, 0], R[0, 0])])


def quat2rot(quats):
  """Numeric quaternion(s) -> rotation matrix / matrices, shape (..., 3, 3)."""
  q = np.atleast_2d(np.asarray(quats, dtype=np.float64))
  w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
  R = np.empty((q.shape[0], 3, 3))
  R[:, 0, 0] = w * w + x * x - y * y - z * z
  R[:, 1, 1] = w * w - x * x + y * y - z * z
  R[:, 2, 2] = w * w - x * x - y * y + z * z
  R[:, 0, 1] = 2 * (x * y - w * z)
  R[:, 1, 0] = 2 * (x * y + w * z)
  R[:, 0, 2] = 2 * (x * z + w * y)
  R[:, 2, 0] = 2 * (x * z - w * y)
  R[:, 1, 2] = 2 * (y * z - w * x)
  R[:, 2, 1] = 2 * (y * z + w * x)
  return R[0] if np.ndim(quats) < 2 else R


rotations_from_quats = quat2rot


def euler2quat(eulers):
  """Numeric (roll, pitch, yaw) -> quaternion with non-negative scalar part."""
  e = np.atleast_2d(np.asarray(eulers, dtype=np.float64))
  cr, sr = np.cos(e[:, 0] / 2), np.sin(e[:, 0] / 2)
  cp, sp_ = np.cos(e[:, 1] / 2), np.sin(e[:, 1] / 2)
  cy, sy = np.cos(e[:, 2] / 2), np.sin(e[:, 2] / 2)
  q = np.stack([cr * cp * cy + sr * sp_ * sy,
                sr * cp * cy - cr * sp_ * sy,
                cr * sp_ * cy + sr * cp * sy,
                cr * cp * sy - sr * sp_ * cy], axis=1)
  q[q[:, 0] < 0] *= -1
  return q[0] if np.ndim(eulers) < 2 else q


def euler2rot(eulers):
  return quat2rot(euler2quat(eulers))
