"""Inputs of one live filter for tools/ab_step.cpp (doubles: x0[D], P0[E*E], Q[E*E], R[Z*Z], z[Z]):
   python tools/ab_inputs.py <kind> <out.bin>     kind 4 (gyro), 10 (accelerometer), 12 (ECEF position)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.live_kf import LiveKalman as L      # noqa: E402

kind = int(sys.argv[1])
z = {4: np.zeros(3), 10: np.array([0.0, 0.0, -9.81]), 12: L.initial_x[:3]}[kind]
parts = [L.initial_x, np.diag(L.initial_P_diag).reshape(-1), np.asarray(L.Q).reshape(-1), np.atleast_2d(L.obs_noise[kind]).reshape(-1), z]
np.concatenate([np.asarray(p, dtype=np.float64) for p in parts]).tofile(sys.argv[2])
