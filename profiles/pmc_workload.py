#!/usr/bin/env python3
"""One process that launches EVERY kernel configuration bench.py times, a few launches each, in labelled sections -- the
workload of the rocprofv3 passes of profiles/collect.sh (kernel-trace, FETCH_SIZE, WRITE_SIZE, two SQ counter sets), so that
every timed kernel gets the same counters from the launch configuration the bench actually times (config 4: the
8 192 x 2 100 chunk itself, not a stand-in).

Prints a manifest line per section: `SECTION <label> <library> <n_dispatches> <algorithmic_bytes_per_dispatch>
<filter_steps_per_dispatch> <warmup_dispatches>`: the section issued <warmup> product-kernel dispatches (k_step_* / k_predict /
k_run / k_rts*) that belong to nobody, then <n> that are the section's; profiles/summarize_sections.py joins manifest and
rocpd database by dispatch order.
RN_GEN_DIR selects an A/B build.  PMC_SECTIONS=a,b,c restricts the sections (default: all)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench         # noqa: E402
from rednose_amd.helpers.ekf_sym import BatchedEKF    # noqa: E402

dev = torch.device("cuda:0")
only = [s for s in os.environ.get("PMC_SECTIONS", "").split(",") if s]


def want(label):
  return not only or label in only


def section(label, lib, n, nbytes, steps, warmup):
  torch.cuda.synchronize()
  print(f"SECTION {label} {lib} {n} {nbytes:.0f} {steps} {warmup}", flush=True)


def stepwise(label, model, n, reps, only_kind=None, ring=0):
  """ring > 0: the same launches with the rewind ring on -- every call is ONE launch of the step that writes its own checkpoint (k_stepc_{kind}); the
  section's algorithmic bytes then include the checkpoint's x, P and observation (written once)."""
  if not want(label):
    return
  M = bench.model_class(model)
  gen = bench.gen_dir([model])
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev,
                 quaternion_idxs=list(getattr(M, "quaternion_idxs", [])), **({"rewind_to_keep": ring} if ring else {}))
  if model == "live":
    x0, P0, sched = bench.live_stream(torch, M, gen, n, reps + 21, dev, 0)
    if only_kind is not None:
      pool = [s for s in sched if s[0] == only_kind]
      sched = [(only_kind, 0.01 * i, pool[i % len(pool)][2].clone()) for i in range(reps + 21)]
  else:
    x0, P0, sched = bench.kinematic_stream(torch, M, n, reps + 21, dev, 0)
  f.init_state(x0, P0, None)
  Rs = {k: np.atleast_2d(v) for k, v in M.obs_noise.items()}
  tp = None
  for i, (k, t, z) in enumerate(sched):       # 21 warm-up launches (a whole 10-tick pattern of the live stream), then the section
    if i == 21:
      zd = [f.zdims[s[0]] for s in sched[21:]]
      section(label, M.name, reps, 8.0 * ((3 if ring else 2) * (D + E * E) + (3 if ring else 2) * float(np.mean(zd))) * n, n, 21)
    f.predict_and_update_batch(t if tp is None or t >= tp else tp, k, z.clone(), Rs[k])
    tp = t
  torch.cuda.synchronize()


def fused(label, model, n, T):
  if not want(label):
    return
  M = bench.model_class(model)
  gen = bench.gen_dir([model])
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev)
  Z = int(np.atleast_2d(M.obs_noise[1]).shape[0])
  zs = torch.randn((T, n, Z), dtype=torch.float64, device=dev) * 0.1
  ts, kinds = np.arange(1, T + 1) * 0.01, np.ones(T, dtype=np.int32)
  for rep in range(2):
    f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
    if rep == 1:
      # two dispatches belong to nobody: rep 0, and a 16-step launch queued right in front of the section's own -- like bench.fused_run_extra, so that
      # the one launch that is measured runs on a device that is awake (a lone 0.5 ms launch after an idle gap runs at ramping clocks: 552 us against
      # 372 us for the 2-state model, same kernel)
      section(label, M.name, 1, 8.0 * 2 * Z * n * T + 8.0 * 2 * (D + E * E) * n, n * T, 2)
      zc = zs.clone()
      f.run(ts[:16], kinds[:16], zs[:16].clone(), {1: M.obs_noise[1]})
      f.filter_time = 0.0
      f.run(ts, kinds, zc, {1: M.obs_noise[1]})
    else:
      f.run(ts, kinds, zs.clone(), {1: M.obs_noise[1]})
  torch.cuda.synchronize()


def msckf(label, n, reps):
  if not want(label):
    return
  from examples.feature_kf import WideFeatureKalman as FK
  gen = bench.gen_dir(["feature36"])
  f = BatchedEKF(gen, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), 6, 6, batch=n, device=dev, **FK.filter_kwargs())
  lm = torch.tensor([2.0, 1.0, 8.0], dtype=torch.float64, device=dev) + torch.randn((n, 3), dtype=torch.float64, device=dev)
  for i in range(reps + 5):
    if i == 5:
      section(label, FK.name, reps, 8.0 * (2 * (36 + 36 * 36) + 6 + 3 + 3) * n, n, 5)
    f.predict_and_update_batch(0.01 * (i + 1), 2, 0.05 * torch.randn((n, 6), dtype=torch.float64, device=dev), FK.obs_noise[2], extra_args=lm)
  torch.cuda.synchronize()


def config4(n=8192, T=2100):
  """One chunk of config 4 exactly as bench.config4_extra launches it: forward k_run writing trace + gate flags, backward
  smoother (k_rts4) in place on that trace."""
  if not (want("config4_forward") or want("config4_backward") or want("config4_backward_dt_gt0") or want("live_run_notrace")):
    return
  from examples.live_kf import LiveKalman as L
  gen = bench.gen_dir(["live_maha"])
  f = BatchedEKF(gen, "live_maha", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=n, device=dev, quaternion_idxs=[3], maha_test_kinds=[12])
  gdev = torch.Generator(device=dev).manual_seed(4242)
  hacc = bench.live_true_accel(L, gen, "live_maha")
  x0 = bench.live_x0(torch, L, n, dev, gdev)
  kinds, ts = bench.live_schedule(T)
  zs = bench.live_observations(torch, L, hacc, kinds, n, dev, gdev, outlier_frac=0.02)
  Rs = {int(k): np.atleast_2d(L.obs_noise[int(k)]) for k in set(kinds.tolist())}
  if want("live_run_notrace"):
    for rep in range(2):
      f.init_state(x0, np.diag(L.initial_P_diag), None)
      if rep == 1:
        section("live_run_notrace", "live_maha", 1, 8.0 * 2 * 3 * n * 252 + 8.0 * 2 * (23 + 484) * n, n * 252, 1)
      f.run(ts[:252], kinds[:252], zs[:252].clone(), Rs)
  if want("config4_forward") or want("config4_backward") or want("config4_backward_dt_gt0"):
    tx = torch.empty((T, n, 23), dtype=torch.float64, device=dev)
    tP = torch.empty((T, n, 22, 22), dtype=torch.float64, device=dev)
    f.init_state(x0, np.diag(L.initial_P_diag), None)
    f.run(ts[:21], kinds[:21], zs[:21].clone(), Rs, flags=True, out=(tx[:21], tP[:21]))       # warm-up (code object load)
    f.init_state(x0, np.diag(L.initial_P_diag), None)
    section("config4_forward", "live_maha", 1, n * T * 8.0 * ((23 + 484) + 2 * 3) + n * T, n * T, 1)
    f.run(ts, kinds, zs.clone(), Rs, flags=True, out=(tx, tP))
    section("config4_backward", "live_maha", 1, n * (T - 1) * 8.0 * 2 * (23 + 484), n * (T - 1), 0)
    f._rts_on(tx, tP, ts, n, None)      # pylint: disable=protected-access
    # the same kernel with every step advancing time: no step takes the identity-gain path (bench.py: roofline_backward_dt_gt0); in place on the
    # smoothed trace of the launch above
    section("config4_backward_dt_gt0", "live_maha", 1, n * (T - 1) * 8.0 * 2 * (23 + 484), n * (T - 1), 0)
    f._rts_on(tx, tP, 0.01 * np.arange(T), n, None)      # pylint: disable=protected-access
  torch.cuda.synchronize()


def calibration():
  """1 GiB fp64 copy (past the 256 MiB Infinity Cache): 2^30 B read + 2^30 B written per launch, the known byte count the
  FETCH_SIZE / WRITE_SIZE corrections are checked against in the same pass (MI355X_MICROARCH.md, HBM section)."""
  if not want("calibration"):
    return
  a = torch.ones(2**27, dtype=torch.float64, device=dev)
  b = torch.empty_like(a)
  b.copy_(a)
  torch.cuda.synchronize()
  print("CALIBRATION copy 3 1073741824", flush=True)
  for _ in range(3):
    b.copy_(a)
  torch.cuda.synchronize()


if __name__ == "__main__":
  stepwise("kinematic6_b65536", "kinematic6", 65536, 20)
  stepwise("kinematic6_b1048576", "kinematic6", 1 << 20, 5)
  stepwise("kinematic6_ring_b65536", "kinematic6", 65536, 20, ring=8)
  stepwise("kinematic_b65536", "kinematic", 65536, 20)
  stepwise("kinematic9_b65536", "kinematic9", 65536, 10)
  stepwise("live_b16384", "live", 16384, 21)
  stepwise("live_dt_gt0_b16384", "live", 16384, 10, only_kind=4)
  msckf("feature36_b16384", 16384, 10)
  fused("kinematic_fused_b65536", "kinematic", 65536, 2000)
  fused("kinematic6_fused_b65536", "kinematic6", 65536, 500)
  config4()
  calibration()
  print("WORKLOAD done", flush=True)
