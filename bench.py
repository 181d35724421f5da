#!/usr/bin/env python3
"""bench.py -- EKF predict+update steps/sec at batch N on MI355X, with roofline and CPU baseline.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
For N > 1 the driver launches one rank per GPU with torch.distributed.run (RCCL); the batch axis shards
with no data-path collective (filters are independent), so scaling is weak: every rank owns
`--batch` filters; the only collectives are the barrier and a MAX/SUM all-reduce of the timing.

Workload (BASELINE.json configs[1]): kinematic6 (6-state 3-D pos/vel, 3-D position observation), batch
65 536 per GPU, fp64.  A step = ONE fused predict(dt) + update(kind) launch over the whole batch through
the generated library's C ABI ({name}_batch_predict_update_{kind}); state round-trips HBM every step
(the reference's per-call semantics).  Inputs (x, P, the observation stream) are resident in HBM before
the timed region.

roofline: HBM-bound.  Algorithmic bytes per filter-step actually moved with a shared R and scalar dt:
reads x(6)+P(36)+z(3), writes x(6)+P(36)+y(3) = 90 doubles = 720 B (SURVEY.md 8d quotes 800 B when dt and R
are per-filter arrays; bytes that are not moved are not counted).  achieved = 720 B x batch / mean launch
duration measured with HIP events on the launch stream over the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "oracle")):
  if p not in sys.path:
    sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(name, kind, K6, batch, budget_s=6.0):
  """Oracle (C restatement of ekf_c.c + reference-generated sympy C, gcc -O2 as in the reference's SConstruct)
  timed on the host cores of this box: 1 thread (the reference's execution model) and all cores."""
  from oracle_lib import OracleLib
  lib = OracleLib(name)
  try:
    gomp = ctypes.CDLL("libgomp.so.1")
  except OSError:
    gomp = None
  rng = np.random.default_rng(1)
  n = min(batch, 65536)
  x = np.tile(K6.initial_x, (n, 1)) + rng.normal(size=(n, lib.D)) * 0.1
  P = np.tile(np.diag(K6.initial_P_diag), (n, 1, 1))
  Z = lib.zdim(kind)
  R = np.atleast_2d(K6.obs_noise[kind])
  out = {}
  for label, threads in (("1core", 1), ("allcores", min(os.cpu_count() or 1, 64))):
    if gomp is not None:
      gomp.omp_set_num_threads(int(threads))
    elif threads != 1:
      continue
    zpool = rng.normal(size=(8, n, Z))
    steps, el = 0, 0.0
    while el < budget_s and steps < 4000:
      z = zpool[steps % 8].copy()
      t0 = time.perf_counter()
      lib.batch_step(kind, x, P, z, R, K6.Q, 0.01)
      el += time.perf_counter() - t0
      steps += 1
    out[label] = dict(value=n * steps / el, cores=int(threads), steps=steps, seconds=el)
  flav = "reference-generated sympy C + C restatement of ekf_c.c" if lib.flavour == "ref" else "port (sympy C99 + C restatement of ekf_c.c)"
  return out, n, flav


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--warmup", type=int, default=100)
  ap.add_argument("--batch", type=int, default=65536, help="filters per GPU")
  ap.add_argument("--model", default="kinematic6", choices=["kinematic6", "kinematic"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  args = ap.parse_args()

  import torch
  import torch.distributed as dist

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a HIP device: rednose_amd has no CPU path")
  torch.cuda.set_device(local_rank)
  dev = torch.device(f"cuda:{local_rank}")
  if world > 1:
    dist.init_process_group(backend="nccl", device_id=dev)

  from examples import ensure_generated
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  if args.model == "kinematic6":
    from examples.kinematic6_kf import Kinematic6Kalman as M
  else:
    from examples.kinematic_kf import KinematicKalman as M
  kind = 1
  if rank == 0:
    gen = ensure_generated([args.model])
  if world > 1:
    dist.barrier()
  gen = ensure_generated([args.model])

  n, K, W = args.batch, args.steps, args.warmup
  D = M.initial_x.shape[0]
  E = M.initial_P_diag.shape[0]
  R = np.atleast_2d(M.obs_noise[kind])
  Z = R.shape[0]
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev)

  # synthetic stream: truth v_i(t) = sin(5t + phi_i) per axis, z = pos + N(0, 0.1^2)
  gcpu = torch.Generator().manual_seed(1234 + rank)
  gdev = torch.Generator(device=dev).manual_seed(1234 + rank)
  phi = torch.rand((n, Z), generator=gcpu, dtype=torch.float64).to(dev) * (2 * np.pi)
  x0 = torch.as_tensor(M.initial_x, dtype=torch.float64).repeat(n, 1) + 0.1 * torch.randn((n, D), generator=gcpu, dtype=torch.float64)
  f.init_state(x0, np.diag(M.initial_P_diag), None)
  dt = 0.01
  total = W + K
  ts = torch.arange(total, dtype=torch.float64, device=dev) * dt
  # pos(t) = integral of sin(5 s + phi) ds, closed form; one (total, n, Z) block resident in HBM
  pos = (torch.cos(phi)[None] - torch.cos(5.0 * ts[:, None, None] + phi[None])) / 5.0
  zs = pos + 0.1 * torch.randn(pos.shape, generator=gdev, dtype=torch.float64, device=dev)
  del pos

  def step(i):
    f.predict_and_update_batch(float(i) * dt, kind, zs[i], R)

  for i in range(W):
    step(i)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for i in range(W, W + K):
    step(i)
  ev1.record()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  dev_ms = ev0.elapsed_time(ev1)

  # sanity: the filter must have tracked the truth (guards against timing a broken kernel)
  X = f.x
  assert torch.isfinite(X).all() and torch.isfinite(f.P).all()

  stats = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
  wall_max, dev_ms_max = float(stats[0]), float(stats[1])

  if rank == 0:
    bytes_per_step = 8 * (2 * (D + E * E) + 2 * Z)
    launch_s = dev_ms_max * 1e-3 / K
    achieved = bytes_per_step * n / launch_s / 1e9
    traffic = None
    tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
      with open(tf, encoding="utf-8") as fh:
        rec = json.load(fh)
      key = f"{M.name}_b{n}"
      if key in rec:
        traffic = rec[key]["hbm_bytes_per_launch"]
    out = {
      "metric": "EKF predict+update steps/sec at batch N",
      "value": n * world * K / wall_max,
      "unit": "steps/s",
      "n_gpus": world,
      "steps": K,
      "warmup": W,
      "ms_per_step": wall_max * 1e3 / K,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f64",
      "data": "synthetic",
      "config": {"workload": f"{M.name} (D={D}, E={E}, Z={Z}) fused predict+update, step-granular (state round-trips HBM each step), "
                             f"batch {n} per GPU, shared R, scalar dt", "batch_per_gpu": n, "global_batch": n * world,
                 "parallelism": f"batch-sharded x{world}, no data-path collective"},
      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                   "traffic": traffic, "kernel": f"k_step_{kind}<true>", "algorithmic_bytes_per_launch": bytes_per_step * n,
                   "launch_us": launch_s * 1e6},
    }
    if not args.no_cpu_baseline and world == 1:
      cb, ncpu, flav = cpu_baseline(M.name, kind, M, n)
      one = cb["1core"]
      out["cpu_baseline"] = {"value": one["value"], "unit": "steps/s", "cores": 1, "kind": "port",
                             "sample": f"{ncpu} filters x {one['steps']} steps ({one['seconds']:.1f} s), {flav}, gcc -O2",
                             "all_cores": cb.get("allcores")}
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
