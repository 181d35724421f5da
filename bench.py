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
  quat = list(getattr(K6, 'quaternion_idxs', []))
  x = np.tile(K6.initial_x, (n, 1)) + (0.0 if quat else 0.1) * rng.normal(size=(n, lib.D))
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
      lib.batch_step(kind, x, P, z, R, K6.Q, 0.01, quat_idx=quat[0] if quat else -1)
      el += time.perf_counter() - t0
      steps += 1
    out[label] = dict(value=n * steps / el, cores=int(threads), steps=steps, seconds=el)
  flav = "reference-generated sympy C + C restatement of ekf_c.c" if lib.flavour == "ref" else "port (sympy C99 + C restatement of ekf_c.c)"
  return out, n, flav


def kinematic_stream(torch, M, n, total, dev, rank):
  """truth v_i(t) = sin(5t + phi_i) per axis, z = pos + N(0, 0.1^2); one kind, dt = 0.01."""
  D = M.initial_x.shape[0]
  Z = np.atleast_2d(M.obs_noise[1]).shape[0]
  gcpu = torch.Generator().manual_seed(1234 + rank)
  gdev = torch.Generator(device=dev).manual_seed(1234 + rank)
  phi = torch.rand((n, Z), generator=gcpu, dtype=torch.float64).to(dev) * (2 * np.pi)
  x0 = torch.as_tensor(M.initial_x, dtype=torch.float64).repeat(n, 1) + 0.1 * torch.randn((n, D), generator=gcpu, dtype=torch.float64)
  ts = torch.arange(total, dtype=torch.float64, device=dev) * 0.01
  pos = (torch.cos(phi)[None] - torch.cos(5.0 * ts[:, None, None] + phi[None])) / 5.0   # closed-form integral of v
  zs = pos + 0.1 * torch.randn(pos.shape, generator=gdev, dtype=torch.float64, device=dev)
  sched = [(1, 0.01 * i, zs[i]) for i in range(total)]
  return x0, np.diag(M.initial_P_diag), sched


def live_stream(torch, M, f, n, total, dev, rank):
  """stationary device, per 10 ms tick a PHONE_GYRO(4) then a PHONE_ACCEL(10) observation at the same
  time (second has dt = 0), every 10th tick an ECEF_POS(12); per-filter initial attitude error <= 0.05 rad."""
  from rednose_amd.helpers.ekf_sym import EKF_sym
  gdev = torch.Generator(device=dev).manual_seed(2025 + rank)
  x_true = M.initial_x.copy()
  # expected specific force at the true state through the library's own h_10 (GPU, batch of one)
  s = EKF_sym(f._folder, f.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), 23, 22)
  hacc = np.zeros(3)
  s.hs[10](np.ascontiguousarray(x_true), np.zeros(1), hacc)
  e = (torch.rand((n, 3), generator=gdev, dtype=torch.float64, device=dev) - 0.5) * 0.1
  x0 = torch.as_tensor(x_true, dtype=torch.float64, device=dev).repeat(n, 1)
  q = torch.cat([torch.ones((n, 1), dtype=torch.float64, device=dev), 0.5 * e], dim=1)
  x0[:, 3:7] = q / q.norm(dim=1, keepdim=True)
  sched = []
  tick = 0
  while len(sched) < total:
    t = 0.01 * tick
    sched.append((4, t, 0.025 * torch.randn((n, 3), generator=gdev, dtype=torch.float64, device=dev)))
    sched.append((10, t, torch.as_tensor(hacc, device=dev)[None] + 0.5 * torch.randn((n, 3), generator=gdev, dtype=torch.float64, device=dev)))
    if tick % 10 == 9:
      sched.append((12, t, torch.as_tensor(x_true[:3], device=dev)[None] + 5.0 * torch.randn((n, 3), generator=gdev, dtype=torch.float64, device=dev)))
    tick += 1
  return x0, np.diag(M.initial_P_diag), sched[:total]


def run_model(torch, dist, args, model, n, K, W, dev, rank, world):
  """Warm up W steps, time exactly K steps (barrier + synchronize on both sides).  Returns timing dict (max over ranks)."""
  from examples import ensure_generated
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  if model == "kinematic6":
    from examples.kinematic6_kf import Kinematic6Kalman as M
  elif model == "kinematic":
    from examples.kinematic_kf import KinematicKalman as M
  elif model == "kinematic9":
    from examples.kinematic9_kf import Kinematic9Kalman as M
  else:
    from examples.live_kf import LiveKalman as M
  if rank == 0:
    ensure_generated([model], **({'folder': os.environ['RN_GEN_DIR']} if 'RN_GEN_DIR' in os.environ else {}))
  if world > 1:
    dist.barrier()
  gen = ensure_generated([model], **({'folder': os.environ['RN_GEN_DIR']} if 'RN_GEN_DIR' in os.environ else {}))
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  quat = list(getattr(M, "quaternion_idxs", []))
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev, quaternion_idxs=quat)
  f._folder = gen
  total = W + K
  if model == "live":
    x0, P0, sched = live_stream(torch, M, f, n, total, dev, rank)
  else:
    x0, P0, sched = kinematic_stream(torch, M, n, total, dev, rank)
  f.init_state(x0, P0, None)
  Rs = {k: np.atleast_2d(v) for k, v in M.obs_noise.items()}

  # the timed loop calls the library's C entry point through pre-bound arguments (what a C/C++ caller does);
  # BatchedEKF.predict_and_update_batch adds several microseconds of Python argument handling per call
  bound = {k: f.bind_step(k, Rs[k]) for k in sorted(set(s_[0] for s_ in sched))}
  sched = [(k, t, z.contiguous()) for (k, t, z) in sched]
  t_prev = [None]

  def step(i):
    kind, t, z = sched[i]
    dt = 0.0 if t_prev[0] is None else t - t_prev[0]
    t_prev[0] = t
    bound[kind](z, dt)

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
  assert torch.isfinite(f.x).all() and torch.isfinite(f.P).all(), "filter diverged: refusing to report a timing"
  stats = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
  zdims = [f.zdims[sched[i][0]] for i in range(W, W + K)]
  bytes_per_step = 8.0 * (2 * (D + E * E) + 2 * float(np.mean(zdims)))
  return dict(M=M, D=D, E=E, Z=float(np.mean(zdims)), wall=float(stats[0]), dev_ms=float(stats[1]), bytes_per_step=bytes_per_step,
              kinds=sorted(set(s[0] for s in sched[W:W + K])))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--warmup", type=int, default=100)
  ap.add_argument("--batch", type=int, default=None, help="filters per GPU (default 65536; 16384 for --model live)")
  ap.add_argument("--model", default="kinematic6", choices=["kinematic6", "kinematic", "kinematic9", "live"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-extras", action="store_true", help="skip the additional configs reported under 'extra'")
  args = ap.parse_args()

  import torch
  import torch.distributed as dist

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a HIP device: rednose_amd has no CPU path")
  ndev = torch.cuda.device_count()
  dev_index = local_rank % ndev                    # one rank per GPU; the modulo only matters for single-GPU dry runs
  torch.cuda.set_device(dev_index)
  dev = torch.device(f"cuda:{dev_index}")
  if world > 1:
    # "nccl" is RCCL on ROCm.  RN_BENCH_BACKEND=gloo allows a functional dry run of this path with several ranks on ONE
    # GPU (RCCL refuses duplicate devices); it is never used for reported numbers.
    backend = os.environ.get("RN_BENCH_BACKEND", "nccl")
    if backend == "nccl":
      dist.init_process_group(backend="nccl", device_id=dev)
    else:
      dist.init_process_group(backend=backend)

  K, W = args.steps, args.warmup
  n = args.batch or (16384 if args.model == "live" else 65536)
  r = run_model(torch, dist, args, args.model, n, K, W, dev, rank, world)
  M, D, E = r["M"], r["D"], r["E"]

  extra = {}
  if not args.no_extras and world == 1:
    others = {"kinematic6": [("live", 16384, 420, 42), ("kinematic", 65536, 500, 50), ("kinematic6", 1 << 20, 200, 20),
                             ("kinematic9", 65536, 300, 30)],
              "live": [], "kinematic": [], "kinematic9": []}[args.model]
    for om, on, oK, oW in others:
      o = run_model(torch, dist, args, om, on, oK, oW, dev, rank, world)
      ls = o["dev_ms"] * 1e-3 / oK
      extra[om if on != (1 << 20) else om + "_1M"] = {"batch": on, "steps": oK, "value": on * oK / o["wall"], "unit": "steps/s", "launch_us": ls * 1e6,
                   "algorithmic_bytes_per_filter_step": o["bytes_per_step"], "achieved_GBs": o["bytes_per_step"] * on / ls / 1e9,
                   "frac_of_8TBs": o["bytes_per_step"] * on / ls / 1e9 / HBM_PEAK_GBS, "kinds": o["kinds"]}

  if not args.no_extras and world == 1 and args.model != "live":
    # fused multi-step mode ({name}_batch_run): x and P stay on chip for T steps, only z / y cross HBM
    from examples import ensure_generated
    from rednose_amd.helpers.ekf_sym import BatchedEKF
    gen = ensure_generated([args.model])
    f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev)
    T = 500
    Z = int(np.atleast_2d(M.obs_noise[1]).shape[0])
    zs = torch.randn((T, n, Z), dtype=torch.float64, device=dev) * 0.1
    ts = np.arange(1, T + 1) * 0.01
    f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
    f.run(ts, np.ones(T, dtype=np.int32), zs.clone(), {1: M.obs_noise[1]})      # warm-up
    torch.cuda.synchronize()
    f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f.run(ts, np.ones(T, dtype=np.int32), zs, {1: M.obs_noise[1]})
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    moved = 8.0 * (2 * Z) * n * T + 8.0 * 2 * (D + E * E) * n
    extra["fused_run"] = {"model": M.name, "batch": n, "T": T, "value": n * T / (ms * 1e-3), "unit": "steps/s", "ms": ms,
                          "hbm_bytes_moved": moved, "achieved_GBs": moved / (ms * 1e-3) / 1e9,
                          "note": "state resident in VGPRs for T steps; bound by fp64 VALU issue, not HBM"}

  if not args.no_extras and world == 1 and args.model == "kinematic6":
    # MSCKF: null-space projected feature-track updates of the windowed-camera example, 36 error
    # states, per-filter landmarks; timed through the generic Python method (no pre-bound variant takes extra arguments)
    from examples import ensure_generated
    from examples.feature_kf import WideFeatureKalman as FK
    from rednose_amd.helpers.ekf_sym import BatchedEKF
    genf = ensure_generated(["feature36"])
    nf, Kf = 16384, 100
    ff = BatchedEKF(genf, FK.name, FK.Q, FK.initial_x, np.diag(FK.initial_P_diag), 6, 6, batch=nf, device=dev, **FK.filter_kwargs())
    lm = torch.tensor([2.0, 1.0, 8.0], dtype=torch.float64, device=dev) + torch.randn((nf, 3), dtype=torch.float64, device=dev)
    zf = [0.05 * torch.randn((nf, 6), dtype=torch.float64, device=dev) for _ in range(8)]
    for i in range(10):
      ff.predict_and_update_batch(0.01 * (i + 1), 2, zf[i % 8].clone(), FK.obs_noise[2], extra_args=lm)
    zc = [zf[i % 8].clone() for i in range(Kf)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(Kf):
      ff.predict_and_update_batch(0.01 * (i + 11), 2, zc[i], FK.obs_noise[2], extra_args=lm)
    e1.record()
    torch.cuda.synchronize()
    assert torch.isfinite(ff.x).all() and torch.isfinite(ff.P).all()
    msf = e0.elapsed_time(e1) / Kf
    bf = 8.0 * (2 * (36 + 36 * 36) + 6 + 3 + 3)
    extra["feature36_msckf"] = {"batch": nf, "steps": Kf, "value": nf / (msf * 1e-3), "unit": "steps/s", "launch_us": msf * 1e3,
                                "algorithmic_bytes_per_filter_step": bf, "frac_of_8TBs": bf * nf / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "fused predict + feature-track update (Z = 6 projected to 3), one filter per wavefront"}
    del ff, zf, zc

    # BASELINE config 4: live with the Mahalanobis gate on ECEF_POS, forward pass keeping the filtered trace, then the
    # batched RTS backward pass.  T = 210 steps (1 s of IMU@100Hz + GNSS@10Hz), batch 16384, 2 % GNSS outliers.
    from examples import ensure_generated
    from examples.live_kf import LiveKalman as L
    from rednose_amd.helpers.ekf_sym import BatchedEKF
    gen = ensure_generated(["live_maha"])
    nb = 16384
    f = BatchedEKF(gen, "live_maha", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=nb, device=dev, quaternion_idxs=[3],
                   maha_test_kinds=[12])
    f._folder = gen
    x0, P0, sched = live_stream(torch, L, f, nb, 210, dev, rank)
    kinds = np.array([s_[0] for s_ in sched], dtype=np.int32)
    tsl = np.array([s_[1] for s_ in sched])
    zsl = torch.stack([s_[2].expand(nb, 3) for s_ in sched]).contiguous()
    gsel = torch.rand((int((kinds == 12).sum()), nb), device=dev) < 0.02
    zsl[torch.as_tensor(np.where(kinds == 12)[0], device=dev)] += gsel[..., None] * 500.0 * torch.randn((gsel.shape[0], nb, 3), dtype=torch.float64, device=dev)
    Rs = {int(k): np.atleast_2d(L.obs_noise[int(k)]) for k in set(kinds.tolist())}
    res = {}
    for rep in range(2):
      f.init_state(x0, P0, None)
      e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
      e0.record()
      _, tx, tP, fl = f.run(tsl, kinds, zsl.clone(), Rs, trace=True, flags=True)
      e1.record()
      xs, Ps = f.rts_smooth(tx, tP, tsl, inplace=True)
      e2.record()
      torch.cuda.synchronize()
      res = {"fwd_ms": e0.elapsed_time(e1), "bwd_ms": e1.elapsed_time(e2)}
    assert torch.isfinite(xs).all() and torch.isfinite(Ps).all()
    T4 = len(kinds)
    extra["live_maha_rts"] = {"batch": nb, "T": T4, "forward_steps_per_s": nb * T4 / (res["fwd_ms"] * 1e-3),
                              "backward_steps_per_s": nb * (T4 - 1) / (res["bwd_ms"] * 1e-3), "forward_ms": res["fwd_ms"], "backward_ms": res["bwd_ms"],
                              "gated_fraction_of_gnss": float(fl[torch.as_tensor(np.where(kinds == 12)[0], device=dev)].float().mean()),
                              "trace_bytes": int(tx.numel() + tP.numel()) * 8,
                              "note": "forward = fused batch_run with filtered trace + gate flags; backward = batch_rts recomputing the predicted pairs"}
    del tx, tP, xs, Ps, zsl

  if rank == 0:
    launch_s = r["dev_ms"] * 1e-3 / K
    achieved = r["bytes_per_step"] * n / launch_s / 1e9
    traffic = None
    tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
      with open(tf, encoding="utf-8") as fh:
        rec = json.load(fh)
      key = f"{M.name}_b{n}"
      if key in rec:
        traffic = rec[key]["hbm_bytes_per_launch"]
    kern = "k_step_1<true>" if args.model != "live" else "k_step_{4,10,12}<true> (stream mix)"
    out = {
      "metric": "EKF predict+update steps/sec at batch N",
      "value": n * world * K / r["wall"],
      "unit": "steps/s",
      "n_gpus": world,
      "steps": K,
      "warmup": W,
      "ms_per_step": r["wall"] * 1e3 / K,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f64",
      "data": "synthetic",
      "config": {"workload": f"{M.name} (D={D}, E={E}, Z={r['Z']:g}) fused predict+update, step-granular (state round-trips HBM each step), "
                             f"batch {n} per GPU, shared R, scalar dt", "batch_per_gpu": n, "global_batch": n * world,
                 "parallelism": f"batch-sharded x{world}, no data-path collective"},
      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                   "traffic": traffic, "kernel": kern, "algorithmic_bytes_per_launch": r["bytes_per_step"] * n,
                   "launch_us": launch_s * 1e6},
    }
    if not args.no_cpu_baseline and world == 1:
      kind = 1 if args.model != "live" else 10
      cb, ncpu, flav = cpu_baseline(M.name, kind, M, n)
      one = cb["1core"]
      out["cpu_baseline"] = {"value": one["value"], "unit": "steps/s", "cores": 1, "kind": "port",
                             "sample": f"{ncpu} filters x {one['steps']} steps ({one['seconds']:.1f} s), {flav}, gcc -O2",
                             "all_cores": cb.get("allcores")}
    if extra:
      out["extra"] = extra
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
