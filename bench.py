#!/usr/bin/env python3
"""bench.py -- EKF predict+update steps/sec at batch N on MI355X, with roofline and CPU baseline.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
For N > 1 the driver launches one rank per GPU with torch.distributed.run (RCCL); the batch axis shards with no
data-path collective (filters are independent).  Default scaling is weak: every rank owns `--batch` filters;
`--global-batch G` fixes the total instead (strong scaling, rank r owns rednose_amd.helpers.sharding.shard_range(G, r, N)).
The only collectives are the barrier and the SUM-of-steps / MAX-of-seconds all-reduce of sharding.aggregate_throughput.

Workload (BASELINE.json configs[1]): kinematic6 (6-state 3-D pos/vel, 3-D position observation), batch 65 536 per GPU,
fp64.  A step = ONE fused predict(dt) + update(kind) launch over the whole batch through the generated library's C ABI
({name}_batch_predict_update_{kind}); state round-trips HBM every step (the reference's per-call semantics).  Inputs (x, P,
the observation stream) are resident in HBM before the timed region.

roofline: HBM-bound.  Algorithmic bytes per filter-step actually moved with a shared R and scalar dt: reads x(6)+P(36)+z(3),
writes x(6)+P(36)+y(3) = 90 doubles = 720 B (SURVEY.md 8d quotes 800 B when dt and R are per-filter arrays; bytes that are
not moved are not counted).  achieved = 720 B x batch / mean launch duration measured with HIP events on the launch stream
over the timed region.  `traffic` comes from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
profiles/collect.sh, corrected as MI355X_MICROARCH.md prescribes) and is reported only while the digest recorded there is the
digest of the library being timed -- a number measured on another build is not printed.

"extra" (1 GPU only) reports the other BASELINE configs, each with its own roofline object: live step-granular stream
(config 3) and its dt > 0 launch alone, the 2-state kinematic model step-granular and fused (K steps per launch, the default
fast path for tiny models), the fused run of kinematic6 (bound by fp64 VALU issue: the fraction is of the vector fp64 issue
rate), 1 M filters, kinematic9, the MSCKF model, and config 4 at its stated size: live with the Mahalanobis gate,
16 384 filters x 2 100 steps, forward pass keeping the filtered trace + RTS backward pass, swept in batch chunks -- with
`roofline_backward_dt_gt0` (the smoother on a chunk whose steps all advance time: no step takes the identity-gain path) and
`packed_trace` (the same sweep with the opt-in packed-triangle trace between the passes) next to the stream's two objects.
Every HBM-bound object carries `frac` (HIP events) and, where a host clock brackets the same launches, `frac_wall`.
`kinematic6_ring`: the headline step with the reference's rewind ring on (one fused step + checkpoint launch per call).
`scalar_abi`: microseconds per predict + update of ONE filter through the reference's scalar host-pointer entry points (a latency, not part of any rate).
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "oracle")):
  if p not in sys.path:
    sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEADY_GROUP = 50              # launches per group of the steady-state warm-up (run_model)
STEADY_MIN_S = 0.1             # ... its minimum duration (clocks keep rising for milliseconds after the first launches)
STEADY_CAP_S = 0.5             # ... and its time limit
MEDIAN_GROUP = 10              # fewest launches between two markers of the timed region (launch-duration distribution)
COLD_LAUNCHES = 20             # launches of the cold sample (run_model): the first ones after >= 100 ms of idle
FP64_VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9     # 256 CUs x 4 SIMDs x 16 fp64 lanes per clock x 2.4 GHz = 39.3 T lane-instructions/s
#                                               (78.6 TFLOP/s vector fp64 counts an FMA as two)


def hbm_roofline(bytes_per_launch, launch_s, kernel, traffic=None, wall_s=None, traffic_source=None, **more):
  """`frac` prices the algorithmic bytes against the launch duration the HIP events measured; `frac_wall` (when `wall_s`, the host's
  wall clock per step over the same region, is given) against what a caller's clock sees -- event interval + whatever the queue left idle.
  `traffic` is a PMC counter figure per launch; `traffic_source` says where it was read from (it is not re-measured in this run)."""
  a = bytes_per_launch / launch_s / 1e9
  out = dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS, traffic=traffic, kernel=kernel,
             algorithmic_bytes_per_launch=bytes_per_launch, launch_us=launch_s * 1e6, **more)
  if traffic is not None:
    out["traffic_source"] = traffic_source
  if wall_s is not None:
    out["frac_wall"] = bytes_per_launch / wall_s / 1e9 / HBM_PEAK_GBS
    out["wall_us"] = wall_s * 1e6
  return out


TRAFFIC_CARRIED = {}      # section label -> note, for counters taken on an earlier build of the library that is being timed
TRAFFIC_SOURCE = {}       # section label -> where its counter figure came from (file, section, digest of the library it was taken on)


def measured_traffic(label, lib_name, gen):
  """HBM bytes per launch of the section `label` of profiles/pmc_workload.py from the committed PMC passes -- only if they were
  taken on the library that is being timed (digest of generated/{lib_name}.digest), or on an earlier build that the record lists
  under `carried_to` together with what changed since (`carried_note`): the JSON line then says so under `traffic_carried`."""
  tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
  if not os.path.exists(tf):
    return None
  with open(tf, encoding="utf-8") as fh:
    rec = json.load(fh).get(label)
  dg = os.path.join(gen, f"{lib_name}.digest")
  if rec is None or not os.path.exists(dg):
    return None
  with open(dg, encoding="utf-8") as fh:
    now = fh.read().strip()
  if rec.get("lib_digest") != now:
    if now not in rec.get("carried_to", []):
      return None
    TRAFFIC_CARRIED[label] = f"counters taken on build {rec['lib_digest'][:12]} of lib{lib_name}.so; since then: {rec.get('carried_note', '?')}"
  TRAFFIC_SOURCE[label] = (f"profiles/pmc_traffic.json[{label}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/collect.sh on "
                           f"lib{lib_name}.so digest {rec['lib_digest'][:12]} (an earlier GPU call; not re-measured in this run)")
  return rec["hbm_bytes_per_launch"]


def traffic_kw(label, lib_name, gen, scale=1.0):
  """`traffic` / `traffic_source` keyword arguments of hbm_roofline for one section of profiles/pmc_traffic.json."""
  per = measured_traffic(label, lib_name, gen)
  return dict(traffic=None) if per is None else dict(traffic=per * scale, traffic_source=TRAFFIC_SOURCE.get(label))


def fp64_valu_instructions(lib, kernel="k_run", unroll=1):
  """fp64 VALU instructions per filter-step of the fused run of a single-kind lane-per-filter model, from the disassembly
  (llvm-objdump) of the kernel an untraced batch_run launches: every v_*_f64 instruction of the kernel divided by the number of
  step bodies it contains.  The step loop is unrolled (`unroll` = {name}_run_unroll()) and hipcc may peel or unswitch it on top
  of that, so the bodies are counted: a step body holds as many v_rcp_f64 (one per pivot of S) as the model's step kernel
  k_step_*<true>, which holds exactly one body."""
  objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
  if not os.path.exists(objdump):
    return None
  llvm = "/opt/rocm/lib/llvm/bin"
  try:
    fb, co = f"/tmp/rn_bench_{os.getpid()}.hipfb", f"/tmp/rn_bench_{os.getpid()}.co"
    subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True, capture_output=True)
    subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    os.unlink(co)
    os.unlink(fb)
  except Exception:      # pylint: disable=broad-except
    return None
  f64, rcp, cur = {}, {}, None
  for line in dis.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
      cur = m.group(1)
      continue
    if cur is not None and re.search(r"\bv_\w+_f64", line):
      f64[cur] = f64.get(cur, 0) + 1
      if "v_rcp_f64" in line:
        rcp[cur] = rcp.get(cur, 0) + 1
  # an untraced run of a lane-per-filter model launches k_run_blk when the library has it (emit_small.run_kernel_blk)
  want = "k_run_blk" if kernel == "k_run" and any("k_run_blk" in nm for nm in f64) else kernel
  mine = [nm for nm in f64 if want in nm and (want != "k_run" or "k_run_blk" not in nm)]
  fp64_valu_instructions.kernel = want
  if not mine:
    return None
  nm = max(mine, key=lambda k_: f64[k_])
  step = [k_ for k_ in f64 if "k_step_" in k_ and "ILb1E" in k_]          # k_step_<kind><true>
  per_body = min((rcp.get(k_, 0) for k_ in step), default=0)
  bodies = rcp.get(nm, 0) // per_body if per_body and rcp.get(nm, 0) % per_body == 0 and rcp.get(nm, 0) else max(1, unroll)
  return f64[nm] / max(1, bodies)


def cpu_baseline(name, kind, K6, batch, budget_s=5.0, suffix="", cflags=None):
  """Oracle (C restatement of ekf_c.c + reference-generated sympy C, gcc flags of the reference's SConstruct unless `cflags`)
  timed on the host cores of this box: 1 thread (the reference's execution model) and all cores."""
  from oracle_lib import OracleLib
  lib = OracleLib(name, suffix=suffix, cflags=cflags)
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
  flav = "reference-generated sympy C + C restatement of ekf_c.c" if lib.flavour == "ref" else "our sympy C99 + C restatement of ekf_c.c"
  return out, n, flav, ("reference" if lib.flavour == "ref" else "port")


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


def live_true_accel(M, gen, name="live"):
  """Expected specific force at the true (initial) state through the library's own h_10 (GPU, batch of one)."""
  from rednose_amd.helpers.ekf_sym import EKF_sym
  s = EKF_sym(gen, name, M.Q, M.initial_x, np.diag(M.initial_P_diag), 23, 22)
  hacc = np.zeros(3)
  s.hs[10](np.ascontiguousarray(M.initial_x.copy()), np.zeros(1), hacc)
  return hacc


def live_x0(torch, M, n, dev, gdev):
  e = (torch.rand((n, 3), generator=gdev, dtype=torch.float64, device=dev) - 0.5) * 0.1
  x0 = torch.as_tensor(M.initial_x, dtype=torch.float64, device=dev).repeat(n, 1)
  q = torch.cat([torch.ones((n, 1), dtype=torch.float64, device=dev), 0.5 * e], dim=1)
  x0[:, 3:7] = q / q.norm(dim=1, keepdim=True)
  return x0


def live_schedule(total):
  """per 10 ms tick a PHONE_GYRO(4) then a PHONE_ACCEL(10) observation at the same time (second has
  dt = 0), every 10th tick an ECEF_POS(12): 2 100 steps per 10 s."""
  kinds, ts, tick = [], [], 0
  while len(kinds) < total:
    t = 0.01 * tick
    kinds += [4, 10]
    ts += [t, t]
    if tick % 10 == 9:
      kinds.append(12)
      ts.append(t)
    tick += 1
  return np.array(kinds[:total], dtype=np.int32), np.array(ts[:total])


def live_observations(torch, M, hacc, kinds, n, dev, gdev, outlier_frac=0.0):
  """(T, n, 3) observations of a stationary device: gyro N(0, 0.025^2), accel h_10(x_true) + N(0, 0.5^2), position
  pos_true + N(0, 5^2); `outlier_frac` of the position fixes replaced by pos_true + N(0, 500^2) (config 4)."""
  T = len(kinds)
  zs = torch.randn((T, n, 3), generator=gdev, dtype=torch.float64, device=dev)
  kd = torch.as_tensor(kinds, device=dev)
  scale = torch.where(kd == 4, 0.025, torch.where(kd == 10, 0.5, 5.0)).to(torch.float64)
  mean = torch.zeros((T, 3), dtype=torch.float64, device=dev)
  mean[kd == 10] = torch.as_tensor(hacc, device=dev)
  mean[kd == 12] = torch.as_tensor(M.initial_x[:3], device=dev)
  zs = zs * scale[:, None, None] + mean[:, None, :]
  if outlier_frac > 0:
    gi = torch.nonzero(kd == 12).flatten()
    sel = torch.rand((len(gi), n), generator=gdev, device=dev) < outlier_frac
    zs[gi] += sel[..., None] * 500.0 * torch.randn((len(gi), n, 3), generator=gdev, dtype=torch.float64, device=dev)
  return zs


def live_stream(torch, M, gen, n, total, dev, rank):
  gdev = torch.Generator(device=dev).manual_seed(2025 + rank)
  hacc = live_true_accel(M, gen)
  x0 = live_x0(torch, M, n, dev, gdev)
  kinds, ts = live_schedule(total)
  zs = live_observations(torch, M, hacc, kinds, n, dev, gdev)
  return x0, np.diag(M.initial_P_diag), [(int(kinds[i]), float(ts[i]), zs[i]) for i in range(total)]


def model_class(model):
  if model == "kinematic6":
    from examples.kinematic6_kf import Kinematic6Kalman as M
  elif model == "kinematic":
    from examples.kinematic_kf import KinematicKalman as M
  elif model == "kinematic9":
    from examples.kinematic9_kf import Kinematic9Kalman as M
  else:
    from examples.live_kf import LiveKalman as M
  return M


def gen_dir(names):
  if os.environ.get("RN_NO_GEN") and "RN_GEN_DIR" in os.environ:      # A/B against a library built from an OLDER emitter: use it as it is
    return os.path.abspath(os.environ["RN_GEN_DIR"])
  from examples import ensure_generated
  return ensure_generated(names, **({'folder': os.environ['RN_GEN_DIR']} if 'RN_GEN_DIR' in os.environ else {}))


def run_model(torch, dist, model, n, K, W, dev, rank, world, only_kind=None):
  """Warm up W steps, time exactly K steps (barrier + synchronize on both sides).  Returns a timing dict; `wall` and `dev_ms`
  are this rank's (the aggregation over ranks is the caller's: sharding.aggregate_throughput)."""
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  M = model_class(model)
  if rank == 0:
    gen_dir([model])
  if world > 1:
    dist.barrier()
  gen = gen_dir([model])
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  quat = list(getattr(M, "quaternion_idxs", []))
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev, quaternion_idxs=quat)
  total = W + K
  if model == "live":
    x0, P0, sched = live_stream(torch, M, gen, n, total, dev, rank)
    if only_kind is not None:      # one kind, every launch advancing time (the launch that runs the covariance predict)
      pool = [s_ for s_ in sched if s_[0] == only_kind]
      sched = [(only_kind, 0.01 * i, pool[i % len(pool)][2].clone()) for i in range(total)]
  else:
    x0, P0, sched = kinematic_stream(torch, M, n, total, dev, rank)
  f.init_state(x0, P0, None)
  Rs = {k: np.atleast_2d(v) for k, v in M.obs_noise.items()}

  # the timed loop calls the library's C entry point through pre-bound arguments (what a C/C++ caller does);
  # BatchedEKF.predict_and_update_batch adds several microseconds of Python argument handling per call
  bound = {k: f.bind_step(k, Rs[k]) for k in sorted(set(s_[0] for s_ in sched))}
  sched = [(k, t, z.clone()) for (k, t, z) in sched]       # own allocation per step: the C ABI wants 16-byte aligned observations, and row
  #                                                          i of a (T, n, Z) tensor is not when n * Z is odd (uneven shards)
  t_prev = [None]

  def step(i):
    kind, t, z = sched[i]
    dt = 0.0 if t_prev[0] is None else t - t_prev[0]
    t_prev[0] = t
    bound[kind](z, dt)

  # Steady state before the timed region, whatever --warmup says: a short run (--steps 20 --warmup 5 is 0.25 ms of GPU work) is
  # otherwise timed on a device that is still raising its clocks (round 3: 10.7 us per launch in such a run against 9.05 us in a
  # long one, same kernel).  Extra launches of the same entry point on this filter, in groups of STEADY_GROUP with HIP events
  # around each group, for at least STEADY_MIN_S seconds and until two consecutive groups agree within 2 % -- at most STEADY_CAP_S.
  # They come FIRST and start from the initial state every time; then the initial state is put back and the --warmup steps of the
  # schedule run, so what precedes the timed region is W step launches and the filter is exactly where the schedule expects it
  # (the nonlinear live filter does not survive being moved half a second ahead of its observations).  Not part of `steps`.
  steady = dict(launches=0, seconds=0.0, group=STEADY_GROUP, converged=False, last_group_us=None)
  # observations of the untimed launches: pristine copies of the first rows of the schedule, restored into scratch buffers BEFORE each
  # group's first event -- a launch overwrites z with the residual y, and a filter fed its own residuals as observations drifts away
  pool0 = [(k, z.clone()) for (k, _, z) in (sched[:W] if W else sched[:1])]
  scratch = {}       # observation shape -> scratch buffers

  def untimed_group(first, count):
    # every group starts from the initial state: thousands of extra launches on repeated observations (each with dt = 0.01) are
    # not a trajectory any filter was tuned for -- live drifts to non-finite states after a few thousand
    f.x.copy_(x_keep)
    f.P.copy_(P_keep)
    used, plan = {}, []
    for j in range(count):
      k_, z0_ = pool0[(first + j) % len(pool0)]
      bufs = scratch.setdefault(tuple(z0_.shape), [])
      i = used.get(tuple(z0_.shape), 0)
      used[tuple(z0_.shape)] = i + 1
      if i == len(bufs):
        bufs.append(torch.empty_like(z0_))
      bufs[i].copy_(z0_)
      plan.append((k_, bufs[i]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k_, buf in plan:
      bound[k_](buf, 0.01)
    b.record()
    return a, b

  x_keep, P_keep = f.x.clone(), f.P.clone()
  # The COLD figure, reported next to the steady one (top-level "cold_launch_us"): the mean of the first COLD_LAUNCHES launches after the
  # device has idled >= 100 ms -- what a caller that steps a batch once in a while sees (empty queue, clocks down): host -> doorbell ->
  # command processor -> dispatch is exposed until the host is ahead (profiles/r4_short_run_per_launch.txt: 35.8 / 19.7 / 18.4 us, then flat).
  torch.cuda.synchronize()
  time.sleep(0.12)
  a_, b_ = untimed_group(0, COLD_LAUNCHES)
  torch.cuda.synchronize()
  cold_us = a_.elapsed_time(b_) * 1e3 / COLD_LAUNCHES
  f.x.copy_(x_keep)
  f.P.copy_(P_keep)
  if os.environ.get("RN_BENCH_NO_STEADY") != "1":
    t_warm = time.perf_counter()
    prev = None
    while time.perf_counter() - t_warm < STEADY_CAP_S:
      a, b = untimed_group(steady["launches"], STEADY_GROUP)
      torch.cuda.synchronize()
      cur = a.elapsed_time(b) * 1e3 / STEADY_GROUP
      steady["launches"] += STEADY_GROUP
      steady["last_group_us"] = cur
      if prev is not None and abs(cur - prev) <= 0.02 * prev and time.perf_counter() - t_warm >= STEADY_MIN_S:
        steady["converged"] = True
        break
      prev = cur
    steady["seconds"] = time.perf_counter() - t_warm
    f.x.copy_(x_keep)
    f.P.copy_(P_keep)
  for i in range(W):
    step(i)
  G = int(os.environ.get("RN_BENCH_MARK_EVERY", 0)) or max(MEDIAN_GROUP, -(-K // 50))
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  spare = [torch.cuda.Event(enable_timing=True) for _ in range(K // G + 1)]      # created outside the timed region
  if dist is not None and dist.is_initialized():
    dist.barrier()
  torch.cuda.synchronize()
  # The timed region: exactly K steps between ev0 and ev1 (mean launch duration = their interval / K).  A marker every G launches
  # inside it (at most 50 of them: a marker is one queue packet, no kernel) gives the distribution: launch_us_median is the median
  # of the per-group means -- the same launches, the same observation buffers, nothing replayed.
  marks = [(W, ev0)]
  t0 = time.perf_counter()
  ev0.record()
  for i in range(W, W + K):
    step(i)
    if (i + 1 - W) % G == 0 and i + 1 < W + K:
      m_ = spare.pop()
      m_.record()
      marks.append((i + 1, m_))
  ev1.record()
  marks.append((W + K, ev1))
  torch.cuda.synchronize()
  if dist is not None and dist.is_initialized():
    dist.barrier()
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  dev_ms = ev0.elapsed_time(ev1)
  groups = [a[1].elapsed_time(b[1]) * 1e3 / (b[0] - a[0]) for a, b in zip(marks[:-1], marks[1:])]
  assert torch.isfinite(f.x).all() and torch.isfinite(f.P).all(), "filter diverged: refusing to report a timing"
  zdims = [f.zdims[sched[i][0]] for i in range(W, W + K)]
  bytes_per_step = 8.0 * (2 * (D + E * E) + 2 * float(np.mean(zdims)))
  return dict(M=M, D=D, E=E, Z=float(np.mean(zdims)), wall=wall, dev_ms=dev_ms, bytes_per_step=bytes_per_step, gen=gen,
              kinds=sorted(set(s[0] for s in sched[W:W + K])), steady=steady, cold_launch_us=cold_us,
              launch_us_median=float(np.median(groups)) if groups else None, launch_us_groups=len(groups),
              group_us=[round(g_, 3) for g_ in groups] if os.environ.get("RN_BENCH_MARK_EVERY") else None)


def scalar_abi_extra(gen):
  """The drop-in boundary for ONE filter: the reference's scalar host-pointer ABI ({name}_predict + {name}_update_{kind}, what rednose's EKF_sym
  binds: rednose/helpers/ekf_sym.py:149-165) as a batch of one on the GPU -- host buffers in, host buffers out, PCIe and the launch / wait
  latency included.  Microseconds per predict + update; not a throughput figure."""
  import ctypes
  import time
  from examples.kinematic_kf import KinematicKalman as K
  from examples.live_kf import LiveKalman as L
  dp = ctypes.POINTER(ctypes.c_double)
  ptr = lambda a: a.ctypes.data_as(dp)      # noqa: E731
  out = {}
  for name, M, E, kind, Z in (("kinematic", K, 2, 1, 1), ("live", L, 22, 4, 3)):
    lib = ctypes.CDLL(os.path.join(gen, f"lib{name}.so"))
    pred, upd = getattr(lib, f"{name}_predict"), getattr(lib, f"{name}_update_{kind}")
    pred.argtypes, pred.restype, upd.argtypes, upd.restype = [dp, dp, dp, ctypes.c_double], None, [dp] * 5, None
    x0 = np.array(M.initial_x, dtype=np.float64)
    P0 = np.eye(E) if name == "kinematic" else np.diag(np.asarray(M.initial_P_diag, dtype=np.float64))
    Q, R = np.ascontiguousarray(M.Q, dtype=np.float64), np.ascontiguousarray(np.atleast_2d(M.obs_noise[kind]), dtype=np.float64)
    x, P, z = x0.copy(), P0.copy(), np.zeros(Z)

    def step():
      x[:] = x0; P[:] = P0; z[:] = 0.0
      pred(ptr(x), ptr(P), ptr(Q), 0.01)
      upd(ptr(x), ptr(P), ptr(z), ptr(R), None)
    for _ in range(100):
      step()
    t0 = time.perf_counter()
    for _ in range(500):
      step()
    us = (time.perf_counter() - t0) / 500 * 1e6
    if getattr(lib, f"{name}_last_error")() != 0:
      raise RuntimeError(f"{name}: scalar ABI call failed")
    out[name] = {"predict_plus_update_us": round(us, 2), "kind": kind}
  out["note"] = ("one filter through the reference's scalar host-pointer entry points (two calls: predict, update), host buffers in / out, "
                 "launch + wait + PCIe included; the arguments are packed into a pinned host buffer the kernels work on in place")
  return out


def ring_extra(torch, gen, n, K, dev):
  """The headline step with the reference's rewind ring ON (EKFSym keeps a checkpoint of every call, ekf_sym.cc:142-156,191; BatchedEKF(rewind_to_keep=..)):
  BatchedEKF.predict_and_update_batch on kinematic6, one launch per call -- the fused step that writes its own checkpoint
  ({name}_batch_predict_update_{kind}_ckpt).  Algorithmic bytes per filter-step: the plain step's 720 B + the checkpoint's x, P and observation written (360 B)."""
  import time
  from examples.kinematic6_kf import Kinematic6Kalman as K6
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  f = BatchedEKF(gen, "kinematic6", K6.Q, K6.initial_x, np.diag(K6.initial_P_diag), 6, 6, batch=n, device=dev, rewind_to_keep=8)
  R = np.ascontiguousarray(K6.obs_noise[1], dtype=np.float64)
  zs = torch.randn((64, n, 3), dtype=torch.float64, device=dev)      # a pool of observation buffers (the kernel overwrites its own with the residuals)
  t = 0.0
  for i in range(200):
    t += 0.01
    f.predict_and_update_batch(t, 1, zs[i % 64], R)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  for i in range(K):
    t += 0.01
    f.predict_and_update_batch(t, 1, zs[i % 64], R)
  e1.record()
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) / K
  per = 8.0 * (2 * (6 + 36 + 3) + (6 + 36 + 3))
  return {"batch": n, "steps": K, "value": n / wall, "unit": "steps/s", "rewind_to_keep": 8, "algorithmic_bytes_per_filter_step": per,
          "roofline": hbm_roofline(per * n, e0.elapsed_time(e1) * 1e-3 / K, "k_stepc_1<true>", wall_s=wall, **traffic_kw(f"kinematic6_ring_b{n}", "kinematic6", gen)),
          "note": "the headline step through BatchedEKF.predict_and_update_batch with the rewind ring on: one launch per call (step + checkpoint); "
                  "the same calls as step + three copies took 26 us before the checkpointing kernel existed"}


def fused_run_extra(torch, model, n, T, dev):
  """{name}_batch_run: x and P stay in registers for T steps, only z / y cross HBM.  Bound by fp64 VALU issue."""
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  M = model_class(model)
  gen = gen_dir([model])
  D, E = M.initial_x.shape[0], M.initial_P_diag.shape[0]
  f = BatchedEKF(gen, M.name, M.Q, M.initial_x, np.diag(M.initial_P_diag), D, E, batch=n, device=dev)
  Z = int(np.atleast_2d(M.obs_noise[1]).shape[0])
  zs = torch.randn((T, n, Z), dtype=torch.float64, device=dev) * 0.1
  # The schedule (kinds, dts, R per step) is staged in HBM BEFORE the timed region, like the observations: the timed interval is
  # the library call alone.  (Through BatchedEKF.run the interval also held the host-side staging of the schedule -- a Python
  # loop over T steps and three small uploads, 1.5 ms for T = 2 000 -- during which the GPU idled: rounds 1-3 reported the
  # 2-state run at 54 G steps/s when the kernel itself sustained 138 G.)
  kd = torch.ones(T, dtype=torch.int32, device=dev)
  dd = torch.full((T,), 0.01, dtype=torch.float64, device=dev)
  Rd = torch.from_numpy(np.tile(np.atleast_2d(M.obs_noise[1]).reshape(1, Z * Z), (T, 1))).to(dev)
  best = None
  for rep in range(4):          # the first repetition warms up (module load, first touch of the buffers)
    zc = zs.clone()
    f.init_state(M.initial_x, np.diag(M.initial_P_diag), 0.0)
    torch.cuda.synchronize()
    # An untimed launch of the same entry point (16 steps) goes into the queue right in front of the first event: the timed launch then starts on a
    # device that is awake -- a single 0.5 ms launch on a device that idled through init_state + synchronize carries 40-70 us of wake-up
    # (top-level `cold_launch_us`), which is a property of the idle device, not of the kernel (rocprofv3 reads the kernel's own duration:
    # profiles/r6_sections_kernel_trace.txt).  Its 16 steps are not counted.
    zw = zs[:16].clone()
    f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), min(16, T), f._p(zw), f._p(Rd), n, f.norm_quats,      # pylint: disable=protected-access
            None, None, None, None, None, f._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f._call("batch_run", f._p(f.x), f._p(f.P), f._p(f.Q), f._p(kd), f._p(dd), T, f._p(zc), f._p(Rd), n, f.norm_quats,      # pylint: disable=protected-access
            None, None, None, None, None, f._stream())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    if rep:
      best = ms if best is None or ms < best else best
  assert torch.isfinite(f.x).all()
  moved = 8.0 * (2 * Z) * n * T + 8.0 * 2 * (D + E * E) * n
  rate = n * T / (best * 1e-3)
  insts = fp64_valu_instructions(os.path.join(gen, f"lib{M.name}.so"), unroll=getattr(f._lib, f"{M.name}_run_unroll")())      # pylint: disable=protected-access
  kname = getattr(fp64_valu_instructions, "kernel", "k_run")
  roof = {"bound": "fp64-valu", "achieved": None, "peak": FP64_VALU_LANE_OPS / 1e12, "unit": "T fp64 lane-instructions/s", "frac": None,
          "fp64_valu_instructions_per_filter_step": insts, "hbm_GBs": moved / (best * 1e-3) / 1e9, "hbm_frac": moved / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "kernel": kname, "traffic": measured_traffic(f"{model}_fused_b{n}", M.name, gen)}
  if insts:
    roof["achieved"] = insts * rate / 1e12
    roof["frac"] = insts * rate / FP64_VALU_LANE_OPS
    roof["fp64_frac"] = roof["frac"]
    if roof["hbm_frac"] > roof["frac"]:          # the 2-state model: 16 B per filter-step against ~30 fp64 instructions
      roof.update(bound="hbm", achieved=roof["hbm_GBs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=roof["hbm_frac"])
  return {"model": M.name, "batch": n, "T": T, "value": rate, "unit": "steps/s", "ms": best, "hbm_bytes_moved": moved, "roofline": roof,
          "note": "state resident in VGPRs for T steps (x / P cross HBM once per launch, z / y once per step); one timed launch behind an untimed 16-step launch of "
                  "the same entry point (device awake); priced against fp64 VALU issue "
                  "(every v_*_f64 instruction of the kernel, llvm-objdump, per step of its unrolled loop) and against the HBM "
                  "bytes of z / y; `bound` names the larger fraction"}


def config4_extra(torch, dev, rank, nb=16384, T=2100, chunk=8192):
  """BASELINE config 4 at its stated size: live with the Mahalanobis gate on ECEF_POS, 2 % GNSS outliers, forward pass keeping
  the filtered trace, RTS backward pass -- swept in chunks of `chunk` filters (the trace of a chunk is T x chunk x 4 056 B:
  70 GB at 8 192 -- 1 024 wavefronts of 8 filters, one on every SIMD; filters are independent, the result is that of one sweep).  The trace buffers are allocated once, outside
  the timed region; forward and backward times are the sums of per-chunk HIP-event intervals."""
  from examples.live_kf import LiveKalman as L
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  gen = gen_dir(["live_maha"])
  f = BatchedEKF(gen, "live_maha", L.Q, L.initial_x, np.diag(L.initial_P_diag), 23, 22, batch=nb, device=dev, quaternion_idxs=[3],
                 maha_test_kinds=[12])
  gdev = torch.Generator(device=dev).manual_seed(4242 + rank)
  hacc = live_true_accel(L, gen, "live_maha")
  x0 = live_x0(torch, L, nb, dev, gdev)
  kinds, ts = live_schedule(T)
  zs = live_observations(torch, L, hacc, kinds, nb, dev, gdev, outlier_frac=0.02)
  Rs = {int(k): np.atleast_2d(L.obs_noise[int(k)]) for k in set(kinds.tolist())}
  chunk = min(chunk, nb)
  tx = torch.empty((T, chunk, 23), dtype=torch.float64, device=dev)
  tP = torch.empty((T, chunk, 22, 22), dtype=torch.float64, device=dev)
  gi = torch.as_tensor(np.where(kinds == 12)[0], device=dev)
  res = None
  for rep in range(2):                      # first sweep warms up (library load, allocator); the second is reported
    f.init_state(x0, np.diag(L.initial_P_diag), None)
    fwd = bwd = 0.0
    gated = []
    finite = True
    for lo in range(0, nb, chunk):
      hi = min(nb, lo + chunk)
      bx, bP = (tx, tP) if hi - lo == chunk else (tx[:, :hi - lo].contiguous(), tP[:, :hi - lo].contiguous())
      zc = zs[:, lo:hi].contiguous()
      f.filter_time = None
      e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
      e0.record()
      _, _, _, fl = f.run(ts, kinds, zc, Rs, flags=True, out=(bx, bP), filters=(lo, hi))
      e1.record()
      xs, Ps = f.rts_smooth(bx, bP, ts, inplace=True) if hi - lo == nb else f._rts_on(bx, bP, ts, hi - lo, None)   # pylint: disable=protected-access
      e2.record()
      torch.cuda.synchronize()
      fwd += e0.elapsed_time(e1)
      bwd += e1.elapsed_time(e2)
      gated.append((fl[gi] & 1).float().mean().item())
      finite = finite and bool(torch.isfinite(xs[0]).all()) and bool(torch.isfinite(Ps[0]).all()) and bool(torch.isfinite(xs[-1]).all())
    assert finite, "config 4: non-finite smoothed estimate"
    res = dict(fwd_ms=fwd, bwd_ms=bwd, gated=float(np.mean(gated)))
  # The backward pass again on ONE chunk with every step advancing time (ts = 0.01 k): no step takes the identity-gain path of k_rts4
  # (dt == 0: Ck = I, emit_rts4.py), every step pays the factorisation, the substitutions and both products -- the cost of the full
  # step stays visible next to the stream's figure, in which 1.1 of every 2.1 steps have dt = 0.  The trace is the last chunk's
  # smoothed one (valid states and covariances; the pass runs in place on it).
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  f._rts_on(tx, tP, 0.01 * np.arange(T), chunk, None)      # pylint: disable=protected-access
  e1.record()
  torch.cuda.synchronize()
  full_ms = e0.elapsed_time(e1)
  assert bool(torch.isfinite(tP[0]).all()) and bool(torch.isfinite(tx[0]).all()), "config 4 (all steps dt > 0): non-finite smoothed estimate"
  del tP
  # The opt-in packed-triangle trace (batch_run_tri / batch_rts_tri: lower triangles, 253 instead of 484 doubles per covariance): the same sweep of
  # chunks, second sweep reported.  Same kernels, same arithmetic (the packed results are the full ones' lower triangles bit for bit: tests/test_gpu_tri.py).
  packed = None
  if f.has_tri_trace():
    tT = torch.empty((T, chunk, f.dim_tri), dtype=torch.float64, device=dev)
    for rep in range(2):
      f.init_state(x0, np.diag(L.initial_P_diag), None)
      pf = pb = 0.0
      for lo in range(0, nb, chunk):
        hi = min(nb, lo + chunk)
        bx, bT = (tx, tT) if hi - lo == chunk else (tx[:, :hi - lo].contiguous(), tT[:, :hi - lo].contiguous())
        zc = zs[:, lo:hi].contiguous()
        f.filter_time = None
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        f.run(ts, kinds, zc, Rs, flags=True, out=(bx, bT), filters=(lo, hi), packed=True)
        e1.record()
        f._rts_on(bx, bT, ts, hi - lo, None, packed=True)      # pylint: disable=protected-access
        e2.record()
        torch.cuda.synchronize()
        pf += e0.elapsed_time(e1)
        pb += e1.elapsed_time(e2)
      assert bool(torch.isfinite(bT[0]).all()) and bool(torch.isfinite(bx[0]).all())
      packed = dict(fwd_ms=pf, bwd_ms=pb)
    del tT
  del tx
  def chunk_traffic(label):
    """PMC traffic of the 8 192 x 2 100 chunk launch (profiles/pmc_workload.py), times the chunks swept here (sum over launches, like the bytes)."""
    return traffic_kw(label, "live_maha", gen, scale=nb / chunk) if (chunk == 8192 and T == 2100) else dict(traffic=None)

  try:          # the smoother kernel this library was built with (emit_rts4's k_rts4, or its fallback, the lane-group rn::k_rts_group)
    with open(os.path.join(gen, "live_maha.kernels.txt"), encoding="utf-8") as fh:
      kt = fh.read()
      rts_kernel = "k_rts4" if "k_rts4" in kt else "rn::k_rts_group"
      run_kernel = "k_run2" if "k_run2 " in kt else "k_run"
  except OSError:
    rts_kernel, run_kernel = "k_rts*", "k_run*"
  fwd_bytes = nb * T * 8.0 * ((23 + 484) + 2 * 3) + nb * T           # filtered trace written, z read, y written, flags
  bwd_bytes = nb * (T - 1) * 8.0 * 2 * (23 + 484)                     # filtered pair read, smoothed pair written
  # fp64 work of a FULL backward step, counted on the algorithm (E = 22): L D L^T E^3 / 6, two triangular solves with E right-hand sides E^3,
  # T = Ck D E^3, U = T Ck^T on the lower blocks E^3 / 2 -> 28.4 k FMAs per filter-step; a dt = 0 step of the stream takes the identity-gain
  # path (one subtraction and one addition per entry of P: not counted).  Priced against the vector fp64 issue rate like the fused runs.
  E_ = 22
  fma_full = E_ ** 3 * (1 / 6 + 1 + 1 + 1 / 2)
  dts = np.diff(ts)
  try:
    with open(os.path.join(gen, "live_maha.hip"), encoding="utf-8") as fh:
      id0 = "if (dt == 0.0 && !first)" in fh.read()      # this build's k_rts4 has the identity-gain path (tuning knob rts_dt0, emit_rts4.dt0_path)
  except OSError:
    id0 = False
  full_steps_frac = float(np.mean(dts[:-1] != 0.0)) if id0 else 1.0      # (the newest step always takes the full path)
  bwd_rate = nb * (T - 1) / (res["bwd_ms"] * 1e-3)
  full_rate = chunk * (T - 1) / (full_ms * 1e-3)
  rb = hbm_roofline(bwd_bytes, res["bwd_ms"] * 1e-3, rts_kernel, **chunk_traffic("config4_backward"))
  rb.update(fp64_fma_per_full_step=fma_full, full_steps_fraction=full_steps_frac, fp64_frac=fma_full * full_steps_frac * bwd_rate / FP64_VALU_LANE_OPS,
            fp64_note="algorithmic FMAs of the steps that take the full path (dt != 0) x steps/s / 39.3 T fp64 lane-instructions/s; the kernel issues ~1.5 x that "
                      "(11 of 16 DPP lanes busy, scalar phase on one lane per filter)")
  rfull = hbm_roofline(chunk * (T - 1) * 8.0 * 2 * (23 + 484), full_ms * 1e-3, rts_kernel + " (every step dt > 0: full solve on all steps)",
                       **(traffic_kw("config4_backward_dt_gt0", "live_maha", gen) if (chunk == 8192 and T == 2100) else dict(traffic=None)))
  rfull.update(fp64_frac=fma_full * full_rate / FP64_VALU_LANE_OPS, steps_per_s=full_rate, chunk_filters=chunk)
  return {"batch": nb, "T": T, "chunk_filters": chunk,
          "forward_steps_per_s": nb * T / (res["fwd_ms"] * 1e-3), "backward_steps_per_s": nb * (T - 1) / (res["bwd_ms"] * 1e-3),
          "combined_steps_per_s": nb * T / ((res["fwd_ms"] + res["bwd_ms"]) * 1e-3),
          "forward_ms": res["fwd_ms"], "backward_ms": res["bwd_ms"], "gated_fraction_of_gnss": res["gated"],
          "trace_bytes_per_chunk": int(T * chunk * (23 + 484) * 8),
          "roofline_forward": hbm_roofline(fwd_bytes, res["fwd_ms"] * 1e-3, f"{run_kernel} (trace + gate flags)", **chunk_traffic("config4_forward")),
          "roofline_backward": rb,
          "roofline_backward_dt_gt0": rfull,
          **({"packed_trace": {
            "forward_steps_per_s": nb * T / (packed["fwd_ms"] * 1e-3), "backward_steps_per_s": nb * (T - 1) / (packed["bwd_ms"] * 1e-3),
            "combined_steps_per_s": nb * T / ((packed["fwd_ms"] + packed["bwd_ms"]) * 1e-3), "forward_ms": packed["fwd_ms"], "backward_ms": packed["bwd_ms"],
            "roofline_forward": hbm_roofline(nb * T * 8.0 * ((23 + 253) + 2 * 3) + nb * T, packed["fwd_ms"] * 1e-3, "k_run2_tri (packed trace + gate flags)", traffic=None),
            "roofline_backward": hbm_roofline(nb * (T - 1) * 8.0 * 2 * (23 + 253), packed["bwd_ms"] * 1e-3, "k_rts4_tri", traffic=None),
            "note": "opt-in record layout (run / rts_smooth / smooth(packed=True)): lower triangles only, 2 256 B forward and 4 416 B backward per filter-step "
                    "actually moved -- the fractions are priced on THOSE bytes, the steps/s compare directly with the full layout above"}} if packed else {}),
          "note": "forward = fused batch_run writing the filtered trace + gate flags; backward = batch_rts recomputing the predicted pairs; "
                  "bytes: forward 4 104 B + 1 flag per filter-step, backward 8 112 B per filter-step (both paths of k_rts4 read the filtered pair and "
                  "write the smoothed pair); roofline_backward_dt_gt0 = the same kernel on one chunk whose steps all advance time"}


def msckf_extra(torch, dev):
  from examples.feature_kf import WideFeatureKalman as FK
  from rednose_amd.helpers.ekf_sym import BatchedEKF
  genf = gen_dir(["feature36"])
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
  return {"batch": nf, "steps": Kf, "value": nf / (msf * 1e-3), "unit": "steps/s",
          "roofline": hbm_roofline(bf * nf, msf * 1e-3, "k_step_2<true>", **traffic_kw(f"feature36_b{nf}", FK.name, genf)),
          "note": "fused predict + feature-track update (Z = 6 projected to 3), one filter per wavefront, per-filter landmarks"}


def _free_port():
  import socket
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def visible_gpus():
  """Device count without importing torch into this (launcher) process: a forked HIP context must not leak into the ranks."""
  res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"],
                       capture_output=True, text=True)
  try:
    return int(res.stdout.strip().split()[-1])
  except (ValueError, IndexError):
    return 0


def rank_command(n, argv):
  """The command line the driver itself uses for N > 1: one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
  return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv, capture=False):
  """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run with N ranks (RCCL).  Fails
  loudly when the node has fewer than N GPUs (RN_BENCH_BACKEND=gloo, the functional dry run of tests/test_sharding.py, may
  oversubscribe one GPU; it is never a reported number)."""
  have = visible_gpus()
  if have < n and os.environ.get("RN_BENCH_BACKEND", "nccl") == "nccl":
    raise SystemExit(f"bench.py: --gpus {n} asked, {have} GPU(s) visible on this node: not run")
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
  env.pop("WORLD_SIZE", None)
  res = subprocess.run(rank_command(n, argv), env=env, cwd=REPO, capture_output=capture, text=True)
  if capture:
    return res
  raise SystemExit(res.returncode)


def sweep(args):
  """--sweep 1,2,4,8: BASELINE.json configs[4] in one invocation.  Weak points keep `--batch` filters per GPU, strong points keep
  the total of the largest count (524 288 at 8 x 65 536).  A count the node cannot run is reported as "not run" -- never
  extrapolated.  Efficiency is not computed here (the driver does that from the per-count values)."""
  counts = sorted({int(c) for c in args.sweep.split(",") if c.strip()})
  have = visible_gpus()
  oversub = os.environ.get("RN_BENCH_BACKEND", "nccl") != "nccl"
  per_gpu = args.batch or (16384 if args.model == "live" else 65536)
  total = args.global_batch or per_gpu * max(counts)
  base = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--model", args.model, "--no-extras", "--no-cpu-baseline"]
  points = {"weak": [], "strong": []}
  for mode, size in (("weak", ["--batch", str(per_gpu)]), ("strong", ["--global-batch", str(total)])):
    for g in counts:
      if g > have and not oversub:
        points[mode].append({"n_gpus": g, "status": "not run", "reason": f"{have} GPU(s) visible"})
        continue
      argv = ["--gpus", str(g)] + base + size
      if g == 1:
        res = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, cwd=REPO,
                             env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
      else:
        res = spawn_ranks(g, argv, capture=True)
      lines = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
      if res.returncode != 0 or len(lines) != 1:
        points[mode].append({"n_gpus": g, "status": "failed", "rc": res.returncode, "stderr_tail": res.stderr[-400:]})
        continue
      o = json.loads(lines[0])
      points[mode].append({"n_gpus": o["n_gpus"], "status": "ok", "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                           "global_batch": o["config"]["global_batch"], "batch_rank0": o["config"]["batch_per_gpu"],
                           "roofline_frac_rank_max": o["roofline"]["frac"]})
  ok = [p_ for p_ in points["weak"] if p_["status"] == "ok"]
  head = ok[-1] if ok else {}
  print(json.dumps({"metric": "EKF predict+update steps/sec at batch N", "value": head.get("value"), "unit": "steps/s",
                    "n_gpus": head.get("n_gpus"), "steps": args.steps, "warmup": args.warmup, "ms_per_step": head.get("ms_per_step"),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": f"{args.model} sweep over GPU counts {counts}: weak = {per_gpu} filters per GPU, strong = {total} filters in total",
                               "parallelism": "batch-sharded, no data-path collective (RCCL only for the barrier and the SUM/MAX of the timing)"},
                    "sweep": points}))
  return 0


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--warmup", type=int, default=100)
  ap.add_argument("--batch", type=int, default=None, help="filters per GPU (default 65536; 16384 for --model live): weak scaling")
  ap.add_argument("--global-batch", type=int, default=None, help="total filters over all GPUs (strong scaling): rank r owns shard_range(G, r, N)")
  ap.add_argument("--model", default="kinematic6", choices=["kinematic6", "kinematic", "kinematic9", "live"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-extras", action="store_true", help="skip the additional configs reported under 'extra'")
  ap.add_argument("--sweep", default=None, metavar="1,2,4,8",
                  help="run every listed GPU count in this one invocation (weak: --batch per GPU; strong: --global-batch or "
                       "batch x the largest count) and print ONE JSON line holding all points; counts this node cannot run are marked 'not run'")
  args = ap.parse_args()

  if args.sweep:
    return sweep(args)
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    return spawn_ranks(args.gpus, sys.argv[1:])

  import torch
  import torch.distributed as dist
  from rednose_amd.helpers import sharding

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a line "
                     "whose n_gpus differs from what was asked for")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a HIP device: rednose_amd has no CPU path")
  ndev = torch.cuda.device_count()
  if world > ndev and os.environ.get("RN_BENCH_BACKEND", "nccl") == "nccl":
    raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible; one rank per GPU over RCCL is the only measured mode")
  dev_index = local_rank % ndev                    # one rank per GPU; the modulo only matters for single-GPU gloo dry runs
  torch.cuda.set_device(dev_index)
  dev = torch.device(f"cuda:{dev_index}")
  # RN_BENCH_FORCE_DIST=1 runs the N = 1 line through the SAME process-group code as N > 1 (RCCL communicator of one rank, barrier,
  # device-tensor all-reduce): it says nothing about scaling, it proves the branch the 8-GPU run takes loads librccl and reduces.
  force_dist = world == 1 and os.environ.get("RN_BENCH_FORCE_DIST") == "1"
  use_dist = world > 1 or force_dist
  if use_dist:
    # "nccl" is RCCL on ROCm.  RN_BENCH_BACKEND=gloo allows a functional dry run of this path with several ranks on ONE
    # GPU (RCCL refuses duplicate devices); it is never used for reported numbers.
    backend = os.environ.get("RN_BENCH_BACKEND", "nccl")
    kw = dict(init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1) if force_dist else {}
    if backend == "nccl":
      dist.init_process_group(backend="nccl", device_id=dev, **kw)
    else:
      dist.init_process_group(backend=backend, **kw)

  K, W = args.steps, args.warmup
  if args.global_batch:
    lo, hi = sharding.shard_range(args.global_batch, rank, world)
    n, scaling = hi - lo, "strong"
  else:
    n, scaling = args.batch or (16384 if args.model == "live" else 65536), "weak"
  r = run_model(torch, dist if use_dist else None, args.model, n, K, W, dev, rank, world)
  M, D, E = r["M"], r["D"], r["E"]
  agg_dev = dev if (use_dist and os.environ.get("RN_BENCH_BACKEND", "nccl") == "nccl") else None
  value, steps_total, wall_max = sharding.aggregate_throughput(n * K, r["wall"], dist if use_dist else None, device=agg_dev)
  _, _, dev_ms_max = sharding.aggregate_throughput(1.0, r["dev_ms"], dist if use_dist else None, device=agg_dev)

  extra = {}
  if not args.no_extras and world == 1 and args.model == "kinematic6":
    def stepwise(om, on, oK, oW, kernel, key=None, only_kind=None, note=None):
      o = run_model(torch, None, om, on, oK, oW, dev, rank, world, only_kind=only_kind)
      ls = o["dev_ms"] * 1e-3 / oK
      rec = {"batch": on, "steps": oK, "value": on * oK / o["wall"], "unit": "steps/s", "kinds": o["kinds"],
             "algorithmic_bytes_per_filter_step": o["bytes_per_step"],
             "roofline": hbm_roofline(o["bytes_per_step"] * on, ls, kernel, **traffic_kw(f"{'kinematic6' if key == 'kinematic6_1M' else (key or om)}_b{on}", o["M"].name, o["gen"]),
                                      wall_s=o["wall"] / oK, launch_us_median=o["launch_us_median"])}
      if note:
        rec["note"] = note
      extra[key or om] = rec
    stepwise("live", 16384, 420, 42, "k_step_{4,10,12}<true> (IMU + GNSS stream mix)")
    stepwise("live", 16384, 200, 20, "k_step_4<true>", key="live_dt_gt0", only_kind=4,
             note="every launch advances time: the launch that runs the covariance predict (in the stream, 1.1 of every 2.1 launches have dt = 0)")
    stepwise("kinematic", 65536, 500, 50, "k_step_1<true>",
             note="2-state model, 112 B per filter-step: one launch per step is bounded by launch latency; see kinematic_fused")
    stepwise("kinematic6", 1 << 20, 200, 20, "k_step_1<true>", key="kinematic6_1M")
    stepwise("kinematic9", 65536, 300, 30, "k_step_1<true>")
    extra["kinematic_fused"] = fused_run_extra(torch, "kinematic", 65536, 2000, dev)
    extra["fused_run"] = fused_run_extra(torch, "kinematic6", n, 500, dev)
    extra["feature36_msckf"] = msckf_extra(torch, dev)
    extra["scalar_abi"] = scalar_abi_extra(gen_dir(["kinematic", "live"]))
    extra["kinematic6_ring"] = ring_extra(torch, gen_dir(["kinematic6"]), 65536, 1000, dev)
    extra["live_maha_rts"] = config4_extra(torch, dev, rank)

  if rank == 0:
    launch_s = dev_ms_max * 1e-3 / K
    kern = "k_step_1<true>" if args.model != "live" else "k_step_{4,10,12}<true> (stream mix)"
    out = {
      "metric": "EKF predict+update steps/sec at batch N",
      "value": value,
      "unit": "steps/s",
      "n_gpus": dist.get_world_size() if use_dist else 1,        # the ranks that actually ran (the process group's own count)
      "steps": K,
      "warmup": W,
      # launches of the same entry point that ran BEFORE the W warm-up steps and are part of neither W nor K (the cold sample and the
      # steady-state groups: steady_state_warmup below), and the mean duration of a launch on the idle device -- top-level, so that a
      # reader of the line sees what "warmup" does not say
      "untimed_launches": int(r["steady"]["launches"]) + COLD_LAUNCHES,
      "cold_launch_us": round(r["cold_launch_us"], 3),
      "ms_per_step": wall_max * 1e3 / K,
      "higher_is_better": True,
      "scaling": scaling,
      "vs_baseline": None,
      "dtype": "f64",
      "data": "synthetic",
      "config": {"workload": f"{M.name} (D={D}, E={E}, Z={r['Z']:g}) fused predict+update, step-granular (state round-trips HBM each step), "
                             f"batch {n} on rank 0, shared R, scalar dt", "batch_per_gpu": n,
                 "steady_state": f"device warmed to steady state before the timed region: {r['steady']['launches']} extra untimed launches (<= {STEADY_CAP_S} s), see untimed_launches / cold_launch_us / steady_state_warmup",
                 "global_batch": int(round(steps_total / K)), "parallelism": f"batch-sharded x{world}, no data-path collective"},
      # frac: HIP events over the K timed launches; frac_wall: the same bytes against ms_per_step (host wall clock, max over ranks)
      "roofline": hbm_roofline(r["bytes_per_step"] * n, launch_s, kern, **traffic_kw(f"{M.name}_b{n}", M.name, r["gen"]),
                               wall_s=wall_max / K, launch_us_median=r["launch_us_median"], launch_us_median_groups=r["launch_us_groups"]),
      "steady_state_warmup": r["steady"],
    }
    if r.get("group_us"):
      out["roofline"]["group_us"] = r["group_us"]
    if force_dist:
      out["forced_process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if not args.no_cpu_baseline and world == 1:
      kind = 1 if args.model != "live" else 10
      cb, ncpu, flav, ckind = cpu_baseline(M.name, kind, M, n)
      one = cb["1core"]
      out["cpu_baseline"] = {"value": one["value"], "unit": "steps/s", "cores": 1, "kind": ckind,
                             "sample": f"{ncpu} filters x {one['steps']} steps ({one['seconds']:.1f} s), {flav}, gcc -g -fPIC -O2 (the reference's flags)",
                             "all_cores": cb.get("allcores")}
      if not args.no_extras:
        c3, _, _, _ = cpu_baseline(M.name, kind, M, n, budget_s=3.0, suffix="_o3native", cflags=["-fPIC", "-O3", "-march=native"])
        out["cpu_baseline"]["O3_march_native"] = {"1core": c3["1core"], "all_cores": c3.get("allcores")}
        if args.model == "kinematic6":
          from examples.live_kf import LiveKalman as L
          cl, nl, _, _ = cpu_baseline("live", 10, L, 16384, budget_s=3.0)
          out["cpu_baseline"]["live_kind10"] = {"1core": cl["1core"], "all_cores": cl.get("allcores"),
                                                "sample": f"{nl} filters, fused predict + PHONE_ACCEL update, gcc -O2"}
    if extra:
      out["extra"] = extra
    if TRAFFIC_CARRIED:
      out["traffic_carried"] = dict(TRAFFIC_CARRIED)
    print(json.dumps(out))
  if use_dist:
    dist.destroy_process_group()


if __name__ == "__main__":
  sys.exit(main())
