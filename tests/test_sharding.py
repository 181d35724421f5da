"""World-size-2 gloo test (CPU) of the multi-GPU path: batch-axis sharding + throughput aggregation + checksum gather.
The per-rank compute is the oracle here (no GPU in this container); what is under test is the sharding arithmetic and
the collectives bench.py relies on (SUM of steps, MAX of elapsed), with the rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest

from conftest import REPO


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n_total, q):
  import sys
  for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  import torch.distributed as dist
  from rednose_amd.helpers.sharding import shard_range, aggregate_throughput, state_checksum, gather_checksums
  from oracle_lib import OracleLib
  dist.init_process_group("gloo", rank=rank, world_size=world)
  lo, hi = shard_range(n_total, rank, world)
  rng = np.random.default_rng(7)                         # same global stream on every rank, sliced
  x = rng.normal(size=(n_total, 6))[lo:hi].copy()
  P = np.tile(np.eye(6), (hi - lo, 1, 1))
  z = rng.normal(size=(n_total, 3))[lo:hi].copy()
  OracleLib("kinematic6").batch_step(1, x, P, z, np.eye(3) * 0.01, np.eye(6) * 0.1, 0.01)
  sps, steps, secs = aggregate_throughput(hi - lo, 1.0 + rank, dist)
  sums = gather_checksums(state_checksum(x), dist)
  q.put((rank, lo, hi, sps, steps, secs, sums, x))
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
  import torch.multiprocessing as mp
  from rednose_amd.helpers.sharding import shard_range, state_checksum
  from oracle_lib import OracleLib
  n_total, world = 1001, 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  # slices tile the batch axis exactly
  assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total
  assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
  # aggregate = sum(steps) / max(seconds), identical on both ranks
  for r in res:
    assert r[4] == n_total and r[5] == 2.0 and abs(r[3] - n_total / 2.0) < 1e-9
  # sharded result == unsharded result, bit for bit (no cross-filter coupling anywhere)
  rng = np.random.default_rng(7)
  x = rng.normal(size=(n_total, 6)); P = np.tile(np.eye(6), (n_total, 1, 1)); z = rng.normal(size=(n_total, 3))
  OracleLib("kinematic6").batch_step(1, x, P, z, np.eye(3) * 0.01, np.eye(6) * 0.1, 0.01)
  assert np.array_equal(np.concatenate([res[0][7], res[1][7]]), x)
  assert res[0][6] == res[1][6] == [state_checksum(res[0][7]), state_checksum(res[1][7])]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["driver", "self"])
@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(mode, launcher):
  """bench.py's N > 1 path end to end: two ranks launched the way the driver launches them (torch.distributed.run,
  127.0.0.1 rendezvous), both on the ONE leased GPU over gloo (RCCL refuses two ranks per device; RN_BENCH_BACKEND exists for
  exactly this dry run).  Checks the aggregated JSON line: SUM of steps over ranks, one line, rank 0 only."""
  import json
  import subprocess
  import sys
  env = dict(os.environ, RN_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
  size = ["--batch", "4096"] if mode == "weak" else ["--global-batch", "8190"]
  tail = [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + size
  if launcher == "driver":
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + tail
  else:                      # `python bench.py --gpus 2` on its own: bench.py starts the two ranks itself
    cmd = [sys.executable] + tail
    env.pop("WORLD_SIZE", None)
  res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=REPO, timeout=600)
  assert res.returncode == 0, res.stderr[-3000:]
  lines = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
  assert len(lines) == 1, res.stdout[-2000:]
  out = json.loads(lines[0])
  total = 8192 if mode == "weak" else 8190
  assert out["n_gpus"] == 2 and out["steps"] == 30 and out["scaling"] == mode
  assert out["config"]["global_batch"] == total
  assert abs(out["value"] - total * 30 / (out["ms_per_step"] * 1e-3 * 30)) < 1e-6 * out["value"]
  assert 0 < out["roofline"]["frac"] < 1 and "extra" not in out


def test_bench_gpus_flag_is_honoured_without_a_launcher():
  """`python bench.py --gpus 2` must never silently measure one GPU: without enough devices it refuses (here: none)."""
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RN_BENCH_BACKEND")}
  res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True, env=env,
                       cwd=REPO, timeout=300)
  assert res.returncode != 0 and "not run" in (res.stderr + res.stdout)
  # a launcher that started a different number of ranks than --gpus asks for is refused too
  res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0"), cwd=REPO, timeout=300)
  assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


def test_bench_sweep_marks_missing_counts_not_run():
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RN_BENCH_BACKEND")}
  env["HIP_VISIBLE_DEVICES"] = ""
  env["CUDA_VISIBLE_DEVICES"] = ""
  res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--sweep", "1,2,4,8", "--steps", "2"], capture_output=True, text=True,
                       env=env, cwd=REPO, timeout=300)
  assert res.returncode == 0, res.stderr[-2000:]
  out = json.loads([ln for ln in res.stdout.split("\n") if ln.startswith("{")][0])
  for mode in ("weak", "strong"):
    assert [p["n_gpus"] for p in out["sweep"][mode]] == [1, 2, 4, 8]
    assert all(p["status"] == "not run" for p in out["sweep"][mode])
  assert out["value"] is None


@pytest.mark.gpu
def test_bench_sweep_on_the_leased_gpu():
  """--sweep on the one leased GPU: the 1-GPU points run (weak and strong), the 2-GPU points are 'not run'."""
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RN_BENCH_BACKEND")}
  res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--sweep", "1,2", "--steps", "30", "--warmup", "5", "--batch", "8192"],
                       capture_output=True, text=True, env=env, cwd=REPO, timeout=900)
  assert res.returncode == 0, res.stderr[-2000:]
  out = json.loads([ln for ln in res.stdout.split("\n") if ln.startswith("{")][0])
  import torch
  have = torch.cuda.device_count()
  for mode, total in (("weak", 8192), ("strong", 16384)):
    p1, p2 = out["sweep"][mode]
    assert p1["status"] == "ok" and p1["n_gpus"] == 1 and p1["global_batch"] == total and p1["value"] > 0
    assert p2["status"] == ("ok" if have >= 2 else "not run")
  assert out["n_gpus"] == (2 if have >= 2 else 1)


@pytest.mark.gpu
def test_bench_rccl_branch_on_one_gpu():
  """RN_BENCH_FORCE_DIST=1: the N = 1 line goes through the process-group branch of the N > 1 run -- a one-rank RCCL communicator
  on the leased GPU, barrier, device-tensor all-reduce of sharding.aggregate_throughput.  Not a scaling measurement (one GPU): it
  shows that the branch the 8-GPU run takes loads librccl, creates a communicator and reduces, and that the line keeps its shape."""
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RN_BENCH_BACKEND")}
  common = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--batch", "8192", "--no-extras",
            "--no-cpu-baseline"]
  lines = {}
  for label, extra_env in (("plain", {}), ("forced", {"RN_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})):
    res = subprocess.run(common, capture_output=True, text=True, env=dict(env, **extra_env), cwd=REPO, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    js = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
    assert len(js) == 1, res.stdout[-2000:]
    lines[label] = json.loads(js[0])
  plain, forced = lines["plain"], lines["forced"]
  assert forced["forced_process_group"] == {"backend": "nccl", "world_size": 1}
  assert forced["n_gpus"] == 1 and forced["steps"] == 40 and forced["config"]["global_batch"] == 8192
  assert set(forced) - {"forced_process_group"} == set(plain)
  assert set(forced["roofline"]) == set(plain["roofline"]) and 0 < forced["roofline"]["frac"] < 1.2
  assert abs(forced["value"] - 8192 * 40 / (forced["ms_per_step"] * 1e-3 * 40)) < 1e-6 * forced["value"]
