"""Batch-axis sharding across GPUs: one process per GPU, filters are independent, no data-path collective.

SURVEY.md 8e: rank r owns the contiguous slice [r*N/G, (r+1)*N/G) of the batch axis.  The only collectives are in
the measurement path: SUM of processed steps, MAX of elapsed seconds, and an optional all-gather of a per-rank state
checksum.  Backend "nccl" is RCCL on ROCm; the same code runs over gloo on CPU (tests/test_sharding.py).
"""
import numpy as np


def shard_range(n_total, rank, world):
  """Contiguous slice of the batch axis owned by `rank`: sizes differ by at most one, order preserved."""
  base, rem = divmod(int(n_total), int(world))
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def aggregate_throughput(steps_local, seconds_local, dist=None, device=None):
  """Whole-job steps/s = sum of steps over ranks / max of elapsed over ranks."""
  import torch
  t = torch.tensor([float(steps_local)], dtype=torch.float64, device=device)
  s = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
  if dist is not None and dist.is_initialized():      # a one-rank group reduces too (bench.py RN_BENCH_FORCE_DIST: the RCCL branch on one GPU)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
  return float(t[0]) / float(s[0]), float(t[0]), float(s[0])


def state_checksum(x):
  """Order-independent 64-bit checksum of a float64 array (sum of the raw words mod 2^64)."""
  a = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).view(np.uint64)
  return int(np.add.reduce(a.ravel(), dtype=np.uint64))


def gather_checksums(local_checksum, dist=None, device=None):
  import torch
  if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
    return [int(local_checksum)]
  # int64 transport of the uint64 word
  v = torch.tensor([np.uint64(local_checksum).astype(np.int64)], dtype=torch.int64, device=device)
  out = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
  dist.all_gather(out, v)
  return [int(np.int64(o.item()).astype(np.uint64)) for o in out]
