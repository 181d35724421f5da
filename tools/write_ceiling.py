"""Achievable rate of a pure store stream on this box (what bounds config 4's forward pass, which writes 4 056 B of trace per
filter-step and reads 24): torch fill_ (vectorised stores) and hipMemset-style zero_ over tensors well past the Infinity Cache."""
import torch
for gb in (1, 8, 64):
  n = gb * (1 << 30) // 8
  a = torch.empty(n, dtype=torch.float64, device="cuda")
  for name, fn in (("fill_", lambda: a.fill_(1.5)), ("zero_", lambda: a.zero_()), ("copy from 8 KB broadcast", lambda: a.view(-1, 1024).copy_(b))):
    b = torch.ones(1024, dtype=torch.float64, device="cuda")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
      fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{gb:3d} GiB {name:26s} {ms:9.3f} ms  {n * 8 / ms / 1e9:6.2f} TB/s written")
  del a
