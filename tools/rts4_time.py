"""GPU side of tools/rts4_dev.py: times `rts4_dev_batch_rts` of one or more development builds of the register-broadcast smoother on the
same synthetic trace (live: 8 192 filters x T steps, random SPD covariances around the filter's initial one), checks the variants against
the first one, and prints the phase timeline of builds made with RTS4_TL=1.   python tools/rts4_time.py dir1 [dir2 ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from examples.live_kf import LiveKalman as L

name = os.environ.get("RTS4_MODEL", "live_maha")
n, T = int(os.environ.get("RTS4_N", 8192)), int(os.environ.get("RTS4_T", 300))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
tx = torch.as_tensor(L.initial_x, device=dev).repeat(T, n, 1).contiguous()
tx[:, :, 3:7] += 0.01 * torch.randn((T, n, 4), dtype=torch.float64, device=dev, generator=g)
A = torch.randn((n, 22, 22), dtype=torch.float64, device=dev, generator=g) * 0.01
P = torch.diag(torch.as_tensor(L.initial_P_diag, device=dev)) * 1e-2 + A @ A.transpose(1, 2)
tP = P.repeat(T, 1, 1, 1).contiguous()
if os.environ.get("RTS4_SCHED"):      # config 4's time stamps: gyro + accelerometer at the same tick, a position fix every tenth -- 1.1 of 2.1 steps have dt = 0
  import bench
  ts = torch.as_tensor(bench.live_schedule(T)[1], device=dev)
else:
  ts = torch.as_tensor(np.arange(T) * 0.01, device=dev)
Q = torch.as_tensor(np.ascontiguousarray(L.Q), device=dev)
ref = None
vp = ctypes.c_void_p
for d in sys.argv[1:]:
  lib = ctypes.CDLL(os.path.join(d, f"lib{name}_rts4.so"))
  fn = lib.rts4_dev_batch_rts
  fn.argtypes = [vp, vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, vp]
  xs, Ps = torch.empty_like(tx), torch.empty_like(tP)
  best = 1e9
  for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(tx.data_ptr(), tP.data_ptr(), ts.data_ptr(), T, Q.data_ptr(), n, 3, xs.data_ptr(), Ps.data_ptr(), None, None, None)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    best = min(best, e0.elapsed_time(e1))
  fin = bool(torch.isfinite(Ps).all() and torch.isfinite(xs).all())
  if ref is None:
    ref = (xs.clone(), Ps.clone())
    dif = ""
  else:
    dif = f"  vs first: x {float((xs - ref[0]).abs().max()):.2e}  P {float(((Ps - ref[1]).abs() / ref[1].abs().amax(dim=(2, 3), keepdim=True)).max()):.2e} of the matrix maximum"
  print(f"{d}: {best:.3f} ms = {n * (T - 1) / best / 1e3:.1f} M steps/s ({n * (T - 1) * 8112 / best / 1e6 / 8000:.3f} of 8 TB/s), "
        f"{best * 1e3 / (T - 1) * 1024 * 4 * 2 / n:.2f} us per step and wavefront pair, finite {fin}{dif}")
  if hasattr(lib, "rts4_dev_timeline"):
    buf = (ctypes.c_ulonglong * (256 * 16))()
    assert lib.rts4_dev_timeline(ctypes.cast(buf, vp)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.float64) / 100.0
    rel = a[:, :11] - a[:, :1]
    names = ["step start", "Pk copy issued, x in LDS", "scalars done, Pk landed", "rows of Pk in registers", "A -> image, rows of Pk1_k", "x out, delta, D",
             "L D L^T done", "substitutions done (Ck in image)", "T = Ck D done", "U = T Ck^T done (in image)", "Pk_n stored, rows carried (step end)"]
    prev = 0.0
    for i, nm in enumerate(names):
      m = rel[:, i].mean()
      print(f"    {nm:40s} {m:8.2f} us (+{m - prev:6.2f})  min {rel[:, i].min():7.2f} max {rel[:, i].max():7.2f}")
      prev = m
