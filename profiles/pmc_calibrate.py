"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this box (gfx950).

MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read;
other widths and WRITE_SIZE are uncalibrated -> measure a copy of known size in the same PMC pass.
A 1 GiB fp64 tensor (past the 256 MiB Infinity Cache) is copied 5 times: each copy reads 2^30 B and writes 2^30 B.
"""
import torch

a = torch.ones(2**27, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
for _ in range(5):
  b.copy_(a)
torch.cuda.synchronize()
print("calibration copy done")
