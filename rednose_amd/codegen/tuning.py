"""Generation-time tuning knobs of the kernel emitters, with what round-1 measurements said about each.

Every knob has ONE default (the measured best on MI355X) and can be overridden for A/B runs through the
environment variable RN_TUNE, e.g.  RN_TUNE="wide_ft=8,wide_lb=2" python bench.py --model live
(bench.py regenerates into RN_GEN_DIR when set, so variants do not overwrite generated/).

  knob          default  measured alternatives (live = 23/22-state ESKF, batch 16 384; k6 = kinematic6, batch 65 536)
  wide_struct   2        1 = scalars evaluated redundantly in all 32 lanes of a group: live 122 us/launch vs 47 us
  wide_ft       16       filters per wavefront tile: 8 -> 52.8 us, 32 -> LDS allows only 3 waves per CU
  wide_lb       0        second argument of __launch_bounds__ (waves per SIMD): 2 forces <= 256 registers, hipcc then
                         spills 64-172 VGPRs to scratch: 87-138 us (0 = unconstrained, 1 wave per SIMD, 47 us)
  wide_db       1        double-buffered asynchronous P prefetch; 0 = single buffer (only sensible with wide_lb=2)
  wide_inline   1        0 = phase functions __noinline__: each fits 256 registers but pays scratch frames: 237 us
  wide_fpw      0        filters per wavefront in the matrix phase: 0 = 64 // dim_err (dim_err-lane groups when that is > 2, e.g. 7 filters
                         for 9 error states; live's 22 error states fit twice: two 32-lane groups); 2 = always two groups
  small_waves   0        amdgpu_waves_per_eu(n, n) on the lane-per-filter step kernels: 1 -> k6 35 us/launch vs 9.5 us
  small_lpf     1        lanes per filter in the family-S step kernels: 2 = lane PAIR per filter (emit_small2.py: half the
                         rows per lane, DPP exchanges, 2 waves per SIMD): k6 9.8-10.0 us/launch vs 9.4 us -- parity-green but
                         not faster, the two waves of a SIMD still move in lockstep through load / compute / store
"""
import os
from dataclasses import dataclass, fields


@dataclass(frozen=True)
class Tuning:
  wide_struct: int = 2
  wide_ft: int = 16
  wide_lb: int = 0
  wide_db: int = 1
  wide_inline: int = 1
  wide_fpw: int = 0
  small_waves: int = 0
  small_lpf: int = 1


def current():
  vals = {}
  names = {f.name for f in fields(Tuning)}
  for kv in os.environ.get("RN_TUNE", "").split(","):
    if "=" in kv:
      k, v = (t.strip() for t in kv.split("=", 1))
      if k not in names:
        raise ValueError(f"RN_TUNE: unknown knob {k!r} (known: {sorted(names)})")
      vals[k] = int(v)
  return Tuning(**vals)
