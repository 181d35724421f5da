"""Generation-time tuning knobs of the kernel emitters, with what round-1 measurements said about each.

Every knob has ONE default (the measured best on MI355X) and can be overridden for A/B runs through the
environment variable RN_TUNE, e.g.  RN_TUNE="wide_ft=8,wide_lb=2" python bench.py --model live
(bench.py regenerates into RN_GEN_DIR when set, so variants do not overwrite generated/).

  knob          default  measured alternatives (live = 23/22-state ESKF, batch 16 384; k6 = kinematic6, batch 65 536)
  wide_ft       0 (auto) filters per wavefront tile; auto = 8 above 40 error states (LDS budget), 16 above 16 error states (live: 8 -> 52.8 us, 32 -> LDS allows only
                         3 waves per CU), two groups below (kinematic9, 7 filters per group: 14 -> 25.5-26.3 us with a single
                         buffer, 7 -> 31.2, 21 -> 28.6, 28 -> 28.3; with the double buffer 16 -> 30.5, 42 -> 36.4, 63 -> 47.7)
  wide_lb       0        second argument of __launch_bounds__ (waves per SIMD): 2 forces <= 256 registers, hipcc then
                         spills 64-172 VGPRs to scratch: 87-138 us (0 = unconstrained, 1 wave per SIMD, 47 us)
  wide_db       -1 (auto) double-buffered asynchronous P prefetch (1) or single buffer (0); auto = double from 17 to 40 error states
                         (one wavefront per SIMD there: nothing else hides the HBM latency), single below
  wide_inline   1        0 = phase functions __noinline__: each fits 256 registers but pays scratch frames: 237 us
  wide_fpw      0        filters per wavefront in the matrix phase: 0 = 64 // dim_err (dim_err-lane groups when that is > 2, e.g. 7 filters
                         for 9 error states: kinematic9 30.2 us vs 45.8 us with two groups; live's 22 error states fit twice: two
                         32-lane groups); 2 = always two groups
  wide_lean     0        1 = covariance rows stay in LDS (in-place rank-Z pass, only the columns He touches are read, Q from
                         LDS): 199-229 VGPRs instead of 256 + 84..234 AGPRs, two waves per SIMD without spills -- but live
                         46.4 us/launch at 1 wave/SIMD and 44.4 us at 2 waves/SIMD (ft=4, lb=2, db=0; smaller tiles repeat the
                         16-lane scalar phase more often: 7.9 M VALU instructions vs 5.4 M) against 40.0 us for the default;
                         2 = rows in registers, lean algebra (no column array, one fused rank-Z pass): 41.3 us.  Parity-green.
  wide_unroll   2        unroll factor of the lean in-place pass (1: 65.9 us, 4: same as 2)
  wide_lean_q   0        1 = the lean predict takes its column of Q from registers instead of an LDS copy: with wide_lean=1,
                         wide_ft=8, wide_lb=2, wide_db=0 the block needs 19 KB of LDS and <= 256 VGPRs, so all 2 048 tiles of
                         16 384 filters are resident at two wavefronts per SIMD: live 38.0 us per launch against 40.0 us for the
                         default in the same call (-5 %; not the default: the budget does not hold for larger models)
  small_waves   0        amdgpu_waves_per_eu(n, n) on the lane-per-filter step kernels: 1 -> k6 35 us/launch vs 9.5 us
  small_max_e   7        largest error-state count served lane-per-filter (8 spills, see emit.py); below it the lane-group family also works (k6 with
                         small_max_e=4: 13.9-15.7 us/launch, parity-green, against 9.1 us lane-per-filter)
Also measured, not kept: SOFTWARE PIPELINING over the tiles of a wavefront (double-buffered LDS image, the next tile's
global_load_lds issued before the current tile is computed, counted s_waitcnt vmcnt(23) so that the previous tile's stores stay
in flight, two tiles per wavefront): k6 at 65 536 filters 10.7 us per launch against 9.2 us (half as many wavefronts, 53 KB of
LDS each), no difference from 196 608 filters up (1 M filters: 134.5 vs 134.0 us on the same box).  The same kernel with the
arithmetic removed takes 7.3 of the 9.2 us: the load -> store skeleton dominates.  Box-to-box spread of identical builds is up
to 9 % (1 M filters: 134 us on one MI355X, 147 us on another), so only same-call comparisons are quoted here.
Also measured, not kept (family W, live): the double-buffered P prefetch is drained early in the compute phase -- hipcc inserts
s_waitcnt vmcnt(0) where the per-filter R (an ordinary global load issued AFTER the global_load_lds prefetch) is first used,
because vmcnt retires in order.  With R copied into the LDS slot in phase 1, two distinct LDS objects ping-ponged by an
unrolled group loop and a counted vmcnt(8) that leaves the previous group's write-back in flight, the ISA shows no wait between
prefetch and write-back any more -- and the kernel time does not move (same-call A/B: 41.0 vs 41.1 us).  With the arithmetic
removed the live kernel takes 19.8 us (8 dependent HBM round trips per wavefront), the matrix part of predict adds 5, of the
update 13, the scalar phases 2.7: the step is bound by the dependent-instruction latency of a wavefront that is alone on its
SIMD (LDS turnarounds, the in-lane 3 x 3 factorisation), not by HBM latency or bandwidth.
Also measured, not kept: records straight between HBM and registers (per-lane 16-byte loads / stores, no LDS staging): k6
12.5 us per launch against 9.2 us (1 M filters: 173 vs 134 us) -- a lane's record is 288 bytes, so every wave-instruction
touches 64 different cache lines.
Also measured, not kept: TWO WAVEFRONTS per 64-filter tile, lane l of both = filter l, both run predict / gains / state
redundantly from the shared LDS image and each finishes half of the covariance rows (no exchange, code specialised per
wavefront, ~65 % of the fp64 work per wavefront, 2 wavefronts per SIMD at 256 VGPRs with 14 spills): k6 11.1 us per launch
against 9.2 us -- the extra LDS reads, workgroup barriers and spills cost more than the second wavefront hides.
Also measured, not kept as a knob: delaying every other group of 8 wavefronts with s_sleep so that load / compute / store
phases of the two halves interleave (k6: 1.0 us delay -> 9.4 us, 2.9 us -> 10.8 us, none 9.1 us): the phases are latency-,
not bandwidth-bound, so staggering only adds the delay.
"""
import os
from dataclasses import dataclass, fields


@dataclass(frozen=True)
class Tuning:
  wide_ft: int = 0
  wide_lb: int = 0
  wide_db: int = -1
  wide_inline: int = 1
  wide_fpw: int = 0
  wide_lean: int = 0
  wide_unroll: int = 2
  wide_lean_q: int = 0
  small_waves: int = 0
  small_max_e: int = 7
  wide_step3: int = 0        # EXPERIMENT (off): step-granular kernels in the fused run's layout (emit_wide3.step_kernels: 8 lanes x 3 rows per
                             # filter, 8 filters per wavefront, the next tile's P prefetched into the LDS image during the update).
                             # Parity-green (tests/test_gpu_live.py, 9 tests) and SLOWER on live at 16 384 filters: 44.8 us per launch
                             # of the IMU / GNSS stream mix with 1 024 wavefronts x 2 tiles (=1), 48.5 us with 2 048 x 1 tile (=2),
                             # against 38.1 us for the three-phase kernels: one wavefront per SIMD runs load -> compute -> store
                             # serially and 1 024 of them do it in step, so HBM idles while they compute
  wide_timeline: int = 0     # debug: lane 0 of the first 256 workgroups stamps s_memtime / the 100 MHz wall clock at every phase boundary of
                             # the three-phase step kernels into a device buffer read back by {name}_debug_timeline (tools/timeline.py)


NO_MODEL_DEFAULTS = set()  # models whose build with the per-model defaults spilled registers: gen_code falls back to the general structure
_MODEL_DEFAULTS = {}       # knob values chosen per model by model_defaults() while that model is being emitted


def model_defaults(spec):
  """Per-model defaults, applied by emit() around the emission of one library (RN_TUNE still overrides every knob).

  Two 32-lane groups per wavefront and at most 24 error states, no feature-track kinds (live: 22): the register-lean
  structure -- rows of P stay in LDS, Q column in registers, tiles of 8 filters, single P buffer, <= 256 registers -- puts
  TWO wavefronts on every SIMD.  Measured on live, 16 384 filters (round 2, phase timeline of both builds in one call,
  tools/timeline.py): a lone wavefront spends 4.5 us per pair of filters (0.5 waiting for the previous write-back, 1.5
  predict, 2.0 update, 0.5 issuing stores: a chain of dependent LDS and fp64 operations that issues one instruction every
  ~8 cycles), two co-resident wavefronts 6.4 us per pair EACH, i.e. 3.2 us per pair and SIMD: the dt > 0 launch ends after
  32.4 us instead of 42.8 us."""
  if spec is None or spec.name in NO_MODEL_DEFAULTS:
    return {}
  msckf = any(k.He_sym is not None for k in spec.kinds)
  if not msckf and 22 <= spec.dim_err <= 24:
    return dict(wide_lean=1, wide_lean_q=1, wide_ft=8, wide_lb=2, wide_db=0)
  if spec.dim_err > 32:
    # One filter per wavefront pass (33 .. 64 error states, e.g. the 36-state MSCKF example).  Measured on feature36, 16 384
    # filters, fused predict + feature-track update (tools/msckf_time.py, one call): general structure with tiles of 16 filters
    # 156.7 us per launch (28 % of the HBM roofline), tiles of 8 106.7, tiles of 4 118.4 / 109.9 (double / single buffer);
    # register-lean structure at two wavefronts per SIMD: tiles of 8 113.3, of 4 **84.2 us (52 %)**, of 3 92.0, of 2 96.0.
    # With one filter per pass the scalar phase is short relative to the covariance passes, so small tiles (more wavefronts,
    # 8 per CU at 19 KB of LDS each) win; below 4 the scalar phase, repeated per tile at 4 active lanes, takes over.
    return dict(wide_lean=1, wide_lean_q=1, wide_ft=4, wide_lb=2, wide_db=0)
  return {}


class using_model:
  """Context manager: emit one model's kernels with its per-model defaults."""

  def __init__(self, spec):
    self.vals = model_defaults(spec)

  def __enter__(self):
    _MODEL_DEFAULTS.clear()
    _MODEL_DEFAULTS.update(self.vals)

  def __exit__(self, *exc):
    _MODEL_DEFAULTS.clear()


def current():
  vals = dict(_MODEL_DEFAULTS)
  if os.environ.get("RN_TUNE_NO_MODEL_DEFAULTS"):
    vals = {}
  names = {f.name for f in fields(Tuning)}
  for kv in os.environ.get("RN_TUNE", "").split(","):
    if "=" in kv:
      k, v = (t.strip() for t in kv.split("=", 1))
      if k not in names:
        raise ValueError(f"RN_TUNE: unknown knob {k!r} (known: {sorted(names)})")
      vals[k] = int(v)
  return Tuning(**vals)
