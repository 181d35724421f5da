"""Generation-time tuning knobs of the kernel emitters.

Every knob has ONE default (the measured best on MI355X) and can be overridden for A/B runs through the environment variable
RN_TUNE, e.g.  RN_TUNE="wide_ft=8,wide_lb=2" python bench.py --model live   (bench.py regenerates into RN_GEN_DIR when set, so
variants do not overwrite generated/).  What each alternative measured: profiles/tuning_notes.md.

  wide_ft      filters per wavefront tile of the three-phase step kernels (0 = auto by state count)
  wide_lb      second argument of __launch_bounds__ (wavefronts per SIMD the register budget is set for; 0 = unconstrained)
  wide_db      double-buffered asynchronous P prefetch (1), single buffer (0), auto (-1)
  wide_inline  0 = phase functions __noinline__
  wide_fpw     filters per wavefront in the matrix phase (0 = 64 // dim_err)
  wide_lean    1 = covariance rows stay in LDS (register-lean structure, two wavefronts per SIMD)
  wide_lean_q  1 = the lean predict takes its column of Q from registers instead of an LDS copy
  small_waves  amdgpu_waves_per_eu on the lane-per-filter step kernels
  small_max_e  largest error-state count served lane-per-filter
  nt_trace     1 = nontemporal stores for the fused run's covariance trace (lane-group models)
  run_block    steps per block of the lane-per-filter fused run without trace (0 = auto by model size, -1 = that kernel is not emitted)
  exact_math   1 = IEEE division / square root and ocml sin / cos in place of the fast primitives (reference build for the accuracy tests; full sin / cos range)
  run2         1 = fused run of lane-group models up to 22 error states as TWO wavefronts per tile, matrix + scalar (emit_run2), 0 = k_run (emit_wide3)
  run2_prio    s_setprio level of emit_run2's scalar wavefront (its chain of dependent instructions issues ahead of the co-resident matrix wavefront's FMAs); 0 = none
  rts4         1 = smoother of lane-group models with register-broadcast operands (emit_rts4: 16 lanes x 2 rows, 4 filters per wavefront, two wavefronts per SIMD), 0 = rn::k_rts_group
  tri_trace    1 = models with both k_run2 and k_rts4 also get k_run2_tri / k_rts4_tri: the filtered trace between batch_run_tri and batch_rts_tri as packed
               lower triangles (E (E + 1) / 2 doubles per covariance); 0 = not generated
  rts_dt0      1 = backward steps with dt == 0 of a model whose predict(dt = 0) is the identity take the identity-gain path of k_rts4 (Ck = I: no
               factorisation, no products); 0 = the full solve on every step
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
  wide_lean_q: int = 0
  small_waves: int = 0
  small_max_e: int = 7
  run_block: int = 0
  run2_prio: int = 0              # (measured: 20.5-20.9 ms per config-4 chunk at 3 against 20.4 at 0 -- the scalar wavefront is not on the critical path)
  run2: int = 1              # fused run with a scalar wavefront beside the matrix wavefront (emit_run2: two wavefronts per SIMD); 0 = emit_wide3's k_run
  rts4: int = 1              # smoother with every cross-lane operand by row_newbcast, two wavefronts per SIMD (emit_rts4: 8 .. 22 error states); 0 = rn::k_rts_group
  tri_trace: int = 1         # packed-triangle trace kernels (k_run2_tri, k_rts4_tri) for the models that have both structures
  rts_dt0: int = 1           # identity-gain path for dt == 0 steps in k_rts4 (models with identity_at_dt0 only); 0 = full solve on every step
  exact_math: int = 0        # 1 = IEEE division / sqrt and the library's sin / cos instead of the hardware-seed + Newton primitives and rn::sincos_fast (a reference build for tests: tests/test_gpu_live.py)
  nt_trace: int = 1          # the fused run's covariance trace leaves with nontemporal stores (config 4 forward: 23.5 vs 24.5 ms per chunk, same call)
  wide_timeline: int = 0     # debug: lane 0 of the first 256 workgroups stamps s_memtime / the 100 MHz wall clock at every phase boundary of
                             # the three-phase step kernels into a device buffer read back by {name}_debug_timeline (tools/timeline.py)


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
  if spec is None:
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

  def __init__(self, spec, enabled=True):
    self.vals = model_defaults(spec) if enabled else {}

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
