/*
 * rednose_amd_filter.h -- the C ABI of a generated rednose_amd filter library (lib{name}.so).
 *
 * Every symbol is prefixed with the filter name chosen at gen_code() time, exactly like the reference
 * (/root/reference/rednose/helpers/ekf_sym.py:149-171).  This header documents the ABI with the macro
 * RN_FN(name, sym) == name##_##sym; the concrete per-model prototype lists that gen_code writes next to
 * each library are committed for the shipped models as include/kinematic.h, include/kinematic6.h and
 * include/live.h, and tests/test_abi.py checks that each library exports every symbol they declare.
 *
 * Conventions
 *   - fp64 everywhere, dense row-major: x is D doubles, P is E x E, z is Z, R is Z x Z
 *     (/root/reference/rednose/templates/ekf_c.c:4-6,39-44).
 *   - in-place semantics of the reference are kept: x and P are updated in place, z is overwritten with the
 *     residual y = z - h(x) (ekf_c.c:118-120); Q, R, ea are read-only; no allocation crosses the boundary.
 *   - section 1 takes HOST pointers and one filter (the reference's existing ABI, run as a batch of one on
 *     the GPU); section 2 takes DEVICE pointers and n filters laid out as the natural batch of the same
 *     buffers: x (n, D), P (n, E, E), z (n, Z), R (Z, Z) shared or (n, Z, Z), contiguous, 16-byte aligned.
 *   - section 2 functions are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and return 0 on success or a status (1 HIP error, 2 bad argument, 3 misaligned pointer); the
 *     reference returns void and asserts (/root/reference/rednose/helpers/ekf_sym.cc:13,25-27,204).
 *     {name}_last_error() / {name}_last_error_string() report the last failure of the calling thread,
 *     including failures inside the void section-1 functions.
 *   - there is no CPU implementation behind any of these symbols.
 *
 * Asymmetric covariances.  The reference never symmetrises P: its predict and update multiply with both halves, and its
 * innovation covariance S = H P H^T + R is solved as a general matrix (ekf_c.c:24,100-101,115) -- a P that is not exactly
 * symmetric (its own Joseph form leaves ~1e-12 of asymmetry after a few thousand steps) is a legal input.  What each entry
 * point does with one (tests/test_gpu_asymmetric.py feeds SPD + a skew part through every one of them):
 *   - step-granular entry points -- the section-1 functions, batch_predict, batch_update_k, batch_predict_update_k, their
 *     _masked twins, batch_maha_k -- follow the reference entry for entry: both halves of P, S factored as a general matrix
 *     (L D U, no pivoting).  Same input, same result as the oracle to 1e-10 of the row maximum, symmetric or not.
 *   - batch_run keeps P in registers for T steps with arithmetic that uses P = P^T (one transposition per predict, the gain
 *     taken from the rows).  It reads (P + P^T) / 2 of the caller's matrix, ONCE, when the state enters the registers; the
 *     result is the reference's on THAT matrix, to rounding.  Against the reference run on the asymmetric matrix itself the
 *     difference is first order in (P - P^T) / 2, like any perturbation of P0 of that size.  The final P and the covariance
 *     trace come back symmetric to rounding (lane-per-filter models: exactly).
 *     WHICH MODE IS THE REFERENCE'S: the step-granular one.  A caller that needs the reference's result for a T-step schedule on a
 *     covariance with a skew part walks the schedule with batch_predict_update_k -- BatchedEKF.run(..., exact=True) does exactly that
 *     (trace, flags and residuals included; tests/test_gpu_asymmetric.py::test_run_exact_is_the_reference_on_the_asymmetric_matrix_itself:
 *     1e-10 against the oracle on the asymmetric matrix); batch_run is the fast path for covariances that are symmetric up to
 *     rounding, which is every covariance a filter produces itself.
 *   - batch_rts factors the predicted covariance by Cholesky and keeps Pk1_n - Pk1_k as a packed triangle.  It reads the
 *     LOWER triangle (diagonal included) of every covariance it is given -- Pf[k], P_last -- mirrored, for the gain Ck and
 *     the correction Ck (Pk1_n - Pk1_k) Ck^T; the filtered covariance itself enters the sum as given:
 *     Ps[k] = Pf[k] + correction.  On a trace written by batch_run (symmetric to rounding) that is the reference's
 *     rts_smooth to rounding.  (The reference's own backward step is triangle-dependent on such input too: its gain comes from
 *     scipy.linalg.solve(..., assume_a='pos'), i.e. LAPACK posv on ONE triangle of Pk1_k, ekf_sym.py:677.)
 *
 * Elementary functions of the model's expressions.  sqrt, reciprocals and negative half-integer powers are evaluated from the
 * hardware seeds with Newton steps (within an ulp or two of libm; IEEE answers kept at 0 and infinity).  sin / cos of one
 * argument come from one in-line routine accurate to 2e-16 ABSOLUTE for |a| <= 2^45 rad; beyond that (neighbouring doubles are
 * 0.008 rad apart there), and for NaN / inf, both are NaN -- the reference's libm returns the sine of the exact double instead.
 * A filter whose state drives a trigonometric argument that far comes back non-finite and is flagged (flag bit 1), not silently wrong.
 * The opt-out: a library generated under RN_TUNE=exact_math=1 uses IEEE division / sqrt and the library's sin / cos (full range) in
 * every kernel.  Measured against such a build (tests/test_gpu_live.py, profiles/r5_live_fast_vs_ieee.json): single calls of the live
 * filter agree to 1e-17 of the row maximum, a free-running 84-step stream to 3.4e-13 on P.
 */
#ifndef REDNOSE_AMD_FILTER_H
#define REDNOSE_AMD_FILTER_H

#include <stdint.h>

#define RN_FN(name, sym) name##_##sym

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------
 * Section 1 -- the reference's scalar ABI, unchanged signatures (HOST pointers, one filter)
 * Each call runs as a batch of one on the GPU and returns when its results are in the caller's buffers: the arguments are packed
 * into one pinned host buffer that is mapped into the device's address space, the kernel works on it in place, the call waits for
 * the null stream (one launch + one wait: 15-25 us per call on an MI355X; thread-safe: one mutex per library).  Errors are recorded
 * ({name}_last_error / _last_error_string) and leave the caller's buffers untouched -- without a HIP device every call does that.
 * --------------------------------------------------------------------------------------------------- */
#define RN_DECLARE_SCALAR_ABI(name)                                                                              \
  /* replaces {name}_predict, ekf_sym.py:162-165 -> predict(), ekf_c.c:8-33 */                                     \
  void RN_FN(name, predict)(double *in_x, double *in_P, double *in_Q, double dt);                                  \
  /* sympy routine wrappers, ekf_sym.py:155-161 (typedefs in rednose/helpers/ekf.h:21-25) */                      \
  void RN_FN(name, f_fun)(double *state, double dt, double *out);        /* next nominal state, D            */   \
  void RN_FN(name, F_fun)(double *state, double dt, double *out);        /* error-state transition, E x E    */   \
  void RN_FN(name, err_fun)(double *nom_x, double *delta_x, double *out);    /* inject error, D              */   \
  void RN_FN(name, inv_err_fun)(double *nom_x, double *true_x, double *out); /* extract error, E             */   \
  void RN_FN(name, H_mod_fun)(double *state, double *out);               /* d nominal / d error, D x E       */

/* per observation kind k (the integer is part of the symbol):
 *   replaces {name}_update_{k}, ekf_sym.py:149-152 -> update<Z,3,MAHA>(), ekf_c.c:37-121 */
/* one per gen_code global_var v: void {name}_set_{v}(double x)  -- ekf_sym.py:166-171; writes a device global */

#define RN_DECLARE_SCALAR_KIND(name, k)                                                                          \
  void RN_FN(name, update_##k)(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);             \
  void RN_FN(name, h_##k)(double *state, double *ea, double *out);       /* predicted observation, Z         */   \
  void RN_FN(name, H_##k)(double *state, double *ea, double *out);       /* observation Jacobian, Z x D      */

/* ---------------------------------------------------------------------------------------------------
 * Section 2 -- batched entry points (DEVICE pointers, n independent filters) -- new with this engine.
 * They replace n sequential calls of the section-1 functions made by the reference's orchestrators
 * (EKFSym::predict / ::update, /root/reference/rednose/helpers/ekf_sym.cc:196-219).
 * --------------------------------------------------------------------------------------------------- */
#define RN_DECLARE_BATCH_ABI(name)                                                                               \
  void RN_FN(name, dims)(int *dims);                  /* dims[0..2] = DIM, EDIM, MEDIM (ekf_sym.py:122-124) */     \
  int RN_FN(name, num_kinds)(void);                                                                              \
  void RN_FN(name, kinds)(int *out);                  /* observation kinds, EKF::kinds (ekf.h:18)            */   \
  int RN_FN(name, kind_zdim)(int kind);               /* Z of a kind, -1 if unknown                          */   \
  int RN_FN(name, kind_maha)(int kind);               /* 1 if generated with the Mahalanobis gate            */   \
  int RN_FN(name, kind_eadim)(int kind);              /* extra arguments per observation of a kind (0: `ea` ignored) */ \
  void RN_FN(name, msckf_dims)(int *dims);            /* dim_main, dim_main_err, dim_augment, dim_augment_err, N (ekf_sym.py:57-73) */ \
  int RN_FN(name, last_error)(void);                                                                             \
  const char *RN_FN(name, last_error_string)(void);                                                              \
  void RN_FN(name, clear_error)(void);                                                                           \
  /* P <- F P F^T + dt Q, x <- f(x, dt) for n filters; dt_vec (n) may be NULL => scalar dt for all;            \
   * norm_quats != 0 renormalises the quaternion slices given to gen_code (EKFSym::normalize_quaternions,      \
   * ekf_sym.cc:69-77,207) */                                                                                  \
  int RN_FN(name, batch_predict)(double *x, double *P, const double *Q, const double *dt_vec, double dt,         \
                                 int64_t n, int norm_quats, void *stream);                                       \
  /* the same for the filters with active[i] != 0 only (see RN_DECLARE_BATCH_KIND_MASKED) */                     \
  int RN_FN(name, batch_predict_masked)(double *x, double *P, const double *Q, const double *dt_vec, double dt,  \
                                        int64_t n, int norm_quats, const uint8_t *active, void *stream);

/* fused multi-step run and offline smoothing (SURVEY.md 8b "proposed new batched exports") */
#define RN_DECLARE_BATCH_RUN(name)                                                                               \
  int RN_FN(name, zmax)(void);                        /* largest Z over the kinds: row stride of z in batch_run  */  \
  int RN_FN(name, run_unroll)(void);                  /* steps per iteration of batch_run's loop (instruction accounting) */ \
  int RN_FN(name, has_tri_trace)(void);               /* 1: the library has RN_DECLARE_BATCH_TRI's entry points (packed-triangle covariance trace) */ \
  int RN_FN(name, has_batch_run)(void);               /* 0: the fused kernel of this model did not fit the register file -- batch_run \
                                                         returns 4 (unsupported); walk the schedule with batch_predict_update_k */ \
  int RN_FN(name, predict_identity_at_dt0)(void);     /* 1: predict(dt = 0) is the identity on (x, P) for this model (f(x, 0) == x, F(x, 0) == I \
                                                         symbolically): a batch_run step with dt = 0 is an update alone -- the n observations of one  \
                                                         EKFSym::predict_and_update_batch call (ekf_sym.cc:172-180) can be one launch */ \
  /* T predict+update steps in ONE launch, x and P resident on chip between steps.  kinds (T) int32, dts (T),   \
   * R (T, zmax*zmax; the leading Z*Z entries of row t are that step's row-major R) and z (T, n, zmax; in: z,   \
   * out: y) are DEVICE arrays; the schedule is shared by all filters.  flags (T, n), trace_x (T, n, D) and      \
   * trace_P (T, n, E, E) -- the FILTERED pair after each step, i.e. Estimate.xk/Pk of ekf_sym.h:32-42 -- may be  \
   * NULL.  ea (T, n, EA) DEVICE array or NULL: per filter and step extra arguments of the kinds that take them (EA = the   \
   * largest kind_eadim; MSCKF feature tracks: the landmark); augment (T) int32 DEVICE array or NULL: non-zero = MSCKF window \
   * shift after that step (EKF_sym.augment, ekf_sym.py:365-391,527-528).  Replaces T calls of                              \
   * EKFSym::predict_and_update_batch (ekf_sym.cc:158-194). */                                                              \
  int RN_FN(name, batch_run)(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts,       \
                             int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags,    \
                             double *trace_x, double *trace_P, const double *ea, const int32_t *augment,          \
                             void *stream);                                                                       \
  /* Rauch-Tung-Striebel backward pass over a filtered trace; replaces the Python-only EKF_sym.rts_smooth        \
   * (/root/reference/rednose/helpers/ekf_sym.py:651-690).  xs/Ps may alias xf/Pf.  norm_quats: bit 0 = renormalise the   \
   * recomputed predicted states (as the forward pass did), bit 1 = the reference's norm_quats (smoothed states).       \
   * x_last (n, D) / P_last (n, E, E), both optional (NULL): the PREDICTED pair of the last step, which the reference     \
   * returns verbatim as the newest smoothed estimate (estimates[-1][0], [2], ekf_sym.py:658-659); NULL = recomputed from  \
   * the filtered pair of step T-2 (exact unless an MSCKF window shift happened in between).  MSCKF models: only the     \
   * main block / main states are smoothed, the rest of each filtered estimate passes through (ekf_sym.py:675-686). */    \
  int RN_FN(name, batch_rts)(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q,     \
                             int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last,             \
                             const double *P_last, void *stream);

/* Packed-triangle trace (libraries with {name}_has_tri_trace() == 1: lane-group models of 13 .. 22 error states, e.g. live): the filtered
 * covariance trace between the forward and the backward pass as LOWER TRIANGLES packed row-major -- entry (i, j <= i) of a covariance at
 * i (i + 1) / 2 + j, E (E + 1) / 2 doubles per record instead of E * E.  Nothing is lost against batch_run + batch_rts: the fused run's covariance
 * is symmetric by contract and batch_rts reads lower triangles only (see "Asymmetric covariances" above).  batch_run_tri: trace_P is
 * (T, n, E (E + 1) / 2), everything else as batch_run.  batch_rts_tri: Pf, Ps and P_last are packed, everything else as batch_rts (Ps may alias
 * Pf).  batch_tri_unpack / batch_tri_pack convert `count` records to / from full symmetric matrices (pack takes the lower triangle). */
#define RN_DECLARE_BATCH_TRI(name)                                                                               \
  int RN_FN(name, batch_run_tri)(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts,   \
                                 int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, \
                                 double *trace_x, double *trace_P, const double *ea, const int32_t *augment,       \
                                 void *stream);                                                                    \
  int RN_FN(name, batch_rts_tri)(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q,  \
                                 int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last,          \
                                 const double *P_last, void *stream);                                              \
  int RN_FN(name, batch_tri_unpack)(const double *tri, double *full, int64_t count, void *stream);                 \
  int RN_FN(name, batch_tri_pack)(const double *full, double *tri, int64_t count, void *stream);

/* MSCKF models only (gen_code msckf_params): window shift of EKF_sym.augment (ekf_sym.py:365-391) on n filters, in place */
#define RN_DECLARE_BATCH_MSCKF(name)                                                                             \
  int RN_FN(name, batch_augment)(double *x, double *P, int64_t n, void *stream);

/* feature-track kinds of an MSCKF model additionally export the reference's extra-argument Jacobian:
 *   void {name}_He_{k}(double *state, double *ea, double *out)   -- Z x 3, ekf_sym.py:110-113; their updates project the
 *   residual, H and R on the left null space of it (ekf_c.c:66-76) and write Z - 3 residual rows back into z (the last 3 entries of
 *   z pass through).  BASIS of that residual: the reference writes A^T (z - h) with A = Hea^T.fullPivLu().kernel() (ekf_c.c:71-73,120),
 *   a basis that is not orthonormal; these kernels use an ORTHONORMAL basis Q of the same space (Householder QR of Hea).  x and P do
 *   not depend on the choice (tested to 1e-10); the residuals are related by y_ref = (A^T Q) y, and the quantity that does not
 *   depend on the basis agrees: |y|^2 = y_ref^T (A^T A)^-1 y_ref = |projection of z - h on the null space|^2
 *   (tests/test_gpu_msckf.py::test_both_kinds_vs_oracle_strict checks it against the reference's A for every filter). */
#define RN_DECLARE_BATCH_KIND(name, k)                                                                           \
  /* update only.  ea: (n, kind_eadim) extra arguments, one row per filter (NULL when the kind takes none).    \
   * flags (n bytes, may be NULL): bit0 = Mahalanobis gate fired (R inflated, ekf_c.c:88-94),                  \
   * bit1 = non-finite state after the update, bit2 = null-space projection failed (rank-deficient extra-argument  \
   * Jacobian; the measurement is ignored like in ekf_sym.py:589-591) */                                        \
  int RN_FN(name, batch_update_##k)(double *x, double *P, double *z, const double *R, int r_per_filter,          \
                                    const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);  \
  /* Mahalanobis distance d2 = y^T (He P He^T + R)^-1 y of an observation, nothing modified; replaces the Python-only \
   * EKF_sym.maha_test (/root/reference/rednose/helpers/ekf_sym.py:626-649) for n filters */                      \
  int RN_FN(name, batch_maha_##k)(const double *x, const double *P, const double *z, const double *R,            \
                                  int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);      \
  /* fused predict + update: ONE launch, state crosses HBM once (EKFSym::predict_and_update_batch with a       \
   * single observation, ekf_sym.cc:158-194) */                                                                \
  int RN_FN(name, batch_predict_update_##k)(double *x, double *P, const double *Q, const double *dt_vec,         \
                                            double dt, double *z, const double *R, int r_per_filter,             \
                                            const double *ea, int64_t n, int norm_quats, uint8_t *flags,         \
                                            void *stream);                                                       \
  /* the same launch also writing the call's CHECKPOINT -- what EKFSym::checkpoint keeps of a call (ekf_sym.cc:142-156, 191): \
   * ckpt_z (n, Z) the observations as they came (z itself leaves as the residuals), ckpt_x (n, D) / ckpt_P (n, E, E) the   \
   * filtered pair; DEVICE, 16-byte aligned, distinct from x / P / z.  One launch and 1.5 x the bytes of the plain step      \
   * instead of the step plus three copies (2 x the bytes, four launches) */                                    \
  int RN_FN(name, batch_predict_update_##k##_ckpt)(double *x, double *P, const double *Q, const double *dt_vec,  \
                                                   double dt, double *z, const double *R, int r_per_filter,      \
                                                   const double *ea, int64_t n, int norm_quats, uint8_t *flags,  \
                                                   double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);

/* Per-filter timelines.  Every filter of the reference is its own instance with its own filter_time and its own rewind ring
 * (EKFSym::predict_and_update_batch / rewind, /root/reference/rednose/helpers/ekf_sym.cc:83-156); a batch fed from n INDEPENDENT
 * logs therefore needs calls that advance only SOME filters, each by its own dt.  The `_masked` twins of the step-granular entry
 * points take `active` (n bytes, DEVICE): filters with active[i] == 0 pass through untouched -- x, P and z[i] leave exactly as
 * they came -- and get flags[i] = 16 (bit 4); the others see the plain entry point's arithmetic with dt_vec[i].  active == NULL is
 * the plain entry point.  (Bit 5 of the orchestrators' flags, "observation too old for this filter's ring, ignored"
 * -- ekf_sym.cc:87-94 -- is set on the host side: BatchedEKF / EKFSymBatch.) */
/* Checkpoint rings of per-filter timelines: `rec` doubles of record i of a flat DEVICE array (row stride flat_stride) <-> entry
 * slot[i] of filter i in a ring laid out (K, n, ring_stride), for the filters with active[i] != 0 (NULL: all); to_ring != 0
 * stores, 0 loads.  One launch moves one array of a checkpoint (x: D, P: E * E, z: Z of the kind into a ring of stride zmax) for
 * the whole batch, every filter at its own ring position -- the batched form of EKFSym::checkpoint / ::rewind's state copy
 * (ekf_sym.cc:119-156), whose lists are per filter instance. */
#define RN_DECLARE_BATCH_RING(name)                                                                              \
  int RN_FN(name, batch_ring_copy)(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride,          \
                                   int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n,            \
                                   int to_ring, void *stream);                                                    \
  /* flags[i] = value for the filters with mask[i] != 0 (both n bytes, DEVICE): how an orchestrator marks the filters whose observation was  \
   * too old for their ring (bits 4 | 5) on the stream, without moving the flag bytes through the host */          \
  int RN_FN(name, batch_flags_set)(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream);

#define RN_DECLARE_BATCH_KIND_MASKED(name, k)                                                                    \
  int RN_FN(name, batch_update_##k##_masked)(double *x, double *P, double *z, const double *R, int r_per_filter,  \
                                             const double *ea, int64_t n, int norm_quats, uint8_t *flags,         \
                                             const uint8_t *active, void *stream);                                \
  int RN_FN(name, batch_predict_update_##k##_masked)(double *x, double *P, const double *Q, const double *dt_vec, \
                                                     double dt, double *z, const double *R, int r_per_filter,     \
                                                     const double *ea, int64_t n, int norm_quats, uint8_t *flags, \
                                                     const uint8_t *active, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REDNOSE_AMD_FILTER_H */
