// ekf_hip_rts.h -- batched Rauch-Tung-Striebel backward pass (hand-written HIP, gfx950).
//
// Restates the reference's Python-only smoother (/root/reference/rednose/helpers/ekf_sym.py:651-690) for N
// independent filters on the GPU.  Per backward step k (T-2 ... 0), with estimates[k] = (xk_km1, xk_k, Pk_km1, Pk_k, t):
//     Fk      = F(xk_k, t[k+1] - t[k])                                              (:673)
//     Ck      = solve(Pk1_k, Fk Pk_k^T)^T                                           (:677)
//     delta   = Ck inv_err(xk1_k, xk1_n);  xk_n = err(xk_k, delta)                  (:680-684)
//     Pk_n    = Pk_k + Ck (Pk1_n - Pk1_k) Ck^T                                      (:686)
// including its quirks: the recursion starts from the PREDICTED state/covariance of the last step
// (estimates[-1][0], [2], :658-659), and with norm_quats every xk1_n is renormalised in place, so all returned
// states except the oldest are normalised (:665-667).  norm_quats is a bit mask here: bit 0 renormalises the RECOMPUTED
// predicted state (what the forward pass did when the filter has quaternion_idxs: the reference reads those pairs from
// the stored estimates), bit 1 is the reference's norm_quats argument (the smoothed states; the reference normalises
// the hard-coded slice 3:7, :666-667 -- here the model's quaternion slices).
// Memory plan (SURVEY.md section 7 "smoother trace capacity"): only the FILTERED trace (xk_k, Pk_k, t) is stored by
// the forward pass; the predicted pair (xk1_k, Pk1_k) is recomputed here from the filtered pair of step k with the
// same f/F code -- half the trace (140 GB instead of 279 GB at 2100 x 16384 live steps).  Outputs may alias inputs.
//
// Mapping: 32-lane group per filter (2 filters per wavefront), all E x E work matrices in the wave's private LDS,
// lane c owns row/column c.  Pk1_k is SPD: Cholesky (unrolled, row-owner) + one forward/back substitution per lane
// (lane c solves for column c of Ck^T).  No MFMA: 22 x 22 fp64 blocks, and the pass is bound by LDS/VALU latency.
#pragma once

#include "ekf_hip_rt.h"

namespace rn {

// Model must provide: D, E; f(x, dt, out[D]); F(x, dt, out[E*E]); err(nom, delta, out[D]);
// inv_err(nom, tru, out[E]); normalize(x) (quaternion slices, no-op if none).
template <class Model>
__global__ __launch_bounds__(64) void k_rts(const double* __restrict__ xf, const double* __restrict__ Pf,
                                            const double* __restrict__ ts, const int64_t T,
                                            const double* __restrict__ gQ, const int64_t n, const int norm_quats,
                                            double* __restrict__ xs, double* __restrict__ Ps) {
  constexpr int D = Model::D, E = Model::E, EE = E * E, GL = 32, FPW = 2;
  constexpr int DP = D + (D & 1);
  __shared__ __attribute__((aligned(16))) double s_A[FPW * EE];   // Pk_k in, Pk_n out (HBM staging)
  __shared__ __attribute__((aligned(16))) double s_F[FPW * EE];   // Fk
  __shared__ __attribute__((aligned(16))) double s_M[FPW * EE];   // M = Fk Pk_k^T
  __shared__ __attribute__((aligned(16))) double s_L[FPW * EE];   // Pk1_k -> its Cholesky factor
  __shared__ __attribute__((aligned(16))) double s_C[FPW * EE];   // Ck
  __shared__ __attribute__((aligned(16))) double s_D[FPW * EE];   // Pk1_n - Pk1_k
  __shared__ __attribute__((aligned(16))) double s_N[FPW * EE];   // Pk1_n (smoothed covariance of step k+1)
  __shared__ __attribute__((aligned(16))) double s_Q[EE];
  __shared__ __attribute__((aligned(16))) double s_x[FPW * DP];
  __shared__ __attribute__((aligned(16))) double s_d[FPW * E];
  __shared__ __attribute__((aligned(16))) double s_il[FPW * E];

  const int lane = threadIdx.x;
  const int g = lane / GL;
  const int c = lane % GL;
  const bool act = c < E;
  const int cc = act ? c : 0;
  copy_g2l<EE>(gQ, EE, s_Q, lane);
  const int64_t tiles = (n + FPW - 1) / FPW;

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * FPW;
    const int cnt = (n - base) < FPW ? (int)(n - base) : FPW;
    const int gg = g < cnt ? g : 0;
    const bool on = act && g < cnt;
    double* A = s_A + gg * EE;
    double* Fm = s_F + gg * EE;
    double* M = s_M + gg * EE;
    double* L = s_L + gg * EE;
    double* C = s_C + gg * EE;
    double* Dm = s_D + gg * EE;
    double* Nn = s_N + gg * EE;
    double* sd = s_d + gg * E;
    double* sil = s_il + gg * E;

    double xn1[D];          // xk1_n: smoothed state of step k+1 (replicated per lane)
#pragma unroll
    for (int i = 0; i < D; i++) xn1[i] = 0.0;

    for (int64_t k = T - 2; k >= -1; k--) {
      if (k < 0) {
        // the oldest smoothed state goes out un-normalised (ekf_sym.py:665-667 never reaches it)
        if (T >= 1) {
          if (c == 0 && g < cnt) {
#pragma unroll
            for (int i = 0; i < D; i++) s_x[g * D + i] = xn1[i];
          }
          wave_lds_sync();
          if (T >= 2) {
            copy_l2g<FPW * D>(xs + base * D, cnt * D, s_x, lane);
            copy_l2g<FPW * EE>(Ps + base * EE, cnt * EE, s_N, lane);
          }
          wave_lds_sync();
        }
        break;
      }
      // ---- load the filtered pair of step k ---------------------------------------------------------
      copy_g2l<FPW * EE>(Pf + (k * n + base) * EE, cnt * EE, s_A, lane);
      copy_g2l<FPW * D>(xf + (k * n + base) * D, cnt * D, s_x, lane);
      const double dt = ts[k + 1] - ts[k];
      wave_lds_sync();
      double xk[D], x1k[D];
#pragma unroll
      for (int i = 0; i < D; i++) xk[i] = s_x[gg * D + i];
      double prow[E];       // row c of Pk_k
#pragma unroll
      for (int j = 0; j < E; j++) prow[j] = A[cc * E + j];

      // ---- recompute the predicted pair of step k+1 exactly as the forward pass did ----------------------
      const bool first = (k == T - 2);     // recursion start: smoothed(T-1) := predicted(T-1)  (estimates[-1][0], [2])
      if constexpr (Model::SPARSE) {
        // generated sparse predict (F has ~33 non-trivial entries of 484): the same code the forward kernels run
        double mcol[E], p1col[E];
#pragma unroll
        for (int i = 0; i < D; i++) x1k[i] = xk[i];
        Model::predict_cov(x1k, prow, mcol, p1col, L, s_Q, dt, cc, on);        // L <- Pk1_k (full matrix), x1k <- f(xk)
        if (norm_quats & 1) Model::normalize(x1k);
        if (on) {
#pragma unroll
          for (int i = 0; i < E; i++) {
            M[i * E + c] = mcol[i];
            if (first) Nn[i * E + c] = p1col[i];
            Dm[i * E + c] = Nn[i * E + c] - p1col[i];
          }
        }
#pragma unroll
        for (int j = 0; j < E; j++) prow[j] = A[cc * E + j];                 // predict_cov consumed the row registers
      } else {
        Model::f(xk, dt, x1k);
        if (norm_quats & 1) Model::normalize(x1k);
        if (c == 0 && g < cnt) Model::F(xk, dt, Fm);
        wave_lds_sync();
        // M = Fk Pk_k^T : lane c forms column c, M[i][c] = sum_m F[i][m] Pk[c][m]; i loops stay rolled (operands in LDS)
#pragma unroll 1
        for (int i = 0; i < E; i++) {
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < E; m++) s += Fm[i * E + m] * prow[m];
          if (on) M[i * E + c] = s;
        }
        wave_lds_sync();
        // Pk1_k = Fk (Pk_k Fk^T) + dt Q : column c, using row c of M (= column c of Pk_k Fk^T); Dm = Pk1_n - Pk1_k
        double mrow[E];
#pragma unroll
        for (int m = 0; m < E; m++) mrow[m] = M[cc * E + m];
#pragma unroll 1
        for (int i = 0; i < E; i++) {
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < E; m++) s += Fm[i * E + m] * mrow[m];
          const double v = s + dt * s_Q[i * E + cc];
          if (on) {
            L[i * E + c] = v;
            if (first) Nn[i * E + c] = v;
            Dm[i * E + c] = Nn[i * E + c] - v;
          }
        }
      }
      wave_lds_sync();
      if (k == T - 2) {
#pragma unroll
        for (int i = 0; i < D; i++) xn1[i] = x1k[i];
      }
      if (norm_quats & 2) Model::normalize(xn1);
      // smoothed step k+1 is final now: write it out (state after the in-place renormalisation)
      if (c == 0 && g < cnt) {
#pragma unroll
        for (int i = 0; i < D; i++) s_x[g * D + i] = xn1[i];
      }
      wave_lds_sync();
      copy_l2g<FPW * D>(xs + ((k + 1) * n + base) * D, cnt * D, s_x, lane);
      copy_l2g<FPW * EE>(Ps + ((k + 1) * n + base) * EE, cnt * EE, s_N, lane);

      // ---- Cholesky of Pk1_k in LDS (left-looking by columns), lane c owns row c; loops stay rolled -------------
#pragma unroll 1
      for (int j = 0; j < E; j++) {
        double s = L[cc * E + j], s2 = 0.0;
        int m = 0;
#pragma unroll 2
        for (; m + 1 < j; m += 2) {
          s -= L[cc * E + m] * L[j * E + m];
          s2 -= L[cc * E + m + 1] * L[j * E + m + 1];
        }
        if (m < j) s -= L[cc * E + m] * L[j * E + m];
        s += s2;
        if (c == j && g < cnt) {
          const double ljj = sqrt(s);
          L[j * E + j] = ljj;
          sil[j] = 1.0 / ljj;
        }
        wave_lds_sync();
        if (c > j && on) L[c * E + j] = s * sil[j];
        wave_lds_sync();
      }
      // ---- Ck^T = Pk1_k^-1 M : lane c solves for column c IN PLACE in its own column of M -----------------------
#pragma unroll 1
      for (int i = 0; i < E; i++) {
        double s = M[i * E + cc], s2 = 0.0;
        int m = 0;
#pragma unroll 2
        for (; m + 1 < i; m += 2) {
          s -= L[i * E + m] * M[m * E + cc];
          s2 -= L[i * E + m + 1] * M[(m + 1) * E + cc];
        }
        if (m < i) s -= L[i * E + m] * M[m * E + cc];
        if (on) M[i * E + c] = (s + s2) * sil[i];
      }
#pragma unroll 1
      for (int i = E - 1; i >= 0; i--) {
        double s = M[i * E + cc], s2 = 0.0;
        int m = i + 1;
#pragma unroll 2
        for (; m + 1 < E; m += 2) {
          s -= L[m * E + i] * M[m * E + cc];
          s2 -= L[(m + 1) * E + i] * M[(m + 1) * E + cc];
        }
        if (m < E) s -= L[m * E + i] * M[m * E + cc];
        if (on) M[i * E + c] = (s + s2) * sil[i];
      }
      // column c of X = Ck^T is row c of Ck
      double ck[E];
#pragma unroll
      for (int j = 0; j < E; j++) ck[j] = M[j * E + cc];
      if (on) {
#pragma unroll
        for (int j = 0; j < E; j++) C[c * E + j] = ck[j];
      }
      // ---- state: delta = Ck inv_err(xk1_k, xk1_n); xk_n = err(xk_k, delta) -------------------------------
      {
        double delta[E];
        Model::inv_err(x1k, xn1, delta);
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < E; j++) s += ck[j] * delta[j];
        if (on) sd[c] = s;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < E; j++) delta[j] = sd[j];
        Model::err(xk, delta, xn1);          // xk_n, becomes xk1_n of the next (older) step
      }
      // ---- covariance: Pk_n = Pk_k + (Ck Dm) Ck^T, row c -----------------------------------------------
      wave_lds_sync();           // every lane has taken its right-hand side out of M: M becomes the T = Ck Dm buffer
#pragma unroll 1
      for (int m = 0; m < E; m++) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};      // four independent chains: one accumulator would serialise 22 fp64 FMAs
#pragma unroll
        for (int j = 0; j < E; j++) acc[j & 3] += ck[j] * Dm[j * E + m];
        if (on) M[c * E + m] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      }
      wave_lds_sync();
      {
        double trow[E];
#pragma unroll
        for (int j = 0; j < E; j++) trow[j] = M[cc * E + j];
#pragma unroll 1
        for (int m = 0; m < E; m++) {
          double acc[4] = {A[cc * E + m], 0.0, 0.0, 0.0};
#pragma unroll
          for (int j = 0; j < E; j++) acc[j & 3] += trow[j] * C[m * E + j];
          if (on) Nn[c * E + m] = (acc[0] + acc[1]) + (acc[2] + acc[3]);   // Pk1_n of the next (older) step
        }
      }
      wave_lds_sync();
    }
    // T == 1: nothing to smooth, the single estimate's predicted pair is not available -> copy filtered through
    if (T == 1) {
      copy_g2l<FPW * EE>(Pf + base * EE, cnt * EE, s_A, lane);
      copy_g2l<FPW * D>(xf + base * D, cnt * D, s_x, lane);
      wave_lds_sync();
      copy_l2g<FPW * EE>(Ps + base * EE, cnt * EE, s_A, lane);
      copy_l2g<FPW * D>(xs + base * D, cnt * D, s_x, lane);
      wave_lds_sync();
    }
  }
}

// ---- lane-group models with a generated sparse predict (Model::SPARSE): register-resident solves -----------------------
// Same recursion and quirks as k_rts below; what differs is where the work lives.  Lane c keeps its row of the Cholesky
// factor, its right-hand side / solution column and its rows of T = Ck (Pk1_n - Pk1_k) and Pk_n in REGISTERS; LDS holds
// only what other lanes must see (the factor, Ck, the difference matrix, Pk1_n) and every LDS read is a broadcast or a
// lane-private row, issued from fully unrolled code so that none of them sits on a dependent chain:
//   * Cholesky, left-looking: at column j every lane reads pivot row j (final since step j-1) and forms BOTH its own
//     entry and the pivot redundantly -> no publish / wait round trip per column (the rolled version paid two wave
//     syncs + sqrt + division per column); reciprocal square root by v_rsq_f64 + Newton.
//   * forward / back substitution, right-looking on the register column: as soon as y[m] is final, all later entries are
//     updated by independent FMAs.
//   * the filtered pair of step k-1 streams HBM -> LDS (global_load_lds) while step k computes; Pk_k's row is taken to
//     registers first, so one buffer suffices.
// LDS: 4 E x E matrices per filter (36 KB per wave for E = 22, 4 waves per CU) instead of 7 (51 KB, 3 waves).
// Measured (live, 16 384 filters): 26.5 us per step and wavefront against ~42 us for k_rts; timing with phases removed
// puts ~20 us of that in the factorisation -- a serial chain of (j + 10) DEPENDENT fp64 operations per column j
// (dot product, v_rsq_f64 + two Newton steps, scaling) plus an LDS write -> broadcast-read turnaround, on a wavefront
// that is alone on its SIMD (LDS holds 8 filters per CU).  Things that did NOT help, measured: 4 partial sums per dot
// product (+12 %: more instructions, same chain through the pivot), __builtin_amdgcn_sched_barrier / asm memory fences /
// never-taken aliasing stores between unrolled iterations (each sends the register allocator to 6 KB of scratch per lane),
// per-column predicates (one exec mask per column is hoisted out of the step loop: 190 SGPR spills; a single predicate
// and harmless stores above the pivot brought that to 61).  A variant with only two LDS matrices per filter (rows of
// Pk1_n carried in registers, the prefetch landing in the dead difference matrix: 18 KB per wave, enough for two waves
// per SIMD) needs <= 256 registers for that, and the generated predict alone keeps ~270 live: 1 264 spilled VGPRs under
// every -amdgpu-sched-strategy, so it was dropped.  A later variant -- right-looking factorisation with redundantly tracked
// diagonals, the two products as rolled loops with LDS-resident multipliers, and the factor read through an LDS pointer
// "redefined" by an empty asm once per column (which stops hipcc from hoisting the substitution loads: 0 spills, 256 + 156
// registers) -- ran at 21.2 us per step and wavefront (96 M steps/s) but returned wrong states for SOME inputs of the
// 24-error-state random model (tools/lds_poison.hip + a numpy restatement found it; 11, 13, 17 and 22 states were fine).
// The pinned LDS pointer alone, on this otherwise unchanged kernel (0 spills, 23.4 us per step and wavefront, 87 M steps/s),
// reproduces exactly that failure -- wrong for 24 states, right for the others -- so it is the asm-pinned address-space-3
// pointer that miscompiles there (cause not found), and this version, spills and all, stays.
template <class Model>
__global__ __launch_bounds__(64) void k_rts_wide(const double* __restrict__ xf, const double* __restrict__ Pf,
                                                 const double* __restrict__ ts, const int64_t T,
                                                 const double* __restrict__ gQ, const int64_t n, const int norm_quats,
                                                 double* __restrict__ xs, double* __restrict__ Ps) {
  constexpr int D = Model::D, E = Model::E, EE = E * E, GL = 32, FPW = 2;
  constexpr int DP = D + (D & 1);
  constexpr int ABUF = (FPW * EE + 3) / 2 * 2, XBUF = (FPW * D + 3) / 2 * 2;
  __shared__ __attribute__((aligned(16))) double s_A[ABUF];       // Pk_k staging, prefetched one step ahead
  __shared__ __attribute__((aligned(16))) double s_L[FPW * EE];   // Pk1_k -> its Cholesky factor -> Ck
  __shared__ __attribute__((aligned(16))) double s_D[FPW * EE];   // Pk1_n - Pk1_k
  __shared__ __attribute__((aligned(16))) double s_N[FPW * EE];   // Pk1_n; also the staging of the smoothed output
  __shared__ __attribute__((aligned(16))) double s_Q[EE];
  __shared__ __attribute__((aligned(16))) double s_xin[XBUF];     // xk_k staging, prefetched
  __shared__ __attribute__((aligned(16))) double s_x[FPW * DP];   // smoothed state going out
  __shared__ __attribute__((aligned(16))) double s_xk[2 * FPW * DP];  // xk_k and xk1_k parked during the factorisation
  __shared__ __attribute__((aligned(16))) double s_xn[FPW * DP];      // xk1_n, the smoothed state of step k+1
  __shared__ __attribute__((aligned(16))) double s_il[FPW * E];       // reciprocal pivots of the factor
  __shared__ __attribute__((aligned(16))) double s_d[FPW * E];

  const int lane = threadIdx.x;
  const int g = lane / GL;
  const int c = lane % GL;
  const bool act = c < E;
  const int cc = act ? c : 0;
  copy_g2l<EE>(gQ, EE, s_Q, lane);
  const int64_t tiles = (n + FPW - 1) / FPW;

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * FPW;
    const int cnt = (n - base) < FPW ? (int)(n - base) : FPW;
    const int gg = g < cnt ? g : 0;
    const bool on = act && g < cnt;
    double* L = s_L + gg * EE;
    double* Dm = s_D + gg * EE;
    double* Nn = s_N + gg * EE;
    double* sd = s_d + gg * E;
    double* sxk = s_xk + gg * 2 * DP;
    double* sxn = s_xn + gg * DP;
    double* sil = s_il + gg * E;

    if (T >= 2) {
      async_copy_g2l_any<ABUF>(Pf + ((T - 2) * n + base) * EE, cnt * EE, s_A, lane);
      async_copy_g2l_any<XBUF>(xf + ((T - 2) * n + base) * D, cnt * D, s_xin, lane);
    }
    for (int64_t k = T - 2; k >= 0; k--) {
      // ---- the filtered pair of step k has landed: row c of Pk_k and xk_k to registers, then reuse the buffers ------
      const int shA = odd_start(Pf + (k * n + base) * EE), shx = odd_start(xf + (k * n + base) * D);
      const double dt = ts[k + 1] - ts[k];
      async_wait();
      wave_lds_sync();
      double xk[D], x1k[D], prow[E];
#pragma unroll
      for (int i = 0; i < D; i++) xk[i] = s_xin[shx + gg * D + i];
#pragma unroll
      for (int j = 0; j < E; j++) prow[j] = s_A[shA + gg * EE + cc * E + j];
      wave_lds_sync();
      if (k >= 1) {
        async_copy_g2l_any<ABUF>(Pf + ((k - 1) * n + base) * EE, cnt * EE, s_A, lane);
        async_copy_g2l_any<XBUF>(xf + ((k - 1) * n + base) * D, cnt * D, s_xin, lane);
      }

      // ---- predicted pair of step k+1, recomputed with the forward kernels' generated sparse predict -------------------
      const bool first = (k == T - 2);     // recursion start: smoothed(T-1) := predicted(T-1)  (estimates[-1][0], [2])
      double y[E];
      {
        double p1col[E], pk[E], xn1[D];
#pragma unroll
        for (int j = 0; j < E; j++) pk[j] = prow[j];
#pragma unroll
        for (int i = 0; i < D; i++) x1k[i] = xk[i];
        Model::predict_cov(x1k, prow, y, p1col, L, s_Q, dt, cc, on);        // L <- Pk1_k, y <- column c of M = Fk Pk_k^T
        if (norm_quats & 1) Model::normalize(x1k);
        if (on) {
#pragma unroll
          for (int i = 0; i < E; i++) {
            if (first) Nn[i * E + c] = p1col[i];
            Dm[i * E + c] = Nn[i * E + c] - p1col[i];
          }
        }
#pragma unroll
        for (int i = 0; i < D; i++) xn1[i] = first ? x1k[i] : sxn[i];
        if (norm_quats & 2) Model::normalize(xn1);
        wave_lds_sync();
        // smoothed step k+1 is final now: write it out (state after the in-place renormalisation).  The replicated state
        // vectors wait in LDS until the state update (in registers they would hold ~140 VGPRs through the factorisation)
        if (c == 0 && g < cnt) {
#pragma unroll
          for (int i = 0; i < D; i++) { s_x[g * D + i] = xn1[i]; sxn[i] = xn1[i]; sxk[i] = xk[i]; sxk[DP + i] = x1k[i]; }
        }
        wave_lds_sync();
        copy_l2g<FPW * D>(xs + ((k + 1) * n + base) * D, cnt * D, s_x, lane);
        copy_l2g<FPW * EE>(Ps + ((k + 1) * n + base) * EE, cnt * EE, s_N, lane);
        // Pk1_n has been consumed (difference formed, output issued): its row c now parks row c of Pk_k until the end of
        // the step, where the smoothed Pk_n is accumulated on top of it
        wave_lds_sync();
        if (on) {
#pragma unroll
          for (int j = 0; j < E; j++) Nn[c * E + j] = pk[j];
        }
      }

      // ---- Cholesky of Pk1_k: lane c owns row c in registers, pivot rows are broadcast from LDS ---------------------------
      double lrow[E];
#pragma unroll
      for (int j = 0; j < E; j++) lrow[j] = L[cc * E + j];
#pragma unroll
      for (int j = 0; j < E; j++) {
        double s = lrow[j], sj = L[j * E + j];
#pragma unroll
        for (int m = 0; m < j; m++) {
          const double r = L[j * E + m];          // final: written by lane j at step m
          s = fma(-lrow[m], r, s);
          sj = fma(-r, r, sj);
        }
        const double ilj = fast_rsqrt(sj);
        lrow[j] = (c == j) ? sj * ilj : s * ilj;
        // every row publishes its entry (rows above the pivot write a value nobody reads: one predicate for the whole
        // factorisation instead of one exec mask per column) and the pivot reciprocal, identical in all lanes of the group;
        // LDS is in-order within the wave, so the next column's broadcast reads see these stores
        if (on) { L[c * E + j] = lrow[j]; sil[j] = ilj; }
      }
      wave_lds_sync();
      // ---- Ck^T = Pk1_k^-1 M: lane c solves for column c in registers (right-looking substitutions) -------------------
#pragma unroll
      for (int m = 0; m < E; m++) {
        y[m] *= sil[m];
#pragma unroll
        for (int i = m + 1; i < E; i++) y[i] = fma(-L[i * E + m], y[m], y[i]);
      }
#pragma unroll
      for (int m = E - 1; m >= 0; m--) {
        y[m] *= sil[m];
#pragma unroll
        for (int i = 0; i < m; i++) y[i] = fma(-L[m * E + i], y[m], y[i]);
      }
      // y is column c of X = Ck^T, i.e. row c of Ck; the factor is dead: its buffer takes Ck^T (column c written by lane c)
      wave_lds_sync();
      if (on) {
#pragma unroll
        for (int j = 0; j < E; j++) L[j * E + c] = y[j];
      }
      // ---- state: delta = Ck inv_err(xk1_k, xk1_n); xk_n = err(xk_k, delta) -------------------------------
      {
        double delta[E], xa[D], xb[D], xn1[D];
#pragma unroll
        for (int i = 0; i < D; i++) { xa[i] = sxk[i]; xb[i] = sxk[DP + i]; xn1[i] = sxn[i]; }
        Model::inv_err(xb, xn1, delta);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < E; j += 2) { s0 = fma(y[j], delta[j], s0); s1 = fma(y[j + 1], delta[j + 1], s1); }
        if (E & 1) s0 = fma(y[E - 1], delta[E - 1], s0);
        if (on) sd[c] = s0 + s1;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < E; j++) delta[j] = sd[j];
        Model::err(xa, delta, xn1);          // xk_n, becomes xk1_n of the next (older) step
        if (c == 0 && g < cnt) {
#pragma unroll
          for (int i = 0; i < D; i++) sxn[i] = xn1[i];
        }
      }
      // ---- covariance: Pk_n = Pk_k + (Ck Dm) Ck^T, row c in registers ---------------------------------------
      double trow[E];
#pragma unroll
      for (int m = 0; m < E; m++) trow[m] = 0.0;
#pragma unroll
      for (int j = 0; j < E; j++) {
#pragma unroll
        for (int m = 0; m < E; m++) trow[m] = fma(y[j], Dm[j * E + m], trow[m]);     // row j of Dm: contiguous broadcast
      }
      double nn[E];
#pragma unroll
      for (int m = 0; m < E; m++) nn[m] = Nn[cc * E + m];
#pragma unroll
      for (int j = 0; j < E; j++) {
#pragma unroll
        for (int m = 0; m < E; m++) nn[m] = fma(trow[j], L[j * E + m], nn[m]);      // row j of Ck^T: contiguous broadcast
      }
      if (on) {
#pragma unroll
        for (int m = 0; m < E; m++) Nn[c * E + m] = nn[m];   // Pk1_n of the next (older) step
      }
      wave_lds_sync();
    }
    // the oldest smoothed state goes out un-normalised (ekf_sym.py:665-667 never reaches it)
    if (T >= 2) {
      wave_lds_sync();
      if (c == 0 && g < cnt) {
#pragma unroll
        for (int i = 0; i < D; i++) s_x[g * D + i] = sxn[i];
      }
      wave_lds_sync();
      copy_l2g<FPW * D>(xs + base * D, cnt * D, s_x, lane);
      copy_l2g<FPW * EE>(Ps + base * EE, cnt * EE, s_N, lane);
      wave_lds_sync();
    }
    // T == 1: nothing to smooth, the single estimate's predicted pair is not available -> copy filtered through
    if (T == 1) {
      copy_g2l<FPW * EE>(Pf + base * EE, cnt * EE, s_L, lane);
      copy_g2l<FPW * D>(xf + base * D, cnt * D, s_x, lane);
      wave_lds_sync();
      copy_l2g<FPW * EE>(Ps + base * EE, cnt * EE, s_L, lane);
      copy_l2g<FPW * D>(xs + base * D, cnt * D, s_x, lane);
      wave_lds_sync();
    }
  }
}

}  // namespace rn
