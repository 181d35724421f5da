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
// Steps with dt == 0 (Model::ID0: models whose predict(dt = 0) is the identity symbolically -- f(x, 0) == x, F(x, 0) == I --, tuning knob
// rts_dt0).  There the predicted pair IS the filtered one, Pk1_k = Pk_k bit for bit, and Ck = solve(Pk1_k, Pk_k^T)^T is the identity up to the
// rounding of the solve (2e-9 in the reference's own live golden, np.linalg.solve).  Every kernel here takes Ck = I on such a step:
//     xk_n = err(xk_k, inv_err(xk1_k, xk1_n))   (NOT xk1_n: err o inv_err is not the identity for finite rotations),   Pk_n = Pk_k + (Pk1_n - Pk1_k)
// -- no factorisation, no substitutions, no products.  In an IMU + GNSS stream with several observations per tick that is half the steps.  The
// recursion's first step always takes the full path.  (The register-broadcast smoother of the lane-group models, codegen/emit_rts4.py, does the same.)
//
// Two kernels: k_rts for lane-per-filter models (the first mapping written: 2 filters per wavefront, every E x E matrix in
// LDS, rolled loops -- these models have at most 7 error states, 4 - 7 wavefronts per SIMD) and k_rts_group for lane-group
// models (see its header).  No MFMA: the blocks are <= 64 x 64 fp64 per filter and on CDNA4 the fp64 matrix rate equals the
// vector rate.
#pragma once

#include "ekf_hip_rt.h"

namespace rn {

// Model must provide: D, E; f(x, dt, out[D]); F(x, dt, out[E*E]); err(nom, delta, out[D]);
// inv_err(nom, tru, out[E]); normalize(x) (quaternion slices, no-op if none).
template <class Model>
__global__ __launch_bounds__(64) void k_rts(const double* __restrict__ xf, const double* __restrict__ Pf,
                                            const double* __restrict__ ts, const int64_t T,
                                            const double* __restrict__ gQ, const int64_t n, const int norm_quats,
                                            double* __restrict__ xs, double* __restrict__ Ps,
                                            const double* __restrict__ xl, const double* __restrict__ Pl) {
  // xl / Pl (optional): the predicted pair of the last step, returned verbatim as the newest smoothed estimate (ekf_sym.py:658-659)
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
      // Contract of batch_rts (include/rednose_amd_filter.h): the LOWER triangle of every covariance handed in is read, mirrored --
      // the recursion factors Pk1_k by Cholesky and the lane-group kernels keep D = Pk1_n - Pk1_k as a packed triangle, so all
      // three smoother kernels take P[r][j] for j <= r and P[j][r] above the diagonal.  (The reference solves with general
      // matrices, ekf_sym.py:677,686; on the symmetric traces a forward pass produces the two agree to rounding.)
      double prow[E];       // row c of Pk_k
#pragma unroll
      for (int j = 0; j < E; j++) prow[j] = (j <= cc) ? A[cc * E + j] : A[j * E + cc];

      // ---- recompute the predicted pair of step k+1 exactly as the forward pass did ----------------------
      const bool first = (k == T - 2);     // recursion start: smoothed(T-1) := predicted(T-1)  (estimates[-1][0], [2])
      {
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
            if (first) Nn[i * E + c] = (Pl != nullptr) ? Pl[(base + gg) * EE + i * E + c] : v;
            Dm[i * E + c] = Nn[i * E + c] - v;
          }
        }
      }
      wave_lds_sync();
      if (on) {             // D from its lower triangle (Pk1_n = Pk_k + correction carries whatever asymmetry the caller's Pk_k had)
#pragma unroll
        for (int i = 0; i < E; i++) {
          if (i < c) Dm[i * E + c] = Dm[c * E + i];
        }
      }
      wave_lds_sync();
      if (k == T - 2) {
#pragma unroll
        for (int i = 0; i < D; i++) xn1[i] = (xl != nullptr) ? xl[(base + gg) * D + i] : x1k[i];
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

      if (Model::ID0 && dt == 0.0 && !first) {
        // identity-gain step (see the head of this file): Ck = I.  Dm is symmetric by now (mirrored from its lower triangle above); the filtered
        // covariance enters the sum as stored.  dt is a scalar of the launch: the branch is uniform.
        double delta[E];
        Model::inv_err(x1k, xn1, delta);
        Model::err(xk, delta, xn1);          // xk_n, becomes xk1_n of the next (older) step
        double nrow[E];
#pragma unroll
        for (int m = 0; m < E; m++) nrow[m] = A[cc * E + m] + Dm[cc * E + m];
        wave_lds_sync();           // copy_l2g above has read s_N
        if (on) {
#pragma unroll
          for (int m = 0; m < E; m++) Nn[c * E + m] = nrow[m];      // Pk1_n of the next (older) step
        }
        wave_lds_sync();
        continue;
      }

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

// =====================================================================================================================
// k_rts_group -- smoother for lane-group models (8 .. 64 error states in the main block), second mapping.
//
// What bounded k_rts_wide (above) was not arithmetic: 4 E x E matrices per filter in LDS (36 KB per wavefront for 22 error
// states) left one wavefront per SIMD, and the fully unrolled register-resident algebra needed 256 + 256 registers and still
// spilled 129 of them -- a scratch access costs a lone wavefront about a microsecond.  This kernel is built around the two
// budgets instead:
//   * LDS: TWO main-block matrices per filter.  s_B holds Pk1_k, is factored in place, and finally takes Ck^T; s_C holds the
//     difference Pk1_n - Pk1_k.  The filtered covariance never passes through LDS: lane c loads row c of Pk_k straight from HBM
//     into registers (a row is contiguous; 16-byte loads when rows are 16-byte aligned) and stores row c of the smoothed
//     covariance the same way.  Q is read from global memory (L1/L2 hits; 1 column per lane and step).
//     22 error states: 15.5 KB of matrices + ~2 KB of vectors per wavefront -> 8 wavefronts per CU.
//   * registers: <= 256 up to 22 error states (__launch_bounds__(64, 2), two wavefronts per SIMD; about six row vectors are
//     live at the widest point, so larger models get the full file and one wavefront per SIMD: Model::WAVES).  Per lane only ONE row / column of a matrix is
//     live at a time next to the right-hand side y: the state-dependent scalars (f, F non-zeros) are evaluated by one lane per
//     filter into an LDS slot -- the phase-1 function of the three-phase step kernels (emit_wide2.scal_predict) -- instead of
//     being replicated in every lane, and the state vectors live in LDS between the places that use them.
// The recursion, its quirks (start from the PREDICTED pair of the last step, in-place renormalisation of every xk1_n) and the
// norm_quats bit mask are those of k_rts.  MSCKF models (ekf_sym.py:675-686): only the main block [:EM, :EM] of the covariance and
// the main states [:DM] are smoothed; everything else of (xk_k, Pk_k) passes through.  The main block of the predicted pair is
// recomputed from the filtered main block (F_main Pk_k[:EM, :EM] F_main^T + dt Q[:EM, :EM], ekf_c.c:24,28), which assumes what
// an MSCKF means: main dynamics that do not read the window clones.  The predicted pair of the LAST step, which the reference
// returns verbatim as the newest smoothed estimate, can be passed in (xl, Pl); otherwise its main block is recomputed as well:
// exact for models without a window; for MSCKF models the window states and the window / cross-covariance blocks of that one
// estimate (index T - 1) are then the FILTERED ones of the trace (F_main P[main, window] is not formed here) -- a predicted
// main block next to filtered cross blocks, so callers that need the newest estimate exactly pass (xl, Pl).
//
// Model: constants D, E, DM, EM, SLOT, OFF_X, OFF_DT and the functions
//   scal(xin, dt, sl, norm)                      one filter's f / F non-zeros -> slot (x' = f(x) [normalised] at sl[OFF_X..])
//   mat_predict(row, sB, gQc, sl, cc, act, y)    row c of Pk_k, gQc = column cc of Q -> y = column c of M = F Pk_k^T, sB <- Pk1_k (EM x EM)
//   inv_err, err, normalize                      as for k_rts

template <int E, int EM>
__device__ __forceinline__ void rts_load_row(const double* __restrict__ p, double (&r)[EM]) {
  if constexpr (E % 2 == 0 && EM % 2 == 0) {
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int j = 0; j < EM / 2; j++) { const double2 v = p2[j]; r[2 * j] = v.x; r[2 * j + 1] = v.y; }
  } else {
#pragma unroll
    for (int j = 0; j < EM; j++) r[j] = p[j];
  }
}

// Row c of the covariance READ FROM ITS LOWER TRIANGLE (the contract of batch_rts, see k_rts): entries j <= c from the row itself,
// entries above the diagonal from column c -- for a fixed j the lanes of a group read consecutive doubles of row j.
template <int E, int EM>
__device__ __forceinline__ void rts_load_row_lower(const double* __restrict__ prow_ptr, const double* __restrict__ pcol_ptr, const int c,
                                                   double (&r)[EM]) {
  // one 8-byte load per entry from the selected side (a value select between a row load and a column load trips an hipcc 7.2
  // backend assertion in some instantiations: "V_CMP_NE_U32 $src_shared_base: incorrect register class")
#pragma unroll
  for (int j = 0; j < EM; j++) {
    const int64_t off = (j <= c) ? (int64_t)j : (int64_t)j * E + (pcol_ptr - prow_ptr);
    r[j] = prow_ptr[off];
  }
}

template <int E, int EM>
__device__ __forceinline__ void rts_store_row(double* __restrict__ p, const double (&r)[EM]) {
  if constexpr (E % 2 == 0 && EM % 2 == 0) {
    double2* __restrict__ p2 = reinterpret_cast<double2*>(p);
#pragma unroll
    for (int j = 0; j < EM / 2; j++) { double2 v; v.x = r[2 * j]; v.y = r[2 * j + 1]; p2[j] = v; }
  } else {
#pragma unroll
    for (int j = 0; j < EM; j++) p[j] = r[j];
  }
}

// The two products of one filter on the matrix cores (see phase H of k_rts_group): in  Bf = X = Ck^T (X[k][i] = Ck[i][k]),
// Cf = Dm;  out Cf = Pk_n = Pk_k + (Ck Dm) Ck^T;  Bf is scratch afterwards.  All 64 lanes of the wavefront work on this filter.
// The result is produced in blocks of JB column tiles (JB = 2: at most 2 TI accumulator tiles live, 16 TI registers):
//   T[:, jb] = Ck Dm[:, jb] depends on no other column of Dm, so it overwrites Dm[:, jb] in place;
//   Pk_n[:, jb] = Pk_k[:, jb] + T X[:, jb] needs all of T but only X[:, jb], so it overwrites X[:, jb]; with more than one block
//   the result is then moved from Bf to Cf (with one block it goes to Cf directly, all of T having been read).
template <int E, int EM>
__device__ __forceinline__ void rts_products(double* Bf, double* Cf, const double* __restrict__ Pkf, const int lane) {
  constexpr int TI = (EM + 15) / 16, KS = (EM + 3) / 4, JB = TI <= 2 ? TI : 2, NB = (TI + JB - 1) / JB;
  typedef double v4d __attribute__((ext_vector_type(4)));
  const int li = lane & 15, lk = lane >> 4;
  // Pk_k in the result layout (the start value of the second product): with a single block it is requested now, so that its L2
  // latency passes under the first product
  v4d pk[NB == 1 ? TI : 1][NB == 1 ? JB : 1];
  if constexpr (NB == 1) {
#pragma unroll
    for (int it = 0; it < TI; it++) {
#pragma unroll
      for (int jt = 0; jt < JB; jt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = 16 * it + 4 * r + lk, j = 16 * jt + li;
          pk[it][jt][r] = (i < EM && j < EM) ? Pkf[i * E + j] : 0.0;
        }
      }
    }
  }
  // ---- T = Ck Dm ------------------------------------------------------------------------------------------------------------
  static_for<NB>([&](auto JBI) {
    constexpr int j0 = decltype(JBI)::value * JB;
    constexpr int nj = (TI - j0) < JB ? (TI - j0) : JB;
    v4d acc[TI][JB];
#pragma unroll
    for (int it = 0; it < TI; it++) {
#pragma unroll
      for (int jt = 0; jt < nj; jt++) acc[it][jt] = v4d{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int kk = 4 * ks + lk;
      double a[TI], b[JB];
#pragma unroll
      for (int t = 0; t < TI; t++) {
        const int i = 16 * t + li;
        const bool ok = kk < EM && i < EM;
        a[t] = ok ? Bf[ok ? kk * EM + i : 0] : 0.0;              // Ck[i][k]
      }
#pragma unroll
      for (int u = 0; u < nj; u++) {
        const int j = 16 * (j0 + u) + li;
        const bool ok = kk < EM && j < EM;
        b[u] = ok ? Cf[ok ? kk * EM + j : 0] : 0.0;              // Dm[k][j]
      }
#pragma unroll
      for (int it = 0; it < TI; it++) {
#pragma unroll
        for (int jt = 0; jt < nj; jt++) acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[it], b[jt], acc[it][jt], 0, 0, 0);
      }
    }
    wave_lds_sync();           // these columns of Dm have been read: they take the same columns of T
#pragma unroll
    for (int it = 0; it < TI; it++) {
#pragma unroll
      for (int jt = 0; jt < nj; jt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = 16 * it + 4 * r + lk, j = 16 * (j0 + jt) + li;
          if (i < EM && j < EM) Cf[i * EM + j] = acc[it][jt][r];
        }
      }
    }
    wave_lds_sync();
  });
  // ---- Pk_n = Pk_k + T Ck^T ---------------------------------------------------------------------------------------------------
  static_for<NB>([&](auto JBI) {
    constexpr int j0 = decltype(JBI)::value * JB;
    constexpr int nj = (TI - j0) < JB ? (TI - j0) : JB;
    v4d acc[TI][JB];
#pragma unroll
    for (int it = 0; it < TI; it++) {
#pragma unroll
      for (int jt = 0; jt < nj; jt++) {
        if constexpr (NB == 1) {
          acc[it][jt] = pk[it][jt];
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = 16 * it + 4 * r + lk, j = 16 * (j0 + jt) + li;
            acc[it][jt][r] = (i < EM && j < EM) ? Pkf[i * E + j] : 0.0;
          }
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int kk = 4 * ks + lk;
      double a[TI], b[JB];
#pragma unroll
      for (int t = 0; t < TI; t++) {
        const int i = 16 * t + li;
        const bool ok = kk < EM && i < EM;
        a[t] = ok ? Cf[ok ? i * EM + kk : 0] : 0.0;              // T[i][k]
      }
#pragma unroll
      for (int u = 0; u < nj; u++) {
        const int j = 16 * (j0 + u) + li;
        const bool ok = kk < EM && j < EM;
        b[u] = ok ? Bf[ok ? kk * EM + j : 0] : 0.0;              // X[k][j] = Ck[j][k]
      }
#pragma unroll
      for (int it = 0; it < TI; it++) {
#pragma unroll
        for (int jt = 0; jt < nj; jt++) acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[it], b[jt], acc[it][jt], 0, 0, 0);
      }
    }
    wave_lds_sync();           // one block: all of T has been read; several: these columns of X have been read
    double* dst = NB == 1 ? Cf : Bf;
#pragma unroll
    for (int it = 0; it < TI; it++) {
#pragma unroll
      for (int jt = 0; jt < nj; jt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = 16 * it + 4 * r + lk, j = 16 * (j0 + jt) + li;
          if (i < EM && j < EM) dst[i * EM + j] = acc[it][jt][r];
        }
      }
    }
    wave_lds_sync();
  });
  if constexpr (NB > 1) {
    for (int i = lane; i < EM * EM; i += 64) Cf[i] = Bf[i];
    wave_lds_sync();
  }
}

// Debug phase timeline (builds with -DRN_RTS_TL, tuning knob wide_timeline): lane 0 of the first 256 workgroups stamps the
// 100 MHz wall clock at the phase boundaries of the backward step k == T / 2 of its first tile (tools/timeline.py).
#ifdef RN_RTS_TL
__device__ unsigned long long g_rts_tl[256 * 16];
#define RN_RTS_STAMP(i) do { if (lane == 0 && blockIdx.x < 256 && tile == blockIdx.x && k == T / 2) rn::g_rts_tl[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define RN_RTS_STAMP(i) do { } while (0)
#endif

template <class Model>
__global__ __launch_bounds__(64, Model::WAVES) void k_rts_group(const double* __restrict__ xf, const double* __restrict__ Pf,
                                                     const double* __restrict__ ts, const int64_t T,
                                                     const double* __restrict__ gQ, const int64_t n, const int norm_quats,
                                                     double* __restrict__ xs, double* __restrict__ Ps,
                                                     const double* __restrict__ xl, const double* __restrict__ Pl) {
  constexpr int D = Model::D, E = Model::E, EE = E * E, DM = Model::DM, EM = Model::EM, MM = EM * EM;
  constexpr int GL = EM <= 16 ? 16 : (EM <= 32 ? 32 : 64), FPW = 64 / GL;
  constexpr int SLOT = Model::SLOT;
  constexpr int MMP = MM + (MM & 1), DP = D + (D & 1), EP = E + (E & 1), EMP = EM + (EM & 1);
  __shared__ __attribute__((aligned(16))) double s_B[FPW * MMP];    // Pk1_k -> Cholesky factor (lower) -> Ck^T
  __shared__ __attribute__((aligned(16))) double s_C[FPW * MMP];    // Pk1_n - Pk1_k
  __shared__ __attribute__((aligned(16))) double s_sl[FPW * SLOT];  // per-filter scalars of the predict (x' at OFF_X)
  __shared__ __attribute__((aligned(16))) double s_xk[FPW * DP];    // xk_k
  __shared__ __attribute__((aligned(16))) double s_xn[FPW * DP];    // xk1_n, then xk_n
  __shared__ __attribute__((aligned(16))) double s_de[FPW * EP];    // inv_err(xk1_k, xk1_n)
  __shared__ __attribute__((aligned(16))) double s_dx[FPW * EP];    // Ck delta
  __shared__ __attribute__((aligned(16))) double s_il[FPW * EMP];   // reciprocal pivots of the factor

  const int lane = threadIdx.x;
  const int g = lane / GL;
  const int c = lane % GL;
  const bool act = c < EM;
  const int cc = act ? c : 0;
  const int64_t tiles = (n + FPW - 1) / FPW;
  const bool inplace_P = (Ps == Pf), inplace_x = (xs == xf);

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t base = tile * FPW;
    const int cnt = (n - base) < FPW ? (int)(n - base) : FPW;
    const int gg = g < cnt ? g : 0;
    const bool on = act && g < cnt;
    const bool lead = (c == 0) && g < cnt;           // the one lane per filter that does the state algebra
    const int64_t fil = base + gg;
    double* B = s_B + gg * MMP;
    double* C = s_C + gg * MMP;
    double* sl = s_sl + gg * SLOT;
    double* sxk = s_xk + gg * DP;
    double* sxn = s_xn + gg * DP;
    double* sde = s_de + gg * EP;
    double* sdx = s_dx + gg * EP;
    double* sil = s_il + gg * EMP;

    // what is not smoothed passes through (MSCKF window blocks; nothing for ordinary models): done up front for the whole
    // trace of this tile, by the tile's own wavefront, before any of its main blocks is written
    if constexpr (EM < E || DM < D) {
      if (!inplace_P || !inplace_x) {
        for (int64_t k = 0; k < T; k++) {
          if (!inplace_P) {
            for (int i = lane; i < cnt * EE; i += 64) {
              const int e = i % EE, r = e / E, q = e % E;
              if (r >= EM || q >= EM) Ps[(k * n + base) * EE + i] = Pf[(k * n + base) * EE + i];
            }
          }
          if (!inplace_x) {
            for (int i = lane; i < cnt * D; i += 64) {
              if (i % D >= DM) xs[(k * n + base) * D + i] = xf[(k * n + base) * D + i];
            }
          }
        }
      }
    }

    for (int64_t k = T - 2; k >= 0; k--) {
      const bool first = (k == T - 2);
      // ---- A. filtered pair of step k: row c of Pk_k to registers, xk_k to LDS --------------------------------------
      const double* Pk = Pf + ((k * n + fil) * EE + (int64_t)cc * E);
      double y[EM];
      const double dt = ts[k + 1] - ts[k];
      RN_RTS_STAMP(0);
      {
        // row c of Pk_k is requested first: its HBM / L2 latency passes under the scalar phase, which
        // keeps one lane per filter busy for about a microsecond (up to 32 error states; beyond, the registers are not there)
        double prow[EM];
        if constexpr (EM <= 32) rts_load_row_lower<E, EM>(Pk, Pf + ((k * n + fil) * EE + cc), cc, prow);
        for (int i = c; i < D; i += GL) sxk[i] = xf[(k * n + fil) * D + i];
        wave_lds_sync();
        // ---- B. f(xk_k), non-zeros of Fk: once per filter -> slot ----------------------------------------------------------
        RN_RTS_STAMP(1);
        if (lead) Model::scal(sxk, dt, sl, norm_quats & 1);
        wave_lds_sync();
        RN_RTS_STAMP(2);
        // ---- C. predicted pair of step k+1 (main block): B <- Pk1_k, y <- column c of M = Fk Pk_k^T -----------------------
        if constexpr (EM > 32) rts_load_row_lower<E, EM>(Pk, Pf + ((k * n + fil) * EE + cc), cc, prow);
        // column cc of Q behind an opaque zero: without it hipcc hoists one 64-bit address per entry out of the step loop
        // (2 EM registers for the whole kernel; beyond the immediate-offset range they cannot share a base) and spills them
        int qz = 0;
        asm volatile("" : "+v"(qz));
        Model::mat_predict(prow, B, gQ + cc + qz, sl, cc, on, y);
      }
      // ---- D. recursion start / difference matrix / smoothed estimate of step k+1 leaves ------------------------------------
      RN_RTS_STAMP(3);
      // Row c of Pk1_n waited in this lane's row of the difference buffer since the end of the previous step (no other lane
      // touches it): no matrix row stays in registers across the phases above.
      double lrow[EM], nrow[EM];
#pragma unroll
      for (int j = 0; j < EM; j++) { lrow[j] = B[cc * EM + j]; nrow[j] = C[cc * EM + j]; }
      if (first) {
        if (Pl != nullptr) {
          rts_load_row<E, EM>(Pl + (fil * EE + (int64_t)cc * E), nrow);
        } else {
#pragma unroll
          for (int j = 0; j < EM; j++) nrow[j] = lrow[j];
        }
        if (g < cnt) {      // (two loops, not one select between a global and an LDS source: hipcc 7.2 trips over the generic pointer)
          if (xl != nullptr) { for (int i = c; i < D; i += GL) sxn[i] = xl[fil * D + i]; }
          else { for (int i = c; i < D; i += GL) sxn[i] = sl[Model::OFF_X + i]; }
        }
        wave_lds_sync();
      }
      if (lead && (norm_quats & 2)) {
        double xn1[D];
#pragma unroll
        for (int i = 0; i < D; i++) xn1[i] = sxn[i];
        Model::normalize(xn1);
#pragma unroll
        for (int i = 0; i < D; i++) sxn[i] = xn1[i];
      }
      if (on) {
#pragma unroll
        for (int j = 0; j < EM; j++) C[c * EM + j] = nrow[j] - lrow[j];
        rts_store_row<E, EM>(Ps + (((k + 1) * n + fil) * EE + (int64_t)c * E), nrow);
      }
      wave_lds_sync();
      if (on) {             // D from its lower triangle (see k_rts): Pk1_n carries whatever asymmetry the caller's covariances had
#pragma unroll
        for (int j = 1; j < EM; j++) {
          if (j > c) C[c * EM + j] = C[j * EM + c];
        }
      }
      wave_lds_sync();
      if (g < cnt) {
        if (first && (EM < E || DM < D)) {
          // window part of the newest estimate: the predicted pair when it was passed in
          if (Pl != nullptr) {
            for (int i = c; i < EE; i += GL) {
              const int r = i / E, q = i % E;
              if (r >= EM || q >= EM) Ps[((k + 1) * n + fil) * EE + i] = Pl[fil * EE + i];
            }
          }
          for (int i = c; i < D; i += GL) xs[((k + 1) * n + fil) * D + i] = sxn[i];
        } else {
          for (int i = c; i < DM; i += GL) xs[((k + 1) * n + fil) * D + i] = sxn[i];
        }
      }

      RN_RTS_STAMP(4);
      if (Model::ID0 && dt == 0.0 && !first) {
        // ---- identity-gain step (see the head of this file): Ck = I on the main block.  The difference buffer holds D = Pk1_n - Pk1_k,
        // symmetric; the filtered row enters the sum as stored.  dt is a scalar of the launch: the branch is uniform. ----
        wave_lds_sync();           // the group has written xk1_n out: the buffer takes xk_n
        // (straight from / to LDS, one lane per filter: staged through register arrays like the full path's state phase, the five state-sized
        // arrays of this body put the 56-state model's one-wavefront build 53 registers over the file)
        if (lead) Model::inv_err(sl + Model::OFF_X, sxn, sde);
        wave_lds_sync();
        if (lead) {
          Model::err(sxk, sde, sxn);                                               // xk_n: becomes xk1_n of the next (older) step
          if constexpr (DM < D) {
#pragma unroll
            for (int i = DM; i < D; i++) sxn[i] = sxk[i];                          // (MSCKF: only the main states are smoothed)
          }
        }
        if (on) {                  // row c of Pk_n = Pk_k + D, eight entries at a time: Pk1_n of the next (older) step
#pragma unroll
          for (int j0 = 0; j0 < EM; j0 += 8) {
            double pr[8];
#pragma unroll
            for (int j = 0; j < 8; j++) pr[j] = (j0 + j < EM) ? Pk[j0 + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j0 + j < EM) C[c * EM + j0 + j] += pr[j]; }
          }
        }
        wave_lds_sync();
        continue;
      }
      // ---- E. Cholesky of Pk1_k, left-looking: lane c owns row c in registers; pivot row j (final since column j - 1) is
      // broadcast from LDS and every lane forms its own entry AND the pivot redundantly -- no publish / wait per column -------
      static_for<EM>([&](auto J) {
        constexpr int j = decltype(J)::value;
        double s0 = lrow[j], s1 = 0.0, p0 = B[j * EM + j], p1 = 0.0;
#pragma unroll
        for (int m = 0; m < j; m++) {
          const double r = B[j * EM + m];
          if (m & 1) { s1 = fma(-lrow[m], r, s1); p1 = fma(-r, r, p1); }
          else       { s0 = fma(-lrow[m], r, s0); p0 = fma(-r, r, p0); }
        }
        const double sj = p0 + p1;
        const double ilj = fast_rsqrt(sj);
        lrow[j] = (c == j) ? sj * ilj : (s0 + s1) * ilj;
        // rows above the pivot store a value nobody reads (one predicate for the whole factorisation).  The LDS serves one
        // wavefront's instructions in order, so the next column's broadcast reads see these stores -- but the COMPILER only
        // knows that through the fence: without it, it may (and for some sizes did) move a later column's loads of another
        // lane's entries above this store, and it also hoists loads until the register file overflows.
        if (on) { B[c * EM + j] = lrow[j]; sil[j] = ilj; }
        wave_lds_sync();
      });
      wave_lds_sync();
      RN_RTS_STAMP(5);
      // ---- F. Ck^T = Pk1_k^-1 M: lane c solves for column c in registers (right-looking substitutions) -------------------
      // Fully unrolled (the register column needs compile-time indices) through static_for -- `#pragma unroll` does not
      // duplicate the wavefront fence, and a loop that stays rolled sends the register column to scratch memory.  Two things
      // keep the register pressure at one pivot's worth: the fence per pivot keeps the coefficient loads of later pivots
      // behind it, and pin() keeps the FMAs of this pivot in front of it.  Without pin() hipcc issues the loads of ALL
      // pivots first and every FMA after the last fence (arithmetic is free to cross a fence): 460 live coefficients,
      // 1 050 spilled registers, a scratch round trip per operand.  Neither costs an instruction.
      // The coefficients of pivot m + 1 (and its reciprocal pivot) are requested before the FMAs of pivot m, so their LDS
      // latency overlaps that arithmetic instead of being exposed once per pivot.
      if constexpr (EM > 32) {
        // one wavefront per SIMD and four 2 EM-register vectors would not fit next to the rest: plain loads per pivot
        static_for<EM>([&](auto Mi) {
          constexpr int m = decltype(Mi)::value;
          y[m] *= sil[m];
#pragma unroll
          for (int i = m + 1; i < EM; i++) y[i] = fma(-B[i * EM + m], y[m], y[i]);
#pragma unroll
          for (int i = m + 1; i < EM; i++) pin(y[i]);
          wave_lds_sync();
        });
        static_for<EM>([&](auto Mi) {
          constexpr int m = EM - 1 - decltype(Mi)::value;
          y[m] *= sil[m];
#pragma unroll
          for (int i = 0; i < m; i++) y[i] = fma(-B[m * EM + i], y[m], y[i]);
#pragma unroll
          for (int i = 0; i < m; i++) pin(y[i]);
          wave_lds_sync();
        });
      } else {
        double cur[EM], nxt[EM], ilc, iln = 0.0;
#pragma unroll
        for (int i = 1; i < EM; i++) cur[i] = B[i * EM + 0];
        ilc = sil[0];
        static_for<EM>([&](auto Mi) {
          constexpr int m = decltype(Mi)::value;
          if constexpr (m + 1 < EM) {
#pragma unroll
            for (int i = m + 2; i < EM; i++) nxt[i] = B[i * EM + (m + 1)];
            iln = sil[m + 1];
          }
          y[m] *= ilc;
#pragma unroll
          for (int i = m + 1; i < EM; i++) y[i] = fma(-cur[i], y[m], y[i]);
#pragma unroll
          for (int i = m + 1; i < EM; i++) pin(y[i]);
          wave_lds_sync();
#pragma unroll
          for (int i = m + 2; i < EM; i++) cur[i] = nxt[i];
          ilc = iln;
        });
#pragma unroll
        for (int i = 0; i < EM - 1; i++) cur[i] = B[(EM - 1) * EM + i];
        ilc = sil[EM - 1];
        static_for<EM>([&](auto Mi) {
          constexpr int m = EM - 1 - decltype(Mi)::value;
          if constexpr (m >= 1) {
#pragma unroll
            for (int i = 0; i < m - 1; i++) nxt[i] = B[(m - 1) * EM + i];
            iln = sil[m - 1];
          }
          y[m] *= ilc;
#pragma unroll
          for (int i = 0; i < m; i++) y[i] = fma(-cur[i], y[m], y[i]);
#pragma unroll
          for (int i = 0; i < m; i++) pin(y[i]);
          wave_lds_sync();
#pragma unroll
          for (int i = 0; i < m - 1; i++) cur[i] = nxt[i];
          ilc = iln;
        });
      }
      RN_RTS_STAMP(6);
      // y is column c of Ck^T, i.e. row c of Ck
      // ---- G. state: delta = Ck inv_err(xk1_k, xk1_n)[:EM]; xk_n[:DM] = err(xk_k, delta)[:DM] -----------------------------
      if (lead) {
        double xb[D], xn1[D], delta[E];
#pragma unroll
        for (int i = 0; i < D; i++) { xb[i] = sl[Model::OFF_X + i]; xn1[i] = sxn[i]; }
        Model::inv_err(xb, xn1, delta);
#pragma unroll
        for (int i = 0; i < E; i++) sde[i] = delta[i];
      }
      wave_lds_sync();
      {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < EM; j += 2) { s0 = fma(y[j], sde[j], s0); s1 = fma(y[j + 1], sde[j + 1], s1); }
        if (EM & 1) s0 = fma(y[EM - 1], sde[EM - 1], s0);
        if (on) sdx[c] = s0 + s1;
      }
      wave_lds_sync();
      if (lead) {
        double xa[D], xnew[D], delta[E];
#pragma unroll
        for (int i = 0; i < D; i++) xa[i] = sxk[i];
#pragma unroll
        for (int i = 0; i < E; i++) delta[i] = (i < EM) ? sdx[i] : sde[i];
        Model::err(xa, delta, xnew);
#pragma unroll
        for (int i = 0; i < D; i++) sxn[i] = (i < DM) ? xnew[i] : xa[i];       // xk_n: becomes xk1_n of the next (older) step
      }
      RN_RTS_STAMP(7);
      // ---- H. covariance: Pk_n = Pk_k + (Ck Dm) Ck^T on the matrix cores -----------------------------------------------------
      // The two EM^3 products are the only GEMM-shaped work of the whole package, and in the row-per-lane form above they
      // were bound by OPERAND DELIVERY, not arithmetic: every FMA needs one broadcast double from LDS (2 LDS cycles per 4 VALU
      // cycles and wavefront, four SIMDs sharing one LDS: measured 80 % LDS-busy, 4.5 us per product and wavefront).
      // v_mfma_f64_16x16x4 has the same peak rate as the vector unit on CDNA4 but distributes its operands itself: one
      // ds_read_b64 per lane feeds a 16 x 16 x 4 block (LDS traffic / 10), all 64 lanes work on one filter (no idle lanes for
      // EM < 32), and the accumulators never leave registers.  Operand / result layout measured on the device
      // (tools/mfma_probe.hip): lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; register r of lane l holds
      // D[4 r + l / 16][l % 16].  Tiles beyond EM are fed zeros (k) or simply not stored (i, j).
      wave_lds_sync();           // every lane is done with the factor: its buffer takes Ck^T (column c written by lane c)
      if (on) {
#pragma unroll
        for (int j = 0; j < EM; j++) B[j * EM + c] = y[j];
      }
      wave_lds_sync();
      RN_RTS_STAMP(8);
      for (int fg = 0; fg < cnt; fg++)
        rts_products<E, EM>(s_B + fg * MMP, s_C + fg * MMP, Pf + (k * n + base + fg) * EE, lane);
      wave_lds_sync();
      RN_RTS_STAMP(9);
    }
    // ---- the oldest smoothed estimate goes out un-normalised (ekf_sym.py:665-667 never reaches it) --------------------------
    if (T >= 2) {
      if (on) {
        double nrow[EM];
#pragma unroll
        for (int j = 0; j < EM; j++) nrow[j] = C[c * EM + j];
        rts_store_row<E, EM>(Ps + ((int64_t)fil * EE + (int64_t)c * E), nrow);
      }
      if (g < cnt) {
        for (int i = c; i < DM; i += GL) xs[fil * D + i] = sxn[i];
      }
      wave_lds_sync();
    }
    // T == 1: nothing to smooth, the single estimate's predicted pair is not available -> the filtered pair passes through
    if (T == 1 && (!inplace_P || !inplace_x)) {
      if (!inplace_P) { for (int i = lane; i < cnt * EE; i += 64) Ps[base * EE + i] = Pf[base * EE + i]; }
      if (!inplace_x) { for (int i = lane; i < cnt * D; i += 64) xs[base * D + i] = xf[base * D + i]; }
    }
  }
}

}  // namespace rn
