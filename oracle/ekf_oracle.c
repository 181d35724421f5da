/*
 * ekf_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's Eigen template
 *   /root/reference/rednose/templates/ekf_c.c:8-33   (predict)
 *   /root/reference/rednose/templates/ekf_c.c:37-121 (update<ZDIM,EADIM,MAHA_TEST>)
 * with run-time dimensions so one object file serves every model.  Eigen is
 * not installed in this image, so the Eigen half of the reference cannot be
 * compiled; the sympy-generated half (f_fun, F_fun, h_k, H_k, H_mod_fun,
 * err_fun, inv_err_fun) IS produced by the reference's own gen_code and is
 * linked in unmodified by oracle/build_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  Nothing under rednose_amd/ links, imports or calls it.
 *
 * Parity pin: the four known-answer values of
 * /root/reference/examples/test_kinematic_kf.py:52-55 and golden vectors
 * produced by the reference's own numpy path (EKF_sym._predict_python /
 * _update_python / rts_smooth, rednose/helpers/ekf_sym.py:533-690), see
 * oracle/make_golden.py and tests/test_oracle.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ekf_oracle.h"

#define OR_MAXE 64
#define OR_MAXZ 16

/* C = A(m x k) * B(k x n), all row-major, plain i-k-j accumulation in k order */
static void mm(const double *A, const double *B, double *C, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int p = 0; p < k; p++) s += A[i * k + p] * B[p * n + j];
      C[i * n + j] = s;
    }
}

/* C = A(m x k) * B^T where B is (n x k) */
static void mm_bt(const double *A, const double *B, double *C, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int p = 0; p < k; p++) s += A[i * k + p] * B[j * k + p];
      C[i * n + j] = s;
    }
}

/*
 * Solve S X = B for X by Gaussian elimination with FULL pivoting, the
 * decomposition the reference asks Eigen for (ekf_c.c:101 fullPivLu().solve,
 * :89 .inverse()).  S is n x n, B is n x m, both row-major and overwritten.
 */
static void fullpiv_solve(double *S, double *B, double *X, int n, int m) {
  int colperm[OR_MAXZ];
  for (int i = 0; i < n; i++) colperm[i] = i;
  for (int k = 0; k < n; k++) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int i = k; i < n; i++)
      for (int j = k; j < n; j++) {
        double a = fabs(S[i * n + j]);
        if (a > best) { best = a; pr = i; pc = j; }
      }
    if (pr != k) {
      for (int j = 0; j < n; j++) { double t = S[k * n + j]; S[k * n + j] = S[pr * n + j]; S[pr * n + j] = t; }
      for (int j = 0; j < m; j++) { double t = B[k * m + j]; B[k * m + j] = B[pr * m + j]; B[pr * m + j] = t; }
    }
    if (pc != k) {
      for (int i = 0; i < n; i++) { double t = S[i * n + k]; S[i * n + k] = S[i * n + pc]; S[i * n + pc] = t; }
      int t = colperm[k]; colperm[k] = colperm[pc]; colperm[pc] = t;
    }
    double piv = S[k * n + k];
    for (int i = k + 1; i < n; i++) {
      double l = S[i * n + k] / piv;
      if (l == 0.0) continue;
      for (int j = k + 1; j < n; j++) S[i * n + j] -= l * S[k * n + j];
      for (int j = 0; j < m; j++) B[i * m + j] -= l * B[k * m + j];
    }
  }
  /* back substitution into the permuted unknowns */
  double Y[OR_MAXZ];
  for (int j = 0; j < m; j++) {
    for (int i = n - 1; i >= 0; i--) {
      double s = B[i * m + j];
      for (int p = i + 1; p < n; p++) s -= S[i * n + p] * Y[p];
      Y[i] = s / S[i * n + i];
    }
    for (int i = 0; i < n; i++) X[colperm[i] * m + j] = Y[i];
  }
}


/*
 * Basis of the right null space of M (rows x cols, row-major) by Gaussian elimination with FULL pivoting -- what
 * Eigen's FullPivLU::kernel() computes for ekf_c.c:71 (A = Hea^T.fullPivLu().kernel()): with P M Q = L U and
 * U = [U1 U2] (U1 rank x rank upper triangular), the kernel vectors are Q [-U1^-1 U2 ; I].  ker is cols x (cols - rank),
 * row-major; returns the kernel dimension.  As in Eigen's FullPivLU: the pivot search walks the remaining corner column by column and
 * keeps the FIRST maximum (its maxCoeff visitor; decides which of several equal entries becomes the pivot), the elimination runs through
 * all min(rows, cols) steps unless the corner is exactly zero, and the rank is decided afterwards -- a pivot counts when
 * |pivot| > eps * min(rows, cols) * |largest pivot met| (threshold() = epsilon * diagonalSize, m_maxpivot the running maximum).  Pivots are
 * counted from the front (with full pivoting a small pivot is followed by smaller ones; Eigen's kernel() would also skip a small pivot in
 * the middle).  Any basis gives the same x and P; only the projected residual y depends on the choice.
 */
static int fullpiv_kernel(const double *M, int rows, int cols, double *ker) {
  double U[OR_MAXZ * OR_MAXZ];
  int colperm[OR_MAXZ];
  memcpy(U, M, sizeof(double) * (size_t)rows * cols);
  for (int j = 0; j < cols; j++) colperm[j] = j;
  int rank = 0, done = 0;
  double maxpiv = 0.0, piv[OR_MAXZ];
  const int steps = rows < cols ? rows : cols;
  for (int k = 0; k < steps; k++) {
    int pr = k, pc = k;
    double best = 0.0;
    for (int j = k; j < cols; j++)
      for (int i = k; i < rows; i++) {
        double a = fabs(U[i * cols + j]);
        if (a > best) { best = a; pr = i; pc = j; }
      }
    if (best == 0.0) break;
    piv[done++] = best;
    if (best > maxpiv) maxpiv = best;
    if (pr != k) for (int j = 0; j < cols; j++) { double t = U[k * cols + j]; U[k * cols + j] = U[pr * cols + j]; U[pr * cols + j] = t; }
    if (pc != k) {
      for (int i = 0; i < rows; i++) { double t = U[i * cols + k]; U[i * cols + k] = U[i * cols + pc]; U[i * cols + pc] = t; }
      int t = colperm[k]; colperm[k] = colperm[pc]; colperm[pc] = t;
    }
    for (int i = k + 1; i < rows; i++) {
      double l = U[i * cols + k] / U[k * cols + k];
      for (int j = k; j < cols; j++) U[i * cols + j] -= l * U[k * cols + j];
    }
  }
  while (rank < done && piv[rank] > 2.220446049250313e-16 * steps * maxpiv) rank++;
  const int nk = cols - rank;
  for (int c = 0; c < nk; c++) {
    double v[OR_MAXZ];
    for (int i = rank - 1; i >= 0; i--) {            /* U1 v = -U2[:, c] */
      double acc = -U[i * cols + rank + c];
      for (int p = i + 1; p < rank; p++) acc -= U[i * cols + p] * v[p];
      v[i] = acc / U[i * cols + i];
    }
    for (int j = 0; j < cols; j++) ker[j * nk + c] = 0.0;
    for (int i = 0; i < rank; i++) ker[colperm[i] * nk + c] = v[i];
    ker[colperm[rank + c] * nk + c] = 1.0;
  }
  return nk;
}

/* ekf_c.c:8-33 */
void oracle_predict(const oracle_model *mdl, double *in_x, double *in_P, const double *in_Q, double dt) {
  const int D = mdl->dim, E = mdl->edim, M = mdl->medim;
  double *nx = (double *)calloc((size_t)D, sizeof(double));
  double *F = (double *)calloc((size_t)E * E, sizeof(double));
  double *Fm = (double *)malloc(sizeof(double) * (size_t)M * M);
  double *T = (double *)malloc(sizeof(double) * (size_t)E * E);
  double *T2 = (double *)malloc(sizeof(double) * (size_t)E * E);
  double *P = (double *)malloc(sizeof(double) * (size_t)E * E);

  mdl->f_fun(in_x, dt, nx);       /* :15 */
  mdl->F_fun(in_x, dt, F);        /* :16 -- Jacobian at the PRE-propagation state */
  memcpy(P, in_P, sizeof(double) * (size_t)E * E);

  for (int i = 0; i < M; i++)
    for (int j = 0; j < M; j++) Fm[i * M + j] = F[i * E + j];   /* :23 */

  /* :24  P[:M,:M] = (F_main * P[:M,:M]) * F_main^T */
  {
    double *Pmm = T2;
    for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) Pmm[i * M + j] = P[i * E + j];
    mm(Fm, Pmm, T, M, M, M);
    double *R2 = (double *)malloc(sizeof(double) * (size_t)M * M);
    mm_bt(T, Fm, R2, M, M, M);
    /* :25  P[:M,M:] = F_main * P[:M,M:]   (uses the ORIGINAL top-right block) */
    int A = E - M;
    if (A > 0) {
      double *Ptr = (double *)malloc(sizeof(double) * (size_t)M * A);
      double *Ntr = (double *)malloc(sizeof(double) * (size_t)M * A);
      double *Pbl = (double *)malloc(sizeof(double) * (size_t)A * M);
      double *Nbl = (double *)malloc(sizeof(double) * (size_t)A * M);
      for (int i = 0; i < M; i++) for (int j = 0; j < A; j++) Ptr[i * A + j] = P[i * E + M + j];
      for (int i = 0; i < A; i++) for (int j = 0; j < M; j++) Pbl[i * M + j] = P[(M + i) * E + j];
      mm(Fm, Ptr, Ntr, M, M, A);
      mm_bt(Pbl, Fm, Nbl, A, M, M);     /* :26 */
      for (int i = 0; i < M; i++) for (int j = 0; j < A; j++) P[i * E + M + j] = Ntr[i * A + j];
      for (int i = 0; i < A; i++) for (int j = 0; j < M; j++) P[(M + i) * E + j] = Nbl[i * M + j];
      free(Ptr); free(Ntr); free(Pbl); free(Nbl);
    }
    for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) P[i * E + j] = R2[i * M + j];
    free(R2);
  }

  for (int i = 0; i < E * E; i++) P[i] = P[i] + dt * in_Q[i];   /* :28 */

  memcpy(in_x, nx, sizeof(double) * (size_t)D);                 /* :31 */
  memcpy(in_P, P, sizeof(double) * (size_t)E * E);              /* :32 */
  free(nx); free(F); free(Fm); free(T); free(T2); free(P);
}

/* ekf_c.c:37-121; Hea_fun == NULL for ordinary kinds, the EADIM (= 3, ekf_sym.py:151) extra-argument Jacobian of a
 * feature-track kind otherwise (:66-76: residual, H and R projected on the left null space of Hea) */
int oracle_update(const oracle_model *mdl, int ZDIM, int maha_test, double maha_thresh,
                  oracle_hfun h_fun, oracle_hfun H_fun, oracle_hfun Hea_fun,
                  double *in_x, double *in_P, double *in_z, const double *in_R, double *in_ea) {
  const int D = mdl->dim, E = mdl->edim;
  int Z = ZDIM;
  int gated = 0;
  double hx[OR_MAXZ] = {0};
  double y[OR_MAXZ];
  double *H = (double *)calloc((size_t)ZDIM * D, sizeof(double));
  double *Hmod = (double *)calloc((size_t)E * D, sizeof(double));
  double *Herr = (double *)malloc(sizeof(double) * (size_t)Z * E);
  double *HP = (double *)malloc(sizeof(double) * (size_t)Z * E);
  double *HPt = (double *)malloc(sizeof(double) * (size_t)Z * E);
  double *KT = (double *)malloc(sizeof(double) * (size_t)Z * E);
  double *IKH = (double *)malloc(sizeof(double) * (size_t)E * E);
  double *T = (double *)malloc(sizeof(double) * (size_t)E * E);
  double *Pn = (double *)malloc(sizeof(double) * (size_t)E * E);
  double *KR = (double *)malloc(sizeof(double) * (size_t)E * Z);
  double *dx = (double *)calloc((size_t)E, sizeof(double));
  double *xn = (double *)calloc((size_t)D, sizeof(double));
  double R[OR_MAXZ * OR_MAXZ], S[OR_MAXZ * OR_MAXZ], S2[OR_MAXZ * OR_MAXZ];

  h_fun(in_x, in_ea, hx);                                   /* :55 */
  H_fun(in_x, in_ea, H);                                    /* :56 */
  for (int i = 0; i < Z; i++) y[i] = in_z[i] - hx[i];       /* :60 */
  memcpy(R, in_R, sizeof(double) * (size_t)Z * Z);          /* :75 */
  if (Hea_fun) {                                            /* :66-72 */
    enum { EADIM = 3 };
    double Hea[OR_MAXZ * EADIM] = {0}, HeaT[EADIM * OR_MAXZ], A[OR_MAXZ * OR_MAXZ];
    Hea_fun(in_x, in_ea, Hea);
    for (int i = 0; i < ZDIM; i++) for (int j = 0; j < EADIM; j++) HeaT[j * ZDIM + i] = Hea[i * EADIM + j];
    const int nk = fullpiv_kernel(HeaT, EADIM, ZDIM, A);   /* A: ZDIM x nk */
    double yp[OR_MAXZ], RA[OR_MAXZ * OR_MAXZ], Rp[OR_MAXZ * OR_MAXZ];
    double *Hp = (double *)calloc((size_t)ZDIM * D, sizeof(double));
    for (int a = 0; a < nk; a++) {                          /* y = A^T y, H = A^T H */
      double acc = 0.0;
      for (int i = 0; i < ZDIM; i++) acc += A[i * nk + a] * y[i];
      yp[a] = acc;
      for (int j = 0; j < D; j++) {
        double h = 0.0;
        for (int i = 0; i < ZDIM; i++) h += A[i * nk + a] * H[i * D + j];
        Hp[a * D + j] = h;
      }
    }
    for (int i = 0; i < ZDIM; i++)                          /* R = A^T R A */
      for (int a = 0; a < nk; a++) {
        double acc = 0.0;
        for (int j = 0; j < ZDIM; j++) acc += R[i * ZDIM + j] * A[j * nk + a];
        RA[i * nk + a] = acc;
      }
    for (int a = 0; a < nk; a++)
      for (int b = 0; b < nk; b++) {
        double acc = 0.0;
        for (int i = 0; i < ZDIM; i++) acc += A[i * nk + a] * RA[i * nk + b];
        Rp[a * nk + b] = acc;
      }
    Z = nk;
    memcpy(y, yp, sizeof(double) * (size_t)Z);
    memcpy(H, Hp, sizeof(double) * (size_t)Z * D);
    memcpy(R, Rp, sizeof(double) * (size_t)Z * Z);
    free(Hp);
  }

  mdl->H_mod_fun(in_x, Hmod);                               /* :83 */
  mm(H, Hmod, Herr, Z, D, E);                               /* :85 H_err = H * H_mod */

  mm(Herr, in_P, HP, Z, E, E);                              /* H_err * P */
  if (maha_test) {                                          /* :88-94 */
    double a[OR_MAXZ * OR_MAXZ], eye[OR_MAXZ * OR_MAXZ], inv[OR_MAXZ * OR_MAXZ];
    mm_bt(HP, Herr, a, Z, E, Z);
    for (int i = 0; i < Z * Z; i++) a[i] += R[i];
    memset(eye, 0, sizeof eye);
    for (int i = 0; i < Z; i++) eye[i * Z + i] = 1.0;
    fullpiv_solve(a, eye, inv, Z, Z);
    double d2 = 0.0;
    for (int i = 0; i < Z; i++) {
      double s = 0.0;
      for (int j = 0; j < Z; j++) s += inv[i * Z + j] * y[j];
      d2 += y[i] * s;
    }
    if (d2 > maha_thresh) {
      for (int i = 0; i < Z * Z; i++) R[i] = 1.0e16 * R[i];
      gated = 1;
    }
  }

  mm_bt(HP, Herr, S, Z, E, Z);                              /* :100 */
  for (int i = 0; i < Z * Z; i++) S[i] += R[i];
  mm_bt(Herr, in_P, HPt, Z, E, E);                          /* H_err * P^T */
  memcpy(S2, S, sizeof(double) * (size_t)Z * Z);
  fullpiv_solve(S2, HPt, KT, Z, E);                         /* :101 KT (Z x E) */

  /* :105 I_KH = I - KT^T * H_err */
  for (int i = 0; i < E; i++)
    for (int j = 0; j < E; j++) {
      double s = 0.0;
      for (int p = 0; p < Z; p++) s += KT[p * E + i] * Herr[p * E + j];
      IKH[i * E + j] = (i == j ? 1.0 : 0.0) - s;
    }
  /* :108-111 */
  for (int i = 0; i < E; i++) {
    double s = 0.0;
    for (int p = 0; p < Z; p++) s += KT[p * E + i] * y[p];
    dx[i] = s;
  }
  mdl->err_fun(in_x, dx, xn);

  /* :115 P = (I_KH * P) * I_KH^T + (KT^T * R) * KT */
  mm(IKH, in_P, T, E, E, E);
  mm_bt(T, IKH, Pn, E, E, E);
  for (int i = 0; i < E; i++)
    for (int j = 0; j < Z; j++) {
      double s = 0.0;
      for (int p = 0; p < Z; p++) s += KT[p * E + i] * R[p * Z + j];
      KR[i * Z + j] = s;
    }
  for (int i = 0; i < E; i++)
    for (int j = 0; j < E; j++) {
      double s = 0.0;
      for (int p = 0; p < Z; p++) s += KR[i * Z + p] * KT[p * E + j];
      Pn[i * E + j] += s;
    }

  memcpy(in_x, xn, sizeof(double) * (size_t)D);             /* :118 */
  memcpy(in_P, Pn, sizeof(double) * (size_t)E * E);         /* :119 */
  memcpy(in_z, y, sizeof(double) * (size_t)Z);              /* :120 y (y.rows() entries) written back into z */

  free(H); free(Hmod); free(Herr); free(HP); free(HPt); free(KT); free(IKH);
  free(T); free(Pn); free(KR); free(dx); free(xn);
  return gated;
}

/* EKFSym::normalize_slice, /root/reference/rednose/helpers/ekf_sym.cc:75-77 */
void oracle_normalize_quat(double *x, int idx) {
  double n = sqrt(x[idx] * x[idx] + x[idx + 1] * x[idx + 1] + x[idx + 2] * x[idx + 2] + x[idx + 3] * x[idx + 3]);
  for (int i = 0; i < 4; i++) x[idx + i] /= n;
}
