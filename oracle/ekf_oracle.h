/*
 * ekf_oracle.h -- CPU ORACLE interface (test infrastructure, not the product).
 * See ekf_oracle.c for the reference lines each function restates.
 */
#ifndef EKF_ORACLE_H
#define EKF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*oracle_hfun)(double *, double *, double *);

typedef struct oracle_model {
  int dim;    /* DIM   -- nominal state size            (ekf_sym.py:122) */
  int edim;   /* EDIM  -- error state / covariance side (ekf_sym.py:123) */
  int medim;  /* MEDIM -- main-block error dim          (ekf_sym.py:124) */
  void (*f_fun)(double *, double, double *);
  void (*F_fun)(double *, double, double *);
  void (*err_fun)(double *, double *, double *);
  void (*inv_err_fun)(double *, double *, double *);
  void (*H_mod_fun)(double *, double *);
} oracle_model;

void oracle_predict(const oracle_model *mdl, double *in_x, double *in_P, const double *in_Q, double dt);

/* returns 1 when the Mahalanobis gate fired (R was inflated), else 0 */
int oracle_update(const oracle_model *mdl, int zdim, int maha_test, double maha_thresh,
                  oracle_hfun h_fun, oracle_hfun H_fun, oracle_hfun Hea_fun /* NULL unless a feature-track kind */,
                  double *in_x, double *in_P, double *in_z, const double *in_R, double *in_ea);

void oracle_normalize_quat(double *x, int idx);

#ifdef __cplusplus
}
#endif
#endif
