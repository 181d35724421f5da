#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void kinematic6_err_fun(double *nom_x, double *delta_x, double *out);
void kinematic6_inv_err_fun(double *nom_x, double *true_x, double *out);
void kinematic6_H_mod_fun(double *state, double *out);
void kinematic6_f_fun(double *state, double dt, double *out);
void kinematic6_F_fun(double *state, double dt, double *out);
void kinematic6_h_1(double *state, double *unused1, double *out);
void kinematic6_H_1(double *state, double *unused1, double *out);
void kinematic6_dims(int *dims);
int kinematic6_kind_zdim(int kind);
int kinematic6_kind_maha(int kind);
int kinematic6_num_kinds(void);
void kinematic6_kinds(int *out);
int kinematic6_last_error(void);
const char *kinematic6_last_error_string(void);
void kinematic6_clear_error(void);
int kinematic6_batch_predict(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, void *stream);
int kinematic6_batch_update_1(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic6_batch_predict_update_1(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic6_batch_predict_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, const uint8_t *active, void *stream);
int kinematic6_batch_update_1_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic6_batch_predict_update_1_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic6_batch_predict_update_1_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int kinematic6_batch_ring_copy(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride, int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n, int to_ring, void *stream);
int kinematic6_batch_flags_set(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream);
int kinematic6_batch_maha_1(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
void kinematic6_msckf_dims(int *dims);
int kinematic6_kind_eadim(int kind);
int kinematic6_zmax(void);
int kinematic6_run_unroll(void);
int kinematic6_has_batch_run(void);
int kinematic6_predict_identity_at_dt0(void);
int kinematic6_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);
int kinematic6_has_tri_trace(void);
int kinematic6_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);
void kinematic6_predict(double *in_x, double *in_P, double *in_Q, double dt);
void kinematic6_update_1(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
#ifdef __cplusplus
}
#endif
