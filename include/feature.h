#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void feature_err_fun(double *nom_x, double *delta_x, double *out);
void feature_inv_err_fun(double *nom_x, double *true_x, double *out);
void feature_H_mod_fun(double *state, double *out);
void feature_f_fun(double *state, double dt, double *out);
void feature_F_fun(double *state, double dt, double *out);
void feature_h_1(double *state, double *unused1, double *out);
void feature_H_1(double *state, double *unused1, double *out);
void feature_h_2(double *state, double *landmark, double *out);
void feature_H_2(double *state, double *landmark, double *out);
void feature_He_2(double *state, double *landmark, double *out);
void feature_dims(int *dims);
int feature_kind_zdim(int kind);
int feature_kind_maha(int kind);
int feature_num_kinds(void);
void feature_kinds(int *out);
int feature_last_error(void);
const char *feature_last_error_string(void);
void feature_clear_error(void);
int feature_batch_predict(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, void *stream);
int feature_batch_update_1(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int feature_batch_predict_update_1(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int feature_batch_update_2(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int feature_batch_predict_update_2(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int feature_batch_predict_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, const uint8_t *active, void *stream);
int feature_batch_update_1_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int feature_batch_predict_update_1_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int feature_batch_update_2_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int feature_batch_predict_update_2_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int feature_batch_predict_update_1_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int feature_batch_predict_update_2_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int feature_batch_ring_copy(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride, int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n, int to_ring, void *stream);
int feature_batch_flags_set(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream);
int feature_batch_maha_1(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int feature_batch_maha_2(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int feature_batch_augment(double *x, double *P, int64_t n, void *stream);
void feature_msckf_dims(int *dims);
int feature_kind_eadim(int kind);
int feature_zmax(void);
int feature_run_unroll(void);
int feature_has_batch_run(void);
int feature_predict_identity_at_dt0(void);
int feature_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);
int feature_has_tri_trace(void);
int feature_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);
void feature_predict(double *in_x, double *in_P, double *in_Q, double dt);
void feature_update_1(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void feature_update_2(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
#ifdef __cplusplus
}
#endif
