#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void live_err_fun(double *nom_x, double *delta_x, double *out);
void live_inv_err_fun(double *nom_x, double *true_x, double *out);
void live_H_mod_fun(double *state, double *out);
void live_f_fun(double *state, double dt, double *out);
void live_F_fun(double *state, double dt, double *out);
void live_h_3(double *state, double *unused1, double *out);
void live_H_3(double *state, double *unused1, double *out);
void live_h_4(double *state, double *unused1, double *out);
void live_H_4(double *state, double *unused1, double *out);
void live_h_9(double *state, double *unused1, double *out);
void live_H_9(double *state, double *unused1, double *out);
void live_h_10(double *state, double *unused1, double *out);
void live_H_10(double *state, double *unused1, double *out);
void live_h_12(double *state, double *unused1, double *out);
void live_H_12(double *state, double *unused1, double *out);
void live_h_13(double *state, double *unused1, double *out);
void live_H_13(double *state, double *unused1, double *out);
void live_h_14(double *state, double *unused1, double *out);
void live_H_14(double *state, double *unused1, double *out);
void live_h_19(double *state, double *unused1, double *out);
void live_H_19(double *state, double *unused1, double *out);
void live_dims(int *dims);
int live_kind_zdim(int kind);
int live_kind_maha(int kind);
int live_num_kinds(void);
void live_kinds(int *out);
int live_last_error(void);
const char *live_last_error_string(void);
void live_clear_error(void);
int live_batch_predict(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, void *stream);
int live_batch_update_3(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_3(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_4(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_4(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_9(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_9(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_10(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_10(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_12(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_12(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_13(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_13(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_14(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_14(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_update_19(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_predict_update_19(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int live_batch_maha_3(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_4(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_9(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_10(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_12(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_13(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_14(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int live_batch_maha_19(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
void live_msckf_dims(int *dims);
int live_kind_eadim(int kind);
int live_zmax(void);
int live_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);
int live_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);
void live_predict(double *in_x, double *in_P, double *in_Q, double dt);
void live_update_3(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_4(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_9(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_10(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_12(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_13(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_14(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void live_update_19(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
#ifdef __cplusplus
}
#endif
