#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void kinematic9_err_fun(double *nom_x, double *delta_x, double *out);
void kinematic9_inv_err_fun(double *nom_x, double *true_x, double *out);
void kinematic9_H_mod_fun(double *state, double *out);
void kinematic9_f_fun(double *state, double dt, double *out);
void kinematic9_F_fun(double *state, double dt, double *out);
void kinematic9_h_1(double *state, double *unused1, double *out);
void kinematic9_H_1(double *state, double *unused1, double *out);
void kinematic9_h_2(double *state, double *unused1, double *out);
void kinematic9_H_2(double *state, double *unused1, double *out);
void kinematic9_h_3(double *state, double *unused1, double *out);
void kinematic9_H_3(double *state, double *unused1, double *out);
void kinematic9_dims(int *dims);
int kinematic9_kind_zdim(int kind);
int kinematic9_kind_maha(int kind);
int kinematic9_num_kinds(void);
void kinematic9_kinds(int *out);
int kinematic9_last_error(void);
const char *kinematic9_last_error_string(void);
void kinematic9_clear_error(void);
int kinematic9_batch_predict(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, void *stream);
int kinematic9_batch_update_1(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_predict_update_1(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_update_2(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_predict_update_2(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_update_3(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_predict_update_3(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, void *stream);
int kinematic9_batch_predict_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, int64_t n, int norm_quats, const uint8_t *active, void *stream);
int kinematic9_batch_update_1_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_predict_update_1_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_update_2_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_predict_update_2_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_update_3_masked(double *x, double *P, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_predict_update_3_masked(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, const uint8_t *active, void *stream);
int kinematic9_batch_predict_update_1_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int kinematic9_batch_predict_update_2_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int kinematic9_batch_predict_update_3_ckpt(double *x, double *P, const double *Q, const double *dt_vec, double dt, double *z, const double *R, int r_per_filter, const double *ea, int64_t n, int norm_quats, uint8_t *flags, double *ckpt_x, double *ckpt_P, double *ckpt_z, void *stream);
int kinematic9_batch_ring_copy(double *ring, int64_t ring_stride, double *flat, int64_t flat_stride, int64_t rec, const int32_t *slot, const uint8_t *active, int64_t n, int to_ring, void *stream);
int kinematic9_batch_flags_set(uint8_t *flags, const uint8_t *mask, int value, int64_t n, void *stream);
int kinematic9_batch_maha_1(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int kinematic9_batch_maha_2(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
int kinematic9_batch_maha_3(const double *x, const double *P, const double *z, const double *R, int r_per_filter, const double *ea, int64_t n, double *d2, void *stream);
void kinematic9_msckf_dims(int *dims);
int kinematic9_kind_eadim(int kind);
int kinematic9_zmax(void);
int kinematic9_run_unroll(void);
int kinematic9_has_batch_run(void);
int kinematic9_predict_identity_at_dt0(void);
int kinematic9_batch_run(double *x, double *P, const double *Q, const int32_t *kinds, const double *dts, int64_t T, double *z, const double *R, int64_t n, int norm_quats, uint8_t *flags, double *trace_x, double *trace_P, const double *ea, const int32_t *augment, void *stream);
int kinematic9_has_tri_trace(void);
int kinematic9_batch_rts(const double *xf, const double *Pf, const double *ts, int64_t T, const double *Q, int64_t n, int norm_quats, double *xs, double *Ps, const double *x_last, const double *P_last, void *stream);
void kinematic9_predict(double *in_x, double *in_P, double *in_Q, double dt);
void kinematic9_update_1(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void kinematic9_update_2(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
void kinematic9_update_3(double *in_x, double *in_P, double *in_z, double *in_R, double *in_ea);
#ifdef __cplusplus
}
#endif
