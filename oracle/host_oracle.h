/*
 * host_oracle.h -- CPU restatement of the host-side (L3) arithmetic around the MD call:
 * rotation tensors, strain preparation, Hooke fallback, replica averaging, tensor files.
 * TEST INFRASTRUCTURE ONLY (see md_oracle.h).  Each function cites the reference lines it
 * follows.  Tensors: rank-2 symmetric in deal.II raw order xx,yy,zz,xy,xz,yz ("raw");
 * rank-4 symmetric as 36 doubles in the order of the reference's init.*.stiff files
 * (kl and mn each running 00,01,02,11,12,22 -- read_write.h:149-171) ("file order").
 */
#ifndef HOST_ORACLE_H
#define HOST_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
/* math_calc.h:23-50 : R = I + K + K^2/(1+a.b), K_ij = a_j b_i - a_i b_j */
void ho_rotation_tensor(const double a[3], const double b[3], double R[9]);
/* math_calc.h:52-71 : sym(R T R^T) */
void ho_rotate_sym2(const double t_raw[6], const double R[9], double out_raw[6]);
/* math_calc.h:73-99 */
void ho_rotate_sym4(const double c_file[36], const double R[9], double out_file[36]);
/* stmd_sync.h:541-557 : rotate with transpose(rotam), then scale by init_length unless hooke */
void ho_prepare_strain(const double eps_raw[6], const double rotam[9], const double init_length[3],
                       int hooke, double out_raw[6]);
/* stmd_problem.h:386-392 : sigma = C : eps */
void ho_hooke(const double c_file[36], const double eps_raw[6], double out_raw[6]);
/* stmd_sync.h:878-922 : mean over replicas of rotate(sigma_r - init_stress_r, rotam_r) */
void ho_store(int nrepl, const double *stress_raw /*[nrepl*6]*/, const double *init_stress_raw,
              const double *rotam /*[nrepl*9]*/, int hooke, double out_raw[6]);
/* read_write.h:123-147,207-224 : one %.16g value per line, order 00,01,02,11,12,22 */
int ho_read_sym2(const char *path, double out_raw[6]);
int ho_write_sym2(const char *path, const double raw[6]);
int ho_read_sym4(const char *path, double out_file[36]);
#ifdef __cplusplus
}
#endif
#endif
