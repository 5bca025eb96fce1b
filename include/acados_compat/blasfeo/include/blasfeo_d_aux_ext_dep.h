/* acados_compat: stands where <blasfeo/include/blasfeo_d_aux_ext_dep.h> stands for
 * src/c_generated_code/main_mobile_robot.c:43,133-135 (d_print_exp_tran_mat of the trajectories).
 * Column-major matrices with leading dimension lda, as BLASFEO's helpers take them. */
#ifndef MPC_AMD_ACADOS_COMPAT_BLASFEO_AUX_H_
#define MPC_AMD_ACADOS_COMPAT_BLASFEO_AUX_H_
#ifdef __cplusplus
extern "C" {
#endif
void d_print_mat(int m, int n, double *A, int lda);
void d_print_tran_mat(int m, int n, double *A, int lda);
void d_print_exp_mat(int m, int n, double *A, int lda);
void d_print_exp_tran_mat(int m, int n, double *A, int lda);
#ifdef __cplusplus
}
#endif
#endif
