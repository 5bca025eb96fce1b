/* acados_compat: stands where <acados_c/external_function_interface.h> stands for
 * src/c_generated_code/acados_solver_mobile_robot.h:37.  The generated capsule holds arrays of these
 * wrappers around the CasADi functions; here the model (H5/H6/H7, mobile_robot_expl_vde_forw.c:63-137)
 * is compiled into the kernels, so the type only keeps the capsule's layout.  See acados/utils/types.h. */
#ifndef MPC_AMD_ACADOS_COMPAT_EXTFUN_H_
#define MPC_AMD_ACADOS_COMPAT_EXTFUN_H_
#include "acados/utils/types.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct external_function_param_casadi {
    void *reserved; /* unused: the dynamics live in libmpc_amd.so's kernels */
} external_function_param_casadi;
#ifdef __cplusplus
}
#endif
#endif
