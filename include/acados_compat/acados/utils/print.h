/* acados_compat: stands where <acados/utils/print.h> stands for
 * src/c_generated_code/main_mobile_robot.c:36,136 (ocp_nlp_out_print).  See acados/utils/types.h. */
#ifndef MPC_AMD_ACADOS_COMPAT_PRINT_H_
#define MPC_AMD_ACADOS_COMPAT_PRINT_H_
#include "acados_c/ocp_nlp_interface.h"
#ifdef __cplusplus
extern "C" {
#endif
/* prints x, u, pi of every stage to stdout (main_mobile_robot.c:136, commented out there) */
void ocp_nlp_out_print(ocp_nlp_dims *dims, ocp_nlp_out *nlp_out);
#ifdef __cplusplus
}
#endif
#endif
