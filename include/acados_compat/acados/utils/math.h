/* acados_compat: stands where <acados/utils/math.h> stands for include/optimizer.hpp:13 and
 * src/c_generated_code/main_mobile_robot.c:37,123 (the MIN / MAX macros).  See acados/utils/types.h. */
#ifndef MPC_AMD_ACADOS_COMPAT_MATH_H_
#define MPC_AMD_ACADOS_COMPAT_MATH_H_
#include "acados/utils/types.h"
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#endif
#endif
