/*
 * acados_compat: stands where <acados_c/sim_interface.h> stands for the reference's callers
 * (src/c_generated_code/acados_sim_solver_mobile_robot.h:34, src/optimizer.cpp:49-52,323-338,
 *  src/c_generated_code/main_sim_mobile_robot.c:38,91-105).  Written for this repository over
 * libmpc_amd.so (mpc_batch_sim_step, include/mpc_batch.h); see acados/utils/types.h.
 * Every object is its own malloc() block (include/optimizer.hpp:30-52 free()s them one by one).
 */
#ifndef MPC_AMD_ACADOS_COMPAT_SIM_INTERFACE_H_
#define MPC_AMD_ACADOS_COMPAT_SIM_INTERFACE_H_

#include "acados/utils/types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { ERK, IRK, GNSF, LIFTED_IRK, INVALID_SIM_SOLVER } sim_solver_t;
typedef struct sim_solver_plan_t { sim_solver_t sim_solver; } sim_solver_plan_t;

typedef struct sim_config { int ns; void *mpc_impl; } sim_config;   /* ns = 4: ERK4 (acados_sim_solver_mobile_robot.c:138-141) */
typedef struct sim_dims_ { int nx, nu, nz; void *mpc_impl; } sim_dims_;  /* handed out as void* like acados does */
typedef struct sim_in { double T; void *mpc_impl; } sim_in;           /* T = step length (the variant's Ts, :77,152-153) */
typedef struct sim_out { void *mpc_impl; } sim_out;
typedef struct sim_opts { int num_stages, num_steps; void *mpc_impl; } sim_opts;
typedef struct sim_solver { sim_config *config; void *dims; void *opts; void *mpc_impl; } sim_solver;

/* "x" (3 doubles), "u" (2 doubles), "T" (1 double: the step length of the next steps)
 * (src/optimizer.cpp:323-324). */
ACADOS_SYMBOL_EXPORT int sim_in_set(void *config, void *dims, sim_in *in, const char *field, void *value);
/* "x" / "xn" (3 doubles): the state after the last <name>_acados_sim_solve (src/optimizer.cpp:338).
 * Forward sensitivities ("S_forw", "Sx", "Su") are not returned by the plant step. */
ACADOS_SYMBOL_EXPORT int sim_out_get(void *config, void *dims, sim_out *out, const char *field, void *value);

#ifdef __cplusplus
}
#endif
#endif
