/*
 * acados_compat: stands where <acados_c/ocp_nlp_interface.h> stands for the reference's callers
 * (include/optimizer.hpp:14, src/optimizer.cpp:60-63,117-119,251-296,
 *  src/c_generated_code/main_mobile_robot.c:38,67-151).  Written for this repository over
 * libmpc_amd.so; see acados/utils/types.h.
 *
 * What the objects are here.  `<name>_acados_create()` (acados_solver_<name>.h) gives every
 * capsule one slot of an mpc_batch handle (include/mpc_batch.h): the iterate x/u/pi lives in HBM
 * inside that handle, exactly like acados keeps it in nlp_out.  The five objects below are small
 * host-side views of that capsule; each one is its own malloc() block, because the reference's
 * Optimizer destructor free()s them one by one (include/optimizer.hpp:30-52).
 *   ocp_nlp_in   the data of the next solve: yref of every stage, the stage-0 state bounds
 *   ocp_nlp_out  a lazily fetched host mirror of the iterate (x, u, pi) + what ocp_nlp_out_set wrote
 *   ocp_nlp_dims N, nx[], nu[], ... as the reference reads them (src/optimizer.cpp:117-119)
 * Only what the generated SQP_RTI solvers of the reference configure is carried: LINEAR_LS cost
 * with the baked W, BGH box bounds with the baked values, ERK4, HPIPM BALANCE.  A set() of another
 * field, or of a value that differs from the baked one where the kernels hold it as a constant,
 * prints one line on stderr and returns MPC_ACADOS_UNSUPPORTED; it never exits the process.
 */
#ifndef MPC_AMD_ACADOS_COMPAT_OCP_NLP_INTERFACE_H_
#define MPC_AMD_ACADOS_COMPAT_OCP_NLP_INTERFACE_H_

#include "acados/utils/types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the plan of the generated solvers (acados_solver_mobile_robot.c:140-173) */
typedef enum { SQP, SQP_RTI, DDP, INVALID_NLP_SOLVER } ocp_nlp_solver_t;
typedef enum { LINEAR_LS, NONLINEAR_LS, CONVEX_OVER_NONLINEAR, EXTERNAL, INVALID_COST } ocp_nlp_cost_t;
typedef enum { CONTINUOUS_MODEL, DISCRETE_MODEL, INVALID_DYNAMICS } ocp_nlp_dynamics_t;
typedef enum { BGH, BGP, INVALID_CONSTRAINT } ocp_nlp_constraints_t;
typedef enum { NO_REGULARIZE, MIRROR, PROJECT, PROJECT_REDUC_HESS, CONVEXIFY, INVALID_REGULARIZE } ocp_nlp_reg_t;
typedef enum { PARTIAL_CONDENSING_HPIPM, FULL_CONDENSING_HPIPM, INVALID_QP_SOLVER } ocp_qp_solver_t;

typedef struct ocp_nlp_plan_t {
    ocp_nlp_solver_t nlp_solver;     /* SQP_RTI   (acados_solver_mobile_robot.c:148) */
    ocp_qp_solver_t ocp_qp_solver;   /* PARTIAL_CONDENSING_HPIPM, qp_cond_N = N (:150, :662-665) */
    ocp_nlp_cost_t nlp_cost;         /* LINEAR_LS, every stage (:152-156) */
    ocp_nlp_dynamics_t nlp_dynamics; /* CONTINUOUS_MODEL + ERK (:160-161) */
    ocp_nlp_constraints_t nlp_constraints; /* BGH (:164-170) */
    ocp_nlp_reg_t regularization;    /* NO_REGULARIZE (:172) */
    int N;
    void *mpc_impl;
} ocp_nlp_plan_t;

typedef struct ocp_nlp_config {
    int N;
    void *mpc_impl;
} ocp_nlp_config;

/* src/optimizer.cpp:117-119 reads N, *nx, *nu */
typedef struct ocp_nlp_dims {
    int *nv; /* [N+1] nx + nu */
    int *nx; /* [N+1] 3 */
    int *nu; /* [N+1] 2 (stage N: 0) */
    int *ni; /* [N+1] two-sided bounds: 3 (x0), 4, .., 0 */
    int *nz; /* [N+1] 0 */
    int *ns; /* [N+1] 0 */
    int *np; /* [N+1] 0 */
    int N;
    void *mpc_impl;
} ocp_nlp_dims;

typedef struct ocp_nlp_in {
    double *Ts; /* [N] shooting intervals (uniform) */
    void *mpc_impl;
} ocp_nlp_in;

typedef struct ocp_nlp_out {
    double inf_norm_res; /* not evaluated by SQP_RTI: 0 */
    void *mpc_impl;
} ocp_nlp_out;

typedef struct ocp_nlp_solver {
    ocp_nlp_config *config;
    void *dims; /* ocp_nlp_dims*, as main_mobile_robot.c:136 passes it on */
    void *opts;
    void *mem;
    void *work;
    void *mpc_impl;
} ocp_nlp_solver;

/* ocp_nlp_cost_model_set(.., stage, "yref", y): stage 0..N-1 takes 5 doubles [x y psi v delta],
 * stage N takes 3 (src/optimizer.cpp:251,272; include/optimizer.hpp:172-175).  "W" / "Vx" / "Vu" /
 * "Vz" / "scaling" / "zl".. are accepted only with the values the solver was generated with. */
ACADOS_SYMBOL_EXPORT int ocp_nlp_cost_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in, int stage,
                                                const char *field, void *value);

/* ocp_nlp_constraints_model_set(.., 0, "lbx" | "ubx", x0): the initial state
 * (src/optimizer.cpp:279-280).  The two must be equal when the solve starts (x0 is eliminated,
 * idxbxe_0 = {0,1,2}, acados_solver_mobile_robot.c:513-519); "idxbx" at stage 0 must be {0,1,2}
 * (main_mobile_robot_25.c).  Bounds of the other stages and "lbu"/"ubu" are compile-time constants
 * of the kernels: accepted only with the generated values. */
ACADOS_SYMBOL_EXPORT int ocp_nlp_constraints_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in,
                                                       int stage, const char *field, void *value);

/* "x" (3), "u" (2), "pi" (3) of one stage -> the iterate the next solve starts from
 * (main_mobile_robot.c:114-119).  pi is kept for ocp_nlp_out_get but does not feed back into an
 * SQP_RTI step with a Gauss-Newton Hessian. */
ACADOS_SYMBOL_EXPORT void ocp_nlp_out_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_out *out, int stage,
                                          const char *field, void *value);

/* "x", "u", "pi" of one stage; "kkt_norm_inf" (0: not evaluated by SQP_RTI)
 * (src/optimizer.cpp:296; main_mobile_robot.c:127-130,150). */
ACADOS_SYMBOL_EXPORT void ocp_nlp_out_get(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_out *out, int stage,
                                          const char *field, void *value);

/* number of doubles ocp_nlp_out_get writes for (stage, field) */
ACADOS_SYMBOL_EXPORT int ocp_nlp_dims_get_from_attr(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_out *out,
                                                    int stage, const char *field);

/* "rti_phase" (int; only 0 = preparation + feedback in one call, main_mobile_robot.c:109,120),
 * "print_level" (ignored). */
ACADOS_SYMBOL_EXPORT void ocp_nlp_solver_opts_set(ocp_nlp_config *config, void *opts, const char *field, void *value);

/* after a solve: "time_tot" (double, s, wall clock of the solve call), "time_qp" / "time_qp_sol"
 * (double, s, device time of the kernels), "time_lin" / "time_reg" / "time_sim" / "time_glob" (0),
 * "sqp_iter" (int, 1), "qp_iter" (int, HPIPM iterations), "status" / "qp_status" / "qp_stat" (int),
 * "stat_m" (2), "stat_n" (2), "statistics" (double[(stat_n + 1) * stat_m], column-major:
 * iter, qp_stat, qp_iter — what <name>_acados_print_stats prints,
 * acados_solver_mobile_robot.c:996-1023) (main_mobile_robot.c:122,151). */
ACADOS_SYMBOL_EXPORT void ocp_nlp_get(ocp_nlp_config *config, ocp_nlp_solver *solver, const char *field,
                                      void *return_value);

#ifdef __cplusplus
}
#endif
#endif
