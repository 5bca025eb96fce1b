/*
 * acados_compat — the subset of acados' public C headers that the reference's callers of the
 * generated solvers include, re-declared over libmpc_amd.so (include/mpc_batch.h).  Written for
 * this repository; nothing here is taken from the acados distribution.
 *
 * This file stands where <acados/utils/types.h> stands for
 *   src/c_generated_code/acados_solver_mobile_robot.h:34 and src/optimizer.cpp:333 (ACADOS_SUCCESS).
 */
#ifndef MPC_AMD_ACADOS_COMPAT_TYPES_H_
#define MPC_AMD_ACADOS_COMPAT_TYPES_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_AMD_ACADOS_COMPAT 1

#if defined(_WIN32)
#define ACADOS_SYMBOL_EXPORT __declspec(dllexport)
#else
#define ACADOS_SYMBOL_EXPORT __attribute__((visibility("default")))
#endif

/* return codes of <name>_acados_solve() (same values as MPC_SUCCESS .. MPC_QP_FAILURE, mpc_batch.h) */
enum return_values {
    ACADOS_SUCCESS = 0,
    ACADOS_NAN_DETECTED = 1,
    ACADOS_MAXITER = 2,
    ACADOS_MINSTEP = 3,
    ACADOS_QP_FAILURE = 4,
    ACADOS_READY = 5,
    /* not an acados code: a set/get with a field or value this layer does not carry */
    MPC_ACADOS_UNSUPPORTED = -100
};

typedef double real_t;
typedef int int_t;

#ifdef __cplusplus
}
#endif
#endif
