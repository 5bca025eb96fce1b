/*
 * mpc_hpipm_args.h — the ONE place that holds the interior-point arguments the reference's
 * solver runs with: HPIPM d_ocp_qp_ipm_arg_set_default(BALANCE), selected by
 * src/c_generated_code/acados_solver_mobile_robot.c:670 ("qp_hpipm_mode" = BALANCE; the call
 * re-applies the mode's defaults), then acados' qp_iter_max override (:682).  QP tolerances are
 * not set by the generated code (scripts/mobile_robot_18_acados_ocp.json:508-511, all null).
 *
 * Included by the product (mpc_amd/csrc/consts.hpp) and by the CPU oracle
 * (oracle/mpc_oracle.hpp), so that a constant corrected here is corrected on both sides.
 *
 * PARITY UNPINNED: HPIPM / acados are neither vendored nor pinned by the reference and are absent
 * from this image; every value below is restated from the published HPIPM
 * sources (ocp_qp/x_ocp_qp_ipm.c: OCP_QP_IPM_ARG_SET_DEFAULT, OCP_QP_IPM_SOLVE, OCP_QP_INIT_VAR;
 * ipm_core/x_core_qp_ipm_aux.c: UPDATE_VAR_QP) and could not be re-checked offline.
 */
#ifndef MPC_HPIPM_ARGS_H
#define MPC_HPIPM_ARGS_H

/* d_ocp_qp_ipm_arg_set_default(BALANCE) */
#define MPC_IPM_MU0 1e1              /* initial barrier value of the cold start: lam = mu0 / t */
#define MPC_IPM_ALPHA_MIN 1e-12      /* exit: step length below this -> HPIPM status MINSTEP */
#define MPC_IPM_RES_G_MAX 1e-6       /* exit tolerance, inf-norm of the stationarity residual */
#define MPC_IPM_RES_B_MAX 1e-8       /* ... of the dynamics residual */
#define MPC_IPM_RES_D_MAX 1e-8       /* ... of the bound residual */
#define MPC_IPM_RES_M_MAX 1e-8       /* ... of the complementarity residual */
#define MPC_IPM_ITER_MAX_BALANCE 30  /* the mode's own iter_max; acados overrides it with qp_iter_max (50) */
#define MPC_IPM_PRED_CORR 1          /* Mehrotra predictor-corrector */
#define MPC_IPM_COND_PRED_CORR 1     /* conditional: fall back to pure centering if mu would grow */
#define MPC_IPM_ITREF_PRED_MAX 0     /* no iterative refinement on the predictor system */
#define MPC_IPM_ITREF_CORR_MAX 2     /* up to 2 refinement solves on the corrector system */
#define MPC_IPM_REG_PRIM 1e-15       /* primal regularisation added to the Hessian diagonal */
#define MPC_IPM_LQ_FACT 1            /* switch to the LQ-based factorisation when the linear system's residual is large */
#define MPC_IPM_LAM_MIN 1e-16
#define MPC_IPM_T_MIN 1e-16
#define MPC_IPM_TAU_MIN 1e-16
#define MPC_IPM_SPLIT_STEP 0         /* one step length for primal and dual variables */
/* The remaining BALANCE arguments (SURVEY.md App. A "[verify]"), with the value ASSUMED here and
 * what it means for this implementation.  None of them is a number the kernels read: each fixes
 * a branch of HPIPM's algorithm, and the branch written in rti_core.hpp / rti_wave*.hpp / the
 * oracle is the one named. */
#define MPC_IPM_T_LAM_MIN 2          /* lam and t are clipped to lam_min / t_min IN THE SOLUTION, i.e. by UPDATE_VAR_QP after
                                        the step (what Lane::sweep_bu / Wave2::fw do), not only inside Gamma / gamma */
#define MPC_IPM_COMP_RES_EXIT 1      /* residuals are evaluated at the returned point: the exit test sees true residuals */
#define MPC_IPM_COMP_RES_PRED 1      /* ... and mu_aff of the predictor is evaluated (needed by cond_pred_corr) */
#define MPC_IPM_ABS_FORM 0           /* HPIPM iterates on steps (delta form); the kernels carry zeta = z + dz, which is the
                                        same iteration in exact arithmetic */
#define MPC_IPM_SQUARE_ROOT_ALG 1    /* square-root Riccati (Cholesky of the stage Hessian); here: LDL^T of H_uu / the scan */
#define MPC_IPM_WARM_START 0         /* every QP starts from OCP_QP_INIT_VAR's cold start (acados qp_solver_warm_start 0) */
#define MPC_IPM_VAR_INIT_SCHEME 0    /* the thr0 clipping below, not the safe / heuristic alternatives */
/* constants inside the algorithm (not arguments) */
#define MPC_IPM_THR0 1e-1            /* OCP_QP_INIT_VAR: slacks of the cold start are clipped to this */
#define MPC_IPM_LQ_SWITCH_TOL 1e-5   /* OCP_QP_IPM_DELTA_STEP: lin-system residual above which lq_fact = 1 refactorises */
#define MPC_IPM_COND_PRED_CORR_FAC 2.0 /* corrector rejected when mu_aff(corrector) > fac * mu_aff(predictor) */
#define MPC_IPM_STEP_SCALE_LO 0.99      /* UPDATE_VAR_QP: alpha *= (1-alpha)*lo + alpha*hi when alpha < 1 */
#define MPC_IPM_STEP_SCALE_HI 0.9999999

#endif /* MPC_HPIPM_ARGS_H */
