"""An independent literal copy of HPIPM's BALANCE arguments, so that a one-sided edit of
include/mpc_hpipm_args.h — which the product AND the oracle include — fails a test.  The values
are restated from the published HPIPM sources (the citation names the upstream file and routine;
HPIPM is not in the reference tree and not in this image: PARITY UNPINNED, as the header says).
The acados side (mode, iter_max override) is pinned to the reference's own files by
tests/test_variants_golden.py."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> (value, where HPIPM sets / uses it)
BALANCE = {
    "MPC_IPM_MU0": (1e1, "ocp_qp/x_ocp_qp_ipm.c OCP_QP_IPM_ARG_SET_DEFAULT, mode BALANCE: mu0"),
    "MPC_IPM_ALPHA_MIN": (1e-12, "same: alpha_min"),
    "MPC_IPM_RES_G_MAX": (1e-6, "same: res_g_max"),
    "MPC_IPM_RES_B_MAX": (1e-8, "same: res_b_max"),
    "MPC_IPM_RES_D_MAX": (1e-8, "same: res_d_max"),
    "MPC_IPM_RES_M_MAX": (1e-8, "same: res_m_max"),
    "MPC_IPM_ITER_MAX_BALANCE": (30, "same: iter_max (acados overrides it: acados_solver_mobile_robot.c:682)"),
    "MPC_IPM_PRED_CORR": (1, "same: pred_corr"),
    "MPC_IPM_COND_PRED_CORR": (1, "same: cond_pred_corr"),
    "MPC_IPM_ITREF_PRED_MAX": (0, "same: itref_pred_max"),
    "MPC_IPM_ITREF_CORR_MAX": (2, "same: itref_corr_max"),
    "MPC_IPM_REG_PRIM": (1e-15, "same: reg_prim"),
    "MPC_IPM_LQ_FACT": (1, "same: lq_fact"),
    "MPC_IPM_LAM_MIN": (1e-16, "same: lam_min"),
    "MPC_IPM_T_MIN": (1e-16, "same: t_min"),
    "MPC_IPM_TAU_MIN": (1e-16, "same: tau_min"),
    "MPC_IPM_SPLIT_STEP": (0, "same: split_step"),
    "MPC_IPM_T_LAM_MIN": (2, "same: t_lam_min"),
    "MPC_IPM_COMP_RES_EXIT": (1, "same: comp_res_exit"),
    "MPC_IPM_COMP_RES_PRED": (1, "same: comp_res_pred"),
    "MPC_IPM_ABS_FORM": (0, "same: abs_form"),
    "MPC_IPM_SQUARE_ROOT_ALG": (1, "same: square_root_alg"),
    "MPC_IPM_WARM_START": (0, "same: warm_start"),
    "MPC_IPM_VAR_INIT_SCHEME": (0, "same: var_init_scheme"),
    "MPC_IPM_THR0": (1e-1, "ocp_qp/x_ocp_qp_ipm.c OCP_QP_INIT_VAR: thr0"),
    "MPC_IPM_LQ_SWITCH_TOL": (1e-5, "ocp_qp/x_ocp_qp_ipm.c OCP_QP_IPM_DELTA_STEP: itref_qp_norm threshold of the lq_fact == 1 branch"),
    "MPC_IPM_COND_PRED_CORR_FAC": (2.0, "OCP_QP_IPM_DELTA_STEP: mu_aff(corrector) > 2.0 * mu_aff(predictor)"),
    "MPC_IPM_STEP_SCALE_LO": (0.99, "ipm_core/x_core_qp_ipm_aux.c UPDATE_VAR_QP: alpha *= (1 - alpha) * 0.99 + alpha * 0.9999999"),
    "MPC_IPM_STEP_SCALE_HI": (0.9999999, "same"),
}


def test_header_matches_the_independent_table():
    text = open(os.path.join(ROOT, "include", "mpc_hpipm_args.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    found = dict(re.findall(r"#define\s+(MPC_IPM_[A-Z0-9_]+)\s+([-+0-9.eE]+)", text))
    assert set(found) == set(BALANCE), set(found) ^ set(BALANCE)
    for name, (value, where) in BALANCE.items():
        assert float(found[name]) == float(value), (name, found[name], value, where)
