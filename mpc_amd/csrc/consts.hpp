// consts.hpp — turns an mpc_ocp_spec_t (include/mpc_batch.h) into the kernel constants:
// what <name>_acados_create bakes (acados_solver_mobile_robot.c:358-602) plus HPIPM's
// d_ocp_qp_ipm_arg_set_default(mode) and acados' qp_iter_max override (:670-682).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mpc_batch.h"
#include "rti_core.hpp"

namespace mpc {

inline int spec_valid(const mpc_ocp_spec_t* s) {
    if (!s) return 0;
    if (s->N < 1 || s->N > MPC_MAX_N) return 0;
    if (!(s->Ts > 0.0)) return 0;
    if (s->qp_iter_max < 0 || s->sqp_max_iter < 1) return 0;
    for (int i = 0; i < 5; ++i) if (!(s->W[i] > 0.0)) return 0;
    for (int i = 0; i < 3; ++i) if (!(s->W_e[i] > 0.0)) return 0;
    for (int i = 0; i < 2; ++i) {
        if (!(s->lbu[i] < s->ubu[i])) return 0;
        if (!(s->lbx[i] < s->ubx[i])) return 0;
    }
    return 1;
}

inline Consts make_consts(const mpc_ocp_spec_t& s) {
    Consts c;
    memset(&c, 0, sizeof(c));
    c.N = s.N;
    c.iter_max = s.qp_iter_max;
    c.Ts = s.Ts;
    if (s.L == 0.0 || s.L == 2.7000000000000002e-01) {
        // literals of mobile_robot_expl_vde_forw.c:74,121
        c.L = 2.7000000000000002e-01;
        c.invL = 3.7037037037037033e+00;
    } else {
        c.L = s.L;
        c.invL = 1.0 / s.L;
    }
    // LINEAR_LS: W = Wc Wc^T, Hess = scaling * Wc Wc^T (ocp_nlp_cost_ls.c); y = [x y psi v delta]
    const int y_of_var[5] = {3, 4, 0, 1, 2};
    for (int i = 0; i < 5; ++i) {
        const double wc = sqrt(s.W[y_of_var[i]]);
        c.wg[i] = (wc * wc) * s.Ts;
    }
    for (int i = 0; i < 3; ++i) {
        const double wc = sqrt(s.W_e[i]);
        c.we[i] = wc * wc;
    }
    c.lb[0] = s.lbu[0]; c.lb[1] = s.lbu[1]; c.lb[2] = s.lbx[0]; c.lb[3] = s.lbx[1];
    c.ub[0] = s.ubu[0]; c.ub[1] = s.ubu[1]; c.ub[2] = s.ubx[0]; c.ub[3] = s.ubx[1];
    // HPIPM modes (x_ocp_qp_ipm.c, OCP_QP_IPM_ARG_SET_DEFAULT)
    c.mu0 = 1e1;
    c.alpha_min = 1e-12;
    c.res_g_max = 1e-6; c.res_b_max = 1e-8; c.res_d_max = 1e-8; c.res_m_max = 1e-8;
    c.reg_prim = 1e-15;
    c.lam_min = 1e-16; c.t_min = 1e-16; c.tau_min = 1e-16;
    c.thr0 = 1e-1;
    if (s.hpipm_mode == MPC_HPIPM_SPEED || s.hpipm_mode == MPC_HPIPM_SPEED_ABS) {
        c.res_g_max = c.res_b_max = c.res_d_max = 1e0;
    } else if (s.hpipm_mode == MPC_HPIPM_ROBUST) {
        c.mu0 = 1e2;
    }
    c.itref_tol_g = c.res_g_max;
    c.lq_tol = 1e-5;
    c.sqp = s.sqp_max_iter > 1 ? 1 : 0;
    c.nlp_tol = s.nlp_tol;
    c.scan_tol_P = getenv("MPC_SCAN_TOL_P") ? atof(getenv("MPC_SCAN_TOL_P")) : 1e-11;
    c.scan_tol_lin = getenv("MPC_SCAN_TOL_LIN") ? atof(getenv("MPC_SCAN_TOL_LIN")) : 1e-8;
    return c;
}

// Workspace geometry (see the layout comment in rti_core.hpp).
inline size_t wave_bytes_for(int N) { return (size_t)(N + 1) * MPC_RPS * 64 * sizeof(double); }
inline size_t n_waves_for(size_t capacity) { return (capacity + 63) / 64; }
inline size_t slab_bytes_for(int N, size_t capacity) { return n_waves_for(capacity) * wave_bytes_for(N); }
inline size_t cap_doubles_for(int N, size_t capacity) { return n_waves_for(capacity) * 64 * (size_t)(N + 1) * MPC_CAP; }
inline size_t islab_ints_for(size_t capacity) { return 14 * n_waves_for(capacity) * 64; }
inline Mem make_mem(char* slab, int* islab, size_t capacity, int N) {
    const size_t S = n_waves_for(capacity) * 64;
    Mem m;
    m.slab = slab;
    m.wave_bytes = wave_bytes_for(N);
    m.STATUS = islab; m.QPITER = islab + S; m.DIAG = islab + 2 * S;
    m.SQPITER = islab + 3 * S; m.ACTIVE = islab + 4 * S;
    m.NLPRES = reinterpret_cast<double*>(islab + 6 * S);   // S is a multiple of 64: 8-byte aligned
    m.cap = nullptr;
    return m;
}
// element (instance i, stage k, row r) as an index in doubles from the slab start
inline size_t slab_index(const Mem& m, size_t i, int k, int r) {
    return (i >> 6) * (m.wave_bytes / sizeof(double)) + ((size_t)(k * MPC_RPS + r) << 6) + (i & 63);
}

}  // namespace mpc
