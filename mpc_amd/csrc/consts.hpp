// consts.hpp — turns an mpc_ocp_spec_t (include/mpc_batch.h) into the kernel constants:
// what <name>_acados_create bakes (acados_solver_mobile_robot.c:358-602) plus HPIPM's
// d_ocp_qp_ipm_arg_set_default(mode) and acados' qp_iter_max override (:670-682).
#pragma once
#include <math.h>
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
    return c;
}

// Rows of every per-stage array (for allocation); order must match Mem.
struct Rows {
    int X, U, PI, X0, YREF, YREFE, AB, C, Z, PIQ, LAM, T, K, F, P, KFF, PV, RG, RB, DZA, DZ, DPI, U0;
    int total;
};
inline Rows rows_for(int N) {
    Rows r;
    r.X = 3 * (N + 1); r.U = 2 * N; r.PI = 3 * N; r.X0 = 3; r.YREF = 5 * N; r.YREFE = 3;
    r.AB = 8 * N; r.C = 3 * N; r.Z = 5 * (N + 1); r.PIQ = 3 * N; r.LAM = 8 * N; r.T = 8 * N;
    r.K = 6 * N; r.F = 3 * N; r.P = 6 * N; r.KFF = 2 * N; r.PV = 3 * N; r.RG = 5 * (N + 1); r.RB = 3 * N;
    r.DZA = 5 * (N + 1); r.DZ = 5 * (N + 1); r.DPI = 3 * N; r.U0 = 2;
    r.total = r.X + r.U + r.PI + r.X0 + r.YREF + r.YREFE + r.AB + r.C + r.Z + r.PIQ + r.LAM + r.T + r.K +
              r.F + r.P + r.KFF + r.PV + r.RG + r.RB + r.DZA + r.DZ + r.DPI + r.U0;
    return r;
}
// Carves one slab of `total*S` doubles (+ 3*S ints) into a Mem.
inline Mem carve(double* slab, int* islab, size_t S, int N) {
    const Rows r = rows_for(N);
    Mem m;
    m.S = S;
    double* p = slab;
    auto take = [&](int rows) { double* q = p; p += (size_t)rows * S; return q; };
    m.X = take(r.X); m.U = take(r.U); m.PI = take(r.PI);
    m.X0 = take(r.X0); m.YREF = take(r.YREF); m.YREFE = take(r.YREFE);
    m.AB = take(r.AB); m.C = take(r.C); m.Z = take(r.Z); m.PIQ = take(r.PIQ); m.LAM = take(r.LAM);
    m.T = take(r.T); m.K = take(r.K); m.F = take(r.F); m.P = take(r.P); m.KFF = take(r.KFF);
    m.PV = take(r.PV); m.RG = take(r.RG); m.RB = take(r.RB); m.DZA = take(r.DZA); m.DZ = take(r.DZ);
    m.DPI = take(r.DPI); m.U0 = take(r.U0);
    m.STATUS = islab; m.QPITER = islab + S; m.DIAG = islab + 2 * S;
    return m;
}

}  // namespace mpc
