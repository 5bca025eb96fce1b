// consts.hpp — turns an mpc_ocp_spec_t (include/mpc_batch.h) into the kernel constants:
// what <name>_acados_create bakes (acados_solver_mobile_robot.c:358-602) plus HPIPM's
// d_ocp_qp_ipm_arg_set_default(mode) and acados' qp_iter_max override (:670-682).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mpc_batch.h"
#include "../../include/mpc_hpipm_args.h"
#include "rti_core.hpp"

namespace mpc {

// Acceptance tests of the wave model's parallel scans (rti_wave.hpp): fixed, part of the solver's
// definition (they decide when an iteration is redone with the sequential recursions).
constexpr double kScanTolP = 1e-11;     // relative residual of the Riccati equation (MPC_SCAN_CHECK_P builds only)
// Acceptance test of a scan direction (rti_wave*.hpp fw): the rows of the linearised KKT system that
// the scans produce must be met to
//   dynamics rows:      kScanTolLin = 1e-8, absolute (= HPIPM's res_b_max: a residual left in these rows
//                       IS the next iterate's dynamics residual, whatever the weights);
//   stationarity rows:  2e-6 * (smallest Hessian diagonal entry), kept within [1e-8, res_g_max / 4].
// A residual r in the stationarity rows moves the solution by at most r / lambda_min(H), and
// lambda_min(H) >= the smallest entry of the diagonal cost Hessian (barrier terms only add to it): the
// bound keeps that below 2e-6 in the worst direction.  With the reference's weights (Ts * W_delta =
// 5e-3) it IS 1e-8; with heavier weights the same solution accuracy allows a proportionally larger
// residual, up to a quarter of the 1e-6 at which HPIPM itself would start refining the direction.
constexpr double kScanTolLin = 1e-8;
constexpr double kScanTolStatPerHess = 2e-6;

// Returns nullptr when the spec is acceptable, else what is wrong with it.
inline const char* spec_problem(const mpc_ocp_spec_t* s) {
    if (!s) return "spec is NULL";
    if (s->N < 1 || s->N > MPC_MAX_N) return "N out of range [1, MPC_MAX_N]";
    if (!(s->Ts > 0.0) || !(s->Ts < 1e3)) return "Ts must be positive and finite";
    if (!(s->L >= 0.0) || !(s->L < 1e3)) return "L must be 0 (the generated literals) or a positive finite wheelbase";
    if (s->qp_iter_max < 0 || s->sqp_max_iter < 1) return "qp_iter_max must be >= 0 and sqp_max_iter >= 1";
    // Only the mode the reference uses is implemented (and even that one is parity unpinned):
    // SPEED / SPEED_ABS use split steps or the absolute form, ROBUST a different factorisation.
    if (s->hpipm_mode != MPC_HPIPM_BALANCE) return "hpipm_mode: only MPC_HPIPM_BALANCE is implemented";
    if (s->sqp_max_iter > 1 && !(s->nlp_tol > 0.0 && s->nlp_tol < 1e3)) return "nlp_tol must be positive and finite when sqp_max_iter > 1";
    for (int i = 0; i < 5; ++i) if (!(s->W[i] > 0.0) || !(s->W[i] < 1e300)) return "W must be positive and finite";
    for (int i = 0; i < 3; ++i) if (!(s->W_e[i] > 0.0) || !(s->W_e[i] < 1e300)) return "W_e must be positive and finite";
    for (int i = 0; i < 2; ++i) {
        if (!(s->lbu[i] < s->ubu[i])) return "lbu must be below ubu";
        if (!(s->lbx[i] < s->ubx[i])) return "lbx must be below ubx";
    }
    return nullptr;
}
inline int spec_valid(const mpc_ocp_spec_t* s) { return spec_problem(s) == nullptr; }

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
    // HPIPM arguments of the reference's mode (include/mpc_hpipm_args.h) + acados' qp_iter_max
    c.mu0 = MPC_IPM_MU0;
    c.alpha_min = MPC_IPM_ALPHA_MIN;
    c.res_g_max = MPC_IPM_RES_G_MAX; c.res_b_max = MPC_IPM_RES_B_MAX;
    c.res_d_max = MPC_IPM_RES_D_MAX; c.res_m_max = MPC_IPM_RES_M_MAX;
    c.reg_prim = MPC_IPM_REG_PRIM;
    c.lam_min = MPC_IPM_LAM_MIN; c.t_min = MPC_IPM_T_MIN; c.tau_min = MPC_IPM_TAU_MIN;
    c.thr0 = MPC_IPM_THR0;
    c.itref_tol_g = c.res_g_max;
    c.lq_tol = MPC_IPM_LQ_SWITCH_TOL;
    c.sqp = s.sqp_max_iter > 1 ? 1 : 0;
    c.nlp_tol = s.nlp_tol;
    for (int i = 0; i < 5; ++i) c.wgr[i] = c.wg[i] + c.reg_prim;
    for (int i = 0; i < 3; ++i) c.wer[i] = c.we[i] + c.reg_prim;
    c.nc_inv = 1.0 / (4 + 8 * (c.N - 1));
    c.scan_tol_P = kScanTolP;
    double wmin = c.wg[0];
    for (int i = 1; i < 5; ++i) wmin = c.wg[i] < wmin ? c.wg[i] : wmin;
    for (int i = 0; i < 3; ++i) wmin = c.we[i] < wmin ? c.we[i] : wmin;
    double tol = kScanTolStatPerHess * wmin;
    if (tol < kScanTolLin) tol = kScanTolLin;
    if (tol > 0.25 * MPC_IPM_RES_G_MAX) tol = 0.25 * MPC_IPM_RES_G_MAX;
    c.scan_tol_lin = tol;
    c.scan_dyn_scale = tol / kScanTolLin;   // dynamics-row residuals are scaled by this, then held against scan_tol_lin
    return c;
}

// Workspace geometry (see the layout comment in rti_core.hpp).
inline size_t wave_bytes_for(int N) { return (size_t)(N + 1) * MPC_RPS * 64 * sizeof(double); }
inline size_t n_waves_for(size_t capacity) { return (capacity + 63) / 64; }
inline size_t slab_bytes_for(int N, size_t capacity) { return n_waves_for(capacity) * wave_bytes_for(N); }
inline size_t cap_doubles_for(int N, size_t capacity) { return n_waves_for(capacity) * 64 * (size_t)(N + 1) * MPC_CAP; }
inline size_t islab_ints_for(size_t capacity) { return 16 * n_waves_for(capacity) * 64; }
// the handle's small per-instance arrays: STATUS, QPITER, DIAG, SQPITER, ACTIVE (int), NLPRES [4],
// REDO (64 + S ints)
inline Mem make_mem(int* islab, size_t capacity, int N) {
    const size_t S = n_waves_for(capacity) * 64;
    Mem m;
    m.slab = nullptr;
    m.wave_bytes = wave_bytes_for(N);
    m.STATUS = islab; m.QPITER = islab + S; m.DIAG = islab + 2 * S;
    m.SQPITER = islab + 3 * S; m.ACTIVE = islab + 4 * S;
    m.NLPRES = reinterpret_cast<double*>(islab + 6 * S);   // S is a multiple of 64: 8-byte aligned
    m.cap = nullptr; m.lam = nullptr; m.u0 = nullptr;
    m.REDO = islab + 14 * S;   // 2 S >= 64 + S ints
    m.IPM = nullptr; m.ipm_slots = 0;
    return m;
}
// element (instance i, stage k, row r) as an index in doubles from the slab start
inline size_t slab_index(const Mem& m, size_t i, int k, int r) {
    return (i >> 6) * (m.wave_bytes / sizeof(double)) + ((size_t)(k * MPC_RPS + r) << 6) + (i & 63);
}

}  // namespace mpc
