// rti_core.hpp — one SQP-RTI step of the kinematic-bicycle OCP for ONE instance ("lane"),
// written for the MI355X lane-per-OCP kernels (rti_kernels.hip).  All per-stage data lives
// in structure-of-arrays global memory, [field][stage*comp][instance], so that the 64 lanes
// of a wavefront read/write 64 consecutive doubles (512 B) per access.
//
// What it computes (reference file:line it replaces):
//   linearize()   H5/H7  mobile_robot_expl_vde_forw.c:63-137 integrated by acados ERK4
//                        (acados_solver_mobile_robot.c:160-161, 636-644), exploiting
//                        A = [[1,0,a],[0,1,b],[0,0,1]]
//   residuals     H8/H9  LINEAR_LS gradient/Hessian (:396-486, scaling :380-385) and BGH box
//                        bounds (:493-580), in absolute variables zeta = z + dz
//   x0 handling   H10    stage-0 state fixed to x0bar (idxbxe_0, :513-519)
//   ipm_*         H11    HPIPM d_ocp_qp_ipm_solve, mode BALANCE (:670-693): Mehrotra
//                        predictor-corrector, conditional centering fallback, Riccati
//   finish        H12    full-step update and status mapping of ocp_nlp_sqp_rti
//
// The same source compiles for the host (tests/hostbuild, test infrastructure) so that the
// algorithm can be checked against the dense oracle without a GPU; the product library only
// contains the HIP build.
#pragma once
#include <math.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif

namespace mpc {

// Per-handle constants (kernel argument, lives in SGPRs / constant memory).
struct Consts {
    int N;
    int iter_max;
    double Ts, L, invL;
    double wg[5];  // stage Hessian diagonal, variable order [v, delta, x, y, psi] (= Ts*W)
    double we[3];  // terminal Hessian diagonal [x, y, psi]
    double lb[4], ub[4];  // bounds on [v, delta, x, y]
    // HPIPM arguments (d_ocp_qp_ipm_arg_set_default(BALANCE) + acados qp_iter_max)
    double mu0, alpha_min, res_g_max, res_b_max, res_d_max, res_m_max;
    double reg_prim, lam_min, t_min, tau_min, thr0;
    double itref_tol_g;   // lin-system residual above which HPIPM would refine (diagnostic)
    double lq_tol;        // lin-system residual above which HPIPM would switch to LQ (diagnostic)
};

// Pointers of one handle; every array is [rows][S] with S = padded capacity.
struct Mem {
    size_t S;
    // capsule (persists across calls)
    double *X, *U, *PI;            // (N+1)*3, N*2, N*3
    // inputs of this call
    const double *X0, *YREF, *YREFE;  // 3, N*5, 3
    // workspace
    double *AB, *C;                // N*8 {a,b,B00,B10,B20,B01,B11,B21}, N*3
    double *Z, *PIQ, *LAM, *T;     // (N+1)*5, N*3, N*8, N*8
    double *K, *F, *P, *KFF, *PV;  // N*6, N*3, N*6 (P_{k+1} at row k), N*2, N*3 (p_{k+1} at row k)
    double *RG, *RB;               // (N+1)*5, N*3
    double *DZA, *DZ, *DPI;        // (N+1)*5, (N+1)*5, N*3
    // outputs
    double* U0;                    // 2
    int *STATUS, *QPITER, *DIAG;   // 1 each
};

#define MPC_DIAG_ITREF 1  /* HPIPM's corrector refinement test would have asked for a solve */
#define MPC_DIAG_LQ 2     /* HPIPM's lq_fact=1 test would have switched to the LQ factorisation */

// ------------------------------------------------------------------------------------
struct Lane {
    const Consts& c;
    const Mem& m;
    size_t i;  // instance index
    MPC_HD Lane(const Consts& c_, const Mem& m_, size_t i_) : c(c_), m(m_), i(i_) {}

    MPC_HD double ld(const double* p, int row) const { return p[(size_t)row * m.S + i]; }
    MPC_HD void st(double* p, int row, double v) const { p[(size_t)row * m.S + i] = v; }

    // ---- H5/H7: ERK4 with forward sensitivities, structure-exploiting but with the generic
    //      integrator's operation order (entries that are structurally 0/1 are exact there too)
    MPC_HD void linearize_stage(int k) const {
        const double x0 = ld(m.X, 3 * k), x1 = ld(m.X, 3 * k + 1), psi = ld(m.X, 3 * k + 2);
        const double v = ld(m.U, 2 * k), dl = ld(m.U, 2 * k + 1);
        const double h = c.Ts;
        const double tg = tan(dl), cd = cos(dl);
        const double vL = v / c.L;
        const double om = vL * tg;           // psi_dot, constant over the RK stages
        const double s20 = c.invL * tg;      // d/dt Su(2,0)
        const double s21 = vL / (cd * cd);   // d/dt Su(2,1)
        const double ah = 0.5 * h, a1 = 1.0 * h;
        const double bw0 = h * (1.0 / 6.0), bw1 = h * (2.0 / 6.0);
        // stage abscissae: psi and Su(2,:) at the 4 RK stages (stages 2 and 3 coincide)
        const double psiA = psi, psiB = psi + ah * om, psiD = psi + a1 * om;
        double sA, cA, sB, cB, sD, cD;
        sincos(psiA, &sA, &cA);
        sincos(psiB, &sB, &cB);
        sincos(psiD, &sD, &cD);
        const double u20B = ah * s20, u20D = a1 * s20;
        const double u21B = ah * s21, u21D = a1 * s21;
        // K_s of the non-trivial components
        // stage 1 (Su2x = 0)
        const double k1x = v * cA, k1y = v * sA;
        const double k1a = -(v * sA), k1b = v * cA;
        const double k1B00 = cA - v * (sA * 0.0), k1B10 = sA + v * (cA * 0.0);
        const double k1B01 = -(v * (sA * 0.0)), k1B11 = v * (cA * 0.0);
        // stages 2,3
        const double k2x = v * cB, k2y = v * sB;
        const double k2a = -(v * sB), k2b = v * cB;
        const double k2B00 = cB - v * (sB * u20B), k2B10 = sB + v * (cB * u20B);
        const double k2B01 = -(v * (sB * u21B)), k2B11 = v * (cB * u21B);
        // stage 4
        const double k4x = v * cD, k4y = v * sD;
        const double k4a = -(v * sD), k4b = v * cD;
        const double k4B00 = cD - v * (sD * u20D), k4B10 = sD + v * (cD * u20D);
        const double k4B01 = -(v * (sD * u21D)), k4B11 = v * (cD * u21D);
        // y+ = y0 + sum_s (h b_s) K_s, accumulated in stage order
        const double xn0 = (((x0 + bw0 * k1x) + bw1 * k2x) + bw1 * k2x) + bw0 * k4x;
        const double xn1 = (((x1 + bw0 * k1y) + bw1 * k2y) + bw1 * k2y) + bw0 * k4y;
        const double xn2 = (((psi + bw0 * om) + bw1 * om) + bw1 * om) + bw0 * om;
        const double a = (((0.0 + bw0 * k1a) + bw1 * k2a) + bw1 * k2a) + bw0 * k4a;
        const double b = (((0.0 + bw0 * k1b) + bw1 * k2b) + bw1 * k2b) + bw0 * k4b;
        const double B00 = (((0.0 + bw0 * k1B00) + bw1 * k2B00) + bw1 * k2B00) + bw0 * k4B00;
        const double B10 = (((0.0 + bw0 * k1B10) + bw1 * k2B10) + bw1 * k2B10) + bw0 * k4B10;
        const double B20 = (((0.0 + bw0 * s20) + bw1 * s20) + bw1 * s20) + bw0 * s20;
        const double B01 = (((0.0 + bw0 * k1B01) + bw1 * k2B01) + bw1 * k2B01) + bw0 * k4B01;
        const double B11 = (((0.0 + bw0 * k1B11) + bw1 * k2B11) + bw1 * k2B11) + bw0 * k4B11;
        const double B21 = (((0.0 + bw0 * s21) + bw1 * s21) + bw1 * s21) + bw0 * s21;
        st(m.AB, 8 * k + 0, a);   st(m.AB, 8 * k + 1, b);
        st(m.AB, 8 * k + 2, B00); st(m.AB, 8 * k + 3, B10); st(m.AB, 8 * k + 4, B20);
        st(m.AB, 8 * k + 5, B01); st(m.AB, 8 * k + 6, B11); st(m.AB, 8 * k + 7, B21);
        // affine term of the absolute formulation: zeta_x+ = A zeta_x + B zeta_u + cvec
        st(m.C, 3 * k + 0, xn0 - (x0 + a * psi) - (B00 * v + B01 * dl));
        st(m.C, 3 * k + 1, xn1 - (x1 + b * psi) - (B10 * v + B11 * dl));
        st(m.C, 3 * k + 2, xn2 - psi - (B20 * v + B21 * dl));
    }

    // ---- per-constraint algebra (COMPUTE_GAMMA_GAMMA_QP / COMPUTE_LAM_T_QP) ----
    struct Con {  // one two-sided bound on one variable
        double lam_l, lam_u, t_l, t_u;
        double rd_l, rd_u;  // res_d
        double rm_l, rm_u;  // res_m (possibly corrected)
        double ti_l, ti_u;
    };
    MPC_HD void con_load(int k, int j, double zj, Con& q) const {
        q.lam_l = ld(m.LAM, 8 * k + j); q.lam_u = ld(m.LAM, 8 * k + 4 + j);
        q.t_l = ld(m.T, 8 * k + j);     q.t_u = ld(m.T, 8 * k + 4 + j);
        q.rd_l = c.lb[j] + q.t_l - zj;
        q.rd_u = q.t_u + zj - c.ub[j];
        q.rm_l = q.lam_l * q.t_l;
        q.rm_u = q.lam_u * q.t_u;
        q.ti_l = 1.0 / q.t_l;
        q.ti_u = 1.0 / q.t_u;
    }
    // corrected complementarity residual: mode 0 = affine (lam*t), 1 = Mehrotra corrector
    // (lam*t + dt_aff*dlam_aff - sigma*mu), 2 = centering only (lam*t - sigma*mu)
    MPC_HD void con_correct(Con& q, int mode, double dza_j, double sigma_mu) const {
        if (mode == 0) return;
        if (mode == 1) {
            const double dt_l = dza_j - q.rd_l, dt_u = -dza_j - q.rd_u;
            const double dl_l = -q.ti_l * (q.lam_l * dt_l + q.rm_l);
            const double dl_u = -q.ti_u * (q.lam_u * dt_u + q.rm_u);
            q.rm_l = q.rm_l + dt_l * dl_l - sigma_mu;
            q.rm_u = q.rm_u + dt_u * dl_u - sigma_mu;
        } else {
            q.rm_l = q.rm_l - sigma_mu;
            q.rm_u = q.rm_u - sigma_mu;
        }
    }
    MPC_HD double con_Gamma(const Con& q) const { return q.ti_l * q.lam_l + q.ti_u * q.lam_u; }
    MPC_HD double con_gamma(const Con& q) const {
        return q.ti_l * (q.rm_l - q.lam_l * q.rd_l) - q.ti_u * (q.rm_u - q.lam_u * q.rd_u);
    }
    struct Step { double dt_l, dt_u, dl_l, dl_u; };
    MPC_HD void con_step(const Con& q, double dzj, Step& s) const {
        s.dt_l = dzj - q.rd_l;
        s.dt_u = -dzj - q.rd_u;
        s.dl_l = -q.ti_l * (q.lam_l * s.dt_l + q.rm_l);
        s.dl_u = -q.ti_u * (q.lam_u * s.dt_u + q.rm_u);
    }

    // ---- residual of stage k in forward order; pim = pi_{k-1}, pik = pi_k (0 at k = N);
    //      z = zeta_k (5), zxn = zeta_x_{k+1}.  Writes RG/RB, returns the norms' contributions.
    struct Norms { double g, b, d, m, musum; };
    MPC_HD static double amax(double cur, double v) { const double a = fabs(v); return a > cur ? a : cur; }

    MPC_HD void residual_stage(int k, const double z[5], const double zxn[3], const double pim[3],
                               const double pik[3], Norms& n) const {
        const int N = c.N;
        double rg[5];
        if (k < N) {
            const double a = ld(m.AB, 8 * k), b = ld(m.AB, 8 * k + 1);
            const double B00 = ld(m.AB, 8 * k + 2), B10 = ld(m.AB, 8 * k + 3), B20 = ld(m.AB, 8 * k + 4);
            const double B01 = ld(m.AB, 8 * k + 5), B11 = ld(m.AB, 8 * k + 6), B21 = ld(m.AB, 8 * k + 7);
            rg[0] = c.wg[0] * (z[0] - ld(m.YREF, 5 * k + 3));
            rg[1] = c.wg[1] * (z[1] - ld(m.YREF, 5 * k + 4));
            rg[2] = c.wg[2] * (z[2] - ld(m.YREF, 5 * k + 0)) - pim[0];
            rg[3] = c.wg[3] * (z[3] - ld(m.YREF, 5 * k + 1)) - pim[1];
            rg[4] = c.wg[4] * (z[4] - ld(m.YREF, 5 * k + 2)) - pim[2];
            const int nb = (k == 0) ? 2 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    const double ll = ld(m.LAM, 8 * k + j), lu = ld(m.LAM, 8 * k + 4 + j);
                    const double tl = ld(m.T, 8 * k + j), tu = ld(m.T, 8 * k + 4 + j);
                    rg[j] += lu - ll;
                    n.d = amax(n.d, c.lb[j] + tl - z[j]);
                    n.d = amax(n.d, tu + z[j] - c.ub[j]);
                    const double ml = ll * tl, mu_ = lu * tu;
                    n.m = amax(n.m, ml);
                    n.m = amax(n.m, mu_);
                    n.musum += ml;
                    n.musum += mu_;
                }
            }
            // res_b = cvec + A zeta_x + B zeta_u - zeta_x+
            const double rb0 = ld(m.C, 3 * k + 0) - zxn[0] + (B00 * z[0] + B01 * z[1] + (z[2] + a * z[4]));
            const double rb1 = ld(m.C, 3 * k + 1) - zxn[1] + (B10 * z[0] + B11 * z[1] + (z[3] + b * z[4]));
            const double rb2 = ld(m.C, 3 * k + 2) - zxn[2] + (B20 * z[0] + B21 * z[1] + z[4]);
            st(m.RB, 3 * k + 0, rb0); st(m.RB, 3 * k + 1, rb1); st(m.RB, 3 * k + 2, rb2);
            n.b = amax(n.b, rb0); n.b = amax(n.b, rb1); n.b = amax(n.b, rb2);
            // res_g += [B A]^T pi_k
            rg[0] += B00 * pik[0] + B10 * pik[1] + B20 * pik[2];
            rg[1] += B01 * pik[0] + B11 * pik[1] + B21 * pik[2];
            rg[2] += pik[0];
            rg[3] += pik[1];
            rg[4] += a * pik[0] + b * pik[1] + pik[2];
        } else {
            rg[0] = rg[1] = 0.0;
            rg[2] = c.we[0] * (z[2] - ld(m.YREFE, 0)) - pim[0];
            rg[3] = c.we[1] * (z[3] - ld(m.YREFE, 1)) - pim[1];
            rg[4] = c.we[2] * (z[4] - ld(m.YREFE, 2)) - pim[2];
        }
        if (k < N) { n.g = amax(n.g, rg[0]); n.g = amax(n.g, rg[1]); }
        if (k > 0) { n.g = amax(n.g, rg[2]); n.g = amax(n.g, rg[3]); n.g = amax(n.g, rg[4]); }
#pragma unroll
        for (int q = 0; q < 5; ++q) st(m.RG, 5 * k + q, rg[q]);
    }

    // ---- OCP_QP_INIT_VAR (cold start) fused with the first residual evaluation ----
    MPC_HD void init_and_residual(Norms& n) const {
        const int N = c.N;
        // zeta = current iterate, stage-0 state = x0bar; clip to the interior of the bounds
        for (int k = 0; k <= N; ++k) {
            double z[5];
            z[0] = (k < N) ? ld(m.U, 2 * k) : 0.0;
            z[1] = (k < N) ? ld(m.U, 2 * k + 1) : 0.0;
            if (k == 0) { z[2] = ld(m.X0, 0); z[3] = ld(m.X0, 1); z[4] = ld(m.X0, 2); }
            else { z[2] = ld(m.X, 3 * k); z[3] = ld(m.X, 3 * k + 1); z[4] = ld(m.X, 3 * k + 2); }
            if (k < N) {
                const int nb = (k == 0) ? 2 : 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nb) {
                        double tl = z[j] - c.lb[j];
                        double tu = c.ub[j] - z[j];
                        if (tl < c.thr0) {
                            if (tu < c.thr0) { z[j] = 0.5 * (c.lb[j] + c.ub[j]); tl = c.thr0; tu = c.thr0; }
                            else { tl = c.thr0; z[j] = c.lb[j] + c.thr0; }
                        } else if (tu < c.thr0) { tu = c.thr0; z[j] = c.ub[j] - c.thr0; }
                        st(m.T, 8 * k + j, tl); st(m.T, 8 * k + 4 + j, tu);
                        st(m.LAM, 8 * k + j, c.mu0 / tl); st(m.LAM, 8 * k + 4 + j, c.mu0 / tu);
                    }
                }
                st(m.PIQ, 3 * k, 0.0); st(m.PIQ, 3 * k + 1, 0.0); st(m.PIQ, 3 * k + 2, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) st(m.Z, 5 * k + q, z[q]);
        }
        residual_sweep(n);
    }

    // residuals of all stages from the stored state (used after init; the update pass has
    // its own fused version)
    MPC_HD void residual_sweep(Norms& n) const {
        const int N = c.N;
        n.g = n.b = n.d = n.m = n.musum = 0.0;
        double pim[3] = {0.0, 0.0, 0.0};
        double z[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) z[q] = ld(m.Z, q);
        for (int k = 0; k <= N; ++k) {
            double zn[5] = {0, 0, 0, 0, 0}, pik[3] = {0, 0, 0};
            if (k < N) {
#pragma unroll
                for (int q = 0; q < 5; ++q) zn[q] = ld(m.Z, 5 * (k + 1) + q);
#pragma unroll
                for (int q = 0; q < 3; ++q) pik[q] = ld(m.PIQ, 3 * k + q);
            }
            residual_stage(k, z, zn + 2, pim, pik, n);
#pragma unroll
            for (int q = 0; q < 5; ++q) z[q] = zn[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) pim[q] = pik[q];
        }
    }

    // ---- backward Riccati sweep.  mode 0: factorise + affine rhs (OCP_QP_FACT_SOLVE_KKT_STEP);
    //      mode 1/2: rhs only with the corrected complementarity (OCP_QP_SOLVE_KKT_STEP).
    MPC_HD void backward(int mode, double sigma_mu) const {
        const int N = c.N;
        // value function of stage k+1: P (sym: 00,10,11,20,21,22), p
        double P00 = c.we[0] + c.reg_prim, P10 = 0.0, P11 = c.we[1] + c.reg_prim, P20 = 0.0, P21 = 0.0,
               P22 = c.we[2] + c.reg_prim;
        double p0 = ld(m.RG, 5 * N + 2), p1 = ld(m.RG, 5 * N + 3), p2 = ld(m.RG, 5 * N + 4);
        for (int k = N - 1; k >= 0; --k) {
            const double a = ld(m.AB, 8 * k), b = ld(m.AB, 8 * k + 1);
            const double B00 = ld(m.AB, 8 * k + 2), B10 = ld(m.AB, 8 * k + 3), B20 = ld(m.AB, 8 * k + 4);
            const double B01 = ld(m.AB, 8 * k + 5), B11 = ld(m.AB, 8 * k + 6), B21 = ld(m.AB, 8 * k + 7);
            if (mode == 0) {
                st(m.P, 6 * k + 0, P00); st(m.P, 6 * k + 1, P10); st(m.P, 6 * k + 2, P11);
                st(m.P, 6 * k + 3, P20); st(m.P, 6 * k + 4, P21); st(m.P, 6 * k + 5, P22);
            } else {
                P00 = ld(m.P, 6 * k + 0); P10 = ld(m.P, 6 * k + 1); P11 = ld(m.P, 6 * k + 2);
                P20 = ld(m.P, 6 * k + 3); P21 = ld(m.P, 6 * k + 4); P22 = ld(m.P, 6 * k + 5);
            }
            st(m.PV, 3 * k + 0, p0); st(m.PV, 3 * k + 1, p1); st(m.PV, 3 * k + 2, p2);
            // w = P rb + p
            const double rb0 = ld(m.RB, 3 * k), rb1 = ld(m.RB, 3 * k + 1), rb2 = ld(m.RB, 3 * k + 2);
            const double w0 = P00 * rb0 + P10 * rb1 + P20 * rb2 + p0;
            const double w1 = P10 * rb0 + P11 * rb1 + P21 * rb2 + p1;
            const double w2 = P20 * rb0 + P21 * rb1 + P22 * rb2 + p2;
            // constraints of this stage: Gamma (diagonal) and gamma (gradient)
            const int nb = (k == 0) ? 2 : 4;
            double Gam[4] = {0, 0, 0, 0}, gam[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Con q;
                    con_load(k, j, ld(m.Z, 5 * k + j), q);
                    if (mode != 0) con_correct(q, mode, ld(m.DZA, 5 * k + j), sigma_mu);
                    Gam[j] = con_Gamma(q);
                    gam[j] = con_gamma(q);
                }
            }
            // gradient
            double hu0 = ld(m.RG, 5 * k + 0) + gam[0] + (B00 * w0 + B10 * w1 + B20 * w2);
            double hu1 = ld(m.RG, 5 * k + 1) + gam[1] + (B01 * w0 + B11 * w1 + B21 * w2);
            double l10, id0, id1;  // Huu = L D L^T
            double K00, K01, K02, K10, K11, K12;
            if (mode == 0) {
                // M = B^T P (2x3)
                const double M00 = B00 * P00 + B10 * P10 + B20 * P20;
                const double M01 = B00 * P10 + B10 * P11 + B20 * P21;
                const double M02 = B00 * P20 + B10 * P21 + B20 * P22;
                const double M10 = B01 * P00 + B11 * P10 + B21 * P20;
                const double M11 = B01 * P10 + B11 * P11 + B21 * P21;
                const double M12 = B01 * P20 + B11 * P21 + B21 * P22;
                const double Huu00 = (c.wg[0] + c.reg_prim) + Gam[0] + (M00 * B00 + M01 * B10 + M02 * B20);
                const double Huu10 = M10 * B00 + M11 * B10 + M12 * B20;
                const double Huu11 = (c.wg[1] + c.reg_prim) + Gam[1] + (M10 * B01 + M11 * B11 + M12 * B21);
                id0 = 1.0 / Huu00;
                l10 = Huu10 * id0;
                id1 = 1.0 / (Huu11 - l10 * Huu10);
                st(m.F, 3 * k + 0, id0); st(m.F, 3 * k + 1, l10); st(m.F, 3 * k + 2, id1);
                if (k > 0) {
                    // Hux = M A (2x3): columns 0,1 unchanged, column 2 = a*col0 + b*col1 + col2
                    const double X02 = M00 * a + M01 * b + M02;
                    const double X12 = M10 * a + M11 * b + M12;
                    // K = -Huu^-1 Hux, column by column through L D L^T
                    solve2(id0, l10, id1, M00, M10, K00, K10);
                    solve2(id0, l10, id1, M01, M11, K01, K11);
                    solve2(id0, l10, id1, X02, X12, K02, K12);
                    st(m.K, 6 * k + 0, K00); st(m.K, 6 * k + 1, K01); st(m.K, 6 * k + 2, K02);
                    st(m.K, 6 * k + 3, K10); st(m.K, 6 * k + 4, K11); st(m.K, 6 * k + 5, K12);
                    // S = P A: columns 0,1 = P's, column 2 = a*P(:,0) + b*P(:,1) + P(:,2)
                    const double S02 = P00 * a + P10 * b + P20;
                    const double S12 = P10 * a + P11 * b + P21;
                    const double S22 = P20 * a + P21 * b + P22;
                    // Hxx = Q + Gamma_x + A^T S (sym); row 2 of A^T S = a*S(0,:) + b*S(1,:) + S(2,:)
                    const double Hxx00 = (c.wg[2] + c.reg_prim) + Gam[2] + P00;
                    const double Hxx10 = P10;
                    const double Hxx11 = (c.wg[3] + c.reg_prim) + Gam[3] + P11;
                    const double Hxx20 = a * P00 + b * P10 + P20;
                    const double Hxx21 = a * P10 + b * P11 + P21;
                    const double Hxx22 = (c.wg[4] + c.reg_prim) + (a * S02 + b * S12 + S22);
                    // P_k = Hxx + Hux^T K
                    const double nP00 = Hxx00 + (M00 * K00 + M10 * K10);
                    const double nP10 = Hxx10 + (M01 * K00 + M11 * K10);
                    const double nP11 = Hxx11 + (M01 * K01 + M11 * K11);
                    const double nP20 = Hxx20 + (X02 * K00 + X12 * K10);
                    const double nP21 = Hxx21 + (X02 * K01 + X12 * K11);
                    const double nP22 = Hxx22 + (X02 * K02 + X12 * K12);
                    P00 = nP00; P10 = nP10; P11 = nP11; P20 = nP20; P21 = nP21; P22 = nP22;
                }
            } else {
                id0 = ld(m.F, 3 * k + 0); l10 = ld(m.F, 3 * k + 1); id1 = ld(m.F, 3 * k + 2);
                if (k > 0) {
                    K00 = ld(m.K, 6 * k + 0); K01 = ld(m.K, 6 * k + 1); K02 = ld(m.K, 6 * k + 2);
                    K10 = ld(m.K, 6 * k + 3); K11 = ld(m.K, 6 * k + 4); K12 = ld(m.K, 6 * k + 5);
                }
            }
            double f0, f1;
            solve2(id0, l10, id1, hu0, hu1, f0, f1);  // kff = -Huu^-1 hu
            st(m.KFF, 2 * k + 0, f0); st(m.KFF, 2 * k + 1, f1);
            if (k > 0) {
                // hx = res_g_x + gamma_x + A^T w ; p_k = hx + K^T hu
                const double hx0 = ld(m.RG, 5 * k + 2) + gam[2] + w0;
                const double hx1 = ld(m.RG, 5 * k + 3) + gam[3] + w1;
                const double hx2 = ld(m.RG, 5 * k + 4) + (a * w0 + b * w1 + w2);
                p0 = hx0 + (K00 * hu0 + K10 * hu1);
                p1 = hx1 + (K01 * hu0 + K11 * hu1);
                p2 = hx2 + (K02 * hu0 + K12 * hu1);
            }
        }
    }
    // x = -(L D L^T)^-1 r for the 2x2 block
    MPC_HD static void solve2(double id0, double l10, double id1, double r0, double r1, double& x0, double& x1) {
        const double y1 = r1 - l10 * r0;
        x1 = -(y1 * id1);
        x0 = -(r0 * id0) - l10 * x1;
    }

    // ---- forward sweep.  Computes the primal step of all stages, the step lengths
    //      (COMPUTE_ALPHA_QP) and the three sums that give mu_aff for any alpha.
    //      mode 0: affine step -> DZA; mode 1/2: final step -> DZ, DPI and the stationarity
    //      residual of the linear system (what HPIPM's refinement loop tests).
    struct FwdOut { double a_prim, a_dual, s1, s2, lin_g; };
    MPC_HD void forward(int mode, double sigma_mu, FwdOut& o) const {
        const int N = c.N;
        double a_prim = -1.0, a_dual = -1.0, s1 = 0.0, s2 = 0.0, lin_g = 0.0;
        double dx0 = 0.0, dx1 = 0.0, dx2 = 0.0;  // stage-0 state is fixed
        double dpm0 = 0.0, dpm1 = 0.0, dpm2 = 0.0;  // dpi_{k-1}
        double* DZo = (mode == 0) ? m.DZA : m.DZ;
        for (int k = 0; k < N; ++k) {
            const double a = ld(m.AB, 8 * k), b = ld(m.AB, 8 * k + 1);
            const double B00 = ld(m.AB, 8 * k + 2), B10 = ld(m.AB, 8 * k + 3), B20 = ld(m.AB, 8 * k + 4);
            const double B01 = ld(m.AB, 8 * k + 5), B11 = ld(m.AB, 8 * k + 6), B21 = ld(m.AB, 8 * k + 7);
            double du0 = ld(m.KFF, 2 * k), du1 = ld(m.KFF, 2 * k + 1);
            if (k > 0) {
                du0 += ld(m.K, 6 * k + 0) * dx0 + ld(m.K, 6 * k + 1) * dx1 + ld(m.K, 6 * k + 2) * dx2;
                du1 += ld(m.K, 6 * k + 3) * dx0 + ld(m.K, 6 * k + 4) * dx1 + ld(m.K, 6 * k + 5) * dx2;
            }
            const double dz[5] = {du0, du1, dx0, dx1, dx2};
#pragma unroll
            for (int q = 0; q < 5; ++q) st(DZo, 5 * k + q, dz[q]);
            // constraints: dt, dlam, step lengths, mu_aff sums
            const int nb = (k == 0) ? 2 : 4;
            double dlam_net[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Con q;
                    con_load(k, j, ld(m.Z, 5 * k + j), q);
                    if (mode != 0) con_correct(q, mode, ld(m.DZA, 5 * k + j), sigma_mu);
                    Step s;
                    con_step(q, dz[j], s);
                    if (a_dual * s.dl_l > q.lam_l) a_dual = q.lam_l / s.dl_l;
                    if (a_prim * s.dt_l > q.t_l) a_prim = q.t_l / s.dt_l;
                    if (a_dual * s.dl_u > q.lam_u) a_dual = q.lam_u / s.dl_u;
                    if (a_prim * s.dt_u > q.t_u) a_prim = q.t_u / s.dt_u;
                    s1 += q.lam_l * s.dt_l + q.t_l * s.dl_l;
                    s1 += q.lam_u * s.dt_u + q.t_u * s.dl_u;
                    s2 += s.dl_l * s.dt_l;
                    s2 += s.dl_u * s.dt_u;
                    dlam_net[j] = s.dl_u - s.dl_l;
                }
            }
            // dx+ = A dx + B du + rb
            const double nx0 = ld(m.RB, 3 * k + 0) + (B00 * du0 + B01 * du1 + (dx0 + a * dx2));
            const double nx1 = ld(m.RB, 3 * k + 1) + (B10 * du0 + B11 * du1 + (dx1 + b * dx2));
            const double nx2 = ld(m.RB, 3 * k + 2) + (B20 * du0 + B21 * du1 + dx2);
            if (mode != 0) {
                // dpi_k = P_{k+1} dx+ + p_{k+1}
                const double P00 = ld(m.P, 6 * k + 0), P10 = ld(m.P, 6 * k + 1), P11 = ld(m.P, 6 * k + 2);
                const double P20 = ld(m.P, 6 * k + 3), P21 = ld(m.P, 6 * k + 4), P22 = ld(m.P, 6 * k + 5);
                const double dp0 = P00 * nx0 + P10 * nx1 + P20 * nx2 + ld(m.PV, 3 * k + 0);
                const double dp1 = P10 * nx0 + P11 * nx1 + P21 * nx2 + ld(m.PV, 3 * k + 1);
                const double dp2 = P20 * nx0 + P21 * nx1 + P22 * nx2 + ld(m.PV, 3 * k + 2);
                st(m.DPI, 3 * k + 0, dp0); st(m.DPI, 3 * k + 1, dp1); st(m.DPI, 3 * k + 2, dp2);
                // stationarity rows of the linear system at this stage
                const double lg0 = c.wg[0] * du0 + ld(m.RG, 5 * k + 0) + dlam_net[0] + (B00 * dp0 + B10 * dp1 + B20 * dp2);
                const double lg1 = c.wg[1] * du1 + ld(m.RG, 5 * k + 1) + dlam_net[1] + (B01 * dp0 + B11 * dp1 + B21 * dp2);
                lin_g = amax(lin_g, lg0); lin_g = amax(lin_g, lg1);
                if (k > 0) {
                    const double lg2 = c.wg[2] * dx0 + ld(m.RG, 5 * k + 2) + dlam_net[2] - dpm0 + dp0;
                    const double lg3 = c.wg[3] * dx1 + ld(m.RG, 5 * k + 3) + dlam_net[3] - dpm1 + dp1;
                    const double lg4 = c.wg[4] * dx2 + ld(m.RG, 5 * k + 4) - dpm2 + (a * dp0 + b * dp1 + dp2);
                    lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
                }
                dpm0 = dp0; dpm1 = dp1; dpm2 = dp2;
            }
            dx0 = nx0; dx1 = nx1; dx2 = nx2;
        }
        st(DZo, 5 * N + 0, 0.0); st(DZo, 5 * N + 1, 0.0);
        st(DZo, 5 * N + 2, dx0); st(DZo, 5 * N + 3, dx1); st(DZo, 5 * N + 4, dx2);
        if (mode != 0) {
            const double lg2 = c.we[0] * dx0 + ld(m.RG, 5 * N + 2) - dpm0;
            const double lg3 = c.we[1] * dx1 + ld(m.RG, 5 * N + 3) - dpm1;
            const double lg4 = c.we[2] * dx2 + ld(m.RG, 5 * N + 4) - dpm2;
            lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
        }
        o.a_prim = -a_prim; o.a_dual = -a_dual; o.s1 = s1; o.s2 = s2; o.lin_g = lin_g;
    }

    // ---- UPDATE_VAR_QP fused with the next residual evaluation (OCP_QP_RES_COMPUTE) ----
    MPC_HD void update_and_residual(int mode, double sigma_mu, double alpha, Norms& n) const {
        const int N = c.N;
        n.g = n.b = n.d = n.m = n.musum = 0.0;
        double pim[3] = {0.0, 0.0, 0.0};
        double z[5];
        update_stage_primal(0, mode, sigma_mu, alpha, z);
        for (int k = 0; k <= N; ++k) {
            double zn[5] = {0, 0, 0, 0, 0}, pik[3] = {0, 0, 0};
            if (k < N) {
                update_stage_primal(k + 1, mode, sigma_mu, alpha, zn);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    pik[q] = ld(m.PIQ, 3 * k + q) + alpha * ld(m.DPI, 3 * k + q);
                    st(m.PIQ, 3 * k + q, pik[q]);
                }
            }
            residual_stage(k, z, zn + 2, pim, pik, n);
#pragma unroll
            for (int q = 0; q < 5; ++q) z[q] = zn[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) pim[q] = pik[q];
        }
    }
    // zeta, lam, t of stage k += alpha * step (dlam, dt recomputed exactly as in forward())
    MPC_HD void update_stage_primal(int k, int mode, double sigma_mu, double alpha, double z[5]) const {
        const int N = c.N;
        double zo[5], dz[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { zo[q] = ld(m.Z, 5 * k + q); dz[q] = ld(m.DZ, 5 * k + q); }
        if (k < N) {
            const int nb = (k == 0) ? 2 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Con q;
                    con_load(k, j, zo[j], q);
                    con_correct(q, mode, ld(m.DZA, 5 * k + j), sigma_mu);
                    Step s;
                    con_step(q, dz[j], s);
                    double ll = q.lam_l + alpha * s.dl_l, lu = q.lam_u + alpha * s.dl_u;
                    double tl = q.t_l + alpha * s.dt_l, tu = q.t_u + alpha * s.dt_u;
                    if (ll < c.lam_min) ll = c.lam_min;
                    if (lu < c.lam_min) lu = c.lam_min;
                    if (tl < c.t_min) tl = c.t_min;
                    if (tu < c.t_min) tu = c.t_min;
                    st(m.LAM, 8 * k + j, ll); st(m.LAM, 8 * k + 4 + j, lu);
                    st(m.T, 8 * k + j, tl);   st(m.T, 8 * k + 4 + j, tu);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            z[q] = zo[q] + alpha * dz[q];
            st(m.Z, 5 * k + q, z[q]);
        }
    }

    // ---- the whole RTI step of this lane ----
    MPC_HD void rti_step() const {
        const int N = c.N;
        for (int k = 0; k < N; ++k) linearize_stage(k);

        Norms n;
        init_and_residual(n);
        const int nc = 4 + 8 * (N - 1);
        const double nc_inv = 1.0 / nc;
        double mu = n.musum * nc_inv;
        double alpha = 1.0;
        int kk = 0, diag = 0;
        while (kk < c.iter_max && alpha > c.alpha_min &&
               (n.g > c.res_g_max || n.b > c.res_b_max || n.d > c.res_d_max ||
                fabs(n.m - c.tau_min) > c.res_m_max)) {
            // --- OCP_QP_IPM_DELTA_STEP ---
            backward(0, 0.0);
            FwdOut fa;
            forward(0, 0.0, fa);
            double a_aff = fa.a_prim < fa.a_dual ? fa.a_prim : fa.a_dual;  // -max(-ap,-ad)
            // mu_aff = sum (lam + a dlam)(t + a dt) / nc
            const double mu_aff0 = (n.musum + a_aff * fa.s1 + (a_aff * a_aff) * fa.s2) * nc_inv;
            const double tmp = mu_aff0 / mu;
            const double sigma = tmp * tmp * tmp;
            double sigma_mu = sigma * mu;
            if (!(sigma_mu > c.tau_min)) sigma_mu = c.tau_min;
            int mode = 1;
            backward(1, sigma_mu);
            FwdOut fc;
            forward(1, sigma_mu, fc);
            alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
            // conditional predictor-corrector: fall back to pure centering
            const double mu_aff1 = (n.musum + alpha * fc.s1 + (alpha * alpha) * fc.s2) * nc_inv;
            if (mu_aff1 > 2.0 * mu_aff0) {
                mode = 2;
                backward(2, sigma_mu);
                forward(2, sigma_mu, fc);
                alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
            }
            if (!(fc.lin_g < c.itref_tol_g)) diag |= MPC_DIAG_ITREF;
            if (!(fc.lin_g < c.lq_tol)) diag |= MPC_DIAG_LQ;
            // --- UPDATE_VAR_QP + OCP_QP_RES_COMPUTE ---
            double a_upd = alpha;
            if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * 0.99 + a_upd * 0.9999999);
            update_and_residual(mode, sigma_mu, a_upd, n);
            mu = n.musum * nc_inv;
            ++kk;
        }
        // status (d_ocp_qp_ipm_solve tail + ocp_qp_hpipm + ocp_nlp_sqp_rti mapping)
        int hs;
        if (kk == c.iter_max) hs = 1;
        else if (alpha <= c.alpha_min) hs = 2;
        else if (mu != mu) hs = 3;
        else hs = 0;
        const bool ok = (hs == 0 || hs == 1);
        m.QPITER[i] = kk;
        m.DIAG[i] = diag;
        m.STATUS[i] = ok ? 0 : 4;
        if (ok) {
            for (int k = 0; k <= N; ++k) {
                if (k < N) {
                    st(m.U, 2 * k, ld(m.Z, 5 * k)); st(m.U, 2 * k + 1, ld(m.Z, 5 * k + 1));
                    st(m.PI, 3 * k, ld(m.PIQ, 3 * k)); st(m.PI, 3 * k + 1, ld(m.PIQ, 3 * k + 1));
                    st(m.PI, 3 * k + 2, ld(m.PIQ, 3 * k + 2));
                }
                st(m.X, 3 * k, ld(m.Z, 5 * k + 2)); st(m.X, 3 * k + 1, ld(m.Z, 5 * k + 3));
                st(m.X, 3 * k + 2, ld(m.Z, 5 * k + 4));
            }
        }
        st(m.U0, 0, ld(m.U, 0));
        st(m.U0, 1, ld(m.U, 1));
    }
};

}  // namespace mpc
