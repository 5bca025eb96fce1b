// rti_core.hpp — one SQP-RTI step of the kinematic-bicycle OCP for ONE instance ("lane"),
// written for the MI355X lane-per-OCP kernels (mpc_batch.hip).  All per-stage data lives in
// global memory as [wave][stage][row][lane], so that the 64 lanes of a wavefront read/write
// 64 consecutive doubles (512 B) per access and address stepping stays on the scalar unit.
//
// What it computes (reference file:line it replaces):
//   linearize()   H5/H7  mobile_robot_expl_vde_forw.c:63-137 integrated by acados ERK4
//                        (acados_solver_mobile_robot.c:160-161, 636-644), exploiting
//                        A = [[1,0,a],[0,1,b],[0,0,1]]
//   residuals     H8/H9  LINEAR_LS gradient/Hessian (:396-486, scaling :380-385) and BGH box
//                        bounds (:493-580), in absolute variables zeta = z + dz
//   x0 handling   H10    stage-0 state fixed to x0bar (idxbxe_0, :513-519)
//   sweep_*       H11    HPIPM d_ocp_qp_ipm_solve, mode BALANCE (:670-693): Mehrotra
//                        predictor-corrector, conditional centering fallback, Riccati.
//                        Four sequential sweeps over the stages per IPM iteration
//                        (BU: update+residual+factorisation, FW affine, BW corrector rhs,
//                        FW final), each software-prefetching the next stage's data.
//   finish        H12    full-step update and status mapping of ocp_nlp_sqp_rti
//
// The same source compiles for the host (tests/hostbuild, test infrastructure) so that the
// algorithm can be checked against the dense oracle without a GPU; the product library only
// contains the HIP build.
#pragma once
#include <math.h>
#include <stddef.h>

#include <type_traits>

#include "../../include/mpc_hpipm_args.h"

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#define MPC_SWEEP __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#define MPC_SWEEP inline
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
    int sqp;              // 1: full SQP (sqp_max_iter > 1) — k_nlp_res linearises and gates the instances
    double nlp_tol;       // exit tolerance on the four NLP residuals (full SQP only)
    // wave model, acceptance tests of the parallel scans (fallback to the sequential recursion):
    double scan_tol_P;    // relative residual of the Riccati equation, stage by stage
    double scan_tol_lin;  // acceptance threshold of a scan direction on the stationarity rows (consts.hpp)
    double scan_dyn_scale; // scan_tol_lin / 1e-8: the dynamics rows are held against 1e-8
    // uniform values the wave kernels would otherwise compute with vector instructions and keep in
    // vector registers for the whole solve (there is no scalar FP64 unit): wg + reg_prim,
    // we + reg_prim, 1 / (number of one-sided bounds)
    double wgr[5], wer[3], nc_inv;
};

// Workspace layout.  One contiguous block per wavefront (64 instances):
//     block[stage k = 0..N][row r = 0..MPC_RPS-1][lane 0..63]      (doubles)
// so a wave's access to (k, r) is one 512-byte line group at byte offset ((k*RPS + r) << 9)
// from the wave's block, plus lane*8.  On the GPU the block is addressed through a buffer
// descriptor: descriptor and row offset live in SGPRs (scalar unit does the row stepping),
// the only per-lane address register is the constant lane*8.  All data one stage visit
// touches sits in one 40-KB span.
enum : int {
    R_X = 0,      // 3   capsule: state trajectory x_k
    R_U = 3,      // 2   capsule: controls u_k
    R_PI = 5,     // 3   capsule: multipliers of the dynamics
    R_YREF = 8,   // 5   input: y_ref of stage k          (stage N: y_ref_e in rows 0..2)
    R_AB = 13,    // 8   a, b, B00, B10, B20, B01, B11, B21 (stage N: x0bar in rows 0..2)
    R_C = 21,     // 3   affine term of the dynamics in absolute variables
    R_Z = 24,     // 5   zeta = z + dz  [v, delta, x, y, psi]
    R_PIQ = 29,   // 3   QP multipliers pi
    R_LAM = 32,   // 8   lam: lower (v, delta, x, y), upper (...)
    R_T = 40,     // 8   slacks t, same order
    R_P = 48,     // 6   value function P_{k+1} (sym: 00,10,11,20,21,22)
    R_KFF = 54,   // 2   feed-forward of the current right-hand side
    R_PV = 56,    // 3   p_{k+1}                              (stage N: u0 out in rows 0..1)
    R_RG = 59,    // 5   res_g
    R_RB = 64,    // 3   res_b
    R_DZA = 67,   // 4   affine step of the bounded variables
    R_DZ = 71,    // 5   final step
    R_DPI = 76,   // 3   final step of pi
    MPC_RPS = 80, // rows per stage
    R_YREFE = R_YREF, R_X0 = R_AB, R_U0 = R_PV  // stage-N aliases (R_U0 must stay outside the carry ranges)
};

struct Mem {
    char* slab;          // lane model only: n_waves blocks of wave_bytes each (null until the first lane-model solve)
    size_t wave_bytes;   // (N+1) * MPC_RPS * 512
    int *STATUS, *QPITER, *DIAG;   // [S] each
    int *SQPITER, *ACTIVE;         // [S] each, full SQP only: QPs solved so far, still iterating
    double* NLPRES;                // [S][4] last NLP residuals (stat, eq, ineq, comp), full SQP only
    // The capsules (the iterate every <name>_acados_solve starts from), AoS:
    // cap[instance][stage 0..N][MPC_CAP] = x 3, u 2, pi 3 — the lane that owns a stage reads and
    // writes one contiguous 64-B line, fully coalesced across the wave.
    double* cap;
    double* lam;         // [instance][stage 0..N][8] bound multipliers, full-SQP handles only (else null)
    double* u0;          // [S][2] u_0 of the last solve (the previous one where the QP failed)
    // Instances whose scan step was declined (rti_wave2.hpp iterate): REDO[0], REDO[1] = two counters
    // used by alternate solves (a scan launch counts in one and clears the other one for the next
    // solve, so no memset sits between two solves), REDO[MPC_REDO_LIST + n] = the n-th instance.
    // Which counter is in use is itself device state, so that a solve recorded once in a HIP graph
    // replays correctly: REDO[MPC_REDO_PS] is read by scan launches and written by list launches
    // (which flip it), REDO[MPC_REDO_PL] is written by scan launches and read by the list launch
    // behind them.  Each word is only written by the kind of launch that does not read it, and the
    // launches of a solve are ordered on one stream.  A list launch that is skipped (latency path,
    // nothing declined) leaves the side unchanged — its counter is then still zero.
    int* REDO;
    // Interior-point state of declined instances (RTI handles): a scan kernel that declines an instance
    // parks the iterate of the iteration it could not certify — per stage z 5, pi 3, lam 8, t 8, and per
    // instance {iterations done, last step length, diag bits} — in slot n of this pool, n = the
    // instance's position in the list, and marks the list entry (MPC_REDO_SAVED); the list kernel then
    // carries on from that iteration instead of solving the QP again from its cold start.  Entries
    // beyond ipm_slots (or with no pool at all: the no-allocation entry point) are solved from the start.
    double* IPM;         // [ipm_slots][(N + 1) * MPC_IPM_STAGE + MPC_IPM_HEAD]
    int ipm_slots;
};
enum { MPC_IPM_STAGE = 24, MPC_IPM_HEAD = 8, MPC_REDO_SAVED = (int)0x40000000 };
enum { MPC_REDO_LIST = 64, MPC_REDO_PS = 2, MPC_REDO_PL = 3, MPC_REDO_WORK = 4 };   // WORK: the work counter of persistent wavefronts (mpc_batch.hip k_wave2)
enum : int { MPC_CAP = 8, C_X = 0, C_U = 3, C_PI = 5 };

// MPC_DIAG_ITREF / MPC_DIAG_LQ / MPC_DIAG_SCAN_FALLBACK / MPC_DIAG_ITREF_SKIPPED: include/mpc_batch.h

// ------------------------------------------------------------------------------------
// Per-stage algebra shared by the two execution models (one OCP per lane: Lane below; one OCP
// per wavefront: rti_wave.hpp).  No memory access in here.
struct Algebra {
    const Consts& c;
    MPC_HD explicit Algebra(const Consts& c_) : c(c_) {}

    // ---- small value types (kept in registers) ----
    struct AB8 { double a, b, B00, B10, B20, B01, B11, B21; };
    struct Sym3 { double p00, p10, p11, p20, p21, p22; };
    struct Bnd { double lam[8], t[8]; };  // [0..3] lower, [4..7] upper of (v, delta, x, y)

    // ---- H5/H7: ERK4 with forward sensitivities, structure-exploiting but with the generic
    //      integrator's operation order (entries that are structurally 0/1 are exact there too)
    MPC_HD void linearize_math(double x0, double x1, double psi, double v, double dl, AB8& ab, double cv[3],
                               double xn[3]) const {
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
        ab.a = a; ab.b = b; ab.B00 = B00; ab.B10 = B10; ab.B20 = B20; ab.B01 = B01; ab.B11 = B11; ab.B21 = B21;
        // affine term of the absolute formulation: zeta_x+ = A zeta_x + B zeta_u + cvec
        cv[0] = xn0 - (x0 + a * psi) - (B00 * v + B01 * dl);
        cv[1] = xn1 - (x1 + b * psi) - (B10 * v + B11 * dl);
        cv[2] = xn2 - psi - (B20 * v + B21 * dl);
        xn[0] = xn0; xn[1] = xn1; xn[2] = xn2;
    }

    // Stage indices are the same in every lane, but after control-flow restructuring the
    // compiler cannot always prove it and would wrap every buffer access in a waterfall loop;
    // reading the value from the first active lane states the uniformity explicitly.
    MPC_HD static int uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_readfirstlane(v);
#else
        return v;
#endif
    }

    // reciprocal: v_rcp_f64 + two Newton steps (what the compiler's own f64 division expands
    // to, minus the scaling/fix-up for subnormal and huge quotients that cannot occur here)
    MPC_HD static double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_amdgcn_rcp(x);
        r = fma(fma(-x, r, 1.0), r, r);
        r = fma(fma(-x, r, 1.0), r, r);
        return r;
#else
        return 1.0 / x;
#endif
    }

    // ---- per-constraint algebra (COMPUTE_GAMMA_GAMMA_QP / COMPUTE_LAM_T_QP) ----
    struct Con {  // one two-sided bound on one variable
        double lam_l, lam_u, t_l, t_u;
        double rd_l, rd_u;  // res_d
        double rm_l, rm_u;  // res_m (possibly corrected)
        double ti_l, ti_u;
    };
    MPC_HD void con_make(int j, double lam_l, double lam_u, double t_l, double t_u, double zj, Con& q) const {
        q.lam_l = lam_l; q.lam_u = lam_u; q.t_l = t_l; q.t_u = t_u;
        q.rd_l = c.lb[j] + t_l - zj;
        q.rd_u = t_u + zj - c.ub[j];
        q.rm_l = lam_l * t_l;
        q.rm_u = lam_u * t_u;
        q.ti_l = rcp(t_l);
        q.ti_u = rcp(t_u);
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

    struct Norms { double g, b, d, m, musum; };
    MPC_HD static double amax(double cur, double v) { return fmax(cur, fabs(v)); }

    // x = -(L D L^T)^-1 r for the 2x2 block
    MPC_HD static void solve2(double id0, double l10, double id1, double r0, double r1, double& x0, double& x1) {
        const double y1 = r1 - l10 * r0;
        x1 = -(y1 * id1);
        x0 = -(r0 * id0) - l10 * x1;
    }

    // ---- Riccati gain of stage k from the value function P of stage k+1:
    //      Huu = R + Gamma_u + B^T P B = L D L^T,  K = -Huu^-1 B^T P A.
    //      Recomputed in every sweep instead of being stored (15 doubles per stage saved;
    //      the arithmetic is identical each time, so every sweep sees the same bits).
    struct Gain {
        double id0, l10, id1;
        double M00, M01, M02, M10, M11, M12;  // M = B^T P
        double X02, X12;                      // third column of Hux = M A
        double K00, K01, K02, K10, K11, K12;
    };
    MPC_HD void gain(const Sym3& P, const AB8& ab, double Gam0, double Gam1, bool needK, Gain& g) const {
        g.M00 = ab.B00 * P.p00 + ab.B10 * P.p10 + ab.B20 * P.p20;
        g.M01 = ab.B00 * P.p10 + ab.B10 * P.p11 + ab.B20 * P.p21;
        g.M02 = ab.B00 * P.p20 + ab.B10 * P.p21 + ab.B20 * P.p22;
        g.M10 = ab.B01 * P.p00 + ab.B11 * P.p10 + ab.B21 * P.p20;
        g.M11 = ab.B01 * P.p10 + ab.B11 * P.p11 + ab.B21 * P.p21;
        g.M12 = ab.B01 * P.p20 + ab.B11 * P.p21 + ab.B21 * P.p22;
        const double Huu00 = c.wgr[0] + Gam0 + (g.M00 * ab.B00 + g.M01 * ab.B10 + g.M02 * ab.B20);
        const double Huu10 = g.M10 * ab.B00 + g.M11 * ab.B10 + g.M12 * ab.B20;
        const double Huu11 = c.wgr[1] + Gam1 + (g.M10 * ab.B01 + g.M11 * ab.B11 + g.M12 * ab.B21);
        g.id0 = rcp(Huu00);
        g.l10 = Huu10 * g.id0;
        g.id1 = rcp(Huu11 - g.l10 * Huu10);
        if (needK) {
            g.X02 = g.M00 * ab.a + g.M01 * ab.b + g.M02;
            g.X12 = g.M10 * ab.a + g.M11 * ab.b + g.M12;
            solve2(g.id0, g.l10, g.id1, g.M00, g.M10, g.K00, g.K10);
            solve2(g.id0, g.l10, g.id1, g.M01, g.M11, g.K01, g.K11);
            solve2(g.id0, g.l10, g.id1, g.X02, g.X12, g.K02, g.K12);
        }
    }

};

// ------------------------------------------------------------------------------------
struct Lane : Algebra {
    const Mem& m;
    size_t wbase;   // first instance of this wavefront (wave-uniform)
    unsigned lane;  // threadIdx.x
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff;
    __device__ __forceinline__ Lane(const Consts& c_, const Mem& m_, size_t wbase_, unsigned lane_)
        : Algebra(c_), m(m_), wbase(wbase_), lane(lane_) {
        char* blk = m.slab + (wbase >> 6) * m.wave_bytes;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)m.wave_bytes, 0x00020000);
        voff = lane_ * 8u;
    }
    __device__ __forceinline__ double ld(int field, int k, int comp) const {
        // rows are split 8 + 3 bits: the low part folds into the instruction's 12-bit immediate
        // offset, so one scalar offset serves 8 rows
        const int row = field + comp;
        const unsigned so = (unsigned)(k * MPC_RPS + (row & ~7)) << 9;
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + ((unsigned)(row & 7) << 9), so, 0));
    }
    __device__ __forceinline__ void st(int field, int k, int comp, double v) const {
        const int row = field + comp;
        const unsigned so = (unsigned)(k * MPC_RPS + (row & ~7)) << 9;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, voff + ((unsigned)(row & 7) << 9), so, 0);
    }
#else
    double* blk;
    MPC_HD Lane(const Consts& c_, const Mem& m_, size_t wbase_, unsigned lane_)
        : Algebra(c_), m(m_), wbase(wbase_), lane(lane_) {
        blk = reinterpret_cast<double*>(m.slab + (wbase >> 6) * m.wave_bytes) + lane_;
    }
    MPC_HD double ld(int field, int k, int comp) const { return blk[(size_t)(k * MPC_RPS + field + comp) << 6]; }
    MPC_HD void st(int field, int k, int comp, double v) const { blk[(size_t)(k * MPC_RPS + field + comp) << 6] = v; }
#endif

    // ---- H5/H7 on the workspace rows of stage k
    MPC_HD void linearize_stage(int k, double* xn_out = nullptr) const {
        AB8 ab;
        double cv[3], xn[3];
        linearize_math(ld(R_X, k, 0), ld(R_X, k, 1), ld(R_X, k, 2), ld(R_U, k, 0), ld(R_U, k, 1), ab, cv, xn);
        st(R_AB, k, 0, ab.a);   st(R_AB, k, 1, ab.b);
        st(R_AB, k, 2, ab.B00); st(R_AB, k, 3, ab.B10); st(R_AB, k, 4, ab.B20);
        st(R_AB, k, 5, ab.B01); st(R_AB, k, 6, ab.B11); st(R_AB, k, 7, ab.B21);
        st(R_C, k, 0, cv[0]); st(R_C, k, 1, cv[1]); st(R_C, k, 2, cv[2]);
        if (xn_out) { xn_out[0] = xn[0]; xn_out[1] = xn[1]; xn_out[2] = xn[2]; }
    }

    // ---- full SQP only (BASELINE config 5; acados ocp_nlp_sqp.c / ocp_nlp_res_compute restated,
    //      no counterpart in the reference's SQP_RTI solvers): linearise every stage at the
    //      iterate and evaluate the NLP residuals with the multipliers of the last QP:
    //      out = { ||cost_grad + dyn_adj + ineq_adj||_inf (stage-0 state rows excluded),
    //              ||phi(x_k,u_k) - x_{k+1}||_inf, max(0, lb - z, z - ub, |x_0 - x0bar|),
    //              max |lam_j * violation_j| }.
    MPC_HD void nlp_res_and_linearize(double out[4]) const {
        const int N = uniform(c.N);
        double r_stat = 0.0, r_eq = 0.0, r_ineq = 0.0, r_comp = 0.0;
        for (int i = 0; i < 3; ++i) r_ineq = amax(r_ineq, ld(R_X0, N, i) - ld(R_X, 0, i));
        double pim0 = 0.0, pim1 = 0.0, pim2 = 0.0;
        for (int k = 0; k <= N; ++k) {
            double z[5], g[5];
            z[0] = (k < N) ? ld(R_U, k, 0) : 0.0;
            z[1] = (k < N) ? ld(R_U, k, 1) : 0.0;
            z[2] = ld(R_X, k, 0); z[3] = ld(R_X, k, 1); z[4] = ld(R_X, k, 2);
            if (k < N) {
#pragma unroll
                for (int i = 0; i < 5; ++i) g[i] = c.wg[i] * (z[i] - ld(R_YREF, k, i < 2 ? 3 + i : i - 2));
            } else {
                g[0] = g[1] = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) g[2 + i] = c.we[i] * (z[2 + i] - ld(R_YREFE, N, i));
            }
            if (k > 0) { g[2] -= pim0; g[3] -= pim1; g[4] -= pim2; }
            if (k < N) {
                const int nb = (k == 0) ? 2 : 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nb) {
                        const double dl = c.lb[j] - z[j], du = z[j] - c.ub[j];
                        const double ll = ld(R_LAM, k, j), lu = ld(R_LAM, k, 4 + j);
                        g[j] += lu - ll;
                        r_ineq = fmax(r_ineq, dl); r_ineq = fmax(r_ineq, du);
                        r_comp = amax(r_comp, ll * dl); r_comp = amax(r_comp, lu * du);
                    }
                }
                double xn[3];
                linearize_stage(k, xn);
                const AB8 ab = ld_ab(k);
#pragma unroll
                for (int j = 0; j < 3; ++j) r_eq = amax(r_eq, xn[j] - ld(R_X, k + 1, j));
                const double p0 = ld(R_PI, k, 0), p1 = ld(R_PI, k, 1), p2 = ld(R_PI, k, 2);
                g[0] += ab.B00 * p0 + ab.B10 * p1 + ab.B20 * p2;
                g[1] += ab.B01 * p0 + ab.B11 * p1 + ab.B21 * p2;
                g[2] += p0; g[3] += p1; g[4] += ab.a * p0 + ab.b * p1 + p2;
                pim0 = p0; pim1 = p1; pim2 = p2;
            }
            if (k < N) { r_stat = amax(r_stat, g[0]); r_stat = amax(r_stat, g[1]); }
            if (k > 0) { r_stat = amax(r_stat, g[2]); r_stat = amax(r_stat, g[3]); r_stat = amax(r_stat, g[4]); }
        }
        out[0] = r_stat; out[1] = r_eq; out[2] = r_ineq; out[3] = r_comp;
    }

    MPC_HD AB8 ld_ab(int k) const {
        AB8 r;
        r.a = ld(R_AB, k, 0); r.b = ld(R_AB, k, 1);
        r.B00 = ld(R_AB, k, 2); r.B10 = ld(R_AB, k, 3); r.B20 = ld(R_AB, k, 4);
        r.B01 = ld(R_AB, k, 5); r.B11 = ld(R_AB, k, 6); r.B21 = ld(R_AB, k, 7);
        return r;
    }
    MPC_HD Sym3 ld_P(int k) const {
        Sym3 r;
        r.p00 = ld(R_P, k, 0); r.p10 = ld(R_P, k, 1); r.p11 = ld(R_P, k, 2);
        r.p20 = ld(R_P, k, 3); r.p21 = ld(R_P, k, 4); r.p22 = ld(R_P, k, 5);
        return r;
    }
    MPC_HD void st_P(int k, const Sym3& P) const {
        st(R_P, k, 0, P.p00); st(R_P, k, 1, P.p10); st(R_P, k, 2, P.p11);
        st(R_P, k, 3, P.p20); st(R_P, k, 4, P.p21); st(R_P, k, 5, P.p22);
    }
    MPC_HD Bnd ld_bnd(int k) const {
        Bnd r;
#pragma unroll
        for (int j = 0; j < 8; ++j) { r.lam[j] = ld(R_LAM, k, j); r.t[j] = ld(R_T, k, j); }
        return r;
    }

    // ---- preparation sweep (forward): ERK4 linearisation + OCP_QP_INIT_VAR cold start ----
    MPC_SWEEP void prepare() const {
        const int N = uniform(c.N);
        for (int k = 0; k <= N; ++k) {
            if (k < N && !c.sqp) linearize_stage(k);   // full SQP: k_nlp_res has done it
            double z[5];
            z[0] = (k < N) ? ld(R_U, k, 0) : 0.0;
            z[1] = (k < N) ? ld(R_U, k, 1) : 0.0;
            if (k == 0) { z[2] = ld(R_X0, c.N, 0); z[3] = ld(R_X0, c.N, 1); z[4] = ld(R_X0, c.N, 2); }
            else { z[2] = ld(R_X, k, 0); z[3] = ld(R_X, k, 1); z[4] = ld(R_X, k, 2); }
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
                        st(R_T, k, j, tl); st(R_T, k, 4 + j, tu);
                        st(R_LAM, k, j, c.mu0 * rcp(tl)); st(R_LAM, k, 4 + j, c.mu0 * rcp(tu));
                    } else {  // stage 0 has no state bounds: benign values keep the unused lanes finite
                        st(R_T, k, j, 1.0); st(R_T, k, 4 + j, 1.0);
                        st(R_LAM, k, j, 0.0); st(R_LAM, k, 4 + j, 0.0);
                    }
                }
                st(R_PIQ, k, 0, 0.0); st(R_PIQ, k, 1, 0.0); st(R_PIQ, k, 2, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) st(R_Z, k, q, z[q]);
        }
    }

    // ---- sweep BU (backward): UPDATE_VAR_QP of the previous step, OCP_QP_RES_COMPUTE and the
    //      factorisation + affine right-hand side of OCP_QP_FACT_SOLVE_KKT_STEP, fused so that
    //      every stage's state is read and written once.  FIRST: no step to apply yet.
    struct BuIn {
        double z[5], dz[5], dza[4];
        Bnd bn;
        double pim[3], dpim[3];
        double yref[5], cv[3];
        AB8 ab;
    };
    template <bool FIRST>
    MPC_HD void ld_bu(int k, BuIn& r) const {
#pragma unroll
        for (int q = 0; q < 5; ++q) r.z[q] = ld(R_Z, k, q);
        r.bn = ld_bnd(k);
        if (!FIRST) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r.dz[q] = ld(R_DZ, k, q);
#pragma unroll
            for (int q = 0; q < 4; ++q) r.dza[q] = ld(R_DZA, k, q);
        }
        if (k > 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                r.pim[q] = ld(R_PIQ, k - 1, q);
                if (!FIRST) r.dpim[q] = ld(R_DPI, k - 1, q);
            }
        } else {
            r.pim[0] = r.pim[1] = r.pim[2] = 0.0;
            r.dpim[0] = r.dpim[1] = r.dpim[2] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) r.yref[q] = ld(R_YREF, k, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) r.cv[q] = ld(R_C, k, q);
        r.ab = ld_ab(k);
    }

    template <bool FIRST>
    MPC_SWEEP void sweep_bu(int mode, double sigma_mu, double alpha, Norms& n) const {
        const int N = uniform(c.N);
        n.g = n.b = n.d = n.m = n.musum = 0.0;
        // ---- terminal stage ----
        double zxn[3], pik[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double zx = ld(R_Z, N, 2 + q);
            double pk = ld(R_PIQ, N - 1, q);
            if (!FIRST) {
                zx += alpha * ld(R_DZ, N, 2 + q);
                pk += alpha * ld(R_DPI, N - 1, q);
                st(R_Z, N, 2 + q, zx);
                st(R_PIQ, N - 1, q, pk);
            }
            zxn[q] = zx; pik[q] = pk;
        }
        double p0 = c.we[0] * (zxn[0] - ld(R_YREFE, c.N, 0)) - pik[0];
        double p1 = c.we[1] * (zxn[1] - ld(R_YREFE, c.N, 1)) - pik[1];
        double p2 = c.we[2] * (zxn[2] - ld(R_YREFE, c.N, 2)) - pik[2];
        st(R_RG, N, 2, p0); st(R_RG, N, 3, p1); st(R_RG, N, 4, p2);
        n.g = amax(n.g, p0); n.g = amax(n.g, p1); n.g = amax(n.g, p2);
        Sym3 P;
        P.p00 = c.we[0] + c.reg_prim; P.p10 = 0.0; P.p11 = c.we[1] + c.reg_prim;
        P.p20 = 0.0; P.p21 = 0.0; P.p22 = c.we[2] + c.reg_prim;

        BuIn cur, nxt;
        ld_bu<FIRST>(N - 1, cur);
        // stage 0 (no state variables, 2 bounds) is peeled off at compile time
        auto body = [&](auto s0tag, const int k_) {
            const int k = uniform(k_);
            constexpr bool S0 = decltype(s0tag)::value;
            constexpr int nb = S0 ? 2 : 4;
            if (!S0) ld_bu<FIRST>(k - 1, nxt);  // software prefetch of the next stage
            const AB8& ab = cur.ab;
            // --- apply the previous step to this stage ---
            double z[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) z[q] = FIRST ? cur.z[q] : cur.z[q] + alpha * cur.dz[q];
            Bnd bn = cur.bn;
            if (!FIRST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nb) {
                        Con q;
                        con_make(j, cur.bn.lam[j], cur.bn.lam[4 + j], cur.bn.t[j], cur.bn.t[4 + j], cur.z[j], q);
                        con_correct(q, mode, cur.dza[j], sigma_mu);
                        Step s;
                        con_step(q, cur.dz[j], s);
                        double ll = q.lam_l + alpha * s.dl_l, lu = q.lam_u + alpha * s.dl_u;
                        double tl = q.t_l + alpha * s.dt_l, tu = q.t_u + alpha * s.dt_u;
                        if (ll < c.lam_min) ll = c.lam_min;
                        if (lu < c.lam_min) lu = c.lam_min;
                        if (tl < c.t_min) tl = c.t_min;
                        if (tu < c.t_min) tu = c.t_min;
                        bn.lam[j] = ll; bn.lam[4 + j] = lu; bn.t[j] = tl; bn.t[4 + j] = tu;
                        st(R_LAM, k, j, ll); st(R_LAM, k, 4 + j, lu);
                        st(R_T, k, j, tl);   st(R_T, k, 4 + j, tu);
                    }
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) st(R_Z, k, q, z[q]);
            }
            double pim[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                pim[q] = FIRST ? cur.pim[q] : cur.pim[q] + alpha * cur.dpim[q];
                if (!FIRST && !S0) st(R_PIQ, k - 1, q, pim[q]);
            }
            // --- residuals of this stage at the new point ---
            double rg[5];
            rg[0] = c.wg[0] * (z[0] - cur.yref[3]);
            rg[1] = c.wg[1] * (z[1] - cur.yref[4]);
            rg[2] = c.wg[2] * (z[2] - cur.yref[0]) - pim[0];
            rg[3] = c.wg[3] * (z[3] - cur.yref[1]) - pim[1];
            rg[4] = c.wg[4] * (z[4] - cur.yref[2]) - pim[2];
            double Gam[4] = {0, 0, 0, 0}, gam[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Con q;
                    con_make(j, bn.lam[j], bn.lam[4 + j], bn.t[j], bn.t[4 + j], z[j], q);
                    rg[j] += q.lam_u - q.lam_l;
                    n.d = amax(n.d, q.rd_l); n.d = amax(n.d, q.rd_u);
                    n.m = amax(n.m, q.rm_l); n.m = amax(n.m, q.rm_u);
                    n.musum += q.rm_l;
                    n.musum += q.rm_u;
                    Gam[j] = con_Gamma(q);
                    gam[j] = con_gamma(q);
                }
            }
            const double rb0 = cur.cv[0] - zxn[0] + (ab.B00 * z[0] + ab.B01 * z[1] + (z[2] + ab.a * z[4]));
            const double rb1 = cur.cv[1] - zxn[1] + (ab.B10 * z[0] + ab.B11 * z[1] + (z[3] + ab.b * z[4]));
            const double rb2 = cur.cv[2] - zxn[2] + (ab.B20 * z[0] + ab.B21 * z[1] + z[4]);
            st(R_RB, k, 0, rb0); st(R_RB, k, 1, rb1); st(R_RB, k, 2, rb2);
            n.b = amax(n.b, rb0); n.b = amax(n.b, rb1); n.b = amax(n.b, rb2);
            rg[0] += ab.B00 * pik[0] + ab.B10 * pik[1] + ab.B20 * pik[2];
            rg[1] += ab.B01 * pik[0] + ab.B11 * pik[1] + ab.B21 * pik[2];
            rg[2] += pik[0];
            rg[3] += pik[1];
            rg[4] += ab.a * pik[0] + ab.b * pik[1] + pik[2];
            n.g = amax(n.g, rg[0]); n.g = amax(n.g, rg[1]);
            if (!S0) { n.g = amax(n.g, rg[2]); n.g = amax(n.g, rg[3]); n.g = amax(n.g, rg[4]); }
#pragma unroll
            for (int q = 0; q < 5; ++q) st(R_RG, k, q, rg[q]);
            // --- Riccati step (factorisation + affine right-hand side) ---
            st_P(k, P);
            const double w0 = P.p00 * rb0 + P.p10 * rb1 + P.p20 * rb2 + p0;
            const double w1 = P.p10 * rb0 + P.p11 * rb1 + P.p21 * rb2 + p1;
            const double w2 = P.p20 * rb0 + P.p21 * rb1 + P.p22 * rb2 + p2;
            const double hu0 = rg[0] + gam[0] + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
            const double hu1 = rg[1] + gam[1] + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
            Gain g;
            gain(P, ab, Gam[0], Gam[1], !S0, g);
            double f0, f1;
            solve2(g.id0, g.l10, g.id1, hu0, hu1, f0, f1);
            st(R_KFF, k, 0, f0); st(R_KFF, k, 1, f1);
            if (!S0) {
                const double a = ab.a, b = ab.b;
                const double S02 = P.p00 * a + P.p10 * b + P.p20;
                const double S12 = P.p10 * a + P.p11 * b + P.p21;
                const double S22 = P.p20 * a + P.p21 * b + P.p22;
                const double Hxx00 = (c.wg[2] + c.reg_prim) + Gam[2] + P.p00;
                const double Hxx10 = P.p10;
                const double Hxx11 = (c.wg[3] + c.reg_prim) + Gam[3] + P.p11;
                const double Hxx20 = a * P.p00 + b * P.p10 + P.p20;
                const double Hxx21 = a * P.p10 + b * P.p11 + P.p21;
                const double Hxx22 = (c.wg[4] + c.reg_prim) + (a * S02 + b * S12 + S22);
                Sym3 Q;
                Q.p00 = Hxx00 + (g.M00 * g.K00 + g.M10 * g.K10);
                Q.p10 = Hxx10 + (g.M01 * g.K00 + g.M11 * g.K10);
                Q.p11 = Hxx11 + (g.M01 * g.K01 + g.M11 * g.K11);
                Q.p20 = Hxx20 + (g.X02 * g.K00 + g.X12 * g.K10);
                Q.p21 = Hxx21 + (g.X02 * g.K01 + g.X12 * g.K11);
                Q.p22 = Hxx22 + (g.X02 * g.K02 + g.X12 * g.K12);
                const double hx0 = rg[2] + gam[2] + w0;
                const double hx1 = rg[3] + gam[3] + w1;
                const double hx2 = rg[4] + (a * w0 + b * w1 + w2);
                p0 = hx0 + (g.K00 * hu0 + g.K10 * hu1);
                p1 = hx1 + (g.K01 * hu0 + g.K11 * hu1);
                p2 = hx2 + (g.K02 * hu0 + g.K12 * hu1);
                P = Q;
            }
            zxn[0] = z[2]; zxn[1] = z[3]; zxn[2] = z[4];
            pik[0] = pim[0]; pik[1] = pim[1]; pik[2] = pim[2];
            if (!S0) cur = nxt;
        };
        for (int k = N - 1; k >= 1; --k) body(std::false_type{}, k);
        body(std::true_type{}, 0);
    }

    // ---- backward sweep with a new right-hand side (OCP_QP_SOLVE_KKT_STEP, backward half):
    //      corrected complementarity -> gamma -> kff, p.  mode 1 / 2 as in con_correct.
    struct BwIn {
        Sym3 P;
        AB8 ab;
        double rg[5], rb[3], z[4], dza[4];
        Bnd bn;
    };
    MPC_HD void ld_bw(int k, BwIn& r) const {
        r.P = ld_P(k);
        r.ab = ld_ab(k);
#pragma unroll
        for (int q = 0; q < 5; ++q) r.rg[q] = ld(R_RG, k, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) r.rb[q] = ld(R_RB, k, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) { r.z[q] = ld(R_Z, k, q); r.dza[q] = ld(R_DZA, k, q); }
        r.bn = ld_bnd(k);
    }
    MPC_SWEEP void sweep_bw(int mode, double sigma_mu) const {
        const int N = uniform(c.N);
        double p0 = ld(R_RG, N, 2), p1 = ld(R_RG, N, 3), p2 = ld(R_RG, N, 4);
        BwIn cur, nxt;
        ld_bw(N - 1, cur);
        auto body = [&](auto s0tag, const int k_) {
            const int k = uniform(k_);
            constexpr bool S0 = decltype(s0tag)::value;
            constexpr int nb = S0 ? 2 : 4;
            if (!S0) ld_bw(k - 1, nxt);
            const AB8& ab = cur.ab;
            const Sym3& P = cur.P;
            st(R_PV, k, 0, p0); st(R_PV, k, 1, p1); st(R_PV, k, 2, p2);
            const double w0 = P.p00 * cur.rb[0] + P.p10 * cur.rb[1] + P.p20 * cur.rb[2] + p0;
            const double w1 = P.p10 * cur.rb[0] + P.p11 * cur.rb[1] + P.p21 * cur.rb[2] + p1;
            const double w2 = P.p20 * cur.rb[0] + P.p21 * cur.rb[1] + P.p22 * cur.rb[2] + p2;
            double Gam[4] = {0, 0, 0, 0}, gam[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Con q;
                    con_make(j, cur.bn.lam[j], cur.bn.lam[4 + j], cur.bn.t[j], cur.bn.t[4 + j], cur.z[j], q);
                    Gam[j] = con_Gamma(q);
                    con_correct(q, mode, cur.dza[j], sigma_mu);
                    gam[j] = con_gamma(q);
                }
            }
            const double hu0 = cur.rg[0] + gam[0] + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
            const double hu1 = cur.rg[1] + gam[1] + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
            Gain g;
            gain(P, ab, Gam[0], Gam[1], !S0, g);
            double f0, f1;
            solve2(g.id0, g.l10, g.id1, hu0, hu1, f0, f1);
            st(R_KFF, k, 0, f0); st(R_KFF, k, 1, f1);
            if (!S0) {
                const double hx0 = cur.rg[2] + gam[2] + w0;
                const double hx1 = cur.rg[3] + gam[3] + w1;
                const double hx2 = cur.rg[4] + (ab.a * w0 + ab.b * w1 + w2);
                p0 = hx0 + (g.K00 * hu0 + g.K10 * hu1);
                p1 = hx1 + (g.K01 * hu0 + g.K11 * hu1);
                p2 = hx2 + (g.K02 * hu0 + g.K12 * hu1);
            }
            if (!S0) cur = nxt;
        };
        for (int k = N - 1; k >= 1; --k) body(std::false_type{}, k);
        body(std::true_type{}, 0);
    }

    // ---- forward sweep: primal step of all stages, step lengths (COMPUTE_ALPHA_QP) and the
    //      three sums that give mu_aff for any alpha.  FINAL = false: affine step -> DZA;
    //      FINAL = true: final step -> DZ, DPI and the stationarity residual of the linear
    //      system (what HPIPM's refinement loop and lq_fact = 1 switch test).
    struct FwIn {
        Sym3 P;
        AB8 ab;
        double kff[2], rb[3], z[4], dza[4], pv[3], rg[5];
        Bnd bn;
    };
    template <bool FINAL>
    MPC_HD void ld_fw(int k, FwIn& r) const {
        r.P = ld_P(k);
        r.ab = ld_ab(k);
        r.kff[0] = ld(R_KFF, k, 0); r.kff[1] = ld(R_KFF, k, 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) r.rb[q] = ld(R_RB, k, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) r.z[q] = ld(R_Z, k, q);
        r.bn = ld_bnd(k);
        if (FINAL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r.dza[q] = ld(R_DZA, k, q);
#pragma unroll
            for (int q = 0; q < 3; ++q) r.pv[q] = ld(R_PV, k, q);
#pragma unroll
            for (int q = 0; q < 5; ++q) r.rg[q] = ld(R_RG, k, q);
        }
    }
    struct FwdOut { double a_prim, a_dual, s1, s2, lin_g; };
    template <bool FINAL>
    MPC_SWEEP void sweep_fw(int mode, double sigma_mu, FwdOut& o) const {
        const int N = uniform(c.N);
        double a_prim = -1.0, a_dual = -1.0, s1 = 0.0, s2 = 0.0, lin_g = 0.0;
        double dx0 = 0.0, dx1 = 0.0, dx2 = 0.0;     // stage-0 state is fixed
        double dpm0 = 0.0, dpm1 = 0.0, dpm2 = 0.0;  // dpi_{k-1}
        const int DZo = FINAL ? R_DZ : R_DZA;
        FwIn cur, nxt;
        ld_fw<FINAL>(0, cur);
        auto body = [&](auto s0tag, const int k_) {
            const int k = uniform(k_);
            constexpr bool S0 = decltype(s0tag)::value;
            constexpr int nb = S0 ? 2 : 4;
            if (k + 1 < N) ld_fw<FINAL>(k + 1, nxt);
            const AB8& ab = cur.ab;
            Con q[4];
            double Gam0 = 0.0, Gam1 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    con_make(j, cur.bn.lam[j], cur.bn.lam[4 + j], cur.bn.t[j], cur.bn.t[4 + j], cur.z[j], q[j]);
                    if (j == 0) Gam0 = con_Gamma(q[j]);
                    if (j == 1) Gam1 = con_Gamma(q[j]);
                    if (FINAL) con_correct(q[j], mode, cur.dza[j], sigma_mu);
                }
            }
            double du0 = cur.kff[0], du1 = cur.kff[1];
            if (!S0) {
                Gain g;
                gain(cur.P, ab, Gam0, Gam1, true, g);
                du0 += g.K00 * dx0 + g.K01 * dx1 + g.K02 * dx2;
                du1 += g.K10 * dx0 + g.K11 * dx1 + g.K12 * dx2;
            }
            const double dz[5] = {du0, du1, dx0, dx1, dx2};
            if (FINAL) {
#pragma unroll
                for (int qq = 0; qq < 5; ++qq) st(DZo, k, qq, dz[qq]);
            } else {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) st(DZo, k, qq, dz[qq]);
            }
            double dlam_net[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nb) {
                    Step s;
                    con_step(q[j], dz[j], s);
                    if (a_dual * s.dl_l > q[j].lam_l) a_dual = q[j].lam_l * rcp(s.dl_l);
                    if (a_prim * s.dt_l > q[j].t_l) a_prim = q[j].t_l * rcp(s.dt_l);
                    if (a_dual * s.dl_u > q[j].lam_u) a_dual = q[j].lam_u * rcp(s.dl_u);
                    if (a_prim * s.dt_u > q[j].t_u) a_prim = q[j].t_u * rcp(s.dt_u);
                    s1 += q[j].lam_l * s.dt_l + q[j].t_l * s.dl_l;
                    s1 += q[j].lam_u * s.dt_u + q[j].t_u * s.dl_u;
                    s2 += s.dl_l * s.dt_l;
                    s2 += s.dl_u * s.dt_u;
                    dlam_net[j] = s.dl_u - s.dl_l;
                }
            }
            const double nx0 = cur.rb[0] + (ab.B00 * du0 + ab.B01 * du1 + (dx0 + ab.a * dx2));
            const double nx1 = cur.rb[1] + (ab.B10 * du0 + ab.B11 * du1 + (dx1 + ab.b * dx2));
            const double nx2 = cur.rb[2] + (ab.B20 * du0 + ab.B21 * du1 + dx2);
            if (FINAL) {
                const Sym3& P = cur.P;
                const double dp0 = P.p00 * nx0 + P.p10 * nx1 + P.p20 * nx2 + cur.pv[0];
                const double dp1 = P.p10 * nx0 + P.p11 * nx1 + P.p21 * nx2 + cur.pv[1];
                const double dp2 = P.p20 * nx0 + P.p21 * nx1 + P.p22 * nx2 + cur.pv[2];
                st(R_DPI, k, 0, dp0); st(R_DPI, k, 1, dp1); st(R_DPI, k, 2, dp2);
                const double lg0 = c.wg[0] * du0 + cur.rg[0] + dlam_net[0] + (ab.B00 * dp0 + ab.B10 * dp1 + ab.B20 * dp2);
                const double lg1 = c.wg[1] * du1 + cur.rg[1] + dlam_net[1] + (ab.B01 * dp0 + ab.B11 * dp1 + ab.B21 * dp2);
                lin_g = amax(lin_g, lg0); lin_g = amax(lin_g, lg1);
                if (!S0) {
                    const double lg2 = c.wg[2] * dx0 + cur.rg[2] + dlam_net[2] - dpm0 + dp0;
                    const double lg3 = c.wg[3] * dx1 + cur.rg[3] + dlam_net[3] - dpm1 + dp1;
                    const double lg4 = c.wg[4] * dx2 + cur.rg[4] - dpm2 + (ab.a * dp0 + ab.b * dp1 + dp2);
                    lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
                }
                dpm0 = dp0; dpm1 = dp1; dpm2 = dp2;
            }
            dx0 = nx0; dx1 = nx1; dx2 = nx2;
            if (k + 1 < N) cur = nxt;
        };
        body(std::true_type{}, 0);
        for (int k = 1; k < N; ++k) body(std::false_type{}, k);
        if (FINAL) {
            st(DZo, N, 2, dx0); st(DZo, N, 3, dx1); st(DZo, N, 4, dx2);
            const double lg2 = c.we[0] * dx0 + ld(R_RG, N, 2) - dpm0;
            const double lg3 = c.we[1] * dx1 + ld(R_RG, N, 3) - dpm1;
            const double lg4 = c.we[2] * dx2 + ld(R_RG, N, 4) - dpm2;
            lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
        }
        o.a_prim = -a_prim; o.a_dual = -a_dual; o.s1 = s1; o.s2 = s2; o.lin_g = lin_g;
    }

    // ---- the RTI step as a resumable state machine -----------------------------------------
    // begin() -> while (cont()) iterate() -> finish().  Between iterations the per-lane scalars
    // can be parked in the workspace (save_state / load_state), which lets the batch driver
    // move a still-running instance to another lane slot (compaction of stragglers).
    struct State {
        int kk, diag;
        double alpha, mu;
        Norms n;
        double home;  // index of the instance this slot is working for
    };
    MPC_HD double nc_inv() const { return 1.0 / (4 + 8 * (c.N - 1)); }

    MPC_HD void begin(State& s, size_t home) const {
        prepare();
        sweep_bu<true>(0, 0.0, 0.0, s.n);
        s.mu = s.n.musum * nc_inv();
        s.alpha = 1.0;
        s.kk = 0;
        s.diag = 0;
        s.home = (double)home;
    }
    // loop condition of d_ocp_qp_ipm_solve
    MPC_HD bool cont(const State& s) const {
        return s.kk < c.iter_max && s.alpha > c.alpha_min &&
               (s.n.g > c.res_g_max || s.n.b > c.res_b_max || s.n.d > c.res_d_max ||
                fabs(s.n.m - c.tau_min) > c.res_m_max);
    }
    // one IPM iteration: OCP_QP_IPM_DELTA_STEP (the factorisation was done by the last BU
    // sweep), UPDATE_VAR_QP, OCP_QP_RES_COMPUTE (+ the next factorisation)
    MPC_HD void iterate(State& s) const {
        const double nci = nc_inv();
        FwdOut fa;
        sweep_fw<false>(0, 0.0, fa);
        const double a_aff = fa.a_prim < fa.a_dual ? fa.a_prim : fa.a_dual;
        const double mu_aff0 = (s.n.musum + a_aff * fa.s1 + (a_aff * a_aff) * fa.s2) * nci;
        const double tmp = mu_aff0 / s.mu;
        const double sigma = tmp * tmp * tmp;
        double sigma_mu = sigma * s.mu;
        if (!(sigma_mu > c.tau_min)) sigma_mu = c.tau_min;
        int mode = 1;
        sweep_bw(1, sigma_mu);
        FwdOut fc;
        sweep_fw<true>(1, sigma_mu, fc);
        double alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
        // conditional predictor-corrector: fall back to pure centering
        const double mu_aff1 = (s.n.musum + alpha * fc.s1 + (alpha * alpha) * fc.s2) * nci;
        if (mu_aff1 > MPC_IPM_COND_PRED_CORR_FAC * mu_aff0) {
            mode = 2;
            sweep_bw(2, sigma_mu);
            sweep_fw<true>(2, sigma_mu, fc);
            alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
        }
        if (!(fc.lin_g < c.itref_tol_g)) s.diag |= MPC_DIAG_ITREF_SKIPPED;
        if (!(fc.lin_g < c.lq_tol)) s.diag |= MPC_DIAG_LQ;
        s.alpha = alpha;
        double a_upd = alpha;
        if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * MPC_IPM_STEP_SCALE_LO + a_upd * MPC_IPM_STEP_SCALE_HI);
        sweep_bu<false>(mode, sigma_mu, a_upd, s.n);
        s.mu = s.n.musum * nci;
        ++s.kk;
    }
    // per-lane scalars parked in unused rows of stage N
    MPC_HD void save_state(const State& s) const {
        const int N = uniform(c.N);
        st(R_LAM, N, 0, (double)s.kk); st(R_LAM, N, 1, (double)s.diag);
        st(R_LAM, N, 2, s.alpha); st(R_LAM, N, 3, s.mu);
        st(R_LAM, N, 4, s.n.g); st(R_LAM, N, 5, s.n.b); st(R_LAM, N, 6, s.n.d); st(R_LAM, N, 7, s.n.m);
        st(R_T, N, 0, s.n.musum); st(R_T, N, 1, s.home);
    }
    MPC_HD void load_state(State& s) const {
        const int N = uniform(c.N);
        s.kk = (int)ld(R_LAM, N, 0); s.diag = (int)ld(R_LAM, N, 1);
        s.alpha = ld(R_LAM, N, 2); s.mu = ld(R_LAM, N, 3);
        s.n.g = ld(R_LAM, N, 4); s.n.b = ld(R_LAM, N, 5); s.n.d = ld(R_LAM, N, 6); s.n.m = ld(R_LAM, N, 7);
        s.n.musum = ld(R_T, N, 0); s.home = ld(R_T, N, 1);
    }
    // status (d_ocp_qp_ipm_solve tail + ocp_qp_hpipm + ocp_nlp_sqp_rti mapping) and the
    // full-step update, written to the instance's HOME capsule (`hm`: the handle's primary
    // workspace), which differs from this lane's slot after a compaction.
    MPC_HD void finish(const State& s, const Mem& hm) const {
        const int N = uniform(c.N);
        int hs;
        if (s.kk == c.iter_max) hs = 1;
        else if (s.alpha <= c.alpha_min) hs = 2;
        else if (s.mu != s.mu) hs = 3;
        else hs = 0;
        // Full SQP in this model keeps its QPs to itself (host-driven rounds, no list kernel to hand an
        // instance to): a QP whose direction HPIPM would have refined was solved WITHOUT the refinement,
        // i.e. off the reference's path.  That is an error, not a result: the instance stops with
        // QP_FAILURE (4) and MPC_DIAG_ITREF_SKIPPED, its iterate stays at the last accepted one.
        const bool ok = (hs == 0 || hs == 1) && !(c.sqp && (s.diag & MPC_DIAG_ITREF_SKIPPED));
        const size_t h = (size_t)s.home;
        hm.QPITER[h] = s.kk;
        hm.DIAG[h] = s.diag;
        hm.STATUS[h] = ok ? (c.sqp ? 2 : 0) : 4;   // full SQP: MAXITER until a residual check passes
        if (c.sqp) {
            hm.SQPITER[h] = hm.SQPITER[h] + 1;
            if (!ok) hm.ACTIVE[h] = 0;
        }
        double* col = reinterpret_cast<double*>(hm.slab + (h >> 6) * hm.wave_bytes) + (h & 63);
        if (ok) {
            for (int k = 0; k <= N; ++k) {
                double* sk = col + ((size_t)(k * MPC_RPS) << 6);
                if (k < N) {
                    sk[(R_U + 0) << 6] = ld(R_Z, k, 0); sk[(R_U + 1) << 6] = ld(R_Z, k, 1);
                    sk[(R_PI + 0) << 6] = ld(R_PIQ, k, 0); sk[(R_PI + 1) << 6] = ld(R_PIQ, k, 1);
                    sk[(R_PI + 2) << 6] = ld(R_PIQ, k, 2);
                }
                sk[(R_X + 0) << 6] = ld(R_Z, k, 2); sk[(R_X + 1) << 6] = ld(R_Z, k, 3);
                sk[(R_X + 2) << 6] = ld(R_Z, k, 4);
                if (c.sqp && k < N) {   // the QP's bound multipliers become the NLP's (next residual check)
#pragma unroll
                    for (int j = 0; j < 8; ++j) sk[(R_LAM + j) << 6] = ld(R_LAM, k, j);
                }
            }
            double* sn = col + ((size_t)(N * MPC_RPS) << 6);
            sn[(R_U0 + 0) << 6] = ld(R_Z, 0, 0);
            sn[(R_U0 + 1) << 6] = ld(R_Z, 0, 1);
        } else {
            double* sn = col + ((size_t)(N * MPC_RPS) << 6);
            sn[(R_U0 + 0) << 6] = col[(R_U + 0) << 6];   // previous u_0 stays in force
            sn[(R_U0 + 1) << 6] = col[(R_U + 1) << 6];
        }
    }

    // whole step on one lane without phases (host build, small batches)
    MPC_HD void rti_step(const Mem& hm) const {
        State s;
        begin(s, wbase + lane);
        while (cont(s)) iterate(s);
        finish(s, hm);
    }
};

// Rows of a stage that carry a running solve from one IPM iteration to the next (everything
// else is recomputed before it is read): YREF, AB, C | Z, PIQ, LAM, T | P, KFF | RG, RB.
// Compaction copies exactly these (two contiguous row ranges) for every stage 0..N.
enum : int { MPC_CARRY_A0 = R_YREF, MPC_CARRY_A1 = R_KFF + 2, MPC_CARRY_B0 = R_RG, MPC_CARRY_B1 = R_RB + 3 };

}  // namespace mpc
