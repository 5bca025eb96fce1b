// rti_wave2.hpp — the wave model with TWO SHOOTING STAGES PER LANE: lane k of a GW-lane group
// owns stages 2k and 2k+1 of one OCP, so a group covers N + 1 <= 2 GW stages and a wavefront
// holds 64 / GW OCPs (N = 50: GW = 32, two OCPs per wavefront).  Device code only (gfx950).
//
// Why: in rti_wave.hpp (one stage per lane) the parallel scans are 3/4 of the instruction
// stream of an IPM iteration and they cost the same whether one or two OCPs sit in the
// wavefront.  With two stages per lane
//   * the Riccati scan runs over GW lanes (one level fewer) on elements that are the local
//     combination of the lane's two stages, and serves two OCPs at once; the odd stage's value
//     function follows from one sequential Riccati step inside the lane;
//   * the affine scans (p backward, dx forward) run on the lane's two-stage closed-loop
//     product Psi_k = Phi_{2k+1} Phi_{2k}; the odd stage is one local matrix-vector product;
//   * wavefront reductions, ballots and neighbour shifts are shared by two OCPs;
//   * everything local to a stage is executed once per stage as before (twice per lane).
// The per-stage arithmetic is rti_core.hpp's Algebra and follows rti_wave.hpp line by line, so the
// results agree with the one-stage-per-lane kernels to rounding (the scans associate differently).
// The sequential recursions (factor_seq / rhs_seq / fwd_seq) are the in-kernel fallback of the
// acceptance test on the linear system's residual and the WAVE_SEQ cross-check model.
#pragma once
#include "rti_wave.hpp"

namespace mpc {

template <int GW, bool SCAN>
struct Wave2 : Algebra {
    using X = XWave<GW>;
    static constexpr int PER_BLOCK = 64 / GW;
    static constexpr int MC_LEVELS = X::MC_LEVELS;   // products of lane distance 1, 2, .., GW/2
    static constexpr bool GUARD = PER_BLOCK > 1;     // several OCPs per wavefront: scans need partner guards
    X x;
    int k, N, Nl;   // lane within the group, horizon, last lane that owns a stage (N / 2)
    // The interior-point arguments are compile-time constants here (only HPIPM's BALANCE mode is
    // implemented, include/mpc_hpipm_args.h), not kernel arguments held in ~30 scalar registers.
    struct K {
        static constexpr double mu0 = MPC_IPM_MU0, alpha_min = MPC_IPM_ALPHA_MIN;
        static constexpr double res_g_max = MPC_IPM_RES_G_MAX, res_b_max = MPC_IPM_RES_B_MAX, res_d_max = MPC_IPM_RES_D_MAX,
                                res_m_max = MPC_IPM_RES_M_MAX;
        static constexpr double reg_prim = MPC_IPM_REG_PRIM, lam_min = MPC_IPM_LAM_MIN, t_min = MPC_IPM_T_MIN, tau_min = MPC_IPM_TAU_MIN;
        static constexpr double thr0 = MPC_IPM_THR0, itref_tol_g = MPC_IPM_RES_G_MAX, lq_tol = MPC_IPM_LQ_SWITCH_TOL;
    };

    struct Stg {
        bool act, mid, isN, is0;   // j <= N, j < N, j == N, j == 0
        double fbu, fbx;           // 1.0 where the stage has input / state bounds, else 0.0
        AB8 ab;
        double z[5];
        Bnd bn;
        double ti[8];
        double rg[5], rb[3];
        Sym3 P;                    // value function of stage j + 1
        double pv[3], kff[2];
        double id0, l10, id1, K00, K01, K02, K10, K11, K12;
    };
    Stg s[2];
    // State that is touched once or twice per IPM iteration lives in LDS, not in registers (two
    // stages per lane = twice the state of rti_wave.hpp's lane): the references and the affine
    // term of the dynamics (read by BU), the multipliers pi (read by BU, updated by the accepted
    // step) and the affine step of the bounded variables (written by the predictor, read by the
    // corrector).  Every lane only reads what it wrote itself: no barriers.  Slots of 2 doubles
    // behind the product cache: stage e: 4e..4e+3 = yref[0..4], cv[0..2]; 8+2e, 9+2e = dza[0..3];
    // 12..14 = pi of both stages.
    // LDS budget: 40 KB per wavefront keeps four wavefronts (one per SIMD) on a CU.  Level 0 of
    // the products (Psi itself) is recomputed by every scan instead of cached: 27 multiply-adds
    // against 5 LDS reads, measured 3.4 % faster at GW = 32 and needed for the budget at GW = 64.
    static constexpr bool CACHE0 = false;
    static constexpr int MC_SLOTS = (X::MC_LEVELS - (CACHE0 ? 0 : 1)) * MPC_MC_PAIRS;
    static constexpr int PK0 = MC_SLOTS > 25 ? MC_SLOTS : 25;   // (the cache area also parks 25 slots during the Riccati scan)
    static constexpr int LDS_SLOTS = PK0 + 15;
    static_assert(LDS_SLOTS <= 40, "more than 40 KB of LDS per wavefront");
    int wdiag = 0;
#ifdef MPC_PHASE_CLOCK   // development build: s_memtime around the phases of an iteration (tools/phase_clock.py)
    long long ph[5] = {0, 0, 0, 0, 0};
#define MPC_PH(i, t_from, t_to) do { t_to = clock64(); ph[i] += t_to - t_from; } while (0)
#else
#define MPC_PH(i, t_from, t_to) do { } while (0)
#endif
    mc_d2* mc = nullptr;

    struct State {
        int kk, diag;
        double alpha, mu, musum;
        bool open;
        bool declined;   // SCAN only: the scan step of this iteration was not accurate enough (see iterate)
        bool force_lq;   // sequential path: HPIPM's lq_fact = 1 switch has fired, LQ factorisation from now on
    };

    __device__ __forceinline__ Wave2(const Consts& c_, int lane_in_group, mc_d2* mc_)
        : Algebra(c_), k(lane_in_group), N(c_.N), Nl(c_.N >> 1), mc(mc_) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * k + e;
            Stg& t = s[e];
            t.act = j <= N; t.mid = j < N; t.isN = j == N; t.is0 = j == 0;
            t.fbu = t.mid ? 1.0 : 0.0;
            t.fbx = (t.mid && !t.is0) ? 1.0 : 0.0;
        }
    }
    // Phase-local opaque constants 0.0 and 1.0.  Every phase of an iteration (residuals,
    // factorisation, predictor, corrector right-hand side, corrector) recomputes the cheap
    // per-bound quantities (lam*t, lb + t - z, ..) from the state instead of keeping them; the
    // compiler would merge those recomputations across phases (they are the same expressions)
    // and keep ~100 doubles per lane alive across the scans.  Writing lam*t as fma(lam, t, 0.0)
    // and lb + t as fma(t, 1.0, lb) with constants it cannot see through gives the same bits at the
    // same instruction count (v_fma instead of v_mul / v_add) and keeps the phases apart.
    struct Opq { double oz, one; };
    __device__ __forceinline__ static Opq opq() {
        Opq o;
        double a = 0.0, b = 1.0;
        asm volatile("" : "+v"(a));
        asm volatile("" : "+v"(b));
        o.oz = a; o.one = b;
        return o;
    }
    // acc + a0 b0 + a1 b1 + a2 b2 as one chain of fused multiply-adds
    __device__ __forceinline__ static double fma3(double acc, double a0, double b0, double a1, double b1, double a2, double b2) {
        return fma(a0, b0, fma(a1, b1, fma(a2, b2, acc)));
    }
    __device__ __forceinline__ static double bf(const Stg& t, int j) { return j < 2 ? t.fbu : t.fbx; }
    __device__ __forceinline__ static bool has_bound(const Stg& t, int j) { return t.mid && (j < 2 || !t.is0); }
    __device__ __forceinline__ void con_cached(const Stg& t, int j, Con& q, const Opq& o) const {
        q.lam_l = t.bn.lam[j]; q.lam_u = t.bn.lam[4 + j]; q.t_l = t.bn.t[j]; q.t_u = t.bn.t[4 + j];
        q.rd_l = fma(q.t_l, o.one, c.lb[j]) - t.z[j];       // = c.lb[j] + t_l - z
        q.rd_u = fma(q.t_u, o.one, t.z[j]) - c.ub[j];       // = t_u + z - c.ub[j]
        q.rm_l = fma(q.lam_l, q.t_l, o.oz);                 // = lam_l * t_l
        q.rm_u = fma(q.lam_u, q.t_u, o.oz);
        q.ti_l = t.ti[j];
        q.ti_u = t.ti[4 + j];
    }
    __device__ __forceinline__ static double con_Gamma_o(const Con& q, const Opq& o) {
        return fma(q.ti_l, q.lam_l, fma(q.ti_u, q.lam_u, o.oz));   // (rti_core's con_Gamma: ti_l*lam_l + ti_u*lam_u)
    }
    __device__ __forceinline__ void pk_st(int slot, double a, double b) const {
        mc_d2 v; v.x = a; v.y = b;
        mc[(PK0 + slot) * 64] = v;
    }
    __device__ __forceinline__ void pk_ld(int slot, double& a, double& b) const {
        const mc_d2 v = mc[(PK0 + slot) * 64];
        a = v.x; b = v.y;
    }
    __device__ __forceinline__ void set_yref(int e, const double y[5]) const {
        double c0, dummy;
        pk_ld(4 * e + 2, dummy, c0);
        pk_st(4 * e, y[0], y[1]); pk_st(4 * e + 1, y[2], y[3]); pk_st(4 * e + 2, y[4], c0);
    }
    __device__ __forceinline__ void get_ref(int e, double y[5], double cv[3]) const {
        pk_ld(4 * e, y[0], y[1]); pk_ld(4 * e + 1, y[2], y[3]); pk_ld(4 * e + 2, y[4], cv[0]); pk_ld(4 * e + 3, cv[1], cv[2]);
    }
    __device__ __forceinline__ void set_dza(int e, const double d[4]) const { pk_st(8 + 2 * e, d[0], d[1]); pk_st(9 + 2 * e, d[2], d[3]); }
    __device__ __forceinline__ void get_dza(int e, double d[4]) const { pk_ld(8 + 2 * e, d[0], d[1]); pk_ld(9 + 2 * e, d[2], d[3]); }
    __device__ __forceinline__ void set_pi(const double p[2][3]) const {
        pk_st(12, p[0][0], p[0][1]); pk_st(13, p[0][2], p[1][0]); pk_st(14, p[1][1], p[1][2]);
    }
    __device__ __forceinline__ void get_pi(double p[2][3]) const {
        pk_ld(12, p[0][0], p[0][1]); pk_ld(13, p[0][2], p[1][0]); pk_ld(14, p[1][1], p[1][2]);
    }
    // During the Riccati scan the product cache is dead (it is rebuilt by the first right-hand side
    // after the factorisation) and the scan's own temporaries need ~110 doubles per lane: the
    // multipliers, slacks and reciprocal slacks of both stages (48 doubles) wait in the cache area.
    __device__ __forceinline__ void park_bounds() const {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mc_d2 v;
                v.x = s[e].bn.lam[2 * q]; v.y = s[e].bn.lam[2 * q + 1]; mc[(12 * e + q) * 64] = v;
                v.x = s[e].bn.t[2 * q]; v.y = s[e].bn.t[2 * q + 1]; mc[(12 * e + 4 + q) * 64] = v;
                v.x = s[e].ti[2 * q]; v.y = s[e].ti[2 * q + 1]; mc[(12 * e + 8 + q) * 64] = v;
            }
        // the primal iterate z of both stages (10 doubles) waits in the slots of the affine step
        // dza, which is dead between the corrector of one iteration and the predictor of the next
        pk_st(8, s[0].z[0], s[0].z[1]); pk_st(9, s[0].z[2], s[0].z[3]); pk_st(10, s[0].z[4], s[1].z[0]);
        pk_st(11, s[1].z[1], s[1].z[2]);
        mc_d2 v; v.x = s[1].z[3]; v.y = s[1].z[4];
        mc[24 * 64] = v;   // slot 24: the last slot of the cache area (PK0 >= 25) or the first dza... see static_assert
    }
    __device__ __forceinline__ void unpark_bounds() {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mc_d2 v;
                v = mc[(12 * e + q) * 64]; s[e].bn.lam[2 * q] = v.x; s[e].bn.lam[2 * q + 1] = v.y;
                v = mc[(12 * e + 4 + q) * 64]; s[e].bn.t[2 * q] = v.x; s[e].bn.t[2 * q + 1] = v.y;
                v = mc[(12 * e + 8 + q) * 64]; s[e].ti[2 * q] = v.x; s[e].ti[2 * q + 1] = v.y;
            }
        pk_ld(8, s[0].z[0], s[0].z[1]); pk_ld(9, s[0].z[2], s[0].z[3]); pk_ld(10, s[0].z[4], s[1].z[0]);
        pk_ld(11, s[1].z[1], s[1].z[2]);
        const mc_d2 v = mc[24 * 64];
        s[1].z[3] = v.x; s[1].z[4] = v.y;
    }
    __device__ __forceinline__ void mc_store(int level_, const M3& M) const {
        if (!CACHE0 && level_ == 0) return;
        const int level = level_ - (CACHE0 ? 0 : 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mc_d2 v; v.x = M.m[2 * q]; v.y = M.m[2 * q + 1];
            mc[(level * MPC_MC_PAIRS + q) * 64] = v;
        }
        mc_d2 v; v.x = M.m[8]; v.y = 0.0;
        mc[(level * MPC_MC_PAIRS + 4) * 64] = v;
    }
    __device__ __forceinline__ M3 mc_load(int level_) const {
        const int level = level_ - (CACHE0 ? 0 : 1);
        M3 M;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const mc_d2 v = mc[(level * MPC_MC_PAIRS + q) * 64];
            M.m[2 * q] = v.x; M.m[2 * q + 1] = v.y;
        }
        M.m[8] = mc[(level * MPC_MC_PAIRS + 4) * 64].x;
        return M;
    }

    // ---- preparation of one stage (Wave::linearize / cold_start) ----
    __device__ __forceinline__ void linearize(int e, const double xi[3], const double ui[2], double xn[3]) {
        Stg& t = s[e];
        double cv[3];
        linearize_math(xi[0], xi[1], xi[2], ui[0], ui[1], t.ab, cv, xn);
        if (!t.mid) { t.ab.a = t.ab.b = t.ab.B00 = t.ab.B10 = t.ab.B20 = t.ab.B01 = t.ab.B11 = t.ab.B21 = 0.0; cv[0] = cv[1] = cv[2] = 0.0; }
        double y4, dummy;
        pk_ld(4 * e + 2, y4, dummy);
        pk_st(4 * e + 2, y4, cv[0]); pk_st(4 * e + 3, cv[1], cv[2]);
    }
    __device__ __forceinline__ void cold_start(int e, const double xi[3], const double ui[2], const double x0bar[3]) {
        Stg& t = s[e];
        t.z[0] = t.mid ? ui[0] : 0.0;
        t.z[1] = t.mid ? ui[1] : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) t.z[2 + q] = t.is0 ? x0bar[q] : xi[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double tl = t.z[j] - c.lb[j];
            double tu = c.ub[j] - t.z[j];
            double zj = t.z[j];
            if (tl < K::thr0) {
                if (tu < K::thr0) { zj = 0.5 * (c.lb[j] + c.ub[j]); tl = K::thr0; tu = K::thr0; }
                else { tl = K::thr0; zj = c.lb[j] + K::thr0; }
            } else if (tu < K::thr0) { tu = K::thr0; zj = c.ub[j] - K::thr0; }
            // (every member is assigned exactly once, by value selects: stores to different members
            //  in the two arms of a branch get merged into one store through a computed address,
            //  which keeps the whole object in scratch memory)
            const bool hb = has_bound(t, j);
            const double til = rcp(tl), tiu = rcp(tu);
            t.z[j] = hb ? zj : t.z[j];
            t.bn.t[j] = hb ? tl : 1.0;
            t.bn.t[4 + j] = hb ? tu : 1.0;
            t.ti[j] = hb ? til : 1.0;
            t.ti[4 + j] = hb ? tiu : 1.0;
            t.bn.lam[j] = hb ? K::mu0 * til : 0.0;
            t.bn.lam[4 + j] = hb ? K::mu0 * tiu : 0.0;
        }
        if (e == 0) {
            const double z4[4] = {0.0, 0.0, 0.0, 0.0};
            const double z23[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            set_dza(0, z4); set_dza(1, z4); set_pi(z23);
        }
        t.pv[0] = t.pv[1] = t.pv[2] = 0.0;
        t.kff[0] = t.kff[1] = 0.0;
        t.id0 = t.l10 = t.id1 = t.K00 = t.K01 = t.K02 = t.K10 = t.K11 = t.K12 = 0.0;
        t.P.p00 = t.P.p10 = t.P.p11 = t.P.p20 = t.P.p21 = t.P.p22 = 0.0;
        t.rb[0] = t.rb[1] = t.rb[2] = 0.0;
    }

    // ---- NLP residuals of acados' full SQP (Wave::nlp_residuals); needs linearize() first ----
    __device__ __forceinline__ void nlp_residuals(const double xi[2][3], const double ui[2][2], const double x0bar[3],
                                                  const double xn[2][3], const double pin[2][3], const double lamn[2][8],
                                                  double out[4]) const {
        double r_stat = 0.0, r_eq = 0.0, r_ineq = 0.0, r_comp = 0.0;
        double pmv[3] = {pin[1][0], pin[1][1], pin[1][2]}, xnv[3] = {xi[0][0], xi[0][1], xi[0][2]};
        x.prev(pmv);   // pi of stage 2k - 1
        x.next(xnv);   // x of stage 2k + 2
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            const double zz[5] = {t.mid ? ui[e][0] : 0.0, t.mid ? ui[e][1] : 0.0, xi[e][0], xi[e][1], xi[e][2]};
            double yref[5], cvu[3];
            get_ref(e, yref, cvu);
            double g[5];
            g[0] = c.wg[0] * (zz[0] - yref[3]);
            g[1] = c.wg[1] * (zz[1] - yref[4]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double pm = e == 0 ? pmv[q] : pin[0][q];
                if (t.is0) pm = 0.0;
                g[2 + q] = (t.isN ? c.we[q] : c.wg[2 + q]) * (zz[2 + q] - yref[q]) - pm;
                const double xnext = e == 0 ? xi[1][q] : xnv[q];
                if (t.mid) r_eq = amax(r_eq, xn[e][q] - xnext);
                if (t.is0) r_ineq = amax(r_ineq, x0bar[q] - xi[e][q]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (has_bound(t, j)) {
                    const double dl = c.lb[j] - zz[j], du = zz[j] - c.ub[j];
                    g[j] += lamn[e][4 + j] - lamn[e][j];
                    r_ineq = fmax(r_ineq, dl); r_ineq = fmax(r_ineq, du);
                    r_comp = amax(r_comp, lamn[e][j] * dl); r_comp = amax(r_comp, lamn[e][4 + j] * du);
                }
            }
            if (t.mid) {
                const AB8& ab = t.ab;
                g[0] += ab.B00 * pin[e][0] + ab.B10 * pin[e][1] + ab.B20 * pin[e][2];
                g[1] += ab.B01 * pin[e][0] + ab.B11 * pin[e][1] + ab.B21 * pin[e][2];
                g[2] += pin[e][0]; g[3] += pin[e][1]; g[4] += ab.a * pin[e][0] + ab.b * pin[e][1] + pin[e][2];
                r_stat = amax(r_stat, g[0]); r_stat = amax(r_stat, g[1]);
            }
            if (t.act && !t.is0) { r_stat = amax(r_stat, g[2]); r_stat = amax(r_stat, g[3]); r_stat = amax(r_stat, g[4]); }
        }
        double rr[4] = {r_stat, r_eq, r_ineq, r_comp};
        x.rmax(rr);
        out[0] = rr[0]; out[1] = rr[1]; out[2] = rr[2]; out[3] = rr[3];
    }

    // ---- BU: residuals at the current point, both stages, and HPIPM's exit test (Wave::bu) ----
    __device__ __forceinline__ void bu(double& musum, bool& open) {
        double pi[2][3];
        get_pi(pi);
        double pim0[3] = {pi[1][0], pi[1][1], pi[1][2]};         // -> pi of stage 2k - 1
        double zxn1[3] = {s[0].z[2], s[0].z[3], s[0].z[4]};      // -> x of stage 2k + 2
        x.prev(pim0);
        x.next(zxn1);
        double pg = 0.0, pb = 0.0, ps = 0.0;
        bool viol = false;
        const Opq o = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
            double pim[3], zxn[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                pim[q] = e == 0 ? pim0[q] : pi[0][q];
                zxn[q] = e == 0 ? s[1].z[2 + q] : zxn1[q];
            }
            if (t.is0) { pim[0] = pim[1] = pim[2] = 0.0; }
            double yref[5], cv[3];
            get_ref(e, yref, cv);
            const double* tpi = pi[e];
            double* rg = t.rg;
            const double* z = t.z;
            const AB8& ab = t.ab;
            rg[0] = c.wg[0] * (z[0] - yref[3]);
            rg[1] = c.wg[1] * (z[1] - yref[4]);
#pragma unroll
            for (int q = 0; q < 3; ++q) rg[2 + q] = (t.isN ? c.we[q] : c.wg[2 + q]) * (z[2 + q] - yref[q]) - pim[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Con q;
                con_cached(t, j, q, o);
                rg[j] += q.lam_u - q.lam_l;
                viol = viol || (has_bound(t, j) && (fabs(q.rd_l) > K::res_d_max || fabs(q.rd_u) > K::res_d_max ||
                                                    fabs(fabs(q.rm_l) - K::tau_min) > K::res_m_max ||
                                                    fabs(fabs(q.rm_u) - K::tau_min) > K::res_m_max));
                ps += q.rm_l;
                ps += q.rm_u;
            }
            const double rb0 = cv[0] - zxn[0] + (ab.B00 * z[0] + ab.B01 * z[1] + (z[2] + ab.a * z[4]));
            const double rb1 = cv[1] - zxn[1] + (ab.B10 * z[0] + ab.B11 * z[1] + (z[3] + ab.b * z[4]));
            const double rb2 = cv[2] - zxn[2] + (ab.B20 * z[0] + ab.B21 * z[1] + z[4]);
            t.rb[0] = t.mid ? rb0 : 0.0; t.rb[1] = t.mid ? rb1 : 0.0; t.rb[2] = t.mid ? rb2 : 0.0;
            if (t.mid) {
                pb = amax(pb, rb0); pb = amax(pb, rb1); pb = amax(pb, rb2);
                rg[0] += ab.B00 * tpi[0] + ab.B10 * tpi[1] + ab.B20 * tpi[2];
                rg[1] += ab.B01 * tpi[0] + ab.B11 * tpi[1] + ab.B21 * tpi[2];
                rg[2] += tpi[0];
                rg[3] += tpi[1];
                rg[4] += ab.a * tpi[0] + ab.b * tpi[1] + tpi[2];
                pg = amax(pg, rg[0]); pg = amax(pg, rg[1]);
            }
            if (t.act && !t.is0) { pg = amax(pg, rg[2]); pg = amax(pg, rg[3]); pg = amax(pg, rg[4]); }
        }
        open = x.gany(pg > K::res_g_max || pb > K::res_b_max || viol);
        double ns[1] = {ps};
        x.rsum(ns);
        musum = ns[0];
    }

    // barrier terms of the current point (exact zeros without a bound)
    __device__ __forceinline__ void barrier_terms(double Gam[2][4], double gam[2][4]) const {
        const Opq o = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (no has_bound selects: entries without a bound carry lam = 0, t = 1, so Gamma and
                //  gamma are exact zeros)
                Con q;
                con_cached(s[e], j, q, o);
                Gam[e][j] = con_Gamma_o(q, o);
                gam[e][j] = con_gamma(q);
            }
    }
    // US: use the parallel scans (compile-time: the sequential fallback of an iteration is a
    // separate, cold copy of the direction computation, not a run-time switch inside every phase)
    template <bool US>
    __device__ __forceinline__ void factor_now(bool lq = false) {
        if (US) {
            factor_scan();
        } else {
            double Gam[2][4], gam[2][4];
            barrier_terms(Gam, gam);
            if (lq) factor_seq(Gam, gam, FACT_LQ);
            else factor_seq(Gam, gam, FACT_CHOL);
        }
    }

    __device__ __forceinline__ static void set_gain(Stg& t, const Gain& g) {
        t.id0 = g.id0; t.l10 = g.l10; t.id1 = g.id1;
        t.K00 = g.K00; t.K01 = g.K01; t.K02 = g.K02; t.K10 = g.K10; t.K11 = g.K11; t.K12 = g.K12;
    }
    // one sequential Riccati step: value function of stage j from P = P_{j+1} and the gains of stage j
    __device__ __forceinline__ Sym3 riccati_step(const Sym3& P, const AB8& ab, double GamX, double GamY, const Gain& g) const {
        const double a = ab.a, b = ab.b;
        const double S02 = P.p00 * a + P.p10 * b + P.p20;
        const double S12 = P.p10 * a + P.p11 * b + P.p21;
        const double S22 = P.p20 * a + P.p21 * b + P.p22;
        const double Hxx00 = c.wgr[2] + GamX + P.p00;
        const double Hxx10 = P.p10;
        const double Hxx11 = c.wgr[3] + GamY + P.p11;
        const double Hxx20 = a * P.p00 + b * P.p10 + P.p20;
        const double Hxx21 = a * P.p10 + b * P.p11 + P.p21;
        const double Hxx22 = c.wgr[4] + (a * S02 + b * S12 + S22);
        Sym3 Po;
        Po.p00 = Hxx00 + (g.M00 * g.K00 + g.M10 * g.K10);
        Po.p10 = Hxx10 + (g.M01 * g.K00 + g.M11 * g.K10);
        Po.p11 = Hxx11 + (g.M01 * g.K01 + g.M11 * g.K11);
        Po.p20 = Hxx20 + (g.X02 * g.K00 + g.X12 * g.K10);
        Po.p21 = Hxx21 + (g.X02 * g.K01 + g.X12 * g.K11);
        Po.p22 = Hxx22 + (g.X02 * g.K02 + g.X12 * g.K12);
        return Po;
    }
    __device__ __forceinline__ Sym3 terminal_P() const {
        Sym3 P;
        P.p00 = c.wer[0]; P.p10 = 0.0; P.p11 = c.wer[1];
        P.p20 = 0.0; P.p21 = 0.0; P.p22 = c.wer[2];
        return P;
    }
    __device__ __forceinline__ static Sym3 zero_P() {
        Sym3 P;
        P.p00 = P.p10 = P.p11 = P.p20 = P.p21 = P.p22 = 0.0;
        return P;
    }

    // ---- sequential recursions over the stages (fallback of the acceptance test / WAVE_SEQ) ----
    // One stage of the backward recursion: in (P, p) of stage j + 1, out (P, p) of stage j.
    // Right-hand side of the stage: g5 = gradient part (u0, u1, x, y, psi; barrier terms included),
    // rb = dynamics part.  (The IPM direction passes its residuals, the refinement the residual of
    // the linear system itself.)
    // `fact`: 0 = right-hand side only, 1 = factorise (value function P, LDL^T of H_uu), 2 = factorise
    // with HPIPM's LQ variant (lq_stage below): the six numbers passed from stage to stage are then the
    // lower Cholesky factor S of the value function (P = S S^T) instead of P itself.
    enum { FACT_NONE = 0, FACT_CHOL = 1, FACT_LQ = 2 };
    __device__ __forceinline__ static Sym3 sst(const Sym3& S) {   // S S^T, S lower: (00, 10, 11, 20, 21, 22)
        Sym3 P;
        P.p00 = S.p00 * S.p00;
        P.p10 = S.p10 * S.p00;
        P.p11 = S.p10 * S.p10 + S.p11 * S.p11;
        P.p20 = S.p20 * S.p00;
        P.p21 = S.p20 * S.p10 + S.p21 * S.p11;
        P.p22 = S.p20 * S.p20 + S.p21 * S.p21 + S.p22 * S.p22;
        return P;
    }
    // One stage of OCP_QP_FACT_LQ_SOLVE_KKT_STEP (HPIPM x_ocp_qp_kkt.c; BALANCE mode's lq_fact = 1 switches
    // to it when the regular factorisation leaves a residual above 1e-5, direction()): the Cholesky
    // factor L of the stage Hessian H = D + G G^T, D = diag(cost Hessian + reg_prim + barrier terms),
    // G = [B A]^T S with S the factor of the next stage's value function, is computed from the square
    // roots — HPIPM: Householder LQ of [D^1/2 | G]; here: diag(D^1/2) updated by the three columns of G
    // with Givens rotations (the same orthogonal triangularisation, column by column) — so H is
    // never formed and its condition number never squared.  Out: the gains of the stage in the form the
    // substitutions use (LDL^T of H_uu, K = -H_uu^-1 H_ux) and S <- L_xx, the factor of this stage's
    // value function.
    __device__ __forceinline__ void lq_stage(Stg& t, const double Gam[4], Sym3& S) {
        const AB8& ab = t.ab;
        double L[5][5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) L[i][j] = 0.0;
        L[0][0] = sqrt(c.wgr[0] + Gam[0]); L[1][1] = sqrt(c.wgr[1] + Gam[1]);
        L[2][2] = sqrt(c.wgr[2] + Gam[2]); L[3][3] = sqrt(c.wgr[3] + Gam[3]); L[4][4] = sqrt(c.wgr[4]);
        // columns of G = [B A]^T S (rows: v, delta, x, y, psi)
        double G[3][5];
        G[0][0] = ab.B00 * S.p00 + ab.B10 * S.p10 + ab.B20 * S.p20; G[1][0] = ab.B10 * S.p11 + ab.B20 * S.p21; G[2][0] = ab.B20 * S.p22;
        G[0][1] = ab.B01 * S.p00 + ab.B11 * S.p10 + ab.B21 * S.p20; G[1][1] = ab.B11 * S.p11 + ab.B21 * S.p21; G[2][1] = ab.B21 * S.p22;
        G[0][2] = S.p00; G[1][2] = 0.0; G[2][2] = 0.0;
        G[0][3] = S.p10; G[1][3] = S.p11; G[2][3] = 0.0;
        G[0][4] = ab.a * S.p00 + ab.b * S.p10 + S.p20; G[1][4] = ab.b * S.p11 + S.p21; G[2][4] = S.p22;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double* x = G[q];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double r = sqrt(L[j][j] * L[j][j] + x[j] * x[j]);
                const double il = rcp(L[j][j]);
                const double cs = r * il, sn = x[j] * il, ics = rcp(cs);
                L[j][j] = r;
#pragma unroll
                for (int i = j + 1; i < 5; ++i) {
                    const double lij = (L[i][j] + sn * x[i]) * ics;
                    x[i] = cs * x[i] - sn * lij;
                    L[i][j] = lij;
                }
            }
        }
        t.id0 = rcp(L[0][0] * L[0][0]);
        t.l10 = L[1][0] * rcp(L[0][0]);
        t.id1 = rcp(L[1][1] * L[1][1]);
        // H_ux = L_uu L_xu^T, column by column of the state
        const double h00 = L[0][0] * L[2][0], h10 = L[1][0] * L[2][0] + L[1][1] * L[2][1];
        const double h01 = L[0][0] * L[3][0], h11 = L[1][0] * L[3][0] + L[1][1] * L[3][1];
        const double h02 = L[0][0] * L[4][0], h12 = L[1][0] * L[4][0] + L[1][1] * L[4][1];
        solve2(t.id0, t.l10, t.id1, h00, h10, t.K00, t.K10);
        solve2(t.id0, t.l10, t.id1, h01, h11, t.K01, t.K11);
        solve2(t.id0, t.l10, t.id1, h02, h12, t.K02, t.K12);
        S.p00 = L[2][2]; S.p10 = L[3][2]; S.p11 = L[3][3]; S.p20 = L[4][2]; S.p21 = L[4][3]; S.p22 = L[4][4];
    }
    __device__ __forceinline__ void seq_back(Stg& t, const double Gam[4], const double g5[5], const double rb[3],
                                             int fact, Sym3& P, double p[3]) {
        if (t.mid) {
            if (fact == FACT_CHOL) t.P = P;
            else if (fact == FACT_LQ) t.P = sst(P);
            t.pv[0] = p[0]; t.pv[1] = p[1]; t.pv[2] = p[2];
            const AB8& ab = t.ab;
            const Sym3& Q = t.P;
            const double w0 = Q.p00 * rb[0] + Q.p10 * rb[1] + Q.p20 * rb[2] + p[0];
            const double w1 = Q.p10 * rb[0] + Q.p11 * rb[1] + Q.p21 * rb[2] + p[1];
            const double w2 = Q.p20 * rb[0] + Q.p21 * rb[1] + Q.p22 * rb[2] + p[2];
            const double hu0 = g5[0] + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
            const double hu1 = g5[1] + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
            if (fact == FACT_CHOL) {
                Gain g;
                gain(t.P, ab, Gam[0], Gam[1], true, g);
                set_gain(t, g);
                if (!t.is0) P = riccati_step(t.P, ab, Gam[2], Gam[3], g);
            } else if (fact == FACT_LQ) {
                lq_stage(t, Gam, P);
            }
            solve2(t.id0, t.l10, t.id1, hu0, hu1, t.kff[0], t.kff[1]);
            if (!t.is0) {
                const double hx0 = g5[2] + w0;
                const double hx1 = g5[3] + w1;
                const double hx2 = g5[4] + (ab.a * w0 + ab.b * w1 + w2);
                p[0] = hx0 + (t.K00 * hu0 + t.K10 * hu1);
                p[1] = hx1 + (t.K01 * hu0 + t.K11 * hu1);
                p[2] = hx2 + (t.K02 * hu0 + t.K12 * hu1);
            }
        } else if (t.isN) {
            if (fact == FACT_CHOL) P = terminal_P();
            else if (fact == FACT_LQ) { P = zero_P(); P.p00 = sqrt(c.wer[0]); P.p11 = sqrt(c.wer[1]); P.p22 = sqrt(c.wer[2]); }
            p[0] = g5[2]; p[1] = g5[3]; p[2] = g5[4];
        }
    }
    __device__ __forceinline__ void back_seq(const double Gam[2][4], const double g5[2][5], const double rb[2][3], int fact) {
        Sym3 Po = terminal_P();
        double po[3] = {0.0, 0.0, 0.0};
        for (int kl = Nl; kl >= 0; --kl) {
            double sh[9] = {Po.p00, Po.p10, Po.p11, Po.p20, Po.p21, Po.p22, po[0], po[1], po[2]};
            x.next(sh);
            if (k == kl) {
                Sym3 P;
                P.p00 = sh[0]; P.p10 = sh[1]; P.p11 = sh[2]; P.p20 = sh[3]; P.p21 = sh[4]; P.p22 = sh[5];
                double p[3] = {sh[6], sh[7], sh[8]};
                seq_back(s[1], Gam[1], g5[1], rb[1], fact, P, p);
                seq_back(s[0], Gam[0], g5[0], rb[0], fact, P, p);
                Po = P; po[0] = p[0]; po[1] = p[1]; po[2] = p[2];
            }
        }
    }
    // the IPM direction's right-hand side: residuals of the stage + barrier terms
    __device__ __forceinline__ void back_seq_ipm(const double Gam[2][4], const double hup[2][2], const double hxp[2][2], int fact) {
        double g5[2][5], rb[2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            g5[e][0] = hup[e][0]; g5[e][1] = hup[e][1]; g5[e][2] = hxp[e][0]; g5[e][3] = hxp[e][1]; g5[e][4] = s[e].rg[4];
            rb[e][0] = s[e].rb[0]; rb[e][1] = s[e].rb[1]; rb[e][2] = s[e].rb[2];
        }
        back_seq(Gam, g5, rb, fact);
    }
    __device__ __forceinline__ void factor_seq(const double Gam[2][4], const double gam[2][4], int fact = FACT_CHOL) {
        double hup[2][2], hxp[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hup[e][0] = s[e].rg[0] + gam[e][0]; hup[e][1] = s[e].rg[1] + gam[e][1];
            hxp[e][0] = s[e].rg[2] + gam[e][2]; hxp[e][1] = s[e].rg[3] + gam[e][3];
        }
        back_seq_ipm(Gam, hup, hxp, fact);
    }
    // one stage of the forward recursion: in dx of stage j, out dx of stage j + 1
    __device__ __forceinline__ void seq_fwd(const Stg& t, const double rb[3], double dx[3], double dl[5]) const {
        if (t.mid) {
            double du0 = t.kff[0], du1 = t.kff[1];
            if (!t.is0) {
                du0 += t.K00 * dx[0] + t.K01 * dx[1] + t.K02 * dx[2];
                du1 += t.K10 * dx[0] + t.K11 * dx[1] + t.K12 * dx[2];
            } else {
                dx[0] = dx[1] = dx[2] = 0.0;   // the stage-0 state is fixed
            }
            dl[0] = du0; dl[1] = du1; dl[2] = dx[0]; dl[3] = dx[1]; dl[4] = dx[2];
            const AB8& ab = t.ab;
            const double n0 = rb[0] + (ab.B00 * du0 + ab.B01 * du1 + (dx[0] + ab.a * dx[2]));
            const double n1 = rb[1] + (ab.B10 * du0 + ab.B11 * du1 + (dx[1] + ab.b * dx[2]));
            const double n2 = rb[2] + (ab.B20 * du0 + ab.B21 * du1 + dx[2]);
            dx[0] = n0; dx[1] = n1; dx[2] = n2;
        } else if (t.isN) {
            dl[2] = dx[0]; dl[3] = dx[1]; dl[4] = dx[2];
        }
    }
    __device__ __forceinline__ void fwd_seq(const double rb[2][3], double nx[2][3], double dl[2][5]) const {
        double out[3] = {0.0, 0.0, 0.0};
        for (int kl = 0; kl <= Nl; ++kl) {
            double sh[3] = {out[0], out[1], out[2]};
            x.prev(sh);
            if (k == kl) {
                double dx[3] = {sh[0], sh[1], sh[2]};
                seq_fwd(s[0], rb[0], dx, dl[0]);
                nx[0][0] = dx[0]; nx[0][1] = dx[1]; nx[0][2] = dx[2];
                seq_fwd(s[1], rb[1], dx, dl[1]);
                nx[1][0] = dx[0]; nx[1][1] = dx[1]; nx[1][2] = dx[2];
                out[0] = dx[0]; out[1] = dx[1]; out[2] = dx[2];
            }
        }
    }
    __device__ __forceinline__ void fwd_seq_ipm(double nx[2][3], double dl[2][5]) const {
        const double rb[2][3] = {{s[0].rb[0], s[0].rb[1], s[0].rb[2]}, {s[1].rb[0], s[1].rb[1], s[1].rb[2]}};
        fwd_seq(rb, nx, dl);
    }

    // ================= parallel-in-time versions (see rti_wave.hpp for the method) =================
    // closed-loop transition of stage e: Phi = A + B K
    // (o: phase-local opaque constants, see opq(): the backward and the forward scan of a solve each
    //  recompute Phi instead of sharing 18 live doubles per stage across the scans)
    __device__ __forceinline__ M3 closed_loop(const Stg& t, const Opq& o) const {
        const AB8& ab = t.ab;
        M3 F;
        F.m[0] = 1.0 + fma(ab.B00, t.K00, fma(ab.B01, t.K10, o.oz)); F.m[1] = fma(ab.B00, t.K01, fma(ab.B01, t.K11, o.oz)); F.m[2] = ab.a + fma(ab.B00, t.K02, fma(ab.B01, t.K12, o.oz));
        F.m[3] = fma(ab.B10, t.K00, fma(ab.B11, t.K10, o.oz)); F.m[4] = 1.0 + fma(ab.B10, t.K01, fma(ab.B11, t.K11, o.oz)); F.m[5] = ab.b + fma(ab.B10, t.K02, fma(ab.B11, t.K12, o.oz));
        F.m[6] = fma(ab.B20, t.K00, fma(ab.B21, t.K10, o.oz)); F.m[7] = fma(ab.B20, t.K01, fma(ab.B21, t.K11, o.oz)); F.m[8] = 1.0 + fma(ab.B20, t.K02, fma(ab.B21, t.K12, o.oz));
        return F;
    }
    // Phi of the recursions: 0 for stage 0 (dx_0 = 0: the stage-0 transition never acts, p_0 is
    // never formed) and from stage N on
    __device__ __forceinline__ M3 closed_loop_z(const Stg& t, const Opq& o) const {
        M3 F = closed_loop(t, o);
        if (!t.mid || t.is0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) F.m[i] = 0.0;
        }
        return F;
    }

    // Combination of two Riccati-scan elements, element 1 = (A, C, J) of the earlier segment (in
    // place), element 2 the later one; `more`: A and C of the result are needed by later levels.
    __device__ __forceinline__ static void combine_full(M3& A, double C[6], double J[6], const M3& A2, const double C2[6],
                                                        const double J2[6], bool more) {
        M3 Cm, Jm;
        Cm.m[0] = C[0]; Cm.m[1] = C[1]; Cm.m[2] = C[3]; Cm.m[3] = C[1]; Cm.m[4] = C[2]; Cm.m[5] = C[4]; Cm.m[6] = C[3]; Cm.m[7] = C[4]; Cm.m[8] = C[5];
        Jm.m[0] = J2[0]; Jm.m[1] = J2[1]; Jm.m[2] = J2[3]; Jm.m[3] = J2[1]; Jm.m[4] = J2[2]; Jm.m[5] = J2[4]; Jm.m[6] = J2[3]; Jm.m[7] = J2[4]; Jm.m[8] = J2[5];
        // (sums that end in "+ accumulator" are written as one fused chain into the accumulator:
        //  three instructions instead of multiply, two fused multiply-adds and an add)
        M3 Xm = m3_mul(Cm, Jm);
        Xm.m[0] = fma3(1.0, Cm.m[0], Jm.m[0], Cm.m[1], Jm.m[3], Cm.m[2], Jm.m[6]);
        Xm.m[4] = fma3(1.0, Cm.m[3], Jm.m[1], Cm.m[4], Jm.m[4], Cm.m[5], Jm.m[7]);
        Xm.m[8] = fma3(1.0, Cm.m[6], Jm.m[2], Cm.m[7], Jm.m[5], Cm.m[8], Jm.m[8]);
        M3 W;   // Xm^-1 (adjugate)
        const double c00 = Xm.m[4] * Xm.m[8] - Xm.m[5] * Xm.m[7];
        const double c01 = Xm.m[5] * Xm.m[6] - Xm.m[3] * Xm.m[8];
        const double c02 = Xm.m[3] * Xm.m[7] - Xm.m[4] * Xm.m[6];
        const double idet = rcp(Xm.m[0] * c00 + Xm.m[1] * c01 + Xm.m[2] * c02);
        W.m[0] = c00 * idet; W.m[3] = c01 * idet; W.m[6] = c02 * idet;
        W.m[1] = (Xm.m[2] * Xm.m[7] - Xm.m[1] * Xm.m[8]) * idet;
        W.m[4] = (Xm.m[0] * Xm.m[8] - Xm.m[2] * Xm.m[6]) * idet;
        W.m[7] = (Xm.m[1] * Xm.m[6] - Xm.m[0] * Xm.m[7]) * idet;
        W.m[2] = (Xm.m[1] * Xm.m[5] - Xm.m[2] * Xm.m[4]) * idet;
        W.m[5] = (Xm.m[2] * Xm.m[3] - Xm.m[0] * Xm.m[5]) * idet;
        W.m[8] = (Xm.m[0] * Xm.m[4] - Xm.m[1] * Xm.m[3]) * idet;
        // J2 W = (J2^-1 + C1)^-1 is symmetric: 6 of its 9 entries are computed, 3 mirrored
        M3 JW;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                JW.m[3 * i + j] = Jm.m[3 * i] * W.m[j] + Jm.m[3 * i + 1] * W.m[3 + j] + Jm.m[3 * i + 2] * W.m[6 + j];
                JW.m[3 * j + i] = JW.m[3 * i + j];
            }
        const M3 U = m3_mul(JW, A);
        J[0] = fma3(J[0], A.m[0], U.m[0], A.m[3], U.m[3], A.m[6], U.m[6]);
        J[1] = fma3(J[1], A.m[1], U.m[0], A.m[4], U.m[3], A.m[7], U.m[6]);
        J[2] = fma3(J[2], A.m[1], U.m[1], A.m[4], U.m[4], A.m[7], U.m[7]);
        J[3] = fma3(J[3], A.m[2], U.m[0], A.m[5], U.m[3], A.m[8], U.m[6]);
        J[4] = fma3(J[4], A.m[2], U.m[1], A.m[5], U.m[4], A.m[8], U.m[7]);
        J[5] = fma3(J[5], A.m[2], U.m[2], A.m[5], U.m[5], A.m[8], U.m[8]);
        if (more) {
            const M3 G = m3_mul(A2, W);
            const M3 T = m3_mul(G, Cm);
            C[0] = fma3(C2[0], T.m[0], A2.m[0], T.m[1], A2.m[1], T.m[2], A2.m[2]);
            C[1] = fma3(C2[1], T.m[3], A2.m[0], T.m[4], A2.m[1], T.m[5], A2.m[2]);
            C[2] = fma3(C2[2], T.m[3], A2.m[3], T.m[4], A2.m[4], T.m[5], A2.m[5]);
            C[3] = fma3(C2[3], T.m[6], A2.m[0], T.m[7], A2.m[1], T.m[8], A2.m[2]);
            C[4] = fma3(C2[4], T.m[6], A2.m[3], T.m[7], A2.m[4], T.m[8], A2.m[5]);
            C[5] = fma3(C2[5], T.m[6], A2.m[6], T.m[7], A2.m[7], T.m[8], A2.m[8]);
            A = m3_mul(G, A);
        }
    }

    // single-stage element in structured form: A = e I + a E02 + b E12, C = B R^-1 B^T, J diagonal
    struct SE { double e, a, b, C[6], j0, j1, j2; };
    __device__ __forceinline__ SE stage_element(const Stg& t, const double Gam[4]) const {
        SE r;
        if (t.mid) {
            const AB8& ab = t.ab;
            r.e = 1.0; r.a = ab.a; r.b = ab.b;
            const double r0 = rcp(c.wgr[0] + Gam[0]), r1 = rcp(c.wgr[1] + Gam[1]);
            const double b00 = ab.B00 * r0, b10 = ab.B10 * r0, b20 = ab.B20 * r0;
            const double b01 = ab.B01 * r1, b11 = ab.B11 * r1, b21 = ab.B21 * r1;
            r.C[0] = b00 * ab.B00 + b01 * ab.B01;
            r.C[1] = b10 * ab.B00 + b11 * ab.B01;
            r.C[2] = b10 * ab.B10 + b11 * ab.B11;
            r.C[3] = b20 * ab.B00 + b21 * ab.B01;
            r.C[4] = b20 * ab.B10 + b21 * ab.B11;
            r.C[5] = b20 * ab.B20 + b21 * ab.B21;
            r.j0 = c.wgr[2] + Gam[2]; r.j1 = c.wgr[3] + Gam[3]; r.j2 = c.wgr[4];
        } else {   // terminal stage (0, 0, Q_N); stages beyond N get the same finite element
            r.e = 0.0; r.a = 0.0; r.b = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) r.C[i] = 0.0;
            r.j0 = c.wer[0]; r.j1 = c.wer[1]; r.j2 = c.wer[2];
        }
        return r;
    }

    __device__ __forceinline__ void factor_scan() {
        // --- the lane's element: stage 2k combined with stage 2k+1, both in structured form
        //     (rti_wave.hpp's first scan level with the partner taken from the same lane) ---
        M3 A;
        double C[6], J[6];
        {
            double Gam0[2][4], gam0[2][4];
            barrier_terms(Gam0, gam0);
            const SE e1 = stage_element(s[0], Gam0[0]), e2 = stage_element(s[1], Gam0[1]);
            park_bounds();
            M3 Cm;
            Cm.m[0] = e1.C[0]; Cm.m[1] = e1.C[1]; Cm.m[2] = e1.C[3]; Cm.m[3] = e1.C[1]; Cm.m[4] = e1.C[2]; Cm.m[5] = e1.C[4]; Cm.m[6] = e1.C[3]; Cm.m[7] = e1.C[4]; Cm.m[8] = e1.C[5];
            M3 Xm;   // I + C1 diag(j2)
#pragma unroll
            for (int i = 0; i < 3; ++i) { Xm.m[3 * i] = Cm.m[3 * i] * e2.j0; Xm.m[3 * i + 1] = Cm.m[3 * i + 1] * e2.j1; Xm.m[3 * i + 2] = Cm.m[3 * i + 2] * e2.j2; }
            Xm.m[0] += 1.0; Xm.m[4] += 1.0; Xm.m[8] += 1.0;
            M3 W;
            const double c00 = Xm.m[4] * Xm.m[8] - Xm.m[5] * Xm.m[7];
            const double c01 = Xm.m[5] * Xm.m[6] - Xm.m[3] * Xm.m[8];
            const double c02 = Xm.m[3] * Xm.m[7] - Xm.m[4] * Xm.m[6];
            const double idet = rcp(Xm.m[0] * c00 + Xm.m[1] * c01 + Xm.m[2] * c02);
            W.m[0] = c00 * idet; W.m[3] = c01 * idet; W.m[6] = c02 * idet;
            W.m[1] = (Xm.m[2] * Xm.m[7] - Xm.m[1] * Xm.m[8]) * idet;
            W.m[4] = (Xm.m[0] * Xm.m[8] - Xm.m[2] * Xm.m[6]) * idet;
            W.m[7] = (Xm.m[1] * Xm.m[6] - Xm.m[0] * Xm.m[7]) * idet;
            W.m[2] = (Xm.m[1] * Xm.m[5] - Xm.m[2] * Xm.m[4]) * idet;
            W.m[5] = (Xm.m[2] * Xm.m[3] - Xm.m[0] * Xm.m[5]) * idet;
            W.m[8] = (Xm.m[0] * Xm.m[4] - Xm.m[1] * Xm.m[3]) * idet;
            // JW = diag(j2) W, symmetric
            M3 JW;
            JW.m[0] = e2.j0 * W.m[0];
            JW.m[3] = e2.j1 * W.m[3]; JW.m[4] = e2.j1 * W.m[4];
            JW.m[6] = e2.j2 * W.m[6]; JW.m[7] = e2.j2 * W.m[7]; JW.m[8] = e2.j2 * W.m[8];
            JW.m[1] = JW.m[3]; JW.m[2] = JW.m[6]; JW.m[5] = JW.m[7];
            // U = JW A1
            M3 U;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                U.m[3 * i] = JW.m[3 * i] * e1.e;
                U.m[3 * i + 1] = JW.m[3 * i + 1] * e1.e;
                U.m[3 * i + 2] = JW.m[3 * i] * e1.a + JW.m[3 * i + 1] * e1.b + JW.m[3 * i + 2] * e1.e;
            }
            // J12 = J1 + A1^T U
            J[0] = e1.j0 + e1.e * U.m[0];
            J[1] = e1.e * U.m[3];
            J[2] = e1.j1 + e1.e * U.m[4];
            J[3] = e1.a * U.m[0] + e1.b * U.m[3] + e1.e * U.m[6];
            J[4] = e1.a * U.m[1] + e1.b * U.m[4] + e1.e * U.m[7];
            J[5] = e1.j2 + (e1.a * U.m[2] + e1.b * U.m[5] + e1.e * U.m[8]);
            // G = A2 W
            M3 G;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                G.m[j] = e2.e * W.m[j] + e2.a * W.m[6 + j];
                G.m[3 + j] = e2.e * W.m[3 + j] + e2.b * W.m[6 + j];
                G.m[6 + j] = e2.e * W.m[6 + j];
            }
            const M3 T = m3_mul(G, Cm);
            // C12 = T A2^T + C2
            C[0] = e2.C[0] + (T.m[0] * e2.e + T.m[2] * e2.a);
            C[1] = e2.C[1] + (T.m[3] * e2.e + T.m[5] * e2.a);
            C[2] = e2.C[2] + (T.m[4] * e2.e + T.m[5] * e2.b);
            C[3] = e2.C[3] + (T.m[6] * e2.e + T.m[8] * e2.a);
            C[4] = e2.C[4] + (T.m[7] * e2.e + T.m[8] * e2.b);
            C[5] = e2.C[5] + T.m[8] * e2.e;
            // A12 = G A1
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                A.m[3 * i] = G.m[3 * i] * e1.e;
                A.m[3 * i + 1] = G.m[3 * i + 1] * e1.e;
                A.m[3 * i + 2] = G.m[3 * i] * e1.a + G.m[3 * i + 1] * e1.b + G.m[3 * i + 2] * e1.e;
            }
        }
        // --- suffix scan over the lanes ---
#pragma unroll
        for (int d = 1; d < GW; d *= 2) {
            M3 A2;
            double C2[6], J2[6];
            const bool more = 2 * d < GW;
            if (more) {
                double ee[21];
#pragma unroll
                for (int i = 0; i < 9; ++i) ee[i] = A.m[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) { ee[9 + i] = C[i]; ee[15 + i] = J[i]; }
                x.down(ee, d);
#pragma unroll
                for (int i = 0; i < 9; ++i) A2.m[i] = ee[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) { C2[i] = ee[9 + i]; J2[i] = ee[15 + i]; }
            } else {   // the last level only needs J of the right-hand segment
#pragma unroll
                for (int i = 0; i < 6; ++i) J2[i] = J[i];
                x.down(J2, d);
#pragma unroll
                for (int i = 0; i < 9; ++i) A2.m[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) C2[i] = 0.0;
            }
            // (one OCP per wavefront: unguarded — an element that reaches the terminal stage has
            //  A = 0 and C = 0 exactly, so W = I and J12 = J1 whatever the missing partner returns)
            if (!GUARD || k + d <= Nl) combine_full(A, C, J, A2, C2, J2, more);
        }
        // --- J = P_{2k}.  Stage 2k+1 needs P_{2k+2} (the next lane's J), stage 2k needs P_{2k+1}:
        //     one sequential Riccati step inside the lane ---
        unpark_bounds();
        double Gam[2][4], gam[2][4];
        barrier_terms(Gam, gam);   // (same bits as before the scan; recomputed rather than kept alive across it)
        {
            double jn[6] = {J[0], J[1], J[2], J[3], J[4], J[5]};
            x.next(jn);
            Stg& t1 = s[1];
            Stg& t0 = s[0];
            if (t1.mid) { t1.P.p00 = jn[0]; t1.P.p10 = jn[1]; t1.P.p11 = jn[2]; t1.P.p20 = jn[3]; t1.P.p21 = jn[4]; t1.P.p22 = jn[5]; }
            else t1.P = zero_P();
            Gain g1;
            gain(t1.P, t1.ab, Gam[1][0], Gam[1][1], true, g1);
            set_gain(t1, g1);
            Sym3 Pm = riccati_step(t1.P, t1.ab, Gam[1][2], Gam[1][3], g1);
            if (!t1.mid) Pm = t1.isN ? terminal_P() : zero_P();
            t0.P = t0.mid ? Pm : zero_P();
            Gain g0;
            gain(t0.P, t0.ab, Gam[0][0], Gam[0][1], true, g0);
            set_gain(t0, g0);
        }
        // --- affine right-hand side ---
        double hup[2][2], hxp[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hup[e][0] = s[e].rg[0] + gam[e][0]; hup[e][1] = s[e].rg[1] + gam[e][1];
            hxp[e][0] = s[e].rg[2] + gam[e][2]; hxp[e][1] = s[e].rg[3] + gam[e][3];
        }
        rhs_scan<true>(hup, hxp);
    }

    // p_j = (hxp + K^T hup) + Phi^T (q + p_{j+1}),  kff = -(Huu)^-1 (hup + B^T (q + p_{j+1})),  q = P_{j+1} rb_j.
    // Lane level: R_k := p_{2k+2} = st_{k+1} + Psi_{k+1}^T R_{k+1} with Psi_k = Phi_{2k+1} Phi_{2k} and
    // st_k = s_{2k} + Phi_{2k}^T s_{2k+1}; the products M^(d)_k = Psi_k Psi_{k-1} .. Psi_{k-d+1} are formed
    // by the FIRST right-hand side after a factorisation and parked in LDS for the other scans.
    template <bool FIRST>
    __device__ __forceinline__ void rhs_scan(const double hup[2][2], const double hxp[2][2]) {
        M3 Fz[2];
        double sv[2][3], q[2][3];
        const Opq o = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
            const Sym3& P = t.P;
            q[e][0] = fma(P.p00, t.rb[0], fma(P.p10, t.rb[1], fma(P.p20, t.rb[2], o.oz)));
            q[e][1] = fma(P.p10, t.rb[0], fma(P.p11, t.rb[1], fma(P.p21, t.rb[2], o.oz)));
            q[e][2] = fma(P.p20, t.rb[0], fma(P.p21, t.rb[1], fma(P.p22, t.rb[2], o.oz)));
            const M3 F = closed_loop(t, o);
            if (t.mid) {
                double r[3];
                m3_tmulv(F, q[e], r);
                sv[e][0] = (hxp[e][0] + (t.K00 * hup[e][0] + t.K10 * hup[e][1])) + r[0];
                sv[e][1] = (hxp[e][1] + (t.K01 * hup[e][0] + t.K11 * hup[e][1])) + r[1];
                sv[e][2] = (t.rg[4] + (t.K02 * hup[e][0] + t.K12 * hup[e][1])) + r[2];
            } else {
                sv[e][0] = t.isN ? t.rg[2] : 0.0; sv[e][1] = t.isN ? t.rg[3] : 0.0; sv[e][2] = t.isN ? t.rg[4] : 0.0;
            }
            Fz[e] = F;
            if (!t.mid || t.is0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) Fz[e].m[i] = 0.0;
            }
        }
        double r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = fma3(sv[0][i], Fz[0].m[i], sv[1][0], Fz[0].m[3 + i], sv[1][1], Fz[0].m[6 + i], sv[1][2]);
        x.next(r);
        if (k >= Nl) { r[0] = r[1] = r[2] = 0.0; }
        M3 M;
        if (FIRST) {
            M = m3_mul(Fz[1], Fz[0]);
            mc_store(0, M);
        }
        int lvl = 0;
#pragma unroll
        for (int d = 1; d < GW; d *= 2, ++lvl) {
            const M3 Md = FIRST ? M : (!CACHE0 && lvl == 0) ? m3_mul(Fz[1], Fz[0]) : mc_load(lvl);
            double cc[3];
            m3_tmulv(Md, r, cc);
            x.down(cc, d);
            if (FIRST && 2 * d < GW) {
                M3 M1 = M;
                x.up(M1.m, d);
                if (!GUARD || k >= d) M = m3_mul(M, M1);
                mc_store(lvl + 1, M);
            }
            if (k + d < Nl) { r[0] += cc[0]; r[1] += cc[1]; r[2] += cc[2]; }
        }
        // r = p_{2k+2}; the odd stage's p_{2k+1} inside the lane
        double pm[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pm[i] = fma3(sv[1][i], Fz[1].m[i], r[0], Fz[1].m[3 + i], r[1], Fz[1].m[6 + i], r[2]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
            if (t.mid) {
                const double p0 = e == 0 ? pm[0] : r[0], p1 = e == 0 ? pm[1] : r[1], p2 = e == 0 ? pm[2] : r[2];
                t.pv[0] = p0; t.pv[1] = p1; t.pv[2] = p2;
                const AB8& ab = t.ab;
                const double w0 = q[e][0] + p0, w1 = q[e][1] + p1, w2 = q[e][2] + p2;
                const double hu0 = hup[e][0] + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
                const double hu1 = hup[e][1] + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
                solve2(t.id0, t.l10, t.id1, hu0, hu1, t.kff[0], t.kff[1]);
            }
        }
    }

    // dx_{j+1} = Phi dx_j + (rb + B kff), dx_0 = 0;  du = kff + K dx.  Lane level:
    // X_k := dx_{2k+2} = Psi_k X_{k-1} + (Phi_{2k+1} v_{2k} + v_{2k+1}).
    __device__ __forceinline__ void fwd_scan(double nx[2][3], double dl[2][5]) const {
        double v[2][3];
        M3 Fz[2];
        const Opq o = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            const AB8& ab = t.ab;
            v[e][0] = t.rb[0] + (ab.B00 * t.kff[0] + ab.B01 * t.kff[1]);
            v[e][1] = t.rb[1] + (ab.B10 * t.kff[0] + ab.B11 * t.kff[1]);
            v[e][2] = t.rb[2] + (ab.B20 * t.kff[0] + ab.B21 * t.kff[1]);
            if (!t.mid) { v[e][0] = v[e][1] = v[e][2] = 0.0; }
            Fz[e] = closed_loop_z(t, o);
        }
        double X[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) X[i] = fma3(v[1][i], Fz[1].m[3 * i], v[0][0], Fz[1].m[3 * i + 1], v[0][1], Fz[1].m[3 * i + 2], v[0][2]);
        int lvl = 0;
#pragma unroll
        for (int d = 1; d < GW; d *= 2, ++lvl) {
            double v1[3] = {X[0], X[1], X[2]};
            x.up(v1, d);
            const M3 Md = (!CACHE0 && lvl == 0) ? m3_mul(Fz[1], Fz[0]) : mc_load(lvl);
            if (!GUARD || (k >= d && k <= Nl)) {
#pragma unroll
                for (int i = 0; i < 3; ++i) X[i] = fma3(X[i], Md.m[3 * i], v1[0], Md.m[3 * i + 1], v1[1], Md.m[3 * i + 2], v1[2]);
            }
        }
        double dxa[3] = {X[0], X[1], X[2]};
        x.prev(dxa);
        if (k == 0) { dxa[0] = dxa[1] = dxa[2] = 0.0; }
        double dxb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dxb[i] = fma3(v[0][i], Fz[0].m[3 * i], dxa[0], Fz[0].m[3 * i + 1], dxa[1], Fz[0].m[3 * i + 2], dxa[2]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            const double* dx = e == 0 ? dxa : dxb;
            if (t.mid) {
                dl[e][0] = t.kff[0] + (t.K00 * dx[0] + t.K01 * dx[1] + t.K02 * dx[2]);
                dl[e][1] = t.kff[1] + (t.K10 * dx[0] + t.K11 * dx[1] + t.K12 * dx[2]);
            }
            if (t.act) { dl[e][2] = dx[0]; dl[e][3] = dx[1]; dl[e][4] = dx[2]; }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { nx[0][q] = dxb[q]; nx[1][q] = X[q]; }
    }

    // ---- BW: corrected complementarity -> new right-hand side, backward (Wave::bw) ----
    template <bool US>
    __device__ __forceinline__ void bw(int mode, double sigma_mu) {
        double hup[2][2], hxp[2][2];
        const Opq o = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            double gam[4], dza[4];
            get_dza(e, dza);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Con q;
                con_cached(t, j, q, o);
                con_correct(q, mode, dza[j], sigma_mu * bf(t, j));
                gam[j] = con_gamma(q);   // exactly 0 without a bound
            }
            hup[e][0] = t.rg[0] + gam[0]; hup[e][1] = t.rg[1] + gam[1];
            hxp[e][0] = t.rg[2] + gam[2]; hxp[e][1] = t.rg[3] + gam[3];
        }
        if (US) rhs_scan<false>(hup, hxp);
        else {
            const double Gam0[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            back_seq_ipm(Gam0, hup, hxp, FACT_NONE);
        }
    }

    // ---- FW: forward substitution, step lengths and the mu_aff sums, both stages (Wave::fw) ----
    struct FwdOut {
        double alpha, s1, s2;
        bool lin_itref, lin_lq, lin_scan;   // refinement performed / residual beyond HPIPM's LQ switch / scan step declined
        bool redo_centering;
        bool accepted;
    };
    // `pre`: (sequential predictor only) the forward sweep has been made already — by the LQ test on the
    // affine direction, direction() — and its result is passed in
    struct Sweep { double nx[2][3], dl[2][5]; };
    template <bool FINAL, bool US>
    __device__ __forceinline__ void fw(int mode, double sigma_mu, FwdOut& o, double musum = 0.0, double mu_aff0 = 0.0,
                                       bool may_retry = false, const Sweep* pre = nullptr) {
        double nx[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        double dl[2][5] = {{0.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0}};
        if (US) fwd_scan(nx, dl);
        else if (pre) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int q = 0; q < 3; ++q) nx[e][q] = pre->nx[e][q];
#pragma unroll
                for (int q = 0; q < 5; ++q) dl[e][q] = pre->dl[e][q];
            }
        } else fwd_seq_ipm(nx, dl);
        double ginv = 1.0;
        double dn = 1.0, dd = -1.0;
        double s1 = 0.0, s2 = 0.0, lin_g = 0.0;
        double dlam_net[2][4];
        const Opq oq = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
            double dza[4] = {0.0, 0.0, 0.0, 0.0};
            if (FINAL) get_dza(e, dza);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Con q;
                con_cached(t, j, q, oq);
                if (FINAL) con_correct(q, mode, dza[j], sigma_mu * bf(t, j));
                Step sp;
                con_step(q, dl[e][j], sp);
                const double gl = (sp.dt_l * q.ti_l) * bf(t, j), gu = (sp.dt_u * q.ti_u) * bf(t, j);
                ginv = fmax(ginv, -gl);
                ginv = fmax(ginv, -gu);
                if (FINAL) {
                    if (dn * sp.dl_l < q.lam_l * dd) { dn = q.lam_l; dd = sp.dl_l; }
                    if (dn * sp.dl_u < q.lam_u * dd) { dn = q.lam_u; dd = sp.dl_u; }
                } else {
                    ginv = fmax(ginv, 1.0 + gl);
                    ginv = fmax(ginv, 1.0 + gu);
                }
                const double cl = q.lam_l * sp.dt_l + q.t_l * sp.dl_l, cu = q.lam_u * sp.dt_u + q.t_u * sp.dl_u;
                s1 += cl;
                s1 += cu;
                s2 += sp.dl_l * sp.dt_l;
                s2 += sp.dl_u * sp.dt_u;
                dlam_net[e][j] = sp.dl_u - sp.dl_l;
            }
        }
        double dpi[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if (FINAL) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const Stg& t = s[e];
                const Sym3& P = t.P;
                dpi[e][0] = t.mid ? P.p00 * nx[e][0] + P.p10 * nx[e][1] + P.p20 * nx[e][2] + t.pv[0] : 0.0;
                dpi[e][1] = t.mid ? P.p10 * nx[e][0] + P.p11 * nx[e][1] + P.p21 * nx[e][2] + t.pv[1] : 0.0;
                dpi[e][2] = t.mid ? P.p20 * nx[e][0] + P.p21 * nx[e][1] + P.p22 * nx[e][2] + t.pv[2] : 0.0;
            }
            double dpm0[3] = {dpi[1][0], dpi[1][1], dpi[1][2]};   // -> dpi of stage 2k - 1
            x.prev(dpm0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const Stg& t = s[e];
                const AB8& ab = t.ab;
                const double* dp = dpi[e];
                const double* dpm = e == 0 ? dpm0 : dpi[0];
                const double* d = dl[e];
                if (t.mid) {
                    // dynamics: the next stage's dx is what the scan delivered to it (nx), checked
                    // against this stage's own step
                    // (scaled so that ONE threshold serves both kinds of rows: consts.hpp kScanTolLin)
                    const double ds = c.scan_dyn_scale;
                    lin_g = amax(lin_g, ds * (t.rb[0] + (ab.B00 * d[0] + ab.B01 * d[1] + (d[2] + ab.a * d[4])) - nx[e][0]));
                    lin_g = amax(lin_g, ds * (t.rb[1] + (ab.B10 * d[0] + ab.B11 * d[1] + (d[3] + ab.b * d[4])) - nx[e][1]));
                    lin_g = amax(lin_g, ds * (t.rb[2] + (ab.B20 * d[0] + ab.B21 * d[1] + d[4]) - nx[e][2]));
                    const double lg0 = c.wg[0] * d[0] + t.rg[0] + dlam_net[e][0] + (ab.B00 * dp[0] + ab.B10 * dp[1] + ab.B20 * dp[2]);
                    const double lg1 = c.wg[1] * d[1] + t.rg[1] + dlam_net[e][1] + (ab.B01 * dp[0] + ab.B11 * dp[1] + ab.B21 * dp[2]);
                    lin_g = amax(lin_g, lg0); lin_g = amax(lin_g, lg1);
                    if (!t.is0) {
                        const double lg2 = c.wg[2] * d[2] + t.rg[2] + dlam_net[e][2] - dpm[0] + dp[0];
                        const double lg3 = c.wg[3] * d[3] + t.rg[3] + dlam_net[e][3] - dpm[1] + dp[1];
                        const double lg4 = c.wg[4] * d[4] + t.rg[4] - dpm[2] + (ab.a * dp[0] + ab.b * dp[1] + dp[2]);
                        lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
                    }
                } else if (t.isN) {
                    const double lg2 = c.we[0] * d[2] + t.rg[2] - dpm[0];
                    const double lg3 = c.we[1] * d[3] + t.rg[3] - dpm[1];
                    const double lg4 = c.we[2] * d[4] + t.rg[4] - dpm[2];
                    lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
                }
            }
        } else {
            set_dza(0, dl[0]); set_dza(1, dl[1]);
        }
        double am[1] = {FINAL ? fmax(ginv, -(dd * rcp(dn))) : ginv}, as[2] = {s1, s2};
        x.rmax(am);
        x.rsum(as);
        o.alpha = rcp(am[0]); o.s1 = as[0]; o.s2 = as[1];
        // The scan step is taken only if the dynamics rows of the linearised KKT system are met to 1e-8
        // (where HPIPM's refinement loop would stop at once) and the stationarity rows to
        // c.scan_tol_lin: 1e-8 with the reference's weights, more with heavier ones, never more than a
        // quarter of HPIPM's 1e-6 (consts.hpp kScanTolLin; lin_g holds the larger of the two norms, the
        // dynamics rows scaled to the common threshold).  The slack and complementarity
        // rows are met to rounding by construction (dt and dlam are computed FROM those rows, lane by
        // lane, by the same code on either path; adding their norms here was measured at +6 % kernel
        // time) and are checked where refinement can act on them, on the sequential path.
        // A declined instance is solved again by the sequential kernel, which refines as HPIPM does.
        o.lin_itref = false;
        o.lin_lq = FINAL ? x.gany(!(lin_g < K::lq_tol)) : false;
        o.lin_scan = FINAL ? x.gany(!(lin_g < c.scan_tol_lin)) : false;
        o.redo_centering = false;
        o.accepted = false;
        if (FINAL) {
            const double alpha = o.alpha;
            if (mode == 1) {
                const double mu_aff1 = (musum + alpha * o.s1 + (alpha * alpha) * o.s2) * nc_inv();
                o.redo_centering = mu_aff1 > MPC_IPM_COND_PRED_CORR_FAC * mu_aff0;
            }
            o.accepted = !o.redo_centering && !(may_retry && o.lin_scan);
            if (o.accepted) {
                double a_upd = alpha;
                if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * MPC_IPM_STEP_SCALE_LO + a_upd * MPC_IPM_STEP_SCALE_HI);
                double pi[2][3];
                get_pi(pi);
                const Opq o2 = opq();
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    Stg& t = s[e];
                    double dza[4];
                    get_dza(e, dza);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (the steps are recomputed, same arithmetic as above, instead of being
                        //  kept alive across the reductions: 32 doubles of register pressure)
                        Con q;
                        con_cached(t, j, q, o2);
                        con_correct(q, mode, dza[j], sigma_mu * bf(t, j));
                        Step sp;
                        con_step(q, dl[e][j], sp);
                        const double aj = a_upd * bf(t, j), lmin = K::lam_min * bf(t, j);
                        double ll = t.bn.lam[j] + aj * sp.dl_l, lu = t.bn.lam[4 + j] + aj * sp.dl_u;
                        double tl = t.bn.t[j] + aj * sp.dt_l, tu = t.bn.t[4 + j] + aj * sp.dt_u;
                        if (ll < lmin) ll = lmin;
                        if (lu < lmin) lu = lmin;
                        if (tl < K::t_min) tl = K::t_min;
                        if (tu < K::t_min) tu = K::t_min;
                        t.bn.lam[j] = ll; t.bn.lam[4 + j] = lu; t.bn.t[j] = tl; t.bn.t[4 + j] = tu;
                        t.ti[j] = rcp(tl); t.ti[4 + j] = rcp(tu);
                    }
#pragma unroll
                    for (int q = 0; q < 5; ++q) t.z[q] = t.z[q] + a_upd * dl[e][q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) pi[e][q] = pi[e][q] + a_upd * dpi[e][q];
                }
                set_pi(pi);
            }
        }
    }

    // ---- the corrector's forward pass on the SEQUENTIAL path, with HPIPM's iterative refinement
    //      of the search direction (OCP_QP_IPM_DELTA_STEP with itref_corr_max = 2, BALANCE mode:
    //      acados_solver_mobile_robot.c:670; HPIPM d_ocp_qp_ipm.c, the itref_corr_max loop).
    // The step (dz, dpi, dlam, dt) is held explicitly; while the residual of the linearised KKT
    // system exceeds (res_g_max, res_b_max, res_d_max, res_m_max) the system is solved again for that
    // residual with the same factorisation and the solution added, at most twice; then the step
    // lengths are recomputed and the step is taken.  Cold code: only instances whose scan step was
    // declined, and the WAVE2_SEQ model, come here.
    struct Resid { double g[2][5], b[2][3], dl[2][4], du[2][4], ml[2][4], mu[2][4]; };
    __device__ __forceinline__ void step_lengths(const Con (&cq)[2][4], const Step (&sp)[2][4], FwdOut& o) const {
        double ginv = 1.0, dn = 1.0, dd = -1.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Con& q = cq[e][j];
                const Step& p = sp[e][j];
                const double f = bf(s[e], j);
                ginv = fmax(ginv, -((p.dt_l * q.ti_l) * f));
                ginv = fmax(ginv, -((p.dt_u * q.ti_u) * f));
                if (dn * p.dl_l < q.lam_l * dd) { dn = q.lam_l; dd = p.dl_l; }
                if (dn * p.dl_u < q.lam_u * dd) { dn = q.lam_u; dd = p.dl_u; }
                s1 += q.lam_l * p.dt_l + q.t_l * p.dl_l;
                s1 += q.lam_u * p.dt_u + q.t_u * p.dl_u;
                s2 += p.dl_l * p.dt_l;
                s2 += p.dl_u * p.dt_u;
            }
        double am[1] = {fmax(ginv, -(dd * rcp(dn)))}, as[2] = {s1, s2};
        x.rmax(am);
        x.rsum(as);
        o.alpha = rcp(am[0]); o.s1 = as[0]; o.s2 = as[1];
    }
    // dpi of both stages from the value function: dpi_j = P_{j+1} dx_{j+1} + p_{j+1}
    __device__ __forceinline__ void step_pi(const double nx[2][3], double dpi[2][3]) const {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            const Sym3& P = t.P;
            dpi[e][0] = t.mid ? P.p00 * nx[e][0] + P.p10 * nx[e][1] + P.p20 * nx[e][2] + t.pv[0] : 0.0;
            dpi[e][1] = t.mid ? P.p10 * nx[e][0] + P.p11 * nx[e][1] + P.p21 * nx[e][2] + t.pv[1] : 0.0;
            dpi[e][2] = t.mid ? P.p20 * nx[e][0] + P.p21 * nx[e][1] + P.p22 * nx[e][2] + t.pv[2] : 0.0;
        }
    }
    // OCP_QP_RES_COMPUTE_LIN: residual of the linearised KKT system at the step; n = inf-norms (g, b, d, m)
    __device__ __forceinline__ void lin_residual(const Con (&cq)[2][4], const Step (&sp)[2][4], const double dz[2][5],
                                                 const double dpi[2][3], Resid& r, double n[4]) const {
        double dpm0[3] = {dpi[1][0], dpi[1][1], dpi[1][2]};   // -> dpi of stage 2k - 1
        double dxn1[3] = {dz[0][2], dz[0][3], dz[0][4]};      // -> dx of stage 2k + 2
        x.prev(dpm0);
        x.next(dxn1);
        n[0] = n[1] = n[2] = n[3] = 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            const AB8& ab = t.ab;
            const double* dp = dpi[e];
            const double* dpm = e == 0 ? dpm0 : dpi[0];
            const double* dxn = e == 0 ? &dz[1][2] : dxn1;
            const double* d = dz[e];
#pragma unroll
            for (int q = 0; q < 5; ++q) r.g[e][q] = 0.0;
            r.b[e][0] = r.b[e][1] = r.b[e][2] = 0.0;
            if (t.mid) {
                r.g[e][0] = c.wg[0] * d[0] + t.rg[0] + (sp[e][0].dl_u - sp[e][0].dl_l) + (ab.B00 * dp[0] + ab.B10 * dp[1] + ab.B20 * dp[2]);
                r.g[e][1] = c.wg[1] * d[1] + t.rg[1] + (sp[e][1].dl_u - sp[e][1].dl_l) + (ab.B01 * dp[0] + ab.B11 * dp[1] + ab.B21 * dp[2]);
                if (!t.is0) {
                    r.g[e][2] = c.wg[2] * d[2] + t.rg[2] + (sp[e][2].dl_u - sp[e][2].dl_l) - dpm[0] + dp[0];
                    r.g[e][3] = c.wg[3] * d[3] + t.rg[3] + (sp[e][3].dl_u - sp[e][3].dl_l) - dpm[1] + dp[1];
                    r.g[e][4] = c.wg[4] * d[4] + t.rg[4] - dpm[2] + (ab.a * dp[0] + ab.b * dp[1] + dp[2]);
                }
                r.b[e][0] = t.rb[0] + (ab.B00 * d[0] + ab.B01 * d[1] + (d[2] + ab.a * d[4])) - dxn[0];
                r.b[e][1] = t.rb[1] + (ab.B10 * d[0] + ab.B11 * d[1] + (d[3] + ab.b * d[4])) - dxn[1];
                r.b[e][2] = t.rb[2] + (ab.B20 * d[0] + ab.B21 * d[1] + d[4]) - dxn[2];
            } else if (t.isN) {
                r.g[e][2] = c.we[0] * d[2] + t.rg[2] - dpm[0];
                r.g[e][3] = c.we[1] * d[3] + t.rg[3] - dpm[1];
                r.g[e][4] = c.we[2] * d[4] + t.rg[4] - dpm[2];
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) n[0] = amax(n[0], r.g[e][q]);
#pragma unroll
            for (int q = 0; q < 3; ++q) n[1] = amax(n[1], r.b[e][q]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Con& q = cq[e][j];
                const Step& p = sp[e][j];
                const double f = bf(t, j);
                r.dl[e][j] = f * ((q.rd_l + p.dt_l) - d[j]);
                r.du[e][j] = f * ((q.rd_u + p.dt_u) + d[j]);
                r.ml[e][j] = f * (q.rm_l + (q.lam_l * p.dt_l + q.t_l * p.dl_l));
                r.mu[e][j] = f * (q.rm_u + (q.lam_u * p.dt_u + q.t_u * p.dl_u));
                n[2] = amax(n[2], r.dl[e][j]); n[2] = amax(n[2], r.du[e][j]);
                n[3] = amax(n[3], r.ml[e][j]); n[3] = amax(n[3], r.mu[e][j]);
            }
        }
        double nn[4] = {n[0], n[1], n[2], n[3]};
        x.rmax(nn);
        n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2]; n[3] = nn[3];
    }
    __device__ __forceinline__ void fw_final_seq(int mode, double sigma_mu, FwdOut& o, double musum, double mu_aff0) {
        double nx[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        double dz[2][5] = {{0.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0}};
        fwd_seq_ipm(nx, dz);
        Con cq[2][4];
        Step sp[2][4];
        const Opq oq = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double dza[4];
            get_dza(e, dza);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                con_cached(s[e], j, cq[e][j], oq);
                con_correct(cq[e][j], mode, dza[j], sigma_mu * bf(s[e], j));
                con_step(cq[e][j], dz[e][j], sp[e][j]);
            }
        }
        double dpi[2][3];
        step_pi(nx, dpi);
        step_lengths(cq, sp, o);
        o.lin_itref = o.lin_lq = o.lin_scan = false;
        o.redo_centering = false;
        o.accepted = false;
        if (mode == 1) {
            const double mu_aff1 = (musum + o.alpha * o.s1 + (o.alpha * o.alpha) * o.s2) * nc_inv();
            o.redo_centering = mu_aff1 > MPC_IPM_COND_PRED_CORR_FAC * mu_aff0;
            if (o.redo_centering) return;
        }
        // iterative refinement
        int itref = 0;
#pragma nounroll
        for (; itref < MPC_IPM_ITREF_CORR_MAX; ++itref) {
            Resid r;
            double n[4];
            lin_residual(cq, sp, dz, dpi, r, n);
            if (n[0] < K::res_g_max && n[1] < K::res_b_max && n[2] < K::res_d_max && n[3] < K::res_m_max) break;
            o.lin_itref = true;
            // OCP_QP_SOLVE_KKT_STEP with the residual as right-hand side
            Con rq[2][4];
            double g5[2][5];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int q = 0; q < 5; ++q) g5[e][q] = r.g[e][q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rq[e][j] = cq[e][j];
                    rq[e][j].rd_l = r.dl[e][j]; rq[e][j].rd_u = r.du[e][j];
                    rq[e][j].rm_l = r.ml[e][j]; rq[e][j].rm_u = r.mu[e][j];
                    g5[e][j] += con_gamma(rq[e][j]);
                }
            }
            const double Gam0[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            back_seq(Gam0, g5, r.b, FACT_NONE);
            double nxr[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            double dzr[2][5] = {{0.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0}};
            fwd_seq(r.b, nxr, dzr);
            double dpr[2][3];
            step_pi(nxr, dpr);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int q = 0; q < 5; ++q) dz[e][q] += dzr[e][q];
#pragma unroll
                for (int q = 0; q < 3; ++q) dpi[e][q] += dpr[e][q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Step pr;
                    con_step(rq[e][j], dzr[e][j], pr);
                    sp[e][j].dt_l += pr.dt_l; sp[e][j].dt_u += pr.dt_u;
                    sp[e][j].dl_l += pr.dl_l; sp[e][j].dl_u += pr.dl_u;
                }
            }
        }
        if (itref > 0) step_lengths(cq, sp, o);
        // UPDATE_VAR_QP
        o.accepted = true;
        double a_upd = o.alpha;
        if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * MPC_IPM_STEP_SCALE_LO + a_upd * MPC_IPM_STEP_SCALE_HI);
        double pi[2][3];
        get_pi(pi);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Step& p = sp[e][j];
                const double aj = a_upd * bf(t, j), lmin = K::lam_min * bf(t, j);
                double ll = t.bn.lam[j] + aj * p.dl_l, lu = t.bn.lam[4 + j] + aj * p.dl_u;
                double tl = t.bn.t[j] + aj * p.dt_l, tu = t.bn.t[4 + j] + aj * p.dt_u;
                if (ll < lmin) ll = lmin;
                if (lu < lmin) lu = lmin;
                if (tl < K::t_min) tl = K::t_min;
                if (tu < K::t_min) tu = K::t_min;
                t.bn.lam[j] = ll; t.bn.lam[4 + j] = lu; t.bn.t[j] = tl; t.bn.t[4 + j] = tu;
                t.ti[j] = rcp(tl); t.ti[4 + j] = rcp(tu);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) t.z[q] = t.z[q] + a_upd * dz[e][q];
#pragma unroll
            for (int q = 0; q < 3; ++q) pi[e][q] = pi[e][q] + a_upd * dpi[e][q];
        }
        set_pi(pi);
    }

    // HPIPM's lq_fact = 1 test (OCP_QP_IPM_DELTA_STEP): the residual of the linearised KKT system at the
    // AFFINE direction the regular factorisation has just produced; any of its four inf-norms above
    // 1e-5 -> refactorise with LQ and stay with LQ for the rest of this QP.  Sequential path only.
    // (`w`: the affine direction's forward sweep, which the predictor then uses as it is)
    __device__ __forceinline__ void affine_sweep_seq(Sweep& w) const {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int q = 0; q < 3; ++q) w.nx[e][q] = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) w.dl[e][q] = 0.0;
        }
        fwd_seq_ipm(w.nx, w.dl);
    }
    __device__ __forceinline__ bool lq_wanted_seq(const Sweep& w) const {
        const double (&nx)[2][3] = w.nx;
        const double (&dz)[2][5] = w.dl;
        Con cq[2][4];
        Step sp[2][4];
        const Opq oq = opq();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                con_cached(s[e], j, cq[e][j], oq);
                con_step(cq[e][j], dz[e][j], sp[e][j]);
            }
        double dpi[2][3];
        step_pi(nx, dpi);
        Resid r;
        double n[4];
        lin_residual(cq, sp, dz, dpi, r, n);
        return !(fmax(fmax(n[0], n[1]), fmax(n[2], n[3])) <= K::lq_tol);
    }

    // ---- the RTI step (Wave::begin / cont / iterate) ----
    __device__ __forceinline__ double nc_inv() const { return c.nc_inv; }
    __device__ __forceinline__ void begin(State& st) {
        bu(st.musum, st.open);
        st.mu = st.musum * nc_inv();
        st.alpha = 1.0;
        st.kk = 0;
        st.diag = 0;
        st.declined = false;
        st.force_lq = false;
    }
    __device__ __forceinline__ bool cont(const State& st) const {
        return st.kk < c.iter_max && st.alpha > K::alpha_min && st.open && !st.declined;
    }
    // Mem::IPM (rti_core.hpp): a scan kernel parks the interior-point iterate of an instance it declines
    // — the declined iteration has changed nothing — and the list kernel carries on from that iteration
    // with the sequential recursions.  `slot`: the instance's record, (N + 1) stages of MPC_IPM_STAGE
    // doubles (z 5, pi 3, lam 8, t 8) + the head {iterations done, last step length, diag bits}.
    __device__ __forceinline__ void save_ipm(double* slot, const State& st) const {
        double pi[2][3];
        get_pi(pi);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Stg& t = s[e];
            if (!t.act) continue;
            double* r = slot + (size_t)(2 * k + e) * MPC_IPM_STAGE;
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = t.z[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) r[5 + q] = pi[e][q];
#pragma unroll
            for (int q = 0; q < 8; ++q) { r[8 + q] = t.bn.lam[q]; r[16 + q] = t.bn.t[q]; }
        }
        if (k == 0) {
            double* hd = slot + (size_t)(N + 1) * MPC_IPM_STAGE;
            hd[0] = (double)st.kk; hd[1] = st.alpha; hd[2] = (double)st.diag;
        }
    }
    // after linearize() + cold_start(): replace the cold iterate by the parked one and take up the count
    __device__ __forceinline__ void load_ipm(const double* slot, State& st) {
        double pi[2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Stg& t = s[e];
            const int j = 2 * k + e;
            const double* r = slot + (size_t)(j <= N ? j : N) * MPC_IPM_STAGE;
#pragma unroll
            for (int q = 0; q < 5; ++q) t.z[q] = t.act ? r[q] : t.z[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) pi[e][q] = t.mid ? r[5 + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool hb = has_bound(t, q & 3);
                const double tq = hb ? r[16 + q] : 1.0;
                t.bn.lam[q] = hb ? r[8 + q] : 0.0;
                t.bn.t[q] = tq;
                t.ti[q] = hb ? rcp(tq) : 1.0;
            }
        }
        set_pi(pi);
        bu(st.musum, st.open);
        st.mu = st.musum * nc_inv();
        const double* hd = slot + (size_t)(N + 1) * MPC_IPM_STAGE;
        st.kk = (int)hd[0]; st.alpha = hd[1]; st.diag = (int)hd[2];
    }
    // The search direction of one IPM iteration (OCP_QP_IPM_DELTA_STEP) and, if it is accepted,
    // the variable update.  US: with the parallel scans; the stationarity residual of the linear
    // system (lin_g, what HPIPM's refinement loop tests against 1e-6) is then checked against
    // 1e-8 and a direction that fails is not taken (returns false, nothing changed).
    template <bool US>
    __device__ __forceinline__ bool direction(State& st, FwdOut& fc, double& alpha) {
        const double nci = nc_inv();
        long long t0 = 0, t1 = 0;
        (void)t0; (void)t1;
        MPC_PH(0, t0, t0);   // (starts the clock: adds 0)
        FwdOut fa;
        if (US) {
            factor_now<US>();
            MPC_PH(0, t0, t1);   // factorisation scan + first right-hand side
            fw<false, US>(0, 0.0, fa);
        } else {
            Sweep aff;   // the affine direction: tested for the LQ switch, then used by the predictor
            factor_now<false>(st.force_lq);
            affine_sweep_seq(aff);
            if (!st.force_lq && lq_wanted_seq(aff)) {   // HPIPM lq_fact = 1: switch to the LQ factorisation
                st.force_lq = true;
                st.diag |= MPC_DIAG_LQ;
                factor_now<false>(true);
                affine_sweep_seq(aff);
            }
            MPC_PH(0, t0, t1);
            fw<false, false>(0, 0.0, fa, 0.0, 0.0, false, &aff);
        }
        MPC_PH(1, t1, t0);   // predictor forward pass
        const double a_aff = fa.alpha;
        const double mu_aff0 = (st.musum + a_aff * fa.s1 + (a_aff * a_aff) * fa.s2) * nci;
        const double tmp = mu_aff0 / st.mu;
        const double sigma = tmp * tmp * tmp;
        double sigma_mu = sigma * st.mu;
        if (!(sigma_mu > K::tau_min)) sigma_mu = K::tau_min;
        // corrector; conditional predictor-corrector: if mu would grow, redo it as pure centering
        int mode = 1;
#pragma nounroll
        for (;;) {
            bw<US>(mode, sigma_mu);
            MPC_PH(2, t0, t1);   // corrector right-hand side
            if (US) fw<true, true>(mode, sigma_mu, fc, st.musum, mu_aff0, true);
            else fw_final_seq(mode, sigma_mu, fc, st.musum, mu_aff0);
            MPC_PH(3, t1, t0);   // corrector forward pass, acceptance test, update
            alpha = fc.alpha;
            if (!fc.redo_centering) break;
            mode = 2;
        }
        return fc.accepted;
    }
    // One IPM iteration.  SCAN: if the scan step is declined NOTHING has changed and the instance
    // stops (st.declined): the kernel hands it to the sequential kernel (k_wave2<.., SCAN = false, ..,
    // LIST = true>), which redoes the whole QP with the sequential recursions and HPIPM's refinement.
    // The hot kernel carries no sequential code (its registers are all spoken for).
    __device__ __forceinline__ void iterate(State& st) {
        FwdOut fc;
        double alpha = 0.0;
        if (SCAN) {
            if (!direction<true>(st, fc, alpha)) {
                st.declined = true;
                return;
            }
        } else {
            direction<false>(st, fc, alpha);
        }
        if (fc.lin_itref) st.diag |= MPC_DIAG_ITREF;
        if (SCAN && fc.lin_lq) st.diag |= MPC_DIAG_LQ;   // (scan path: the corrector's residual; such a step is declined anyway)
        st.alpha = alpha;
        long long t0 = 0, t1 = 0;
        (void)t0; (void)t1;
        MPC_PH(4, t0, t0);
        bu(st.musum, st.open);
        MPC_PH(4, t0, t1);   // residuals and exit test
        st.mu = st.musum * nc_inv();
        ++st.kk;
    }
};

}  // namespace mpc
