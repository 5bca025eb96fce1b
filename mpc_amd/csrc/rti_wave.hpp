// rti_wave.hpp — the SQP-RTI step with ONE OCP PER WAVEFRONT (or per 32-/16-lane group):
// lane = shooting stage.  Device code only (gfx950).
//
// Why a second execution model next to rti_core.hpp's one-OCP-per-lane kernel: with a lane per
// OCP the 31-KB IPM workspace of every instance lives in HBM and is streamed four times per
// IPM iteration (the kernel is HBM-bound on its own workspace), a wave runs as long as its
// slowest lane, and a single instance (batch = 1) uses 1/64 of one SIMD.  Here the whole
// workspace of an instance sits in the registers of the 64 lanes (lane k holds stage k: ~90
// doubles), so an IPM iteration touches no memory at all:
//   * everything that is local to a stage (ERK4 + sensitivities, cold start, residuals,
//     complementarity algebra, step lengths, variable update) runs on all stages at once;
//   * neighbour access (pi_{k-1}, x_{k+1}) is one DPP wave shift;
//   * norms, mu and the step lengths are wavefront reductions (DPP butterfly inside a row,
//     ds_bpermute across rows);
//   * only the two Riccati recursions are sequential: the lane that owns stage k computes
//     its step under `if (lane == k)`, then P/p (backward) or dx (forward) moves one lane over
//     by a DPP shift.
// The arithmetic of every stage is rti_core.hpp's Algebra (same formulas, same order), so the
// result differs from the lane-per-OCP kernel only by the association of the three wave-wide
// sums (mu and the two mu_aff sums) — the same class of deviation as kernel vs oracle.
#pragma once
#include "rti_core.hpp"

namespace mpc {

// lane i <- lane i+1 (the highest lane keeps `old`); DPP wave_shl:1
__device__ __forceinline__ double lane_next(double v, double old = 0.0) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i-1 (lane 0 keeps `old`); DPP wave_shr:1
__device__ __forceinline__ double lane_prev(double v, double old = 0.0) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double lane_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// All-reduce over the GW lanes of a group; every lane ends with the same bits (each step
// combines a lane with its mirror partner, and the operations are commutative).
template <int GW, class Op>
__device__ __forceinline__ double group_reduce(double v, Op op) {
    v = op(v, lane_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, lane_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, lane_dpp<0x141>(v));   // row_half_mirror
    v = op(v, lane_dpp<0x140>(v));   // row_mirror
    if (GW >= 32) v = op(v, __shfl_xor(v, 16, 64));
    if (GW >= 64) v = op(v, __shfl_xor(v, 32, 64));
    return v;
}
struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };

template <int GW>
struct Wave : Algebra {
    int k, N;                    // stage of this lane, horizon
    bool act, mid, isN, is0;     // k <= N, k < N, k == N, k == 0
    // inputs and linearisation of this stage
    double yref[5], cv[3];
    AB8 ab;
    // IPM state (absolute variables, see rti_core.hpp)
    double z[5], pi[3];
    Bnd bn;
    double rg[5], rb[3];
    // Riccati: value function of stage k+1, gains of stage k
    Sym3 P;
    double pv[3], kff[2];
    double id0, l10, id1, K00, K01, K02, K10, K11, K12;
    double dza[4], dz[5], dpi[3];

    struct State {
        int kk, diag;
        double alpha, mu;
        Norms n;
    };

    __device__ __forceinline__ Wave(const Consts& c_, int stage) : Algebra(c_), k(stage), N(c_.N) {
        act = k <= N; mid = k < N; isN = k == N; is0 = k == 0;
    }
    __device__ __forceinline__ bool has_bound(int j) const { return mid && (j < 2 || !is0); }

    // ERK4 linearisation + OCP_QP_INIT_VAR cold start of this stage (Lane::prepare)
    __device__ __forceinline__ void prepare(const double xi[3], const double ui[2], const double x0bar[3]) {
        double xn[3];
        linearize_math(xi[0], xi[1], xi[2], ui[0], ui[1], ab, cv, xn);
        z[0] = mid ? ui[0] : 0.0;
        z[1] = mid ? ui[1] : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) z[2 + q] = is0 ? x0bar[q] : xi[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double tl = z[j] - c.lb[j];
            double tu = c.ub[j] - z[j];
            double zj = z[j];
            if (tl < c.thr0) {
                if (tu < c.thr0) { zj = 0.5 * (c.lb[j] + c.ub[j]); tl = c.thr0; tu = c.thr0; }
                else { tl = c.thr0; zj = c.lb[j] + c.thr0; }
            } else if (tu < c.thr0) { tu = c.thr0; zj = c.ub[j] - c.thr0; }
            if (has_bound(j)) {
                z[j] = zj;
                bn.t[j] = tl; bn.t[4 + j] = tu;
                bn.lam[j] = c.mu0 * rcp(tl); bn.lam[4 + j] = c.mu0 * rcp(tu);
            } else {
                bn.t[j] = 1.0; bn.t[4 + j] = 1.0;
                bn.lam[j] = 0.0; bn.lam[4 + j] = 0.0;
            }
        }
        pi[0] = pi[1] = pi[2] = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) dz[q] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) dza[q] = 0.0;
        dpi[0] = dpi[1] = dpi[2] = 0.0;
        pv[0] = pv[1] = pv[2] = 0.0;
        kff[0] = kff[1] = 0.0;
        id0 = l10 = id1 = K00 = K01 = K02 = K10 = K11 = K12 = 0.0;
        P.p00 = P.p10 = P.p11 = P.p20 = P.p21 = P.p22 = 0.0;
        rb[0] = rb[1] = rb[2] = 0.0;
        if (!mid) { ab.a = ab.b = ab.B00 = ab.B10 = ab.B20 = ab.B01 = ab.B11 = ab.B21 = 0.0; cv[0] = cv[1] = cv[2] = 0.0; }
    }

    // ---- BU: variable update + residuals (all stages at once) + Riccati factorisation with
    //      the affine right-hand side (sequential over stages).  Lane::sweep_bu.
    template <bool FIRST>
    __device__ __forceinline__ void bu(int mode, double sigma_mu, double alpha, Norms& n) {
        if (!FIRST) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Con q;
                con_make(j, bn.lam[j], bn.lam[4 + j], bn.t[j], bn.t[4 + j], z[j], q);
                con_correct(q, mode, dza[j], sigma_mu);
                Step s;
                con_step(q, dz[j], s);
                double ll = q.lam_l + alpha * s.dl_l, lu = q.lam_u + alpha * s.dl_u;
                double tl = q.t_l + alpha * s.dt_l, tu = q.t_u + alpha * s.dt_u;
                if (ll < c.lam_min) ll = c.lam_min;
                if (lu < c.lam_min) lu = c.lam_min;
                if (tl < c.t_min) tl = c.t_min;
                if (tu < c.t_min) tu = c.t_min;
                if (has_bound(j)) { bn.lam[j] = ll; bn.lam[4 + j] = lu; bn.t[j] = tl; bn.t[4 + j] = tu; }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) z[q] = z[q] + alpha * dz[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) pi[q] = pi[q] + alpha * dpi[q];
        }
        // --- residuals at the new point ---
        double pim[3], zxn[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            pim[q] = lane_prev(pi[q]);
            if (is0) pim[q] = 0.0;
            zxn[q] = lane_next(z[2 + q]);
        }
        rg[0] = c.wg[0] * (z[0] - yref[3]);
        rg[1] = c.wg[1] * (z[1] - yref[4]);
#pragma unroll
        for (int q = 0; q < 3; ++q) rg[2 + q] = (isN ? c.we[q] : c.wg[2 + q]) * (z[2 + q] - yref[q]) - pim[q];
        double Gam[4], gam[4];
        double pg = 0.0, pb = 0.0, pd = 0.0, pm = 0.0, ps = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_make(j, bn.lam[j], bn.lam[4 + j], bn.t[j], bn.t[4 + j], z[j], q);
            const bool hb = has_bound(j);
            if (hb) {
                rg[j] += q.lam_u - q.lam_l;
                pd = amax(pd, q.rd_l); pd = amax(pd, q.rd_u);
                pm = amax(pm, q.rm_l); pm = amax(pm, q.rm_u);
                ps += q.rm_l;
                ps += q.rm_u;
            }
            Gam[j] = hb ? con_Gamma(q) : 0.0;
            gam[j] = hb ? con_gamma(q) : 0.0;
        }
        const double rb0 = cv[0] - zxn[0] + (ab.B00 * z[0] + ab.B01 * z[1] + (z[2] + ab.a * z[4]));
        const double rb1 = cv[1] - zxn[1] + (ab.B10 * z[0] + ab.B11 * z[1] + (z[3] + ab.b * z[4]));
        const double rb2 = cv[2] - zxn[2] + (ab.B20 * z[0] + ab.B21 * z[1] + z[4]);
        rb[0] = mid ? rb0 : 0.0; rb[1] = mid ? rb1 : 0.0; rb[2] = mid ? rb2 : 0.0;
        if (mid) {
            pb = amax(pb, rb0); pb = amax(pb, rb1); pb = amax(pb, rb2);
            rg[0] += ab.B00 * pi[0] + ab.B10 * pi[1] + ab.B20 * pi[2];
            rg[1] += ab.B01 * pi[0] + ab.B11 * pi[1] + ab.B21 * pi[2];
            rg[2] += pi[0];
            rg[3] += pi[1];
            rg[4] += ab.a * pi[0] + ab.b * pi[1] + pi[2];
            pg = amax(pg, rg[0]); pg = amax(pg, rg[1]);
        }
        if (act && !is0) { pg = amax(pg, rg[2]); pg = amax(pg, rg[3]); pg = amax(pg, rg[4]); }
        n.g = group_reduce<GW>(pg, OpMax());
        n.b = group_reduce<GW>(pb, OpMax());
        n.d = group_reduce<GW>(pd, OpMax());
        n.m = group_reduce<GW>(pm, OpMax());
        n.musum = group_reduce<GW>(ps, OpSum());

        // --- Riccati factorisation + affine right-hand side, stage N-1 down to 0 ---
        Sym3 Po;
        Po.p00 = c.we[0] + c.reg_prim; Po.p10 = 0.0; Po.p11 = c.we[1] + c.reg_prim;
        Po.p20 = 0.0; Po.p21 = 0.0; Po.p22 = c.we[2] + c.reg_prim;
        double po0 = rg[2], po1 = rg[3], po2 = rg[4];   // meaningful in lane N
        for (int kc = N - 1; kc >= 0; --kc) {
            Sym3 Pi;
            Pi.p00 = lane_next(Po.p00); Pi.p10 = lane_next(Po.p10); Pi.p11 = lane_next(Po.p11);
            Pi.p20 = lane_next(Po.p20); Pi.p21 = lane_next(Po.p21); Pi.p22 = lane_next(Po.p22);
            const double p0 = lane_next(po0), p1 = lane_next(po1), p2 = lane_next(po2);
            if (k == kc) {
                P = Pi;
                const double w0 = P.p00 * rb[0] + P.p10 * rb[1] + P.p20 * rb[2] + p0;
                const double w1 = P.p10 * rb[0] + P.p11 * rb[1] + P.p21 * rb[2] + p1;
                const double w2 = P.p20 * rb[0] + P.p21 * rb[1] + P.p22 * rb[2] + p2;
                const double hu0 = rg[0] + gam[0] + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
                const double hu1 = rg[1] + gam[1] + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
                Gain g;
                gain(P, ab, Gam[0], Gam[1], kc != 0, g);
                id0 = g.id0; l10 = g.l10; id1 = g.id1;
                solve2(g.id0, g.l10, g.id1, hu0, hu1, kff[0], kff[1]);
                if (kc != 0) {
                    K00 = g.K00; K01 = g.K01; K02 = g.K02; K10 = g.K10; K11 = g.K11; K12 = g.K12;
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
                    Po.p00 = Hxx00 + (g.M00 * g.K00 + g.M10 * g.K10);
                    Po.p10 = Hxx10 + (g.M01 * g.K00 + g.M11 * g.K10);
                    Po.p11 = Hxx11 + (g.M01 * g.K01 + g.M11 * g.K11);
                    Po.p20 = Hxx20 + (g.X02 * g.K00 + g.X12 * g.K10);
                    Po.p21 = Hxx21 + (g.X02 * g.K01 + g.X12 * g.K11);
                    Po.p22 = Hxx22 + (g.X02 * g.K02 + g.X12 * g.K12);
                    const double hx0 = rg[2] + gam[2] + w0;
                    const double hx1 = rg[3] + gam[3] + w1;
                    const double hx2 = rg[4] + (a * w0 + b * w1 + w2);
                    po0 = hx0 + (g.K00 * hu0 + g.K10 * hu1);
                    po1 = hx1 + (g.K01 * hu0 + g.K11 * hu1);
                    po2 = hx2 + (g.K02 * hu0 + g.K12 * hu1);
                }
            }
        }
    }

    // ---- BW: corrected complementarity -> new right-hand side, backward.  Lane::sweep_bw.
    __device__ __forceinline__ void bw(int mode, double sigma_mu) {
        double gam[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_make(j, bn.lam[j], bn.lam[4 + j], bn.t[j], bn.t[4 + j], z[j], q);
            con_correct(q, mode, dza[j], sigma_mu);
            gam[j] = has_bound(j) ? con_gamma(q) : 0.0;
        }
        const double q0 = P.p00 * rb[0] + P.p10 * rb[1] + P.p20 * rb[2];
        const double q1 = P.p10 * rb[0] + P.p11 * rb[1] + P.p21 * rb[2];
        const double q2 = P.p20 * rb[0] + P.p21 * rb[1] + P.p22 * rb[2];
        const double hup0 = rg[0] + gam[0], hup1 = rg[1] + gam[1];
        const double hxp0 = rg[2] + gam[2], hxp1 = rg[3] + gam[3];
        double po0 = rg[2], po1 = rg[3], po2 = rg[4];   // meaningful in lane N
        for (int kc = N - 1; kc >= 0; --kc) {
            const double p0 = lane_next(po0), p1 = lane_next(po1), p2 = lane_next(po2);
            if (k == kc) {
                pv[0] = p0; pv[1] = p1; pv[2] = p2;
                const double w0 = q0 + p0, w1 = q1 + p1, w2 = q2 + p2;
                const double hu0 = hup0 + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
                const double hu1 = hup1 + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
                solve2(id0, l10, id1, hu0, hu1, kff[0], kff[1]);
                if (kc != 0) {
                    const double hx0 = hxp0 + w0;
                    const double hx1 = hxp1 + w1;
                    const double hx2 = rg[4] + (ab.a * w0 + ab.b * w1 + w2);
                    po0 = hx0 + (K00 * hu0 + K10 * hu1);
                    po1 = hx1 + (K01 * hu0 + K11 * hu1);
                    po2 = hx2 + (K02 * hu0 + K12 * hu1);
                }
            }
        }
    }

    // ---- FW: forward substitution (sequential), then step lengths and the mu_aff sums on all
    //      stages at once.  Lane::sweep_fw.
    struct FwdOut { double a_prim, a_dual, s1, s2, lin_g; };
    template <bool FINAL>
    __device__ __forceinline__ void fw(int mode, double sigma_mu, FwdOut& o) {
        double nx0 = 0.0, nx1 = 0.0, nx2 = 0.0;   // dx_{k+1}, produced by the lane of stage k
        double dl[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int kc = 0; kc < N; ++kc) {
            double dx0 = lane_prev(nx0), dx1 = lane_prev(nx1), dx2 = lane_prev(nx2);
            if (k == kc) {
                double du0 = kff[0], du1 = kff[1];
                if (kc != 0) {
                    du0 += K00 * dx0 + K01 * dx1 + K02 * dx2;
                    du1 += K10 * dx0 + K11 * dx1 + K12 * dx2;
                } else {
                    dx0 = dx1 = dx2 = 0.0;   // the stage-0 state is fixed
                }
                dl[0] = du0; dl[1] = du1; dl[2] = dx0; dl[3] = dx1; dl[4] = dx2;
                nx0 = rb[0] + (ab.B00 * du0 + ab.B01 * du1 + (dx0 + ab.a * dx2));
                nx1 = rb[1] + (ab.B10 * du0 + ab.B11 * du1 + (dx1 + ab.b * dx2));
                nx2 = rb[2] + (ab.B20 * du0 + ab.B21 * du1 + dx2);
            }
        }
        {
            const double dx0 = lane_prev(nx0), dx1 = lane_prev(nx1), dx2 = lane_prev(nx2);
            if (isN) { dl[2] = dx0; dl[3] = dx1; dl[4] = dx2; }
        }
        double a_prim = -1.0, a_dual = -1.0, s1 = 0.0, s2 = 0.0, lin_g = 0.0;
        double dlam_net[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_make(j, bn.lam[j], bn.lam[4 + j], bn.t[j], bn.t[4 + j], z[j], q);
            if (FINAL) con_correct(q, mode, dza[j], sigma_mu);
            Step s;
            con_step(q, dl[j], s);
            if (has_bound(j)) {
                if (a_dual * s.dl_l > q.lam_l) a_dual = q.lam_l * rcp(s.dl_l);
                if (a_prim * s.dt_l > q.t_l) a_prim = q.t_l * rcp(s.dt_l);
                if (a_dual * s.dl_u > q.lam_u) a_dual = q.lam_u * rcp(s.dl_u);
                if (a_prim * s.dt_u > q.t_u) a_prim = q.t_u * rcp(s.dt_u);
                s1 += q.lam_l * s.dt_l + q.t_l * s.dl_l;
                s1 += q.lam_u * s.dt_u + q.t_u * s.dl_u;
                s2 += s.dl_l * s.dt_l;
                s2 += s.dl_u * s.dt_u;
                dlam_net[j] = s.dl_u - s.dl_l;
            }
        }
        if (FINAL) {
#pragma unroll
            for (int q = 0; q < 5; ++q) dz[q] = dl[q];
            const double dp0 = mid ? P.p00 * nx0 + P.p10 * nx1 + P.p20 * nx2 + pv[0] : 0.0;
            const double dp1 = mid ? P.p10 * nx0 + P.p11 * nx1 + P.p21 * nx2 + pv[1] : 0.0;
            const double dp2 = mid ? P.p20 * nx0 + P.p21 * nx1 + P.p22 * nx2 + pv[2] : 0.0;
            dpi[0] = dp0; dpi[1] = dp1; dpi[2] = dp2;
            const double dpm0 = lane_prev(dp0), dpm1 = lane_prev(dp1), dpm2 = lane_prev(dp2);
            if (mid) {
                const double lg0 = c.wg[0] * dl[0] + rg[0] + dlam_net[0] + (ab.B00 * dp0 + ab.B10 * dp1 + ab.B20 * dp2);
                const double lg1 = c.wg[1] * dl[1] + rg[1] + dlam_net[1] + (ab.B01 * dp0 + ab.B11 * dp1 + ab.B21 * dp2);
                lin_g = amax(lin_g, lg0); lin_g = amax(lin_g, lg1);
                if (!is0) {
                    const double lg2 = c.wg[2] * dl[2] + rg[2] + dlam_net[2] - dpm0 + dp0;
                    const double lg3 = c.wg[3] * dl[3] + rg[3] + dlam_net[3] - dpm1 + dp1;
                    const double lg4 = c.wg[4] * dl[4] + rg[4] - dpm2 + (ab.a * dp0 + ab.b * dp1 + dp2);
                    lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
                }
            } else if (isN) {
                const double lg2 = c.we[0] * dl[2] + rg[2] - dpm0;
                const double lg3 = c.we[1] * dl[3] + rg[3] - dpm1;
                const double lg4 = c.we[2] * dl[4] + rg[4] - dpm2;
                lin_g = amax(lin_g, lg2); lin_g = amax(lin_g, lg3); lin_g = amax(lin_g, lg4);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) dza[q] = dl[q];
        }
        // the running values are negative ratios in (-1, 0]: the binding one is the largest
        o.a_prim = -group_reduce<GW>(a_prim, OpMax());
        o.a_dual = -group_reduce<GW>(a_dual, OpMax());
        o.s1 = group_reduce<GW>(s1, OpSum());
        o.s2 = group_reduce<GW>(s2, OpSum());
        o.lin_g = FINAL ? group_reduce<GW>(lin_g, OpMax()) : 0.0;
    }

    // ---- the RTI step (Lane::begin / cont / iterate) ----
    __device__ __forceinline__ double nc_inv() const { return 1.0 / (4 + 8 * (c.N - 1)); }
    __device__ __forceinline__ void begin(State& s) {
        bu<true>(0, 0.0, 0.0, s.n);
        s.mu = s.n.musum * nc_inv();
        s.alpha = 1.0;
        s.kk = 0;
        s.diag = 0;
    }
    __device__ __forceinline__ bool cont(const State& s) const {
        return s.kk < c.iter_max && s.alpha > c.alpha_min &&
               (s.n.g > c.res_g_max || s.n.b > c.res_b_max || s.n.d > c.res_d_max ||
                fabs(s.n.m - c.tau_min) > c.res_m_max);
    }
    __device__ __forceinline__ void iterate(State& s) {
        const double nci = nc_inv();
        FwdOut fa;
        fw<false>(0, 0.0, fa);
        const double a_aff = fa.a_prim < fa.a_dual ? fa.a_prim : fa.a_dual;
        const double mu_aff0 = (s.n.musum + a_aff * fa.s1 + (a_aff * a_aff) * fa.s2) * nci;
        const double tmp = mu_aff0 / s.mu;
        const double sigma = tmp * tmp * tmp;
        double sigma_mu = sigma * s.mu;
        if (!(sigma_mu > c.tau_min)) sigma_mu = c.tau_min;
        int mode = 1;
        bw(1, sigma_mu);
        FwdOut fc;
        fw<true>(1, sigma_mu, fc);
        double alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
        const double mu_aff1 = (s.n.musum + alpha * fc.s1 + (alpha * alpha) * fc.s2) * nci;
        if (mu_aff1 > 2.0 * mu_aff0) {   // conditional predictor-corrector: pure centering
            mode = 2;
            bw(2, sigma_mu);
            fw<true>(2, sigma_mu, fc);
            alpha = fc.a_prim < fc.a_dual ? fc.a_prim : fc.a_dual;
        }
        if (!(fc.lin_g < c.itref_tol_g)) s.diag |= MPC_DIAG_ITREF;
        if (!(fc.lin_g < c.lq_tol)) s.diag |= MPC_DIAG_LQ;
        s.alpha = alpha;
        double a_upd = alpha;
        if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * 0.99 + a_upd * 0.9999999);
        bu<false>(mode, sigma_mu, a_upd, s.n);
        s.mu = s.n.musum * nci;
        ++s.kk;
    }
};

}  // namespace mpc
