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
//   * mu and the step lengths are wavefront reductions (DPP butterfly inside a row, DPP
//     row_bcast across rows); the residual-norm tests are group ballots;
//   * the Riccati recursion and the two substitutions are parallel scans over the stages
//     (factor_scan / rhs_scan / fwd_scan below), the closed-loop products they share are kept
//     in LDS per factorisation (suffix_cached / prefix_cached); the sequential forms (lane k computes under
//     `if (lane == k)`, P/p or dx move one lane over per step) remain as the in-kernel fallback
//     of the acceptance test and as the WAVE_SEQ cross-check model.
// The per-stage arithmetic is rti_core.hpp's Algebra (same formulas, same order as the
// lane-per-OCP kernel); the scans and the wave-wide sums re-associate, which moves the results
// at the 1e-13 level — the same class of deviation as kernel vs oracle.
// Cross-lane communication goes through a policy (XWave<GW>: DPP / ds_bpermute inside one
// wavefront, GW lanes per OCP).  Horizons beyond 63 use two stages per lane (rti_wave2.hpp).
#pragma once
#include "rti_core.hpp"

namespace mpc {

// DPP moves with bound_ctrl: lanes without a source read 0, and the destination needs no
// preloaded `old` value (which costs one extra v_mov per 32-bit half).
// lane i <- lane i+1 (the highest lane of the wavefront gets 0); DPP wave_shl:1
__device__ __forceinline__ double lane_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i-1 (lane 0 gets 0); DPP wave_shr:1
__device__ __forceinline__ double lane_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double lane_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// All-reduce over the GW lanes of a group; every lane ends with the same bits (each step
// combines a lane with its mirror partner, and the operations are commutative).
__device__ __forceinline__ double lane_down(double v, int d);
// The same for a group width that is no power of two (GW = 21: three OCPs of 33..42 stages per
// wavefront, two stages per lane): the groups are not aligned to DPP rows, so the lanes accumulate
// towards the group's first lane by doubling distances and that lane's value is broadcast.
template <int GW, class Op>
__device__ __forceinline__ double group_reduce_any_width(double v, Op op) {
    const int l = (int)(threadIdx.x & 63), k = l % GW;
#pragma unroll
    for (int d = 1; d < GW; d *= 2) {
        const double o = lane_down(v, d);
        if (k + d < GW) v = op(v, o);
    }
    return __shfl(v, l - k, 64);
}
template <int GW, class Op>
__device__ __forceinline__ double group_reduce(double v, Op op) {
    if ((GW & (GW - 1)) != 0) return group_reduce_any_width<GW>(v, op);
    v = op(v, lane_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, lane_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, lane_dpp<0x141>(v));   // row_half_mirror
    v = op(v, lane_dpp<0x140>(v));   // row_mirror
    // (reading one lane per row with v_readlane and combining the four was measured 3 % slower)
    if (GW == 64) {
        // cross-row steps without LDS: row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3,
        // then the total sits in lane 63 and is broadcast through a scalar register pair
        // (measured 1.5 % faster than two ds_bpermute round trips)
        {
            const double nt = op.neutral(v);
            int lo = __builtin_amdgcn_update_dpp(__double2loint(nt), __double2loint(v), 0x142, 0xa, 0xf, false);
            int hi = __builtin_amdgcn_update_dpp(__double2hiint(nt), __double2hiint(v), 0x142, 0xa, 0xf, false);
            v = op(v, __hiloint2double(hi, lo));
        }
        {
            const double nt = op.neutral(v);
            int lo = __builtin_amdgcn_update_dpp(__double2loint(nt), __double2loint(v), 0x143, 0xc, 0xf, false);
            int hi = __builtin_amdgcn_update_dpp(__double2hiint(nt), __double2hiint(v), 0x143, 0xc, 0xf, false);
            v = op(v, __hiloint2double(hi, lo));
        }
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
        return __hiloint2double(hi, lo);
    }
    if (GW >= 32) v = op(v, __shfl_xor(v, 16, 64));
    return v;
}
// lane i <- lane i+d / lane i-d for any d (ds_bpermute); what a lane without a source gets is
// unspecified (its own value, or 0 for d = 1): the scans never use it.
// (distance 1 through DPP: measured 1.5 % faster than ds_bpermute; distance 2 as two chained DPP
//  shifts another 1.8 %; distance 4 as four: no gain)
__device__ __forceinline__ double lane_down(double v, int d) {
    return d == 1 ? lane_next(v) : d == 2 ? lane_next(lane_next(v)) : __shfl_down(v, d, 64);
}
__device__ __forceinline__ double lane_up(double v, int d) {
    return d == 1 ? lane_prev(v) : d == 2 ? lane_prev(lane_prev(v)) : __shfl_up(v, d, 64);
}

// dense 3x3 helpers for the parallel scans (row-major)
struct M3 { double m[9]; };
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ void m3_mulv(const M3& a, const double v[3], double r[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2];
}
template <class F>
__device__ __forceinline__ M3 m3_map(const M3& a, F f) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = f(a.m[i]);
    return r;
}
// transposed product: r = a^T v
__device__ __forceinline__ void m3_tmulv(const M3& a, const double v[3], double r[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = a.m[i] * v[0] + a.m[3 + i] * v[1] + a.m[6 + i] * v[2];
}
// LDS cache of the closed-loop products of rti_wave2.hpp (and its parked per-stage state):
// [slot][64 lanes] of double2 (ds_*_b128, lane-contiguous); every lane only ever reads what it
// wrote itself, so no barrier is involved.
typedef double mc_d2 __attribute__((ext_vector_type(2)));
#define MPC_MC_PAIRS 5
struct OpMax {
    __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); }
    __device__ __forceinline__ double neutral(double v) const { return v; }
};
struct OpSum {
    __device__ __forceinline__ double operator()(double a, double b) const { return a + b; }
    __device__ __forceinline__ double neutral(double) const { return 0.0; }
};

// ---- cross-lane policies ---------------------------------------------------------------
// XWave<GW>: one OCP per GW lanes of a wavefront (GW = 64, 32, 16) — DPP / ds_bpermute.
template <int GW>
struct XWave {
    static constexpr int LANES = GW;       // lanes per OCP
    static constexpr int THREADS = 64;     // workgroup size
    static constexpr int PER_BLOCK = 64 / GW;
    // closed-loop products of all scan distances kept in LDS (Wave::suffix_cached<true>): one matrix
    // per distance 1, 2, .., GW/2
    static constexpr bool MCACHE = true;
    static constexpr int MC_LEVELS = GW > 32 ? 6 : GW > 16 ? 5 : 4;
    template <int n> __device__ __forceinline__ void next(double (&v)[n]) const {
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = lane_next(v[i]);
    }
    template <int n> __device__ __forceinline__ void prev(double (&v)[n]) const {
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = lane_prev(v[i]);
    }
    template <int n> __device__ __forceinline__ void down(double (&v)[n], int d) const {
#pragma unroll
        for (int i = 0; i < n; ++i)
            v[i] = lane_down(v[i], d);
    }
    template <int n> __device__ __forceinline__ void up(double (&v)[n], int d) const {
#pragma unroll
        for (int i = 0; i < n; ++i)
            v[i] = lane_up(v[i], d);
    }
    template <int n> __device__ __forceinline__ void rmax(double (&v)[n]) const {
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = group_reduce<GW>(v[i], OpMax());
    }
    template <int n> __device__ __forceinline__ void rsum(double (&v)[n]) const {
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = group_reduce<GW>(v[i], OpSum());
    }
    __device__ __forceinline__ bool any(bool p) const { return __any(p) != 0; }   // whole wavefront
    // any lane of THIS lane's group (one ballot instead of a max-reduction)
    __device__ __forceinline__ bool gany(bool p) const {
        const unsigned long long b = __ballot(p);
        if (GW == 64) return b != 0ull;
        const int g = (int)(threadIdx.x & 63) / GW;
        return ((b >> (g * GW)) & (~0ull >> (64 - GW))) != 0ull;
    }
};

template <class X, bool SCAN>
struct Wave : Algebra {
    static constexpr int GW = X::LANES;
    X x;
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
    double ti[8];                // 1 / t of the current slacks (same bits as recomputing it in every pass)
    int wdiag = 0;               // MPC_DIAG_* bits raised inside the sweeps
    static constexpr bool seq_now = false;   // (the in-kernel sequential redo of an iteration is gone: see iterate)
    // 1.0 where this stage has input bounds (k < N) / state bounds (0 < k < N), else 0.0.  Entries
    // without a bound carry lam = 0, t = 1; scaling sigma*mu, the step length and the lam clamp by
    // these factors keeps them there and makes every contribution of such an entry an exact zero,
    // so the sweeps need no per-bound selects.
    double fbu = 0.0, fbx = 0.0;

    struct State {
        int kk, diag;
        double alpha, mu;
        double musum;   // sum of lam*t over all bounds
        bool open;      // some residual norm is above its tolerance (the norm part of HPIPM's loop condition)
        bool declined;  // SCAN: the scan step of this iteration was not accurate enough (see iterate)
    };

    __device__ __forceinline__ Wave(const Consts& c_, int stage, const X& x_)
        : Algebra(c_), x(x_), k(stage), N(c_.N) {
        act = k <= N; mid = k < N; isN = k == N; is0 = k == 0;
        fbu = mid ? 1.0 : 0.0;
        fbx = (mid && !is0) ? 1.0 : 0.0;
    }
    __device__ __forceinline__ double bf(int j) const { return j < 2 ? fbu : fbx; }
    // The cached products M^(d)_k (build_products / suffix_cached) live in registers: the kernel has
    // ~115 accumulation registers to spare, the compiler parks the 6 x 9 doubles there, and a scan
    // level fetches its matrix with 18 register copies instead of 5 LDS reads — measured against
    // the LDS cache of round 1: 65536 x N = 50 5.95 -> 5.82 ms, full SQP 299 -> 291 ms (the LDS
    // pipe is shared by the four wavefronts of a CU and these kernels wait on it), batch = 1 the same.
    M3 Mc[X::MC_LEVELS];
    __device__ __forceinline__ void mc_store(int level, const M3& M) { Mc[level] = M; }
    __device__ __forceinline__ M3 mc_load(int level) const { return Mc[level]; }
    __device__ __forceinline__ bool has_bound(int j) const { return mid && (j < 2 || !is0); }
    // Algebra::con_make with the cached reciprocals
    // Phase-local opaque constants 0.0 and 1.0 (see rti_wave2.hpp opq()): every phase recomputes
    // these cheap per-bound quantities; written with constants the compiler cannot see through, the
    // recomputations are not merged across phases (which would keep them live through the scans).
    // Same bits: fma(t, 1.0, lb) = lb + t, fma(lam, t, 0.0) = lam * t.
    struct Opq { double oz, one; };
    __device__ __forceinline__ static Opq opq() {
        Opq o;
        double a = 0.0, b = 1.0;
        asm volatile("" : "+v"(a));
        asm volatile("" : "+v"(b));
        o.oz = a; o.one = b;
        return o;
    }
    __device__ __forceinline__ void con_cached(int j, Con& q, const Opq& o) const {
        q.lam_l = bn.lam[j]; q.lam_u = bn.lam[4 + j]; q.t_l = bn.t[j]; q.t_u = bn.t[4 + j];
        q.rd_l = fma(q.t_l, o.one, c.lb[j]) - z[j];
        q.rd_u = fma(q.t_u, o.one, z[j]) - c.ub[j];
        q.rm_l = fma(q.lam_l, q.t_l, o.oz);
        q.rm_u = fma(q.lam_u, q.t_u, o.oz);
        q.ti_l = ti[j];
        q.ti_u = ti[4 + j];
    }

    // ERK4 linearisation + OCP_QP_INIT_VAR cold start of this stage (Lane::prepare)
    __device__ __forceinline__ void prepare(const double xi[3], const double ui[2], const double x0bar[3]) {
        double xn[3];
        linearize(xi, ui, xn);
        cold_start(xi, ui, x0bar);
    }
    __device__ __forceinline__ void linearize(const double xi[3], const double ui[2], double xn[3]) {
        linearize_math(xi[0], xi[1], xi[2], ui[0], ui[1], ab, cv, xn);
        if (!mid) { ab.a = ab.b = ab.B00 = ab.B10 = ab.B20 = ab.B01 = ab.B11 = ab.B21 = 0.0; cv[0] = cv[1] = cv[2] = 0.0; }
    }
    // NLP residuals of acados' full SQP (Lane::nlp_res_and_linearize, oracle nlp_residuals) at the
    // iterate (xi, ui) with the multipliers (pin, lamn) of the previous QP; needs linearize() first
    // (xn = phi(x_k, u_k)).  out = {stat, eq, ineq, comp}, identical in all lanes of the group.
    __device__ __forceinline__ void nlp_residuals(const double xi[3], const double ui[2], const double x0bar[3],
                                                  const double xn[3], const double pin[3], const double lamn[8],
                                                  double out[4]) const {
        double r_stat = 0.0, r_eq = 0.0, r_ineq = 0.0, r_comp = 0.0;
        const double zz[5] = {mid ? ui[0] : 0.0, mid ? ui[1] : 0.0, xi[0], xi[1], xi[2]};
        double g[5];
        g[0] = c.wg[0] * (zz[0] - yref[3]);
        g[1] = c.wg[1] * (zz[1] - yref[4]);
        double pmv[3] = {pin[0], pin[1], pin[2]}, xnv[3] = {xi[0], xi[1], xi[2]};
        x.prev(pmv);
        x.next(xnv);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double pm = pmv[q];
            if (is0) pm = 0.0;
            g[2 + q] = (isN ? c.we[q] : c.wg[2 + q]) * (zz[2 + q] - yref[q]) - pm;
            const double xnext = xnv[q];
            if (mid) r_eq = amax(r_eq, xn[q] - xnext);
            if (is0) r_ineq = amax(r_ineq, x0bar[q] - xi[q]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (has_bound(j)) {
                const double dl = c.lb[j] - zz[j], du = zz[j] - c.ub[j];
                g[j] += lamn[4 + j] - lamn[j];
                r_ineq = fmax(r_ineq, dl); r_ineq = fmax(r_ineq, du);
                r_comp = amax(r_comp, lamn[j] * dl); r_comp = amax(r_comp, lamn[4 + j] * du);
            }
        }
        if (mid) {
            g[0] += ab.B00 * pin[0] + ab.B10 * pin[1] + ab.B20 * pin[2];
            g[1] += ab.B01 * pin[0] + ab.B11 * pin[1] + ab.B21 * pin[2];
            g[2] += pin[0]; g[3] += pin[1]; g[4] += ab.a * pin[0] + ab.b * pin[1] + pin[2];
            r_stat = amax(r_stat, g[0]); r_stat = amax(r_stat, g[1]);
        }
        if (act && !is0) { r_stat = amax(r_stat, g[2]); r_stat = amax(r_stat, g[3]); r_stat = amax(r_stat, g[4]); }
        double rr[4] = {r_stat, r_eq, r_ineq, r_comp};
        x.rmax(rr);
        out[0] = rr[0]; out[1] = rr[1]; out[2] = rr[2]; out[3] = rr[3];
    }
    __device__ __forceinline__ void cold_start(const double xi[3], const double ui[2], const double x0bar[3]) {
        z[0] = mid ? ui[0] : 0.0;
        z[1] = mid ? ui[1] : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) z[2 + q] = is0 ? x0bar[q] : xi[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double tl = z[j] - c.lb[j];
            double tu = c.ub[j] - z[j];
            double zj = z[j];
            if (tl < MPC_IPM_THR0) {
                if (tu < MPC_IPM_THR0) { zj = 0.5 * (c.lb[j] + c.ub[j]); tl = MPC_IPM_THR0; tu = MPC_IPM_THR0; }
                else { tl = MPC_IPM_THR0; zj = c.lb[j] + MPC_IPM_THR0; }
            } else if (tu < MPC_IPM_THR0) { tu = MPC_IPM_THR0; zj = c.ub[j] - MPC_IPM_THR0; }
            // every member is written once, through selects: stores to the same member in both arms
            // of a branch are sunk into one store through a phi ADDRESS, which keeps the whole
            // array in scratch
            const bool hb = has_bound(j);
            const double il = rcp(tl), iu = rcp(tu);
            z[j] = hb ? zj : z[j];
            bn.t[j] = hb ? tl : 1.0; bn.t[4 + j] = hb ? tu : 1.0;
            ti[j] = hb ? il : 1.0; ti[4 + j] = hb ? iu : 1.0;
            bn.lam[j] = hb ? MPC_IPM_MU0 * il : 0.0; bn.lam[4 + j] = hb ? MPC_IPM_MU0 * iu : 0.0;
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
    }

    // ---- BU: residuals at the current point (all stages at once) and HPIPM's exit test.
    //      Lane::sweep_bu without its variable update (which happens at the end of the accepted
    //      corrector pass, fw<true>, where the steps are still in registers) and without the
    //      factorisation: that is the head of the NEXT iteration (factor_now), so the iteration
    //      that meets the exit test does not pay for a Riccati scan nobody uses.
    __device__ __forceinline__ void bu(double& musum, bool& open) {
        // --- residuals at the new point ---
        double pim[3] = {pi[0], pi[1], pi[2]}, zxn[3] = {z[2], z[3], z[4]};
        x.prev(pim);
        x.next(zxn);
        if (is0) { pim[0] = pim[1] = pim[2] = 0.0; }
        rg[0] = c.wg[0] * (z[0] - yref[3]);
        rg[1] = c.wg[1] * (z[1] - yref[4]);
#pragma unroll
        for (int q = 0; q < 3; ++q) rg[2 + q] = (isN ? c.we[q] : c.wg[2 + q]) * (z[2 + q] - yref[q]) - pim[q];
        // Entries without a bound (stage N, the state bounds of stage 0) carry lam = 0, t = 1: their
        // lam*t, Gamma and gamma are exact zeros, so only the two norm tests need the predicate.
        // The norms themselves are only compared with tolerances ("max > tol" = "any > tol").
        double pg = 0.0, pb = 0.0, ps = 0.0;
        bool viol = false;
const Opq oq_ = opq();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_cached(j, q, oq_);
            rg[j] += q.lam_u - q.lam_l;
            viol = viol || (has_bound(j) && (fabs(q.rd_l) > MPC_IPM_RES_D_MAX || fabs(q.rd_u) > MPC_IPM_RES_D_MAX ||
                                             fabs(fabs(q.rm_l) - MPC_IPM_TAU_MIN) > MPC_IPM_RES_M_MAX ||
                                             fabs(fabs(q.rm_u) - MPC_IPM_TAU_MIN) > MPC_IPM_RES_M_MAX));
            ps += q.rm_l;
            ps += q.rm_u;
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
        // HPIPM tests the four inf-norms against their tolerances; "max over lanes > tol" is "any
        // lane > tol" (|m - tau_min| > tol is monotone in m for m >= tau_min and false below), so
        // the four max-reductions collapse into one ballot.  Only mu needs a real reduction.
        open = x.gany(pg > MPC_IPM_RES_G_MAX || pb > MPC_IPM_RES_B_MAX || viol);
        double ns[1] = {ps};
        x.rsum(ns);
        musum = ns[0];
    }

    // Barrier terms of the current point (exact zeros for entries without a bound: lam = 0, t = 1)
    // and the Riccati factorisation with the affine right-hand side: the head of an IPM iteration.
    __device__ __forceinline__ void factor_now() {
        double Gam[4], gam[4];
const Opq oq_ = opq();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_cached(j, q, oq_);
            Gam[j] = con_Gamma(q);
            gam[j] = con_gamma(q);
        }
        factor(Gam, gam);
    }

    __device__ __forceinline__ void factor(const double Gam[4], const double gam[4]) {
        if (SCAN && !seq_now) {
            // The scan re-associates the recursion; with strongly active STATE bounds (J ~ 1e8,
            // I + C J ill-conditioned) it can lose digits the sequential recursion keeps.  The
            // guard is the acceptance test on the linear system's residual in iterate(); the
            // optional stage-wise Riccati check (MPC_SCAN_CHECK_P) falls back right here.
            if (!factor_scan(Gam, gam)) {
                wdiag |= MPC_DIAG_SCAN_FALLBACK;
                factor_seq(Gam, gam);
            }
        } else {
            factor_seq(Gam, gam);
        }
    }

    // --- Riccati factorisation + affine right-hand side, stage N-1 down to 0: the lane that
    //     owns the stage computes, P and p move one lane down per step.
    __device__ __forceinline__ void factor_seq(const double Gam[4], const double gam[4]) {
        Sym3 Po;
        Po.p00 = c.wer[0]; Po.p10 = 0.0; Po.p11 = c.wer[1];
        Po.p20 = 0.0; Po.p21 = 0.0; Po.p22 = c.wer[2];
        double po0 = rg[2], po1 = rg[3], po2 = rg[4];   // meaningful in lane N
        for (int kc = N - 1; kc >= 0; --kc) {
            double sh[9] = {Po.p00, Po.p10, Po.p11, Po.p20, Po.p21, Po.p22, po0, po1, po2};
            x.next(sh);
            Sym3 Pi;
            Pi.p00 = sh[0]; Pi.p10 = sh[1]; Pi.p11 = sh[2]; Pi.p20 = sh[3]; Pi.p21 = sh[4]; Pi.p22 = sh[5];
            const double p0 = sh[6], p1 = sh[7], p2 = sh[8];
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
                    const double Hxx00 = c.wgr[2] + Gam[2] + P.p00;
                    const double Hxx10 = P.p10;
                    const double Hxx11 = c.wgr[3] + Gam[3] + P.p11;
                    const double Hxx20 = a * P.p00 + b * P.p10 + P.p20;
                    const double Hxx21 = a * P.p10 + b * P.p11 + P.p21;
                    const double Hxx22 = c.wgr[4] + (a * S02 + b * S12 + S22);
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

    // ================= parallel-in-time versions of the three recursions =================
    // The Riccati recursion is a suffix scan over "conditional value function" elements
    // (A, C, J) of horizon segments [S. Sarkka, A. Garcia-Fernandez, "Temporal parallelization of
    // dynamic programming and linear quadratic control", IEEE TAC 2023]: a segment i..j maps
    // x_i to x_j with  x_j = A x_i + (control effect, C = controllability Gramian-like term)
    // and costs 1/2 x_i^T J x_i when x_j is free.  One stage: A = A_k, C = B_k R_k^-1 B_k^T,
    // J = Q_k; terminal element (0, 0, Q_N).  Combination of segment 1 (k..k+d) followed by
    // segment 2:  W = (I + C1 J2)^-1,
    //     A12 = A2 W A1,   C12 = A2 W C1 A2^T + C2,   J12 = A1^T J2 W A1 + J1.
    // After log2 steps lane k holds J = P_k, the value function of stage k; the gains are then
    // formed per stage exactly as in the sequential code.  The two vector recursions
    // (p backward, dx forward) are scans over affine maps x -> M x + v.
    // Measured against the sequential recursion on the IPM's own matrices (barrier terms up to
    // 1e11): relative deviation of P <= 1e-14.

    // The two affine scans need no "has a partner" guard when a wavefront holds one instance:
    // an element that already reaches the end of its recursion has an exactly zero matrix
    // (lane N / lane 0 start with 0, and 0 * finite = 0), so combining it with whatever the
    // shuffle returns for a missing partner (its own value, 0, or a lane beyond N, which holds
    // (0, 0)) leaves the vector unchanged bit for bit.  Without the branch the compiler needs no
    // merge copies (42 v_mov for 39 FP64 instructions at distance 1).  With several instances per
    // wavefront the guard stays: a neighbour's Inf/NaN must not leak through 0 * NaN.
    static constexpr bool GUARD = X::PER_BLOCK > 1;

    // affine suffix scan: on entry lane k (< N) holds p_k = T p_{k+1} + s, lane N holds (0, p_N),
    // lanes beyond N hold (0, 0); on exit s = p_k.
    __device__ __forceinline__ void affine_suffix_scan(M3& T, double s[3]) const {
#pragma unroll
        for (int d = 1; d < GW; d *= 2) {
            M3 T2 = T;
            double s2[3] = {s[0], s[1], s[2]};
            if (2 * d < GW) x.down(T2.m, d);
            x.down(s2, d);
            if (!GUARD || k + d <= N) {
                double r[3];
                m3_mulv(T, s2, r);
#pragma unroll
                for (int i = 0; i < 3; ++i) s[i] = r[i] + s[i];
                if (2 * d < GW) T = m3_mul(T, T2);
            }
        }
    }
    // affine prefix scan: lane k (< N) holds dx_{k+1} = M dx_k + v with dx_0 = 0, i.e. lane 0 holds
    // (0, v_0); lanes from N on hold (0, 0); on exit v = dx_{k+1}.
    __device__ __forceinline__ void affine_prefix_scan(M3& M, double v[3]) const {
#pragma unroll
        for (int d = 1; d < GW; d *= 2) {
            M3 M1 = M;
            double v1[3] = {v[0], v[1], v[2]};
            if (2 * d < GW) x.up(M1.m, d);
            x.up(v1, d);
            if (!GUARD || (k >= d && mid)) {
                double r[3];
                m3_mulv(M, v1, r);
#pragma unroll
                for (int i = 0; i < 3; ++i) v[i] = r[i] + v[i];
                if (2 * d < GW) M = m3_mul(M, M1);
            }
        }
    }
    // closed-loop transition of this stage: Phi = A + B K
    __device__ __forceinline__ M3 closed_loop() const {
        M3 F;
        F.m[0] = 1.0 + (ab.B00 * K00 + ab.B01 * K10); F.m[1] = ab.B00 * K01 + ab.B01 * K11; F.m[2] = ab.a + (ab.B00 * K02 + ab.B01 * K12);
        F.m[3] = ab.B10 * K00 + ab.B11 * K10; F.m[4] = 1.0 + (ab.B10 * K01 + ab.B11 * K11); F.m[5] = ab.b + (ab.B10 * K02 + ab.B11 * K12);
        F.m[6] = ab.B20 * K00 + ab.B21 * K10; F.m[7] = ab.B20 * K01 + ab.B21 * K11; F.m[8] = 1.0 + (ab.B20 * K02 + ab.B21 * K12);
        return F;
    }

    // ---- X::MCACHE: the products of the closed-loop transitions are shared by all four vector
    //      recursions of an IPM iteration (predictor and corrector, backward and forward), so
    //      they are formed once per factorisation and kept (in registers here: Mc; in LDS in
    //      rti_wave2.hpp):
    //          M^(d)_k = Phi_k Phi_{k-1} .. Phi_{k-d+1},   d = 1, 2, .., GW/2,
    //      with Phi_0 := 0 (dx_0 = 0) and Phi_k := 0 from stage N on.  The forward scan pulls
    //      dx from lane k-d and applies its own M^(d)_k; the backward scan runs on
    //      r_k = p_{k+1}: lane m forms (M^(d)_m)^T r_m, lane k pulls that from lane k+d (the
    //      product over stages k+1..k+d is exactly the one lane k+d holds).  Each scan level
    //      then moves 3 doubles and costs 9 FMA instead of 12 doubles and 36 FMA.
    // backward: on entry s = s_k of p_k = Phi_k^T p_{k+1} + s_k (lane N: p_N, beyond: 0); on exit
    // r = p_{k+1} in the lanes k < N
    // BUILD: the first backward scan after a factorisation also forms and stores the products
    // (the two shuffles of a level travel together: one LDS round trip instead of two).
    template <bool BUILD>
    __device__ __forceinline__ void suffix_cached(const double s[3], double (&r)[3]) {
        r[0] = s[0]; r[1] = s[1]; r[2] = s[2];
        x.next(r);
        if (!mid) { r[0] = r[1] = r[2] = 0.0; }
        M3 M;
        if (BUILD) {
            M = closed_loop();
            if (!mid || is0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) M.m[i] = 0.0;
            }
            mc_store(0, M);
        }
        int lvl = 0;
#pragma unroll
        for (int d = 1; d < GW; d *= 2, ++lvl) {
            const M3 Md = BUILD ? M : mc_load(lvl);
            double cc[3];
            m3_tmulv(Md, r, cc);
            x.down(cc, d);
            if (BUILD && 2 * d < GW) {
                M3 M1 = M;
                x.up(M1.m, d);
                // (unguarded for one instance per wavefront: a product that reaches stage 0 is exactly 0)
                if (!GUARD || k >= d) M = m3_mul(M, M1);
                mc_store(lvl + 1, M);
            }
            if (k + d < N) { r[0] += cc[0]; r[1] += cc[1]; r[2] += cc[2]; }
        }
    }
    // forward: on entry v = v_k of dx_{k+1} = Phi_k dx_k + v_k (0 from lane N on); on exit v = dx_{k+1}
    __device__ __forceinline__ void prefix_cached(double v[3]) const {
        int lvl = 0;
#pragma unroll
        for (int d = 1; d < GW; d *= 2, ++lvl) {
            double v1[3] = {v[0], v[1], v[2]};
            x.up(v1, d);
            const M3 Md = mc_load(lvl);
            if (!GUARD || (k >= d && mid)) {
                double rr[3];
                m3_mulv(Md, v1, rr);
                v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2];
            }
        }
    }

    __device__ __forceinline__ bool factor_scan(const double Gam[4], const double gam[4]) {
        // --- elements ---
        M3 A;
        double C[6], J[6];   // symmetric: 00, 10, 11, 20, 21, 22
        if (mid) {
            A.m[0] = 1.0; A.m[1] = 0.0; A.m[2] = ab.a; A.m[3] = 0.0; A.m[4] = 1.0; A.m[5] = ab.b; A.m[6] = 0.0; A.m[7] = 0.0; A.m[8] = 1.0;
            const double r0 = rcp(c.wgr[0] + Gam[0]), r1 = rcp(c.wgr[1] + Gam[1]);
            const double b00 = ab.B00 * r0, b10 = ab.B10 * r0, b20 = ab.B20 * r0;
            const double b01 = ab.B01 * r1, b11 = ab.B11 * r1, b21 = ab.B21 * r1;
            C[0] = b00 * ab.B00 + b01 * ab.B01;
            C[1] = b10 * ab.B00 + b11 * ab.B01;
            C[2] = b10 * ab.B10 + b11 * ab.B11;
            C[3] = b20 * ab.B00 + b21 * ab.B01;
            C[4] = b20 * ab.B10 + b21 * ab.B11;
            C[5] = b20 * ab.B20 + b21 * ab.B21;
            J[0] = c.wgr[2] + Gam[2]; J[1] = 0.0; J[2] = c.wgr[3] + Gam[3];
            J[3] = 0.0; J[4] = 0.0; J[5] = c.wgr[4];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) A.m[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) { C[i] = 0.0; J[i] = 0.0; }
            J[0] = c.wer[0]; J[2] = c.wer[1]; J[5] = c.wer[2];
        }
        // --- suffix scan, first level (distance 1) on the structured single-stage elements:
        //     A = e I + a E02 + b E12 (e = 1, or 0 from stage N on), J diagonal.  The same
        //     combination as below with the zeros and ones written out: 12 instead of 21 values
        //     shuffled and about 80 multiply-adds fewer.
        {
            const double e1 = fbu, a1 = ab.a, b1 = ab.b;
            double f[12] = {e1, a1, b1, C[0], C[1], C[2], C[3], C[4], C[5], J[0], J[2], J[5]};
            x.down(f, 1);
            if (!GUARD || k + 1 <= N) {
                const double e2 = f[0], a2 = f[1], b2 = f[2];
                const double C2[6] = {f[3], f[4], f[5], f[6], f[7], f[8]};
                const double j20 = f[9], j21 = f[10], j22 = f[11];
                M3 Cm;
                Cm.m[0] = C[0]; Cm.m[1] = C[1]; Cm.m[2] = C[3]; Cm.m[3] = C[1]; Cm.m[4] = C[2]; Cm.m[5] = C[4]; Cm.m[6] = C[3]; Cm.m[7] = C[4]; Cm.m[8] = C[5];
                M3 Xm;   // I + C1 diag(j2)
#pragma unroll
                for (int i = 0; i < 3; ++i) { Xm.m[3 * i] = Cm.m[3 * i] * j20; Xm.m[3 * i + 1] = Cm.m[3 * i + 1] * j21; Xm.m[3 * i + 2] = Cm.m[3 * i + 2] * j22; }
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
                JW.m[0] = j20 * W.m[0];
                JW.m[3] = j21 * W.m[3]; JW.m[4] = j21 * W.m[4];
                JW.m[6] = j22 * W.m[6]; JW.m[7] = j22 * W.m[7]; JW.m[8] = j22 * W.m[8];
                JW.m[1] = JW.m[3]; JW.m[2] = JW.m[6]; JW.m[5] = JW.m[7];
                // U = JW A1
                M3 U;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    U.m[3 * i] = JW.m[3 * i] * e1;
                    U.m[3 * i + 1] = JW.m[3 * i + 1] * e1;
                    U.m[3 * i + 2] = JW.m[3 * i] * a1 + JW.m[3 * i + 1] * b1 + JW.m[3 * i + 2] * e1;
                }
                // J12 = J1 + A1^T U
                const double jd0 = J[0], jd1 = J[2], jd2 = J[5];
                J[0] = jd0 + e1 * U.m[0];
                J[1] = e1 * U.m[3];
                J[2] = jd1 + e1 * U.m[4];
                J[3] = a1 * U.m[0] + b1 * U.m[3] + e1 * U.m[6];
                J[4] = a1 * U.m[1] + b1 * U.m[4] + e1 * U.m[7];
                J[5] = jd2 + (a1 * U.m[2] + b1 * U.m[5] + e1 * U.m[8]);
                // G = A2 W
                M3 G;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    G.m[j] = e2 * W.m[j] + a2 * W.m[6 + j];
                    G.m[3 + j] = e2 * W.m[3 + j] + b2 * W.m[6 + j];
                    G.m[6 + j] = e2 * W.m[6 + j];
                }
                const M3 T = m3_mul(G, Cm);
                // C12 = T A2^T + C2
                C[0] = C2[0] + (T.m[0] * e2 + T.m[2] * a2);
                C[1] = C2[1] + (T.m[3] * e2 + T.m[5] * a2);
                C[2] = C2[2] + (T.m[4] * e2 + T.m[5] * b2);
                C[3] = C2[3] + (T.m[6] * e2 + T.m[8] * a2);
                C[4] = C2[4] + (T.m[7] * e2 + T.m[8] * b2);
                C[5] = C2[5] + T.m[8] * e2;
                // A12 = G A1
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    A.m[3 * i] = G.m[3 * i] * e1;
                    A.m[3 * i + 1] = G.m[3 * i + 1] * e1;
                    A.m[3 * i + 2] = G.m[3 * i] * a1 + G.m[3 * i + 1] * b1 + G.m[3 * i + 2] * e1;
                }
            }
        }
        // --- the remaining levels on full elements ---
#pragma unroll
        for (int d = 2; d < GW; d *= 2) {
            M3 A2;
            double C2[6], J2[6];
            if (2 * d < GW) {
                double e[21];
#pragma unroll
                for (int i = 0; i < 9; ++i) e[i] = A.m[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) { e[9 + i] = C[i]; e[15 + i] = J[i]; }
                x.down(e, d);
#pragma unroll
                for (int i = 0; i < 9; ++i) A2.m[i] = e[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) { C2[i] = e[9 + i]; J2[i] = e[15 + i]; }
            } else {   // the last level only needs J of the right-hand segment
#pragma unroll
                for (int i = 0; i < 6; ++i) J2[i] = J[i];
                x.down(J2, d);
            }
            // (unguarded as in the affine scans: an element that reaches the terminal stage has
            //  A = 0 and C = 0 exactly, so W = I and J12 = J1 whatever the missing partner returns)
            if (!GUARD || k + d <= N) {
                const bool more = 2 * d < GW;   // A and C are only needed by later steps
                // Xm = I + C1 J2
                M3 Cm, Jm;
                Cm.m[0] = C[0]; Cm.m[1] = C[1]; Cm.m[2] = C[3]; Cm.m[3] = C[1]; Cm.m[4] = C[2]; Cm.m[5] = C[4]; Cm.m[6] = C[3]; Cm.m[7] = C[4]; Cm.m[8] = C[5];
                Jm.m[0] = J2[0]; Jm.m[1] = J2[1]; Jm.m[2] = J2[3]; Jm.m[3] = J2[1]; Jm.m[4] = J2[2]; Jm.m[5] = J2[4]; Jm.m[6] = J2[3]; Jm.m[7] = J2[4]; Jm.m[8] = J2[5];
                M3 Xm = m3_mul(Cm, Jm);
                Xm.m[0] += 1.0; Xm.m[4] += 1.0; Xm.m[8] += 1.0;
                // W = Xm^-1 (adjugate)
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
                // J12 = A1^T (J2 W) A1 + J1
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
                J[0] += A.m[0] * U.m[0] + A.m[3] * U.m[3] + A.m[6] * U.m[6];
                J[1] += A.m[1] * U.m[0] + A.m[4] * U.m[3] + A.m[7] * U.m[6];
                J[2] += A.m[1] * U.m[1] + A.m[4] * U.m[4] + A.m[7] * U.m[7];
                J[3] += A.m[2] * U.m[0] + A.m[5] * U.m[3] + A.m[8] * U.m[6];
                J[4] += A.m[2] * U.m[1] + A.m[5] * U.m[4] + A.m[8] * U.m[7];
                J[5] += A.m[2] * U.m[2] + A.m[5] * U.m[5] + A.m[8] * U.m[8];
                if (more) {
                    const M3 G = m3_mul(A2, W);
                    const M3 T = m3_mul(G, Cm);
                    // C12 = (G C1) A2^T + C2
                    C[0] = C2[0] + (T.m[0] * A2.m[0] + T.m[1] * A2.m[1] + T.m[2] * A2.m[2]);
                    C[1] = C2[1] + (T.m[3] * A2.m[0] + T.m[4] * A2.m[1] + T.m[5] * A2.m[2]);
                    C[2] = C2[2] + (T.m[3] * A2.m[3] + T.m[4] * A2.m[4] + T.m[5] * A2.m[5]);
                    C[3] = C2[3] + (T.m[6] * A2.m[0] + T.m[7] * A2.m[1] + T.m[8] * A2.m[2]);
                    C[4] = C2[4] + (T.m[6] * A2.m[3] + T.m[7] * A2.m[4] + T.m[8] * A2.m[5]);
                    C[5] = C2[5] + (T.m[6] * A2.m[6] + T.m[7] * A2.m[7] + T.m[8] * A2.m[8]);
                    A = m3_mul(G, A);
                }
            }
        }
        // --- P_{k+1} to the lane of stage k, gains per stage ---
        {
            double jn[6] = {J[0], J[1], J[2], J[3], J[4], J[5]};
            x.next(jn);
            P.p00 = jn[0]; P.p10 = jn[1]; P.p11 = jn[2]; P.p20 = jn[3]; P.p21 = jn[4]; P.p22 = jn[5];
        }
        Gain g;
        gain(P, ab, Gam[0], Gam[1], true, g);
        id0 = g.id0; l10 = g.l10; id1 = g.id1;
        // (no masking: from stage N on B = 0 makes K an exact zero, and the gain of stage 0 only ever
        //  multiplies dx_0 = 0 or enters quantities of stage 0 that nobody reads)
        K00 = g.K00; K01 = g.K01; K02 = g.K02; K10 = g.K10; K11 = g.K11; K12 = g.K12;
#ifdef MPC_SCAN_CHECK_P
        const bool kk = mid && !is0;
#endif
        // --- a-posteriori check: P_k of the scan against one sequential Riccati step from P_{k+1} ---
        // Off by default: the lin_g acceptance test in iterate() subsumes it (an inaccurate P shows up
        // as a stationarity residual of the linear system) and it costs 3.5 % of the kernel.
#ifdef MPC_SCAN_CHECK_P
        {
            const double a = ab.a, b = ab.b;
            const double S02 = P.p00 * a + P.p10 * b + P.p20;
            const double S12 = P.p10 * a + P.p11 * b + P.p21;
            const double S22 = P.p20 * a + P.p21 * b + P.p22;
            const double Q00 = (c.wgr[2] + Gam[2] + P.p00) + (g.M00 * g.K00 + g.M10 * g.K10);
            const double Q10 = P.p10 + (g.M01 * g.K00 + g.M11 * g.K10);
            const double Q11 = (c.wgr[3] + Gam[3] + P.p11) + (g.M01 * g.K01 + g.M11 * g.K11);
            const double Q20 = (a * P.p00 + b * P.p10 + P.p20) + (g.X02 * g.K00 + g.X12 * g.K10);
            const double Q21 = (a * P.p10 + b * P.p11 + P.p21) + (g.X02 * g.K01 + g.X12 * g.K11);
            const double Q22 = (c.wgr[4] + (a * S02 + b * S12 + S22)) + (g.X02 * g.K02 + g.X12 * g.K12);
            double err = fmax(fmax(fabs(Q00 - J[0]), fabs(Q10 - J[1])), fmax(fabs(Q11 - J[2]), fabs(Q20 - J[3])));
            err = fmax(err, fmax(fabs(Q21 - J[4]), fabs(Q22 - J[5])));
            const double scale = fmax(fmax(fabs(J[0]), fabs(J[2])), fabs(J[5]));
            double bad = (kk && !(err <= c.scan_tol_P * scale)) ? 1.0 : 0.0;   // NaN counts as bad
            double bb[1] = {bad};
            x.rmax(bb);
            if (bb[0] != 0.0) return false;
        }
#endif
        // --- affine right-hand side ---
        const double q0 = P.p00 * rb[0] + P.p10 * rb[1] + P.p20 * rb[2];
        const double q1 = P.p10 * rb[0] + P.p11 * rb[1] + P.p21 * rb[2];
        const double q2 = P.p20 * rb[0] + P.p21 * rb[1] + P.p22 * rb[2];
        rhs_scan<true>(rg[0] + gam[0], rg[1] + gam[1], rg[2] + gam[2], rg[3] + gam[3], q0, q1, q2);
        return true;
    }

    // p_k = (hxp + K^T hup) + Phi^T (q + p_{k+1}),  kff = -(Huu)^-1 (hup + B^T (q + p_{k+1}))
    // FIRST: the first right-hand side after a factorisation (forms the product cache on the way)
    template <bool FIRST>
    __device__ __forceinline__ void rhs_scan(double hup0, double hup1, double hxp0, double hxp1, double q0, double q1, double q2) {
        double p0, p1, p2;
        if (X::MCACHE) {
            // s_k with Phi_k^T q_k written out; Phi_k from the cache (its Phi_0 is zeroed, but s_0 is
            // not needed: p_0 is never formed)
            const M3 F = FIRST ? closed_loop() : mc_load(0);
            double s[3];
            if (mid) {
                const double qv[3] = {q0, q1, q2};
                double r[3];
                m3_tmulv(F, qv, r);
                s[0] = (hxp0 + (K00 * hup0 + K10 * hup1)) + r[0];
                s[1] = (hxp1 + (K01 * hup0 + K11 * hup1)) + r[1];
                s[2] = (rg[4] + (K02 * hup0 + K12 * hup1)) + r[2];
            } else {
                s[0] = isN ? rg[2] : 0.0; s[1] = isN ? rg[3] : 0.0; s[2] = isN ? rg[4] : 0.0;
            }
            double r[3];
            suffix_cached<FIRST>(s, r);
            p0 = r[0]; p1 = r[1]; p2 = r[2];
        } else {
            const M3 F = closed_loop();
            M3 T;   // Phi^T
            T.m[0] = F.m[0]; T.m[1] = F.m[3]; T.m[2] = F.m[6]; T.m[3] = F.m[1]; T.m[4] = F.m[4]; T.m[5] = F.m[7]; T.m[6] = F.m[2]; T.m[7] = F.m[5]; T.m[8] = F.m[8];
            double s[3];
            if (mid) {
                const double qv[3] = {q0, q1, q2};
                double r[3];
                m3_mulv(T, qv, r);
                s[0] = (hxp0 + (K00 * hup0 + K10 * hup1)) + r[0];
                s[1] = (hxp1 + (K01 * hup0 + K11 * hup1)) + r[1];
                s[2] = (rg[4] + (K02 * hup0 + K12 * hup1)) + r[2];
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) T.m[i] = 0.0;
                s[0] = isN ? rg[2] : 0.0; s[1] = isN ? rg[3] : 0.0; s[2] = isN ? rg[4] : 0.0;   // lane N: p_N
            }
            affine_suffix_scan(T, s);
            double sn[3] = {s[0], s[1], s[2]};
            x.next(sn);
            p0 = sn[0]; p1 = sn[1]; p2 = sn[2];
        }
        if (mid) {
            pv[0] = p0; pv[1] = p1; pv[2] = p2;
            const double w0 = q0 + p0, w1 = q1 + p1, w2 = q2 + p2;
            const double hu0 = hup0 + (ab.B00 * w0 + ab.B10 * w1 + ab.B20 * w2);
            const double hu1 = hup1 + (ab.B01 * w0 + ab.B11 * w1 + ab.B21 * w2);
            solve2(id0, l10, id1, hu0, hu1, kff[0], kff[1]);
        }
    }

    // dx_{k+1} = Phi dx_k + (rb + B kff), dx_0 = 0;  du = kff + K dx
    __device__ __forceinline__ void fwd_scan(double& nx0, double& nx1, double& nx2, double dl[5]) {
        double v[3];
        v[0] = rb[0] + (ab.B00 * kff[0] + ab.B01 * kff[1]);
        v[1] = rb[1] + (ab.B10 * kff[0] + ab.B11 * kff[1]);
        v[2] = rb[2] + (ab.B20 * kff[0] + ab.B21 * kff[1]);
        if (!mid) { v[0] = v[1] = v[2] = 0.0; }
        if (X::MCACHE) {
            prefix_cached(v);
        } else {
            M3 M = closed_loop();
            if (!mid || is0) {   // dx_0 = 0: the stage-0 transition never acts
#pragma unroll
                for (int i = 0; i < 9; ++i) M.m[i] = 0.0;
            }
            affine_prefix_scan(M, v);
        }
        nx0 = v[0]; nx1 = v[1]; nx2 = v[2];
        double dxv[3] = {nx0, nx1, nx2};
        x.prev(dxv);
        double dx0 = dxv[0], dx1 = dxv[1], dx2 = dxv[2];
        if (is0) { dx0 = dx1 = dx2 = 0.0; }
        if (mid) {
            dl[0] = kff[0] + (K00 * dx0 + K01 * dx1 + K02 * dx2);
            dl[1] = kff[1] + (K10 * dx0 + K11 * dx1 + K12 * dx2);
        }
        if (act) { dl[2] = dx0; dl[3] = dx1; dl[4] = dx2; }
    }

    // ---- BW: corrected complementarity -> new right-hand side, backward.  Lane::sweep_bw.
    __device__ __forceinline__ void bw(int mode, double sigma_mu) {
        double gam[4];
const Opq oq_ = opq();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_cached(j, q, oq_);
            con_correct(q, mode, dza[j], sigma_mu * bf(j));
            gam[j] = con_gamma(q);   // exactly 0 without a bound
        }
        const double q0 = P.p00 * rb[0] + P.p10 * rb[1] + P.p20 * rb[2];
        const double q1 = P.p10 * rb[0] + P.p11 * rb[1] + P.p21 * rb[2];
        const double q2 = P.p20 * rb[0] + P.p21 * rb[1] + P.p22 * rb[2];
        const double hup0 = rg[0] + gam[0], hup1 = rg[1] + gam[1];
        const double hxp0 = rg[2] + gam[2], hxp1 = rg[3] + gam[3];
        if (SCAN && !seq_now) rhs_scan<false>(hup0, hup1, hxp0, hxp1, q0, q1, q2);
        else rhs_seq(hup0, hup1, hxp0, hxp1, q0, q1, q2);
    }

    __device__ __forceinline__ void rhs_seq(double hup0, double hup1, double hxp0, double hxp1, double q0, double q1, double q2) {
        double po0 = rg[2], po1 = rg[3], po2 = rg[4];   // meaningful in lane N
        for (int kc = N - 1; kc >= 0; --kc) {
            double sh[3] = {po0, po1, po2};
            x.next(sh);
            const double p0 = sh[0], p1 = sh[1], p2 = sh[2];
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

    __device__ __forceinline__ void fwd_seq(double& nx0, double& nx1, double& nx2, double dl[5]) {
        for (int kc = 0; kc < N; ++kc) {
            double sh[3] = {nx0, nx1, nx2};
            x.prev(sh);
            double dx0 = sh[0], dx1 = sh[1], dx2 = sh[2];
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
            double sh[3] = {nx0, nx1, nx2};
            x.prev(sh);
            if (isN) { dl[2] = sh[0]; dl[3] = sh[1]; dl[4] = sh[2]; }
        }
    }

    // ---- FW: forward substitution (sequential), then step lengths and the mu_aff sums on all
    //      stages at once.  Lane::sweep_fw.
    struct FwdOut {
        double alpha, s1, s2;               // min(alpha_prim, alpha_dual), the two mu_aff sums
        bool lin_itref, lin_lq, lin_scan;   // lin_g >= itref_tol_g / lq_tol / scan_tol_lin (FINAL only)
        bool redo_centering;                // FINAL: mu would grow -> redo the corrector as pure centering
        bool accepted;                      // FINAL: the step was taken (variables updated)
    };
    // FINAL: the corrector pass.  It also decides, from the reductions it has just made, whether
    // the direction is the one to take (HPIPM's conditional predictor-corrector test; the scan
    // acceptance test when `may_retry`), and if so performs the variable update right here.
    template <bool FINAL>
    __device__ __forceinline__ void fw(int mode, double sigma_mu, FwdOut& o, double musum = 0.0, double mu_aff0 = 0.0,
                                       bool may_retry = false) {
        double nx0 = 0.0, nx1 = 0.0, nx2 = 0.0;   // dx_{k+1}, produced by the lane of stage k
        double dl[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (SCAN && !seq_now) fwd_scan(nx0, nx1, nx2, dl);
        else fwd_seq(nx0, nx1, nx2, dl);
        // Step length in inverse form: 1/alpha = max(1, max_i -dv_i / v_i) over the slacks and the
        // multipliers (COMPUTE_ALPHA_QP's alpha = min(1, min_{dv_i<0} -v_i/dv_i)).  The slack
        // candidates use the cached 1/t — one multiply and one max each, no selects; the multiplier
        // candidates keep a running ratio n/d compared by cross-multiplication and cost one
        // reciprocal per lane; the reciprocal of the result is taken once, after the reduction.
        double ginv = 1.0;
        double dn = 1.0, dd = -1.0;
        double s1 = 0.0, s2 = 0.0, lin_g = 0.0;
        double dlam_net[4] = {0.0, 0.0, 0.0, 0.0};
        Step st[4];
const Opq oq_ = opq();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Con q;
            con_cached(j, q, oq_);
            if (FINAL) con_correct(q, mode, dza[j], sigma_mu * bf(j));
            Step s;
            con_step(q, dl[j], s);
            st[j] = s;
            // (no has_bound guard: without a bound dlam = 0 exactly and the slack ratios are scaled to 0)
            const double gl = (s.dt_l * q.ti_l) * bf(j), gu = (s.dt_u * q.ti_u) * bf(j);
            ginv = fmax(ginv, -gl);
            ginv = fmax(ginv, -gu);
            if (FINAL) {
                if (dn * s.dl_l < q.lam_l * dd) { dn = q.lam_l; dd = s.dl_l; }
                if (dn * s.dl_u < q.lam_u * dd) { dn = q.lam_u; dd = s.dl_u; }
            } else {
                // affine direction: dlam = -(1/t)(lam dt + lam t), so -dlam/lam = 1 + dt/t
                ginv = fmax(ginv, 1.0 + gl);
                ginv = fmax(ginv, 1.0 + gu);
            }
            s1 += q.lam_l * s.dt_l + q.t_l * s.dl_l;
            s1 += q.lam_u * s.dt_u + q.t_u * s.dl_u;
            s2 += s.dl_l * s.dt_l;
            s2 += s.dl_u * s.dt_u;
            dlam_net[j] = s.dl_u - s.dl_l;
        }
        if (FINAL) {
#pragma unroll
            for (int q = 0; q < 5; ++q) dz[q] = dl[q];
            const double dp0 = mid ? P.p00 * nx0 + P.p10 * nx1 + P.p20 * nx2 + pv[0] : 0.0;
            const double dp1 = mid ? P.p10 * nx0 + P.p11 * nx1 + P.p21 * nx2 + pv[1] : 0.0;
            const double dp2 = mid ? P.p20 * nx0 + P.p21 * nx1 + P.p22 * nx2 + pv[2] : 0.0;
            dpi[0] = dp0; dpi[1] = dp1; dpi[2] = dp2;
            double dpm[3] = {dp0, dp1, dp2};
            x.prev(dpm);
            const double dpm0 = dpm[0], dpm1 = dpm[1], dpm2 = dpm[2];
            if (mid) {
                const double lg0 = c.wg[0] * dl[0] + rg[0] + dlam_net[0] + (ab.B00 * dp0 + ab.B10 * dp1 + ab.B20 * dp2);
                const double lg1 = c.wg[1] * dl[1] + rg[1] + dlam_net[1] + (ab.B01 * dp0 + ab.B11 * dp1 + ab.B21 * dp2);
                lin_g = amax(lin_g, lg0); lin_g = amax(lin_g, lg1);
                if (SCAN) {
                    // dynamics row: the next stage's dx is what the scan delivered (nx), checked
                    // against this stage's own step
                    // (scaled so that ONE threshold serves both kinds of rows: consts.hpp kScanTolLin)
                    const double ds = c.scan_dyn_scale;
                    lin_g = amax(lin_g, ds * (rb[0] + (ab.B00 * dl[0] + ab.B01 * dl[1] + (dl[2] + ab.a * dl[4])) - nx0));
                    lin_g = amax(lin_g, ds * (rb[1] + (ab.B10 * dl[0] + ab.B11 * dl[1] + (dl[3] + ab.b * dl[4])) - nx1));
                    lin_g = amax(lin_g, ds * (rb[2] + (ab.B20 * dl[0] + ab.B21 * dl[1] + dl[4]) - nx2));
                }
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
        // HPIPM steps with alpha = min(alpha_prim, alpha_dual): one reduction on the larger inverse
        double am[1] = {FINAL ? fmax(ginv, -(dd * rcp(dn))) : ginv}, as[2] = {s1, s2};
        x.rmax(am);
        x.rsum(as);
        o.alpha = rcp(am[0]); o.s1 = as[0]; o.s2 = as[1];
        // the linear system's residual is only ever compared with thresholds: ballots
        o.lin_itref = FINAL ? x.gany(!(lin_g < MPC_IPM_RES_G_MAX)) : false;
        o.lin_lq = FINAL ? x.gany(!(lin_g < MPC_IPM_LQ_SWITCH_TOL)) : false;
        o.lin_scan = (FINAL && SCAN) ? x.gany(!(lin_g < c.scan_tol_lin)) : false;
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
                // COMPUTE step: variables move by the damped step length (d_ocp_qp_ipm_solve)
                double a_upd = alpha;
                if (a_upd < 1.0) a_upd = a_upd * ((1.0 - a_upd) * MPC_IPM_STEP_SCALE_LO + a_upd * MPC_IPM_STEP_SCALE_HI);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double aj = a_upd * bf(j), lmin = MPC_IPM_LAM_MIN * bf(j);   // no bound: lam stays 0, t stays 1
                    double ll = bn.lam[j] + aj * st[j].dl_l, lu = bn.lam[4 + j] + aj * st[j].dl_u;
                    double tl = bn.t[j] + aj * st[j].dt_l, tu = bn.t[4 + j] + aj * st[j].dt_u;
                    if (ll < lmin) ll = lmin;
                    if (lu < lmin) lu = lmin;
                    if (tl < MPC_IPM_T_MIN) tl = MPC_IPM_T_MIN;
                    if (tu < MPC_IPM_T_MIN) tu = MPC_IPM_T_MIN;
                    bn.lam[j] = ll; bn.lam[4 + j] = lu; bn.t[j] = tl; bn.t[4 + j] = tu;
                    ti[j] = rcp(tl); ti[4 + j] = rcp(tu);
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) z[q] = z[q] + a_upd * dz[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) pi[q] = pi[q] + a_upd * dpi[q];
            }
        }
    }

    // ---- the RTI step (Lane::begin / cont / iterate) ----
    __device__ __forceinline__ double nc_inv() const { return c.nc_inv; }
    __device__ __forceinline__ void begin(State& s) {
        bu(s.musum, s.open);
        s.mu = s.musum * nc_inv();
        s.alpha = 1.0;
        s.kk = 0;
        s.diag = 0;
        s.declined = false;
    }
    // Mem::IPM: park the interior-point iterate of a declined instance (nothing of the declined iteration
    // has touched it) for the list kernel, which carries on from this iteration (Wave2::load_ipm).
    // `slot`: the instance's record, (N + 1) stages of MPC_IPM_STAGE doubles + the head.
    __device__ __forceinline__ void save_ipm(double* slot, const State& s) const {
        if (!act) return;
        double* r = slot + (size_t)k * MPC_IPM_STAGE;
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = z[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) r[5 + q] = pi[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) { r[8 + q] = bn.lam[q]; r[16 + q] = bn.t[q]; }
        if (is0) {
            double* hd = slot + (size_t)(N + 1) * MPC_IPM_STAGE;
            hd[0] = (double)s.kk; hd[1] = s.alpha; hd[2] = (double)s.diag;
        }
    }
    __device__ __forceinline__ bool cont(const State& s) const {
        return s.kk < c.iter_max && s.alpha > MPC_IPM_ALPHA_MIN && s.open && !s.declined;
    }
    // One IPM iteration.  SCAN build: the stationarity and dynamics rows of the linearised KKT
    // system (lin_g; what HPIPM's refinement loop tests against 1e-6 / 1e-8) are checked against
    // 1e-8.  If the scans were not accurate enough (strongly active state bounds) NOTHING has
    // changed and the instance stops (s.declined): the kernel hands it to the sequential kernel
    // (k_wave2<.., SCAN = false, .., LIST = true>, rti_wave2.hpp), which redoes the whole QP with
    // the sequential recursions and HPIPM's iterative refinement.
    __device__ __forceinline__ void iterate(State& s) {
        const double nci = nc_inv();
        int mode = 1;
        FwdOut fc;
        double alpha = 0.0, sigma_mu = 0.0;
        factor_now();
        FwdOut fa;
        fw<false>(0, 0.0, fa);
        const double a_aff = fa.alpha;
        const double mu_aff0 = (s.musum + a_aff * fa.s1 + (a_aff * a_aff) * fa.s2) * nci;
        const double tmp = mu_aff0 / s.mu;
        const double sigma = tmp * tmp * tmp;
        sigma_mu = sigma * s.mu;
        if (!(sigma_mu > MPC_IPM_TAU_MIN)) sigma_mu = MPC_IPM_TAU_MIN;
        // corrector; conditional predictor-corrector: if mu would grow, redo it as pure centering
#pragma nounroll
        for (;;) {
            bw(mode, sigma_mu);
            fw<true>(mode, sigma_mu, fc, s.musum, mu_aff0, SCAN);
            alpha = fc.alpha;
            if (!fc.redo_centering) break;
            mode = 2;
        }
        if (!fc.accepted) {
            s.declined = true;
            return;
        }
        if (fc.lin_lq) s.diag |= MPC_DIAG_LQ;
        s.alpha = alpha;
        bu(s.musum, s.open);
        s.mu = s.musum * nci;
        ++s.kk;
    }
};

}  // namespace mpc
