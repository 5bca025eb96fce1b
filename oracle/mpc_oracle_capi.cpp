// mpc_oracle_capi.cpp — C entry points of the CPU oracle (ctypes-loadable).
// TEST INFRASTRUCTURE ONLY: see the header of mpc_oracle.hpp.  "parity unpinned" at the
// acados boundary; pinned pieces are listed there.
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "mpc_oracle.hpp"
#include "op_count.hpp"

using namespace mpc_oracle;

// Test hook: tighten / loosen the IPM exit tolerances and iteration cap ("polish" mode of
// SURVEY.md §7 hard part 1).  Values <= 0 leave the mode's default in place.
static double g_tol_override[4] = {0, 0, 0, 0};
static int g_iter_override = 0;
static int g_lq_fact_override = -1;   // -1: the mode's value; 0 / 1 / 2 as HPIPM's lq_fact (tests)
static IpmArg make_arg(const mpc_ocp_spec_t* spec) {
    IpmArg a = ipm_arg_for_mode(spec->hpipm_mode, spec->qp_iter_max);
    if (g_tol_override[0] > 0) a.res_g_max = g_tol_override[0];
    if (g_tol_override[1] > 0) a.res_b_max = g_tol_override[1];
    if (g_tol_override[2] > 0) a.res_d_max = g_tol_override[2];
    if (g_tol_override[3] > 0) a.res_m_max = g_tol_override[3];
    if (g_iter_override > 0) a.iter_max = g_iter_override;
    if (g_lq_fact_override >= 0) a.lq_fact = g_lq_fact_override;
    return a;
}

extern "C" {

void oracle_set_ipm_override(double res_g, double res_b, double res_d, double res_m, int iter_max) {
    g_tol_override[0] = res_g; g_tol_override[1] = res_b; g_tol_override[2] = res_d; g_tol_override[3] = res_m;
    g_iter_override = iter_max;
}

// HPIPM's lq_fact: 0 Cholesky only, 1 Cholesky with the switch to LQ (BALANCE), 2 LQ always; -1 = the mode's
void oracle_set_lq_fact(int lq_fact) { g_lq_fact_override = lq_fact; }

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// H5: vde_forw.  out = [xdot(3), Sxdot(9, col-major), Sudot(6, col-major)].
void oracle_vde_forw(double L, const double* x, const double* Sx, const double* Su, const double* u,
                     double* out) {
    const ModelConst<double> mc = model_const<double>(L);
    vde_forw<double>(mc, x, Sx, Su, u, out, out + 3, out + 12);
}

// H6: ode_fun.
void oracle_ode_fun(double L, const double* x, const double* u, double* out) {
    const ModelConst<double> mc = model_const<double>(L);
    ode_fun<double>(mc, x, u, out);
}

// H7: ERK4 with forward sensitivities; A 3x3 col-major, B 3x2 col-major.
void oracle_erk4(double L, double h, const double* x, const double* u, double* xn, double* A, double* B) {
    const ModelConst<double> mc = model_const<double>(L);
    erk4_forw<double>(mc, h, x, u, xn, A, B);
}

// H16: plant step, B instances, x [B][3] in/out, u [B][2].
void oracle_sim_step(const mpc_ocp_spec_t* spec, int B, double* x, const double* u) {
    const ModelConst<double> mc = model_const<double>(spec->L);
    for (int i = 0; i < B; ++i) {
        double xn[3];
        erk4_plain<double>(mc, spec->Ts, x + 3 * i, u + 2 * i, xn);
        for (int j = 0; j < 3; ++j) x[3 * i + j] = xn[j];
    }
}

// H4/H15: one SQP-RTI step per instance (sqp_max_iter is ignored: see oracle_sqp_batch), AoS like the C-ABI.
// xtraj [B][N+1][3], utraj [B][N][2], pi [B][N][3] in/out (pi may be NULL);
// u0 [B][2], status [B], qp_iter [B] out (may be NULL);
// stats [B][8] out or NULL: {iter, hpipm_status, lq_switch, itref_solves, cond_pc_fallbacks,
//                            max_lin_res_fact, max_lin_res_corr, mu_final}.
int oracle_rti_batch(const mpc_ocp_spec_t* spec, int B, const double* x0, const double* yref,
                     const double* yref_e, double* xtraj, double* utraj, double* pi, double* u0,
                     int* status, int* qp_iter, double* stats, int nthreads) {
    const int N = spec->N;
    if (N < 1 || N > MPC_MAX_N || B < 0) return -1;
    const IpmArg arg = make_arg(spec);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        RtiWork<double> w(N, arg);
        std::vector<double> pitmp(3 * N);
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < B; ++i) {
            double* xt = xtraj + (size_t)i * (N + 1) * 3;
            double* ut = utraj + (size_t)i * N * 2;
            double* pp = pi ? pi + (size_t)i * N * 3 : pitmp.data();
            int it = 0, st = 0;
            IpmStats s;
            st = rti_step<double>(*spec, w, x0 + 3 * i, yref + (size_t)i * N * 5, yref_e + 3 * i, xt, ut,
                                  pp, &it, &s);
            if (u0) { u0[2 * i] = ut[0]; u0[2 * i + 1] = ut[1]; }
            if (status) status[i] = st;
            if (qp_iter) qp_iter[i] = it;
            if (stats) {
                double* o = stats + 8 * (size_t)i;
                o[0] = s.iter; o[1] = s.hpipm_status; o[2] = s.lq_switch; o[3] = s.itref_solves;
                o[4] = s.cond_pc_fallbacks; o[5] = s.max_lin_res_fact; o[6] = s.max_lin_res_corr;
                o[7] = s.mu_final;
            }
        }
    }
    return 0;
}

// Full SQP per instance (config 5).  lam [B][N+1][8] in/out (may be NULL = zeros, not returned);
// sqp_iter [B], res [B][4] out (may be NULL).
int oracle_sqp_batch(const mpc_ocp_spec_t* spec, int B, const double* x0, const double* yref,
                     const double* yref_e, double* xtraj, double* utraj, double* pi, double* lam, double* u0,
                     int* status, int* sqp_iter, int* qp_iter, double* res, int nthreads) {
    const int N = spec->N;
    if (N < 1 || N > MPC_MAX_N || B < 0) return -1;
    const IpmArg arg = make_arg(spec);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        RtiWork<double> w(N, arg);
        std::vector<double> pitmp(3 * N), lamtmp(8 * (N + 1));
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < B; ++i) {
            double* xt = xtraj + (size_t)i * (N + 1) * 3;
            double* ut = utraj + (size_t)i * N * 2;
            double* pp = pi ? pi + (size_t)i * N * 3 : pitmp.data();
            double* ll = lam ? lam + (size_t)i * (N + 1) * 8 : lamtmp.data();
            if (!pi) std::fill(pitmp.begin(), pitmp.end(), 0.0);
            if (!lam) std::fill(lamtmp.begin(), lamtmp.end(), 0.0);
            int sit = 0, qit = 0;
            double r4[4];
            const int st = sqp_solve<double>(*spec, w, x0 + 3 * i, yref + (size_t)i * N * 5, yref_e + 3 * i, xt, ut,
                                             pp, ll, &sit, &qit, r4);
            if (u0) { u0[2 * i] = ut[0]; u0[2 * i + 1] = ut[1]; }
            if (status) status[i] = st;
            if (sqp_iter) sqp_iter[i] = sit;
            if (qp_iter) qp_iter[i] = qit;
            if (res) for (int j = 0; j < 4; ++j) res[4 * (size_t)i + j] = r4[j];
        }
    }
    return 0;
}

// Single instance with a dump of the QP and its solution (G3/G4 tests).
// A [N][9], B [N][6], b [N][3], g [N+1][5], d [N+1][8] (lower 4, upper 4; order u0,u1,x,y),
// dz [N+1][5] (order [u; x]; stage N uses slots 2..4), qpi [N][3], lam/t [N+1][8].
int oracle_rti_debug(const mpc_ocp_spec_t* spec, const double* x0, const double* yref,
                     const double* yref_e, double* xtraj, double* utraj, double* A, double* Bm,
                     double* b, double* g, double* d, double* dz, double* qpi, double* lam, double* t,
                     int* qp_iter, double* res4) {
    const int N = spec->N;
    const IpmArg arg = make_arg(spec);
    RtiWork<double> w(N, arg);
    RtiDebug<double> dbg;
    std::vector<double> pitmp(3 * N);
    IpmStats s;
    int it = 0;
    const int st = rti_step<double>(*spec, w, x0, yref, yref_e, xtraj, utraj, pitmp.data(), &it, &s, &dbg);
    auto cp = [](double* dst, const std::vector<double>& v) { if (dst) for (size_t i = 0; i < v.size(); ++i) dst[i] = v[i]; };
    cp(A, dbg.A); cp(Bm, dbg.B); cp(b, dbg.b); cp(g, dbg.g); cp(d, dbg.d);
    cp(dz, dbg.dz); cp(qpi, dbg.pi); cp(lam, dbg.lam); cp(t, dbg.t);
    if (qp_iter) *qp_iter = it;
    if (res4) for (int i = 0; i < 4; ++i) res4[i] = s.res_final[i];
    return st;
}

// Algorithmic operation count of ONE RTI step of one instance: the same
// template run on an op-counting scalar.  out = {add/sub, mul, div, sqrt, trig, cmp/abs,
// structured add/sub, mul, div (op_count.hpp)}.
int oracle_count_ops(const mpc_ocp_spec_t* spec, const double* x0, const double* yref,
                     const double* yref_e, const double* xtraj, const double* utraj, double* out9,
                     int* qp_iter) {
    using C = opcount::Counted;
    const int N = spec->N;
    const IpmArg arg = make_arg(spec);
    RtiWork<C> w(N, arg);
    std::vector<C> xt((N + 1) * 3), ut(N * 2), pp(N * 3), yr(N * 5), yre(3);
    C x0c[3];
    for (int i = 0; i < 3; ++i) { x0c[i] = C(x0[i]); yre[i] = C(yref_e[i]); }
    for (int i = 0; i < (N + 1) * 3; ++i) xt[i] = C(xtraj[i]);
    for (int i = 0; i < N * 2; ++i) ut[i] = C(utraj[i]);
    for (int i = 0; i < N * 5; ++i) yr[i] = C(yref[i]);
    opcount::reset();
    int it = 0;
    IpmStats s;
    const int st = rti_step<C>(*spec, w, x0c, yr.data(), yre.data(), xt.data(), ut.data(), pp.data(), &it, &s);
    const opcount::Counters& c = opcount::get();
    out9[0] = (double)c.add; out9[1] = (double)c.mul; out9[2] = (double)c.div;
    out9[3] = (double)c.sqrt; out9[4] = (double)c.trig; out9[5] = (double)c.cmp;
    out9[6] = (double)c.s_add; out9[7] = (double)c.s_mul; out9[8] = (double)c.s_div;
    if (qp_iter) *qp_iter = it;
    return st;
}

}  // extern "C"
