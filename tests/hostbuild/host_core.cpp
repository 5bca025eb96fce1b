// host_core.cpp — TEST INFRASTRUCTURE: compiles the product's per-lane algorithm
// (mpc_amd/csrc/rti_core.hpp) for the host so that tests can compare it with the dense oracle
// without a GPU and debug it with gdb.  It is never loaded by the product package; the product
// (libmpc_amd.so) runs this algorithm only as HIP kernels.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../mpc_amd/csrc/consts.hpp"

using namespace mpc;

extern "C" int hostcore_rti_batch(const mpc_ocp_spec_t* spec, int B, const double* x0, const double* yref,
                                  const double* yref_e, double* xtraj, double* utraj, double* pi, double* u0,
                                  int* status, int* qp_iter, int* diag, double* lam, int* sqp_iter, double* res4) {
    if (!spec_valid(spec) || B < 1) return -1;
    const int N = spec->N;
    const Consts c = make_consts(*spec);
    const size_t S = n_waves_for(B) * 64;
    std::vector<double> slab(slab_bytes_for(N, B) / sizeof(double), 0.0);
    std::vector<int> islab(islab_ints_for(B), 0);
    (void)S;
    Mem m = make_mem(islab.data(), B, N);
    m.slab = reinterpret_cast<char*>(slab.data());
    double* D = slab.data();
    for (size_t i = 0; i < (size_t)B; ++i) {
        for (int q = 0; q < 3; ++q) {
            D[slab_index(m, i, N, R_X0 + q)] = x0[3 * i + q];
            D[slab_index(m, i, N, R_YREFE + q)] = yref_e[3 * i + q];
        }
        for (int k = 0; k < N; ++k) {
            for (int q = 0; q < 5; ++q) D[slab_index(m, i, k, R_YREF + q)] = yref[((size_t)i * N + k) * 5 + q];
            for (int q = 0; q < 2; ++q) D[slab_index(m, i, k, R_U + q)] = utraj[((size_t)i * N + k) * 2 + q];
            if (pi) for (int q = 0; q < 3; ++q) D[slab_index(m, i, k, R_PI + q)] = pi[((size_t)i * N + k) * 3 + q];
            if (lam) for (int q = 0; q < 8; ++q) D[slab_index(m, i, k, R_LAM + q)] = lam[((size_t)i * (N + 1) + k) * 8 + q];
        }
        for (int k = 0; k <= N; ++k)
            for (int q = 0; q < 3; ++q) D[slab_index(m, i, k, R_X + q)] = xtraj[((size_t)i * (N + 1) + k) * 3 + q];
    }
    for (size_t i = 0; i < (size_t)B; ++i) {
        Lane lane(c, m, i & ~(size_t)63, (unsigned)(i & 63));
        if (!c.sqp) {
            lane.rti_step(m);
            continue;
        }
        // full SQP: the per-instance sequence of k_nlp_res + k_phase rounds of mpc_batch.hip
        m.ACTIVE[i] = 1; m.SQPITER[i] = 0; m.QPITER[i] = 0; m.STATUS[i] = 2;
        for (int it = 0; it < spec->sqp_max_iter && m.ACTIVE[i]; ++it) {
            double r[4];
            lane.nlp_res_and_linearize(r);
            for (int q = 0; q < 4; ++q) m.NLPRES[4 * i + q] = r[q];
            if (r[0] < c.nlp_tol && r[1] < c.nlp_tol && r[2] < c.nlp_tol && r[3] < c.nlp_tol) {
                m.ACTIVE[i] = 0; m.STATUS[i] = 0;
                lane.st(R_U0, N, 0, lane.ld(R_U, 0, 0));
                lane.st(R_U0, N, 1, lane.ld(R_U, 0, 1));
                break;
            }
            lane.rti_step(m);
        }
    }
    for (size_t i = 0; i < (size_t)B; ++i) {
        for (int k = 0; k <= N; ++k)
            for (int q = 0; q < 3; ++q) xtraj[((size_t)i * (N + 1) + k) * 3 + q] = D[slab_index(m, i, k, R_X + q)];
        for (int k = 0; k < N; ++k) {
            for (int q = 0; q < 2; ++q) utraj[((size_t)i * N + k) * 2 + q] = D[slab_index(m, i, k, R_U + q)];
            if (pi) for (int q = 0; q < 3; ++q) pi[((size_t)i * N + k) * 3 + q] = D[slab_index(m, i, k, R_PI + q)];
        }
        if (u0) { u0[2 * i] = D[slab_index(m, i, N, R_U0)]; u0[2 * i + 1] = D[slab_index(m, i, N, R_U0 + 1)]; }
        if (status) status[i] = m.STATUS[i];
        if (qp_iter) qp_iter[i] = m.QPITER[i];
        if (diag) diag[i] = m.DIAG[i];
        if (sqp_iter) sqp_iter[i] = m.SQPITER[i];
        if (res4) for (int q = 0; q < 4; ++q) res4[4 * i + q] = m.NLPRES[4 * i + q];
        if (lam)
            for (int k = 0; k < N; ++k)
                for (int q = 0; q < 8; ++q) lam[((size_t)i * (N + 1) + k) * 8 + q] = D[slab_index(m, i, k, R_LAM + q)];
    }
    return 0;
}
