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
                                  int* status, int* qp_iter, int* diag) {
    if (!spec_valid(spec) || B < 1) return -1;
    const int N = spec->N;
    const Consts c = make_consts(*spec);
    const size_t S = n_waves_for(B) * 64;
    std::vector<double> slab(slab_bytes_for(N, B) / sizeof(double), 0.0);
    std::vector<int> islab(3 * S, 0);
    Mem m = make_mem(reinterpret_cast<char*>(slab.data()), islab.data(), B, N);
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
        }
        for (int k = 0; k <= N; ++k)
            for (int q = 0; q < 3; ++q) D[slab_index(m, i, k, R_X + q)] = xtraj[((size_t)i * (N + 1) + k) * 3 + q];
    }
    const int iters = spec->sqp_max_iter > 1 ? spec->sqp_max_iter : 1;
    for (size_t i = 0; i < (size_t)B; ++i) {
        Lane lane(c, m, i & ~(size_t)63, (unsigned)(i & 63));
        for (int q = 0; q < iters; ++q) {
            lane.rti_step(m);
            if (m.STATUS[i] != 0) break;
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
    }
    return 0;
}
