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
    const size_t S = (size_t)B;
    const Rows r = rows_for(N);
    std::vector<double> slab((size_t)r.total * S, 0.0);
    std::vector<int> islab(3 * S, 0);
    Mem m = carve(slab.data(), islab.data(), S, N);
    double* X0 = const_cast<double*>(m.X0);
    double* YREF = const_cast<double*>(m.YREF);
    double* YREFE = const_cast<double*>(m.YREFE);
    for (size_t i = 0; i < S; ++i) {
        for (int q = 0; q < 3; ++q) { X0[q * S + i] = x0[3 * i + q]; YREFE[q * S + i] = yref_e[3 * i + q]; }
        for (int q = 0; q < 5 * N; ++q) YREF[q * S + i] = yref[(size_t)i * 5 * N + q];
        for (int q = 0; q < 3 * (N + 1); ++q) m.X[q * S + i] = xtraj[(size_t)i * 3 * (N + 1) + q];
        for (int q = 0; q < 2 * N; ++q) m.U[q * S + i] = utraj[(size_t)i * 2 * N + q];
        if (pi) for (int q = 0; q < 3 * N; ++q) m.PI[q * S + i] = pi[(size_t)i * 3 * N + q];
    }
    const int iters = spec->sqp_max_iter > 1 ? spec->sqp_max_iter : 1;
    for (size_t i = 0; i < S; ++i) {
        Lane lane(c, m, i);
        for (int q = 0; q < iters; ++q) {
            lane.rti_step();
            if (m.STATUS[i] != 0) break;
        }
    }
    for (size_t i = 0; i < S; ++i) {
        for (int q = 0; q < 3 * (N + 1); ++q) xtraj[(size_t)i * 3 * (N + 1) + q] = m.X[q * S + i];
        for (int q = 0; q < 2 * N; ++q) utraj[(size_t)i * 2 * N + q] = m.U[q * S + i];
        if (pi) for (int q = 0; q < 3 * N; ++q) pi[(size_t)i * 3 * N + q] = m.PI[q * S + i];
        if (u0) { u0[2 * i] = m.U0[i]; u0[2 * i + 1] = m.U0[S + i]; }
        if (status) status[i] = m.STATUS[i];
        if (qp_iter) qp_iter[i] = m.QPITER[i];
        if (diag) diag[i] = m.DIAG[i];
    }
    return 0;
}
