// Closed loop through the C++ drop-in (include/optimizer_amd.hpp) on the path files given on the
// command line; prints one line per tick (target index, u, x) with 17 significant digits.
// TEST PROGRAM: run by tests/test_gpu_parity.py::test_cpp_optimizer_shim on the GPU box.
#include <cstdio>
#include <cstdlib>
#include "../../include/optimizer_amd.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s state_refs.txt input_refs.txt n_ticks\n", argv[0]); return 2; }
    using namespace mpc_amd;
    const MatrixXd s = Optimizer::loadTxt(argv[1]);
    const MatrixXd u = Optimizer::loadTxt(argv[2]);
    const int n = atoi(argv[3]);
    Optimizer opt(0.1, 40, 0.5, s(0, 0), s(0, 1), s(0, 2));
    opt.set_path(s, u);
    for (int k = 0; k < n; ++k) {
        const int rc = opt.update_and_solve(opt.x_current);
        opt.integrate_next_states();
        printf("%d %d %.17g %.17g %.17g %.17g %.17g\n", rc, opt.target_waypoint_index, opt.u_current[0], opt.u_current[1],
               opt.x_current[0], opt.x_current[1], opt.x_current[2]);
    }
    // park manoeuvre
    const double xs[5] = {0.63, 0.32, 0.0, 0.0, 0.0};
    opt.x_current[0] = 3.0; opt.x_current[1] = 2.0; opt.x_current[2] = 0.05;
    opt.set_up_park(xs);
    int rc = 0, k = 0;
    for (; k < 400; ++k) {
        rc = opt.update_and_solve_park(xs);
        if (rc != 0) break;
        mpc_ocp_spec_t sp; mpc_spec_variant("park", &sp);
        // plant step with the park variant's Ts (same ERK4)
        static mpc_batch_t* sim = nullptr;
        if (!sim) mpc_batch_create(&sim, &sp, 1, nullptr, 0);
        mpc_batch_sim_step(sim, 1, opt.x_current.data(), opt.u_current);
    }
    printf("park %d %d %.17g %.17g %.17g\n", rc, k, opt.x_current_transformed[0], opt.x_current_transformed[1], opt.x_current_transformed[2]);
    return 0;
}
