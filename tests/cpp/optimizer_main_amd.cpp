// BASELINE.json configs[0]: the 18 cm/s variant (mobile_robot_18: N = 40, Ts = 0.125) in the
// reference's offline closed loop, src/optimizer_main.cpp:3-16 — construct the Optimizer from the
// path files (src/optimizer.cpp:138-160), start on the first waypoint, run(), computeStats() —
// through the C++ drop-in (include/optimizer_amd.hpp).  The reference's own main uses the default
// constructor, whose v_ref = 1.0 names path files that do not exist (include/optimizer.hpp:29); the
// speed is an argument here.  run()'s loop (src/optimizer.cpp:177-206) is written out so that every
// tick can be printed and the run capped.
// TEST PROGRAM: run by tests/test_gpu_parity.py::test_config1_optimizer_main_on_path118.
//   usage: optimizer_main_amd paths_dir T v_ref max_ticks [use18_plant_as_written]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../include/optimizer_amd.hpp"

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s paths_dir T v_ref max_ticks [use18_plant_as_written]\n", argv[0]); return 2; }
    using namespace mpc_amd;
    const double T = atof(argv[2]), v_ref = atof(argv[3]);
    const int max_ticks = atoi(argv[4]);
    Optimizer controller(T, 40, v_ref, 0.83, 0.33, 1.65756, argv[1], "_path1");
    controller.use18_plant_as_written = argc > 5 && atoi(argv[5]) != 0;
    controller.x_current[0] = controller.state_refs(0, 0);
    controller.x_current[1] = controller.state_refs(0, 1);
    controller.x_current[2] = controller.state_refs(0, 2);
    printf("ctor %d %d %d %d %d %.17g\n", controller.N, controller.num_waypoints, (int)controller.input_refs.rows(),
           (int)controller.state_attributes.size(), (int)controller.normals.rows(), controller.density);
    int hsy = 0;
    const int last = (int)controller.state_refs.rows() - 1;
    while (hsy < max_ticks) {
        double e2 = 0.0;
        for (int q = 0; q < 3; ++q) { const double d = controller.state_refs(last, q) - controller.x_current[q]; e2 += d * d; }
        if (controller.target_waypoint_index > controller.num_waypoints || std::sqrt(e2) < 0.1) break;
        const int rc = controller.update_and_solve(controller.x_current);
        controller.integrate_next_states();
        printf("tick %d %d %.17g %.17g %.17g %.17g %.17g\n", rc, controller.target_waypoint_index, controller.u_current[0],
               controller.u_current[1], controller.x_current[0], controller.x_current[1], controller.x_current[2]);
        hsy++;
    }
    const VectorXd stats = controller.computeStats(0);
    printf("stats %d", controller.iter);
    for (int q = 0; q < 7; ++q) printf(" %.17g", stats(q));
    printf("\n");
    return 0;
}
