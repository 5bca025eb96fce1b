// Every `mpc.<member>` that the reference's src/controller.cpp names, called once through the C++
// drop-in (include/optimizer_amd.hpp) in the order a run of the state machine meets them: waypoints
// assigned member by member (controller.cpp:210-226), attribute queries (:1183,:1290), direction
// indices (:567-800), published window (:834), lane change (:964), the tick (:1057), a manoeuvre on
// a turn table in the transformed frame (:1515-1550), parking (:421-425), statistics (:1645).
// Prints one line per step with 17 significant digits.
// TEST PROGRAM: run by tests/test_gpu_parity.py::test_cpp_optimizer_surface on the GPU box, which
// repeats the same sequence on the Python mirror driving the same library.
#include <cstdio>
#include <cstdlib>
#include "../../include/optimizer_amd.hpp"

using namespace mpc_amd;

static void p3(const char* tag, const Vector3d& v) { printf("%s %.17g %.17g %.17g\n", tag, v[0], v[1], v[2]); }

int main(int argc, char** argv) {
#ifdef MPC_HAVE_EIGEN
    fprintf(stderr, "vector types: <Eigen/Dense>\n");
#else
    fprintf(stderr, "vector types: stand-in\n");
#endif
    if (argc < 3) { fprintf(stderr, "usage: %s state_refs.txt input_refs.txt\n", argv[0]); return 2; }
    const MatrixXd s = Optimizer::loadTxt(argv[1]);
    const MatrixXd u = Optimizer::loadTxt(argv[2]);
    const int M = (int)s.rows();
    VectorXd attrs(M);
    MatrixXd nrm(M, 2);
    for (int i = 0; i < M; ++i) {
        attrs(i) = (i / 7) % 10 + ((i / 31) % 2 ? 100 : 0);
        nrm(i, 0) = -std::sin(s(i, 2)); nrm(i, 1) = std::cos(s(i, 2));
    }

    Optimizer mpc(0.1, 40, 0.5);
    mpc.state_refs = s; mpc.input_refs = u; mpc.state_attributes = attrs; mpc.normals = nrm;   // :210-213
    mpc.initialize_current_states(s(0, 0) + 0.01, s(0, 1) - 0.02, s(0, 2) + 4 * M_PI);          // :222
    p3("init", mpc.x_current);
    mpc.target_waypoint_index = mpc.find_closest_waypoint(0, (int)mpc.state_refs.rows() - 1);   // :226
    printf("reset %d\n", mpc.reset_solver());
    printf("dims %d %d %d %.17g %.17g %d %d\n", mpc.N, mpc.target_waypoint_index, mpc.last_waypoint_index, mpc.T, mpc.density,
           mpc.num_waypoints, (int)mpc.input_refs.rows());

    for (int i : {-1, 0, 13, 14, 21, 35, 44, 63, 70, M - 1, M})
        printf("attr %d %d %d %d %d\n", i, (int)mpc.attribute_cmp(i, Optimizer::ATTRIBUTE::CROSSWALK),
               (int)mpc.attribute_cmp(i, Optimizer::ATTRIBUTE::INTERSECTION), (int)mpc.attribute_cmp(i, Optimizer::ATTRIBUTE::DOTTED_CROSSWALK),
               (i >= 0 && i < M) ? (int)mpc.is_not_detectable(i) : -1);
    printf("enum %d %d %d %d %d %d %d %d %d %d\n", Optimizer::NORMAL, Optimizer::CROSSWALK, Optimizer::INTERSECTION, Optimizer::ONEWAY,
           Optimizer::HIGHWAYLEFT, Optimizer::HIGHWAYRIGHT, Optimizer::ROUNDABOUT, Optimizer::STOPLINE, Optimizer::DOTTED,
           Optimizer::DOTTED_CROSSWALK);
    for (double yaw : {-7.0, -3.2, -1.5, -0.8, 0.0, 0.7, 0.79, 1.6, 2.4, 3.1, 3.95, 4.7, 5.5, 6.2, 6.3, 13.0})
        printf("dir %.17g %d %.17g\n", yaw, mpc.NearestDirectionIndex(yaw), mpc.NearestDirection(yaw));

    MatrixXd waypoints(mpc.N, 3);
    mpc.get_current_waypoints(waypoints);                                                      // :834
    double acc = 0;
    for (int j = 0; j < (int)waypoints.rows(); ++j) acc += waypoints(j, 0) + 2 * waypoints(j, 1) + 3 * waypoints(j, 2);
    printf("window %d %.17g %d\n", (int)waypoints.rows(), acc, mpc.get_current_attribute());

    mpc.change_lane(20, 70, true, 0.3);                                                        // :964
    mpc.change_lane(90, 120);
    for (int i : {19, 20, 23, 45, 66, 69, 70, 95, 119})
        printf("lane %d %.17g %.17g\n", i, mpc.state_refs(i, 0), mpc.state_refs(i, 1));

    for (int k = 0; k < 30; ++k) {                                                             // :1057
        const int rc = mpc.update_and_solve(mpc.x_current, true);
        mpc.integrate_next_states();
        // the pose comes back through update_current_states with a wrapped yaw (controller.cpp update_mpc_state)
        mpc.update_current_states(mpc.x_current[0], mpc.x_current[1], mpc.x_current[2] - 2 * M_PI, true);
        printf("tick %d %d %.17g %.17g %.17g %.17g %.17g\n", rc, mpc.target_waypoint_index, mpc.u_current[0], mpc.u_current[1],
               mpc.x_current[0], mpc.x_current[1], mpc.x_current[2]);
    }

    // manoeuvre on a turn table in the frame of the crossing (:1510-1550)
    const int base = mpc.target_waypoint_index;
    MatrixXd turn(60, 3);
    for (int i = 0; i < 60; ++i) {
        Vector3d pt, q;
        for (int c = 0; c < 3; ++c) pt[c] = mpc.state_refs(base + i, c);
        Vector3d f1, f2;
        f1[0] = mpc.state_refs(base, 0); f1[1] = mpc.state_refs(base, 1); f1[2] = mpc.NearestDirection(mpc.state_refs(base, 2));
        f2[0] = f2[1] = f2[2] = 0.0;
        mpc.transform_point(pt, q, f1, f2);
        for (int c = 0; c < 3; ++c) turn(i, c) = q[c];
        if (i == 0) { mpc.frame1 = f1; mpc.frame2 = f2; }
    }
    mpc.left_turn_states = turn;
    mpc.right_turn_states = turn;
    MatrixXd* state_refs_ptr = &mpc.left_turn_states;                                           // :1515
    printf("table %d\n", (int)state_refs_ptr->rows());
    const int keep_target = mpc.target_waypoint_index, keep_last = mpc.last_waypoint_index;
    mpc.target_waypoint_index = 0;
    mpc.last_waypoint_index = 0;
    for (int mode : {0, 2, 1, 7}) {
        mpc.transform_point(mpc.x_current, mpc.x_current_transformed, mpc.frame1, mpc.frame2);   // :1537
        const int rc = mpc.update_and_solve(mpc.x_current_transformed, true, mode);
        printf("turn %d %d %d %.17g %.17g\n", mode, rc, mpc.target_waypoint_index, mpc.u_current[0], mpc.u_current[1]);
        p3("xt", mpc.x_current_transformed);
        if (mode == 2) { mpc.target_waypoint_index = keep_target; mpc.last_waypoint_index = keep_last; }
    }

    // parking with the reference's VectorXd signatures (:415-430)
    VectorXd xs(5);
    xs(0) = 0.63; xs(1) = 0.32; xs(2) = 0.0; xs(3) = 0.0; xs(4) = 0.0;
    const double thresh = 0.1;
    mpc.set_up_park(xs);
    p3("frame1", mpc.frame1);
    p3("frame2", mpc.frame2);
    mpc_ocp_spec_t sp;
    mpc_spec_variant("park", &sp);
    mpc_batch_t* sim = nullptr;
    if (mpc_batch_create(&sim, &sp, 1, nullptr, 0)) return 3;
    int status = 0, i = 0;
    while (1) {
        status = mpc.update_and_solve_park(xs, thresh * thresh);
        if (status == 2 || i > 200) break;
        mpc_batch_sim_step(sim, 1, mpc.x_current.data(), mpc.u_current);
        i++;
    }
    mpc_batch_free(sim);
    printf("park %d %d\n", status, i);
    p3("xt", mpc.x_current_transformed);

    // computeStats as the reference computes it (zero-padded 2*num_waypoints buffers, src/optimizer.cpp:496-499;
    // the tick timing of :476-478), then over the recorded ticks only
    mpc.num_waypoints = (int)mpc.state_refs.rows();   // what the constructor sets when it loads the path files (:160)
    const VectorXd stats = mpc.computeStats(357);                                              // :1645
    printf("stats %d", mpc.iter);
    for (int q = 0; q < 7; ++q) printf(" %.17g", stats(q));
    printf("\nsim %d %d %.17g %.17g\n", (int)mpc.simX.rows(), (int)mpc.simU.rows(), mpc.simX(3, 1), mpc.simU(3, 0));
    printf("time %d %d %.17g %.17g %.17g\n", (int)mpc.time_record.rows(), (int)mpc.time_record.cols(), mpc.time_stats[0], mpc.time_stats[1], mpc.time_stats[2]);
    double tsum = 0.0;
    for (int k = 0; k < (int)mpc.time_record.rows(); ++k) tsum += mpc.time_record(k, 0);
    mpc.reference_stats = false;
    const VectorXd stats2 = mpc.computeStats(357);
    printf("stats_ticks %d", mpc.iter);
    for (int q = 0; q < 7; ++q) printf(" %.17g", stats2(q));
    printf("\ntime %d %d %.17g %.17g %.17g\n", (int)mpc.time_record.rows(), (int)mpc.time_record.cols(), mpc.time_stats[0], mpc.time_stats[1], mpc.time_stats[2]);
    printf("timesum %.17g\n", tsum);
    return 0;
}
