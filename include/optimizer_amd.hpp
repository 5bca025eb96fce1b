// optimizer_amd.hpp — C++ drop-in for the reference's `Optimizer` front-end
// (include/optimizer.hpp:25-457, src/optimizer.cpp) on top of libmpc_amd.so (mpc_batch.h).
//
// Same public names, argument meaning and return codes as the reference class:
//   Optimizer(T, N, v_ref, x_init, y_init, yaw_init); run(); update_and_solve(x, safety_check, mode)
//   -> 0 ok / 1 solver error, result in u_current[2]; integrate_next_states(); reset_solver();
//   find_next_waypoint / find_closest_waypoint; update_current_states; set_up_park /
//   update_and_solve_park -> 0/1/2; transform_point; NearestDirection; loadTxt / saveToFile.
// Differences, on purpose: the acados capsules, nlp_* pointers and string-keyed setter calls are
// replaced by two mpc_batch handles of capacity 1; function-statics (limit, lookahead, count)
// are members, so several Optimizer objects can coexist; the path files are looked up in
// `paths_dir` (constructor argument / $MPC_PATHS_DIR) instead of a path derived from __FILE__.
//
// Vector types: with Eigen available (the reference's build) they are Eigen::Vector3d / VectorXd /
// MatrixXd themselves (column-major, so `Eigen::MatrixXd* p = &mpc.left_turn_states` and
// `mpc.state_refs = Eigen::Map<...>(...).transpose()` of src/controller.cpp:210-213,1515 keep their
// types); without Eigen a minimal stand-in with the members this class needs is used.
// Every `mpc.<member>` that src/controller.cpp names exists here with the reference's signature;
// tests/cpp/optimizer_surface_main.cpp calls each of them (stand-in types: this image has neither
// Eigen nor ROS, so a build of controller.cpp itself against this header has NOT been made).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpc_batch.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(MPC_NO_EIGEN)
#include <Eigen/Dense>
#define MPC_HAVE_EIGEN 1
#endif
#endif

namespace mpc_amd {

#ifdef MPC_HAVE_EIGEN
using Vector3d = Eigen::Vector3d;
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
#else
struct Vector3d {
    double v[3] = {0, 0, 0};
    Vector3d() {}
    Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    const double& operator()(int i) const { return v[i]; }
    double* data() { return v; }
    const double* data() const { return v; }
};
struct VectorXd {
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(int n) : v(n, 0.0) {}
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    const double& operator()(int i) const { return v[i]; }
    int size() const { return (int)v.size(); }
    int rows() const { return (int)v.size(); }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
};
struct MatrixXd {   // row-major
    int r = 0, c = 0;
    std::vector<double> v;
    MatrixXd() {}
    MatrixXd(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return v[(size_t)i * c + j]; }
    const double& operator()(int i, int j) const { return v[(size_t)i * c + j]; }
    int rows() const { return r; }
    int cols() const { return c; }
};
#endif

class Optimizer {
  public:
    Optimizer(double T, int N, double v_ref, double x_init = 0.83, double y_init = 0.33, double yaw_init = 1.65756,
              const std::string& paths_dir = "", const std::string& path_name = "_path1")
        : T(T), v_ref(v_ref) {
        (void)N;   // the horizon comes from the solver variant (src/optimizer.cpp:117-123)
        density = 1 / T / v_ref;
        region_of_acceptance = 0.03076923 * 3 * (0.125 * 1.3) / density;
        t0 = 0.0;
        v_ref_int = (int)(v_ref * 100);
        const char* variant;
        if (v_ref_int >= 30) variant = "mobile_robot";
        else if (v_ref_int == 25) { variant = "mobile_robot_25"; use25 = true; }
        else if (v_ref_int == 18) { variant = "mobile_robot_18"; use18 = true; }
        else { std::cerr << "Invalid reference speed, please use 18, 25 or 50" << std::endl; exit(1); }
        mpc_ocp_spec_t spec, spec_park;
        if (mpc_spec_variant(variant, &spec) || mpc_spec_variant("park", &spec_park)) exit(1);
        if (mpc_batch_create(&mpc_, &spec, 1, nullptr, 0)) { printf("mpc_batch_create() failed. Exiting.\n"); exit(1); }
        if (mpc_batch_create(&mpc_park_, &spec_park, 1, nullptr, 0)) { printf("mpc_batch_create(park) failed. Exiting.\n"); exit(1); }
        this->N = spec.N; nx = MPC_NX; nu = MPC_NU;
        N_park = spec_park.N; nx_park = MPC_NX; nu_park = MPC_NU;
        x_current[0] = x_init; x_current[1] = y_init; x_current[2] = yaw_init;
        x_real = x_current;
        std::string dir = paths_dir;
        if (dir.empty() && getenv("MPC_PATHS_DIR")) dir = getenv("MPC_PATHS_DIR");
        if (!dir.empty()) {
            const std::string sfx = path_name + std::to_string(v_ref_int) + ".txt";
            state_refs = loadTxt(dir + "/state_refs" + sfx);
            input_refs = loadTxt(dir + "/input_refs" + sfx);
            const MatrixXd a = loadTxt(dir + "/wp_attributes" + sfx);   // src/optimizer.cpp:146
            state_attributes = VectorXd((int)a.rows());
            for (int i = 0; i < (int)a.rows(); ++i) state_attributes(i) = a(i, 0);
            normals = loadTxt(dir + "/wp_normals" + sfx);               // :159
            set_path_loaded();
        }
        if (this->v_ref > 0.3) this->v_ref = 0.35;   // src/optimizer.cpp:174
    }
    ~Optimizer() { mpc_batch_free(mpc_); mpc_batch_free(mpc_park_); mpc_batch_free(mpc_plant50_); }
    Optimizer(const Optimizer&) = delete;
    Optimizer& operator=(const Optimizer&) = delete;

    // path given directly instead of files (what src/controller.cpp:210-213 does by assignment)
    void set_path(const MatrixXd& s, const MatrixXd& u) { state_refs = s; input_refs = u; set_path_loaded(); }
    void set_path(const MatrixXd& s, const MatrixXd& u, const VectorXd& attributes, const MatrixXd& wp_normals) {
        state_refs = s; input_refs = u; state_attributes = attributes; normals = wp_normals; set_path_loaded();
    }

    int run() {   // src/optimizer.cpp:177-206
        int hsy = 0;
        while (true) {
            double e2 = 0;
            for (int q = 0; q < 3; ++q) { const double d = state_refs(num_waypoints - 1, q) - x_current[q]; e2 += d * d; }
            if (target_waypoint_index > num_waypoints || std::sqrt(e2) < 0.1) break;
            update_and_solve(x_current);
            integrate_next_states();
            hsy++;
        }
        return 0;
    }

    int update_and_solve(Vector3d& i_current_state, bool safety_check = true, int mode = -1) {   // :208-319
        const auto t_start = std::chrono::high_resolution_clock::now();   // debug timer (:214)
        // mode selects the waypoint table (:215-226); a table that was never filled (the reference
        // leaves all three empty, :147-149, and would index out of range) falls back to state_refs
        if (mode == -1) state_refs_ptr = &state_refs;
        else if (mode == 0) state_refs_ptr = &left_turn_states;
        else if (mode == 1) state_refs_ptr = &straight_states;
        else if (mode == 2) state_refs_ptr = &right_turn_states;
        else { std::cerr << "Invalid mode, proceed with default mode" << std::endl; state_refs_ptr = &state_refs; }
        if (state_refs_ptr->rows() == 0) state_refs_ptr = &state_refs;
        const MatrixXd& refs = *state_refs_ptr;
        find_next_waypoint(target_waypoint_index, i_current_state, safety_check);
        const int rows = (int)refs.rows();
        const int idx = target_waypoint_index < (int)state_refs.rows() ? target_waypoint_index : (int)state_refs.rows() - 1;
        std::vector<double> yref((size_t)N * 5, 0.0);
        double yref_e[3] = {refs(target_waypoint_index, 0), refs(target_waypoint_index, 1),
                            refs(target_waypoint_index, 2)};   // terminal ref = first waypoint (:244-251)
        for (int j = 0; j < N; ++j) {
            double* y = &yref[(size_t)j * 5];
            if (target_waypoint_index + j >= rows) {
                for (int q = 0; q < 3; ++q) y[q] = refs(rows - 1, q);
            } else {
                for (int q = 0; q < 3; ++q) y[q] = refs(idx + j, q);
                y[3] = input_refs(idx + j, 0);
            }
            y[4] = 0.0;   // "set steer rate to 0" (:271)
        }
        double u0[2];
        int st = 0;
        if (mpc_batch_solve(mpc_, 1, i_current_state.data(), yref.data(), yref_e, u0, nullptr, nullptr, &st, nullptr)) st = 4;
        if (shift_warm_start) mpc_batch_shift(mpc_, nullptr);
        status = st;
        if (status != 0) {
            std::cout << "ERROR!!! acados acados_ocp_solver returned status " << status << ". Exiting." << std::endl;
            return 1;
        }
        u_current[0] = u0[0]; u_current[1] = u0[1];
        if (debug) {   // :300-317
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_start).count();
            record_tick(i_current_state, ms);
        }
        return 0;
    }

    // One ERK4 step of the plant (:321-341).  The step length is the solver variant's own Ts.
    // For the 18 cm/s variant the reference's code does not say that: its else-branch (:328-332) calls
    // mobile_robot_acados_sim_solve(sim_capsule) — the N = 50 variant's integrator, Ts = 0.05 — on a
    // capsule that the use18 constructor branch never created (:87-110), i.e. undefined behaviour that
    // cannot be reproduced.  `use18_plant_as_written` = true takes the branch by its letter, as if that
    // capsule had been created: the plant then advances 0.05 s per tick while the OCP assumes 0.125 s.
    void integrate_next_states() {
        if (use18 && use18_plant_as_written) {
            if (!mpc_plant50_) {
                mpc_ocp_spec_t sp;
                if (mpc_spec_variant("mobile_robot", &sp) || mpc_batch_create(&mpc_plant50_, &sp, 1, nullptr, 0))
                    throw std::runtime_error("acados integrator returned status 1. Exiting.");
            }
            if (mpc_batch_sim_step(mpc_plant50_, 1, x_current.data(), u_current))
                throw std::runtime_error("acados integrator returned status 1. Exiting.");
        } else if (mpc_batch_sim_step(mpc_, 1, x_current.data(), u_current)) {
            throw std::runtime_error("acados integrator returned status 1. Exiting.");   // :333-335
        }
        t0 += T;
    }

    int find_closest_waypoint(int min_index = -1, int max_index = -1) {   // :343-382
        const int rows = state_refs.rows();
        if (limit_ < 0) limit_ = (int)std::floor(rdb_circumference / (v_ref * T));
        if (min_index < 0) min_index = std::min(last_waypoint_index, rows - 1);
        if (max_index < 0) max_index = std::min(target_waypoint_index + limit_, rows - 1);
        const double cn = x_current[0] * x_current[0] + x_current[1] * x_current[1];
        double best = std::numeric_limits<double>::max();
        int closest = -1;
        for (int i = max_index; i >= min_index; --i) {
            const double px = state_refs(i, 0), py = state_refs(i, 1);
            const double d = (px * px + py * py) - 2 * (px * x_current[0] + py * x_current[1]) + cn;
            if (d < best) { best = d; closest = i; }
        }
        return closest;
    }

    int find_next_waypoint(int& output_target, const Vector3d& i_current_state, bool safety_check = true,
                           int min_index = -1, int max_index = -1) {   // :384-431
        (void)safety_check;
        if (limit_ < 0) limit_ = (int)std::floor(rdb_circumference / (v_ref * T));
        closest_waypoint_index = find_closest_waypoint(min_index, max_index);
        const double dx = state_refs(closest_waypoint_index, 0) - i_current_state[0];
        const double dy = state_refs(closest_waypoint_index, 1) - i_current_state[1];
        const double dist = std::sqrt(dx * dx + dy * dy);
        if (dist > 1.2) {
            min_index = (int)std::max(closest_waypoint_index - dist * density * 1.2, 0.0);
            closest_waypoint_index = find_closest_waypoint(min_index, max_index);
        }
        int target;
        if (count_ >= 8) { target = closest_waypoint_index + lookahead_; count_ = 0; }
        else { target = target_waypoint_index + 1; count_++; }
        output_target = std::min(target, (int)state_refs_ptr->rows() - 1);
        last_waypoint_index = output_target;
        return 1;
    }

    int update_current_states(double x, double y, double yaw, Vector3d& state, bool safety_check = true) {   // :432-455
        (void)safety_check;
        if (closest_waypoint_index < state_refs.rows() && state_refs.rows() > 0) {
            const double ref_yaw = state_refs(closest_waypoint_index, 2);
            while (ref_yaw - yaw > M_PI) yaw += 2 * M_PI;
            while (ref_yaw - yaw < -M_PI) yaw -= 2 * M_PI;
        }
        state[0] = x; state[1] = y; state[2] = yaw;
        return 1;
    }
    int update_current_states(double x, double y, double yaw, bool safety_check = true) {
        return update_current_states(x, y, yaw, x_current, safety_check);
    }
    int initialize_current_states(double x, double y, double yaw) {   // include/optimizer.hpp:62-64
        return update_current_states(x, y, yaw, x_current, false);
    }

    // ---- waypoint attributes (include/optimizer.hpp:94-105) ----
    enum ATTRIBUTE {
        NORMAL, CROSSWALK, INTERSECTION, ONEWAY, HIGHWAYLEFT, HIGHWAYRIGHT, ROUNDABOUT, STOPLINE, DOTTED, DOTTED_CROSSWALK
    };
    // an attribute + 100 marks the same kind of waypoint where signs / lines cannot be detected
    bool attribute_cmp(int idx, int attr) const {
        if (idx < 0 || idx >= (int)state_attributes.size()) return false;
        const double a = state_attributes(idx);
        return a == attr || a == attr + 100;
    }
    bool is_not_detectable(int idx) const {
        return state_attributes(idx) >= 100 || attribute_cmp(idx, DOTTED_CROSSWALK) || attribute_cmp(idx, INTERSECTION) ||
               attribute_cmp(idx, ROUNDABOUT);
    }
    int get_current_attribute() const { return (int)state_attributes(target_waypoint_index); }   // :440-442
    // the N waypoints from the target on (:443-445); like the reference's block(), the caller keeps
    // target_waypoint_index + N inside the table
    void get_current_waypoints(MatrixXd& output) const {
        if (target_waypoint_index + N > (int)state_refs.rows()) throw std::out_of_range("get_current_waypoints: window past the end of state_refs");
        if ((int)output.rows() != N || (int)output.cols() != 3) output = MatrixXd(N, 3);
        for (int j = 0; j < N; ++j) for (int q = 0; q < 3; ++q) output(j, q) = state_refs(target_waypoint_index + j, q);
    }
    // lateral shift of the waypoints [start_index, end_index) along their normals, ramped in and out
    // over int(density / 8) waypoints (:335-372); to the left unless shift_right
    void change_lane(int start_index, int end_index, bool shift_right = false, double shift_distance = 0.36 - 0.1) {
        if (shift_right) shift_distance *= -1;
        const int ramp_length = static_cast<int>(density * 0.125);
        const int ramp_up_end = start_index + ramp_length, ramp_down_start = end_index - ramp_length;
        for (int i = start_index; i < end_index; i++) {
            double shift = shift_distance;
            if (i < ramp_up_end) shift = shift_distance * (static_cast<double>(i - start_index + 1) / ramp_length);
            else if (i >= ramp_down_start) shift = shift_distance * (static_cast<double>(end_index - i) / ramp_length);
            state_refs(i, 0) += normals(i, 0) * shift;
            state_refs(i, 1) += normals(i, 1) * shift;
        }
    }

    int reset_solver() { return mpc_batch_reset(mpc_, nullptr); }   // include/optimizer.hpp:446-456

    // ---- parking (include/optimizer.hpp:140-267) ----
    void transform_point(const Vector3d& pt, Vector3d& pt_transformed, const Vector3d& f1, const Vector3d& f2) const {
        const double xt = pt[0] - f1[0], yt = pt[1] - f1[1];
        const double rot = f2[2] - f1[2];
        const double c = std::cos(rot), s = std::sin(rot);
        const double rx = c * xt - s * yt, ry = s * xt + c * yt;
        pt_transformed[2] = std::fmod(pt[2] + rot, 2 * M_PI);
        pt_transformed[0] = rx + f2[0];
        pt_transformed[1] = ry + f2[1];
    }
    static double NearestDirection(double yaw) {
        while (yaw > 2 * M_PI) yaw -= 2 * M_PI;
        while (yaw < 0) yaw += 2 * M_PI;
        static const double directions[5] = {0, M_PI / 2, M_PI, 3 * M_PI / 2, 2 * M_PI};
        double minDifference = std::abs(yaw - directions[0]);
        double nearest = directions[0];
        for (int i = 1; i < 5; ++i) {
            const double d = std::abs(yaw - directions[i]);
            if (d < minDifference) { minDifference = d; nearest = directions[i]; }
        }
        while (nearest - yaw > M_PI) nearest -= 2 * M_PI;
        while (nearest - yaw < -M_PI) nearest += 2 * M_PI;
        return nearest;
    }
    static int NearestDirectionIndex(double yaw) {   // :268-294: 0 east, 1 north, 2 west, 3 south
        while (yaw > 2 * M_PI) yaw -= 2 * M_PI;
        while (yaw < 0) yaw += 2 * M_PI;
        int best = 0;
        double best_d = std::abs(yaw);
        for (int i = 1; i < 5; ++i) {
            const double d = std::abs(yaw - i * (M_PI / 2));
            if (d < best_d) { best_d = d; best = i; }
        }
        return best == 4 ? 0 : best;
    }
    // the reference's signatures (Eigen::VectorXd&, include/optimizer.hpp:167,191)
    void set_up_park(const VectorXd& xs) { set_up_park(xs.data()); }
    int update_and_solve_park(const VectorXd& xs, double thresh_sq = 0.0025) { return update_and_solve_park(xs.data(), thresh_sq); }
    void set_up_park(const double xs[5]) {
        mpc_batch_reset(mpc_park_, nullptr);
        for (int q = 0; q < 5; ++q) park_xs_[q] = xs[q];
        frame1[0] = x_current[0]; frame1[1] = x_current[1]; frame1[2] = NearestDirection(x_current[2]);
        frame2[0] = frame2[1] = frame2[2] = 0.0;
    }
    int update_and_solve_park(const double xs[5], double thresh_sq = 0.0025) {
        const auto t_start = std::chrono::high_resolution_clock::now();   // include/optimizer.hpp:192
        transform_point(x_current, x_current_transformed, frame1, frame2);
        while (x_current_transformed[2] - frame2[2] > M_PI) x_current_transformed[2] -= 2 * M_PI;
        while (x_current_transformed[2] - frame2[2] < -M_PI) x_current_transformed[2] += 2 * M_PI;
        double e2 = 0;
        for (int q = 0; q < 3; ++q) { const double d = x_current_transformed[q] - xs[q]; e2 += d * d; }
        if (e2 < thresh_sq) return 2;
        std::vector<double> yref((size_t)N_park * 5);
        for (int j = 0; j < N_park; ++j) for (int q = 0; q < 5; ++q) yref[(size_t)j * 5 + q] = park_xs_[q];
        double u0[2];
        int st = 0;
        if (mpc_batch_solve(mpc_park_, 1, x_current_transformed.data(), yref.data(), park_xs_, u0, nullptr, nullptr, &st, nullptr)) st = 4;
        status = st;
        if (status != 0) return 1;
        u_current[0] = u0[0]; u_current[1] = u0[1];
        if (debug) record_tick(x_current, std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_start).count());   // :223-236
        return 0;
    }

    // ---- statistics of the recorded run (src/optimizer.cpp:472-518) ----
    // [mean speed, mean steer, mean |d speed|, mean |d steer|, mean |x err|, mean |y err|, mean |yaw err|]
    // and the tick timing the reference prints (:476-478): time_stats = {mean, max, min} in ms.
    // reference_stats = true (default) gives the reference's numbers: its error buffers are
    // 2*num_waypoints long and zero beyond the recorded ticks, and .mean() runs over all of them
    // (:166-171, 496-499), so the three error means are sum / (2*num_waypoints); its time_record is a
    // (2*num_waypoints) x nu matrix written through the linear index (:172, 304) and resized to
    // iter x nu (:475), so "average" is sum / (nu*iter) and "min" is 0 (the second column is never
    // written).  reference_stats = false averages over the recorded ticks only.  A run longer than
    // 2*num_waypoints ticks overruns the reference's buffers; here the record grows.
    // Files are written only when `results_dir` is set (the reference writes next to its sources).
    VectorXd computeStats(int hsy = 0) {
        (void)hsy;
        const int n = iter;
        VectorXd stats(7);
        for (int q = 0; q < 7; ++q) stats(q) = 0.0;
        time_stats[0] = time_stats[1] = time_stats[2] = 0.0;
        if (n == 0) return stats;
        double tsum = 0.0, tmax = time_ms_[0], tmin = time_ms_[0];
        for (int k = 0; k < n; ++k) {
            stats(0) += simU_[2 * k]; stats(1) += simU_[2 * k + 1];
            if (k) { stats(2) += std::abs(simU_[2 * k] - simU_[2 * k - 2]); stats(3) += std::abs(simU_[2 * k + 1] - simU_[2 * k - 1]); }
            stats(4) += std::abs(err_[3 * k]); stats(5) += std::abs(err_[3 * k + 1]);
            stats(6) += std::abs(std::atan2(std::sin(err_[3 * k + 2]), std::cos(err_[3 * k + 2])));
            tsum += time_ms_[k]; tmax = std::max(tmax, time_ms_[k]); tmin = std::min(tmin, time_ms_[k]);
        }
        const double nerr = (reference_stats && num_waypoints > 0) ? 2.0 * num_waypoints : (double)n;   // (no buffers at all without waypoints)
        stats(0) /= n; stats(1) /= n; stats(4) /= nerr; stats(5) /= nerr; stats(6) /= nerr;
        if (n > 1) { stats(2) /= n - 1; stats(3) /= n - 1; }
        time_stats[0] = reference_stats ? tsum / (2.0 * n) : tsum / n;
        time_stats[1] = tmax;
        time_stats[2] = reference_stats ? 0.0 : tmin;
        printf("average estimation time %f ms \n", time_stats[0]);
        printf("max estimation time %f ms \n", time_stats[1]);
        printf("min estimation time %f ms \n", time_stats[2]);
        simX = MatrixXd(n, 3); simU = MatrixXd(n, 2); time_record = MatrixXd(n, 2);
        for (int k = 0; k < n; ++k) {
            for (int q = 0; q < 3; ++q) simX(k, q) = simX_[3 * k + q];
            for (int q = 0; q < 2; ++q) simU(k, q) = simU_[2 * k + q];
            time_record(k, 0) = time_ms_[k]; time_record(k, 1) = 0.0;
        }
        if (!results_dir.empty()) {
            saveToFile(simX, results_dir + "/simX.txt");
            saveToFile(simU, results_dir + "/simU.txt");
            MatrixXd st(7, 1);
            for (int q = 0; q < 7; ++q) st(q, 0) = stats(q);
            saveToFile(st, results_dir + "/stats.txt");
            saveToFile(state_refs, results_dir + "/state_refs.txt");
        }
        return stats;
    }

    // ---- text I/O (src/optimizer.cpp:532-577) ----
    static MatrixXd loadTxt(const std::string& filename) {
        std::ifstream file(filename);
        if (!file.is_open()) throw std::runtime_error("Unable to open file: " + filename);
        std::string line;
        std::vector<double> entries;
        int numRows = 0, numCols = -1;
        while (std::getline(file, line)) {
            std::istringstream iss(line);
            double num;
            int n = 0;
            while (iss >> num) { entries.push_back(num); ++n; }
            if (numCols == -1) numCols = n;
            else if (n != numCols) throw std::runtime_error("Inconsistent number of columns");
            numRows++;
        }
        MatrixXd m(numRows, numCols < 0 ? 0 : numCols);
        for (int i = 0; i < numRows; ++i) for (int j = 0; j < numCols; ++j) m(i, j) = entries[(size_t)i * numCols + j];
        return m;
    }
    static void saveToFile(const MatrixXd& data, const std::string& file_path) {
        std::ofstream file(file_path);
        if (!file.is_open()) { std::cerr << "Unable to open file: " << file_path << std::endl; return; }
        for (int i = 0; i < data.rows(); ++i) {
            for (int j = 0; j < data.cols(); ++j) file << (j ? " " : "") << data(i, j);
            file << "\n";
        }
    }

    // public state, names as in the reference
    int status = 0;
    double T, v_ref, density, region_of_acceptance, t0;
    int N = 0, nx = 0, nu = 0, N_park = 0, nx_park = 0, nu_park = 0, iter = 0, v_ref_int = 0;
    bool use25 = false, use18 = false, debug = true;
    int target_waypoint_index = 0, last_waypoint_index = 0, closest_waypoint_index = 0, num_waypoints = 0;
    Vector3d x_current, x_real, frame1, frame2, x_current_transformed;
    double u_current[2] = {0.0, 0.0};
    bool shift_warm_start = false;   // extension (mpc_batch_shift); false = the reference's behaviour
    bool reference_stats = true;     // computeStats: the reference's zero-padded means (see there)
    bool use18_plant_as_written = false;   // integrate_next_states: see there
    double time_stats[3] = {0.0, 0.0, 0.0};   // {mean, max, min} tick time in ms, set by computeStats (:476-478)
    MatrixXd state_refs, input_refs, normals, left_turn_states, right_turn_states, straight_states;
    MatrixXd* state_refs_ptr = &state_refs;
    VectorXd state_attributes;
    MatrixXd simX, simU, time_record;   // filled by computeStats() from the recorded ticks (time_record: ms, column 0)
    std::string results_dir;    // where computeStats() writes its files; empty = nowhere
    static constexpr double rdb_circumference = 3.95;

  private:
    void set_path_loaded() { num_waypoints = (int)state_refs.rows(); state_refs_ptr = &state_refs; }
    // per-tick record behind computeStats (the reference's simX / simU / *_errors, which it sizes
    // 2*num_waypoints and overruns on longer runs; these grow)
    void record_tick(const Vector3d& state, double elapsed_ms) {
        time_ms_.push_back(elapsed_ms);
        const int t = std::min(target_waypoint_index, (int)state_refs.rows() - 1);
        simU_.push_back(u_current[0]); simU_.push_back(u_current[1]);
        for (int q = 0; q < 3; ++q) simX_.push_back(state[q]);
        for (int q = 0; q < 3; ++q) err_.push_back(t >= 0 ? state_refs(t, q) - x_real[q] : 0.0);
        iter++;
    }
    std::vector<double> simX_, simU_, err_, time_ms_;
    mpc_batch_t* mpc_ = nullptr;
    mpc_batch_t* mpc_plant50_ = nullptr;   // use18_plant_as_written only
    mpc_batch_t* mpc_park_ = nullptr;
    int limit_ = -1, lookahead_ = 1, count_ = 0;   // function-statics of the reference, per object
    double park_xs_[5] = {0, 0, 0, 0, 0};
};

}  // namespace mpc_amd
