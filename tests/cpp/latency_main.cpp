// latency_main.cpp — batch = 1 latency of the C-ABI from C++ (no Python in the loop): wall time of
// mpc_batch_solve with host pointers against the device time of the same call.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/mpc_batch.h"

int main(int argc, char** argv) {
    const char* variant = argc > 1 ? argv[1] : "mobile_robot_18";
    mpc_ocp_spec_t sp;
    if (mpc_spec_variant(variant, &sp)) return 1;
    mpc_batch_t* h = nullptr;
    if (mpc_batch_create(&h, &sp, 1, nullptr, 0)) return 2;
    const int N = sp.N;
    std::vector<double> yref(5 * N), wall, dev;
    double x0[3] = {0.1, -0.05, 0.02}, yref_e[3] = {0, 0, 0}, u0[2];
    int st = 0, it = 0;
    for (int t = 0; t < 600; ++t) {
        for (int k = 0; k < N; ++k) {   // a gentle arc ahead of the vehicle, moving with the tick
            const double s = 0.3 * sp.Ts * (k + t), kap = 0.4;
            yref[5 * k] = std::sin(kap * s) / kap; yref[5 * k + 1] = (1 - std::cos(kap * s)) / kap;
            yref[5 * k + 2] = kap * s; yref[5 * k + 3] = 0.3; yref[5 * k + 4] = 0.0;
        }
        yref_e[0] = yref[0]; yref_e[1] = yref[1]; yref_e[2] = yref[2];
        const auto a = std::chrono::steady_clock::now();
        if (mpc_batch_solve(h, 1, x0, yref.data(), yref_e, u0, nullptr, nullptr, &st, &it)) return 3;
        const auto b = std::chrono::steady_clock::now();
        if (t >= 100) { wall.push_back(std::chrono::duration<double, std::milli>(b - a).count()); dev.push_back(mpc_batch_last_kernel_ms(h)); }
        mpc_batch_sim_step(h, 1, x0, u0);
    }
    std::sort(wall.begin(), wall.end()); std::sort(dev.begin(), dev.end());
    printf("%s N=%d batch=1: wall p50 %.4f ms p99 %.4f ms | device p50 %.4f ms | last status %d qp_iter %d\n", variant, N,
           wall[wall.size() / 2], wall[wall.size() * 99 / 100], dev[dev.size() / 2], st, it);
    mpc_batch_free(h);
    return 0;
}
