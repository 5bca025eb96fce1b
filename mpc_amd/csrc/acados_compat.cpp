// acados_compat.cpp — the acados-named entry points of the reference's generated solvers
// (include/acados_compat/*.h) implemented over the C-ABI of this library (include/mpc_batch.h).
//
// What it replaces in the reference (relink, don't rewrite — INTEGRATION.md section 2):
//   src/c_generated_code*/acados_solver_<name>.c      <name>_acados_create_capsule/create/solve/reset/
//                                                     batch_solve/free/free_capsule, the getters
//   src/c_generated_code*/acados_sim_solver_<name>.c  <name>_acados_sim_create/solve/free, the getters
//   libacados' acados_c/ocp_nlp_interface.c           ocp_nlp_cost_model_set, ocp_nlp_constraints_model_set,
//                                                     ocp_nlp_out_set/get, ocp_nlp_get, ocp_nlp_solver_opts_set
//   libacados' acados_c/sim_interface.c               sim_in_set, sim_out_get
// for the call sites src/optimizer.cpp:13-119,251-296,323-338, include/optimizer.hpp:167-222,446-456 and
// src/c_generated_code*/main_*.c.  Host code only: every number comes out of mpc_batch_solve /
// mpc_batch_get / mpc_batch_sim_step, i.e. the HIP kernels; there is no CPU arithmetic path in here.
//
// Model.  A capsule is one slot of a "group" = one mpc_batch handle plus the host arrays of its
// slots (references, x0, u0, status).  <name>_acados_create gives a capsule a group of its own
// (capacity 1: the batch = 1 control loop of Optimizer::update_and_solve).  <name>_acados_batch_solve
// over N capsules moves them — iterates included — into one group of capacity N on the first call
// and is ONE mpc_batch_solve from then on.  The iterate stays in HBM; the host sees it through a
// mirror that is fetched (mpc_batch_get) only when ocp_nlp_out_get / ocp_nlp_out_set ask for more
// than u_0, which every solve returns anyway.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "../../include/mpc_batch.h"
#include "acados/utils/print.h"
#include "acados_c/ocp_nlp_interface.h"
#include "acados_c/sim_interface.h"
#include "acados_sim_solver_mobile_robot.h"
#include "acados_sim_solver_mobile_robot_18.h"
#include "acados_sim_solver_mobile_robot_25.h"
#include "acados_sim_solver_mobile_robot_50.h"
#include "acados_sim_solver_park.h"
#include "acados_solver_mobile_robot.h"
#include "acados_solver_mobile_robot_18.h"
#include "acados_solver_mobile_robot_25.h"
#include "acados_solver_mobile_robot_50.h"
#include "acados_solver_park.h"
#include "blasfeo/include/blasfeo_d_aux_ext_dep.h"

namespace {

struct Ocp;

// One mpc_batch handle and the host arrays of its n slots.
struct Group {
    mpc_batch_t* h = nullptr;
    mpc_ocp_spec_t spec;
    int n = 0, live = 0;
    bool pinned = false;
    double *x0 = nullptr, *yref = nullptr, *yref_e = nullptr, *u0 = nullptr;   // [n][3], [n][N][5], [n][3], [n][2]
    int *status = nullptr, *qp_iter = nullptr;                                  // [n]
    // host mirror of the iterate: valid after a fetch; dirty = it holds values the next solve must load
    std::vector<double> xt, ut, pi;   // [n][N+1][3], [n][N][2], [n][N][3]
    bool mirror_valid = false, dirty = false;
    std::vector<Ocp*> members;
};

struct Ocp {
    const char* variant = nullptr;
    Group* g = nullptr;
    int idx = 0;
    double lbx0[3] = {0, 0, 0}, ubx0[3] = {0, 0, 0};
    int rti_phase = 0;
    bool solved = false;       // u0 / status of the group arrays belong to this capsule's last solve
    int status = 0, qp_iter = 0, sqp_iter = 0;
    double time_tot = 0.0, time_qp = 0.0;
    ocp_nlp_config* config = nullptr;
    ocp_nlp_dims* dims = nullptr;
    ocp_nlp_in* in = nullptr;
    ocp_nlp_out* out = nullptr;
    ocp_nlp_solver* solver = nullptr;
    ocp_nlp_plan_t* plan = nullptr;
    int* opts = nullptr;       // the block handed out as nlp_opts
};

// the layout every acados_solver_<name>.h gives its capsule (mpc_acados_solver_decl.inc)
typedef mobile_robot_solver_capsule OcpCapsule;
typedef mobile_robot_sim_solver_capsule SimCapsule;
#define MPC_SAME_LAYOUT(T, U) static_assert(sizeof(T) == sizeof(U) && offsetof(T, mpc_impl) == offsetof(U, mpc_impl), "capsule layouts differ")
MPC_SAME_LAYOUT(mobile_robot_18_solver_capsule, OcpCapsule);
MPC_SAME_LAYOUT(mobile_robot_25_solver_capsule, OcpCapsule);
MPC_SAME_LAYOUT(mobile_robot_50_solver_capsule, OcpCapsule);
MPC_SAME_LAYOUT(park_solver_capsule, OcpCapsule);
MPC_SAME_LAYOUT(mobile_robot_18_sim_solver_capsule, SimCapsule);
MPC_SAME_LAYOUT(mobile_robot_25_sim_solver_capsule, SimCapsule);
MPC_SAME_LAYOUT(mobile_robot_50_sim_solver_capsule, SimCapsule);
MPC_SAME_LAYOUT(park_sim_solver_capsule, SimCapsule);

void complain(const char* where, const char* what) { fprintf(stderr, "mpc_amd acados_compat: %s: %s\n", where, what); }

template <class T>
T* host_array(size_t count, bool pinned) {
    void* p = nullptr;
    if (pinned) {
        if (hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        memset(p, 0, count * sizeof(T));
    } else {
        p = calloc(count, sizeof(T));
    }
    return static_cast<T*>(p);
}
template <class T>
void host_release(T* p, bool pinned) {
    if (!p) return;
    if (pinned) (void)hipHostFree(p);
    else free(p);
}

void group_destroy(Group* g) {
    if (!g) return;
    if (g->h) mpc_batch_free(g->h);
    host_release(g->x0, g->pinned); host_release(g->yref, g->pinned); host_release(g->yref_e, g->pinned);
    host_release(g->u0, g->pinned); host_release(g->status, g->pinned); host_release(g->qp_iter, g->pinned);
    delete g;
}

// A group of n slots for `spec`.  Batches go through pinned arrays (the kernels then read the
// references in place, include/mpc_batch.h mpc_batch_solve); a single capsule through plain ones
// (mpc_batch_solve stages batches of up to 64 in its own pinned block anyway).
Group* group_create(const mpc_ocp_spec_t& spec, int n, const char* where) {
    Group* g = new Group();
    g->spec = spec;
    g->n = n;
    const int rc = mpc_batch_create(&g->h, &spec, n, nullptr, 0);
    if (rc != 0) {
        char msg[160];
        snprintf(msg, sizeof(msg), "mpc_batch_create(capacity %d) returned %d%s", n, rc, rc == -3 ? " (no GPU)" : "");
        complain(where, msg);
        delete g;
        return nullptr;
    }
    const size_t N = (size_t)spec.N, sn = (size_t)n;
    for (int attempt = 0; attempt < 2; ++attempt) {
        g->pinned = attempt == 0 && n > 64;
        g->x0 = host_array<double>(3 * sn, g->pinned);
        g->yref = host_array<double>(5 * N * sn, g->pinned);
        g->yref_e = host_array<double>(3 * sn, g->pinned);
        g->u0 = host_array<double>(2 * sn, g->pinned);
        g->status = host_array<int>(sn, g->pinned);
        g->qp_iter = host_array<int>(sn, g->pinned);
        if (g->x0 && g->yref && g->yref_e && g->u0 && g->status && g->qp_iter) break;
        host_release(g->x0, g->pinned); host_release(g->yref, g->pinned); host_release(g->yref_e, g->pinned);
        host_release(g->u0, g->pinned); host_release(g->status, g->pinned); host_release(g->qp_iter, g->pinned);
        g->x0 = g->yref = g->yref_e = g->u0 = nullptr;
        g->status = g->qp_iter = nullptr;
        if (!g->pinned) { complain(where, "out of host memory"); group_destroy(g); return nullptr; }
    }
    g->members.assign(sn, nullptr);
    return g;
}

bool ensure_mirror(Group* g, const char* where) {
    if (g->mirror_valid) return true;
    const size_t N = (size_t)g->spec.N, n = (size_t)g->n;
    g->xt.resize(n * (N + 1) * 3); g->ut.resize(n * N * 2); g->pi.resize(n * N * 3);
    const int rc = mpc_batch_get(g->h, g->n, g->xt.data(), g->ut.data(), g->pi.data());
    if (rc != 0) { complain(where, mpc_batch_last_error(g->h)); return false; }
    g->mirror_valid = true;
    return true;
}

void detach(Ocp* o) {
    Group* g = o->g;
    if (!g) return;
    g->members[o->idx] = nullptr;
    if (--g->live == 0) group_destroy(g);
    o->g = nullptr;
}

// Moves the capsules of `list` (same spec) into one new group, iterates and references included.
bool regroup(Ocp* const* list, int n, const mpc_ocp_spec_t& spec, const char* where) {
    for (int i = 0; i < n; ++i)
        if (list[i]->g && !ensure_mirror(list[i]->g, where)) return false;
    Group* ng = group_create(spec, n, where);
    if (!ng) return false;
    const size_t N = (size_t)spec.N;
    ng->xt.assign((size_t)n * (N + 1) * 3, 0.0); ng->ut.assign((size_t)n * N * 2, 0.0); ng->pi.assign((size_t)n * N * 3, 0.0);
    for (int i = 0; i < n; ++i) {
        Ocp* o = list[i];
        Group* og = o->g;
        if (og && og->spec.N == spec.N) {
            const size_t j = (size_t)o->idx, d = (size_t)i;
            memcpy(&ng->xt[d * (N + 1) * 3], &og->xt[j * (N + 1) * 3], sizeof(double) * (N + 1) * 3);
            memcpy(&ng->ut[d * N * 2], &og->ut[j * N * 2], sizeof(double) * N * 2);
            memcpy(&ng->pi[d * N * 3], &og->pi[j * N * 3], sizeof(double) * N * 3);
            memcpy(ng->yref + d * N * 5, og->yref + j * N * 5, sizeof(double) * N * 5);
            memcpy(ng->yref_e + d * 3, og->yref_e + j * 3, sizeof(double) * 3);
            memcpy(ng->u0 + d * 2, og->u0 + j * 2, sizeof(double) * 2);
            ng->status[d] = og->status[j]; ng->qp_iter[d] = og->qp_iter[j];
        }
        detach(o);
        o->g = ng; o->idx = i;
        ng->members[i] = o;
    }
    ng->live = n;
    ng->mirror_valid = true;
    ng->dirty = true;   // the new handle's capsules are all zero: the next solve loads the mirror
    return true;
}

mpc_ocp_spec_t variant_spec(const char* name) {
    mpc_ocp_spec_t s;
    memset(&s, 0, sizeof(s));
    (void)mpc_spec_variant(name, &s);
    return s;
}

// ---- the view objects (each its own malloc block: the reference free()s them one by one) ----
void make_views(Ocp* o) {
    const int N = o->g->spec.N;
    o->config = static_cast<ocp_nlp_config*>(calloc(1, sizeof(ocp_nlp_config)));
    o->config->N = N; o->config->mpc_impl = o;
    char* blk = static_cast<char*>(calloc(1, sizeof(ocp_nlp_dims) + 7 * (size_t)(N + 1) * sizeof(int)));
    o->dims = reinterpret_cast<ocp_nlp_dims*>(blk);
    int* ia = reinterpret_cast<int*>(blk + sizeof(ocp_nlp_dims));
    o->dims->nv = ia; o->dims->nx = ia + (N + 1); o->dims->nu = ia + 2 * (N + 1); o->dims->ni = ia + 3 * (N + 1);
    o->dims->nz = ia + 4 * (N + 1); o->dims->ns = ia + 5 * (N + 1); o->dims->np = ia + 6 * (N + 1);
    for (int k = 0; k <= N; ++k) {
        o->dims->nx[k] = MPC_NX;
        o->dims->nu[k] = k < N ? MPC_NU : 0;
        o->dims->nv[k] = o->dims->nx[k] + o->dims->nu[k];
        o->dims->ni[k] = k == 0 ? MPC_NX + MPC_NU : k < N ? 4 : 0;   // nbx0 + nbu, nbx + nbu, 0
    }
    o->dims->N = N; o->dims->mpc_impl = o;
    blk = static_cast<char*>(calloc(1, sizeof(ocp_nlp_in) + (size_t)N * sizeof(double)));
    o->in = reinterpret_cast<ocp_nlp_in*>(blk);
    o->in->Ts = reinterpret_cast<double*>(blk + sizeof(ocp_nlp_in));
    for (int k = 0; k < N; ++k) o->in->Ts[k] = o->g->spec.Ts;
    o->in->mpc_impl = o;
    o->out = static_cast<ocp_nlp_out*>(calloc(1, sizeof(ocp_nlp_out)));
    o->out->mpc_impl = o;
    o->opts = static_cast<int*>(calloc(4, sizeof(int)));
    o->solver = static_cast<ocp_nlp_solver*>(calloc(1, sizeof(ocp_nlp_solver)));
    o->solver->config = o->config; o->solver->dims = o->dims; o->solver->opts = o->opts; o->solver->mpc_impl = o;
    o->plan = static_cast<ocp_nlp_plan_t*>(calloc(1, sizeof(ocp_nlp_plan_t)));
    o->plan->nlp_solver = SQP_RTI; o->plan->ocp_qp_solver = PARTIAL_CONDENSING_HPIPM; o->plan->nlp_cost = LINEAR_LS;
    o->plan->nlp_dynamics = CONTINUOUS_MODEL; o->plan->nlp_constraints = BGH; o->plan->regularization = NO_REGULARIZE;
    o->plan->N = N; o->plan->mpc_impl = o;
}

Ocp* ocp_of(OcpCapsule* c) { return c ? static_cast<Ocp*>(c->mpc_impl) : nullptr; }

int ocp_create(OcpCapsule* c, const char* variant, int n_time_steps, double* new_time_steps) {
    if (!c) return 1;
    mpc_ocp_spec_t spec = variant_spec(variant);
    if (n_time_steps > 0) spec.N = n_time_steps;
    if (spec.N < 1 || spec.N > MPC_MAX_N) { complain(variant, "horizon outside 1..127"); return 1; }
    if (new_time_steps) {
        for (int k = 1; k < spec.N; ++k)
            if (fabs(new_time_steps[k] - new_time_steps[0]) > 1e-12 * fabs(new_time_steps[0])) {
                complain(variant, "non-uniform time steps are not supported (the kernels hold one Ts)");
                return 1;
            }
        spec.Ts = new_time_steps[0];
    }
    Group* g = group_create(spec, 1, variant);
    if (!g) return 1;
    Ocp* o = new Ocp();
    o->variant = variant;
    o->g = g; o->idx = 0;
    g->members[0] = o; g->live = 1;
    make_views(o);
    c->nlp_in = o->in; c->nlp_out = o->out; c->sens_out = nullptr; c->nlp_solver = o->solver; c->nlp_opts = o->opts;
    c->nlp_solver_plan = o->plan; c->nlp_config = o->config; c->nlp_dims = o->dims; c->nlp_np = 0;
    c->forw_vde_casadi = nullptr; c->expl_ode_fun = nullptr;
    c->mpc_impl = o;
    return 0;
}

int ocp_free(OcpCapsule* c) {
    Ocp* o = ocp_of(c);
    if (!o) return 0;
    detach(o);
    free(o->config); free(o->dims); free(o->in); free(o->out); free(o->solver); free(o->plan); free(o->opts);
    delete o;
    c->nlp_in = nullptr; c->nlp_out = nullptr; c->nlp_solver = nullptr; c->nlp_opts = nullptr; c->nlp_solver_plan = nullptr;
    c->nlp_config = nullptr; c->nlp_dims = nullptr; c->mpc_impl = nullptr;
    return 0;
}

// x0 of the next solve: lbx_0 = ubx_0 (the stage-0 state is eliminated, idxbxe_0 = {0,1,2})
bool take_x0(Ocp* o, const char* where) {
    for (int q = 0; q < 3; ++q)
        if (o->lbx0[q] != o->ubx0[q]) {
            complain(where, "lbx and ubx of stage 0 differ: the initial state must be fixed (idxbxe_0 = {0,1,2})");
            return false;
        }
    memcpy(o->g->x0 + 3 * (size_t)o->idx, o->lbx0, sizeof(o->lbx0));
    return true;
}

int group_solve(Group* g, const char* where) {
    const bool load = g->dirty;
    const int rc = mpc_batch_solve(g->h, g->n, g->x0, g->yref, g->yref_e, g->u0, load ? g->xt.data() : nullptr,
                                   load ? g->ut.data() : nullptr, g->status, g->qp_iter);
    g->dirty = false;
    g->mirror_valid = false;
    if (rc != 0) complain(where, mpc_batch_last_error(g->h));
    return rc;
}

int ocp_solve(OcpCapsule* c, const char* where) {
    Ocp* o = ocp_of(c);
    if (!o) { complain(where, "capsule was not created"); return ACADOS_QP_FAILURE; }
    if (o->rti_phase != 0) { complain(where, "rti_phase 1 / 2 (split preparation / feedback) is not supported"); return ACADOS_QP_FAILURE; }
    const auto t0 = std::chrono::steady_clock::now();
    if (o->g->n != 1 && !regroup(&o, 1, o->g->spec, where)) return ACADOS_QP_FAILURE;
    if (!take_x0(o, where)) return ACADOS_QP_FAILURE;
    Group* g = o->g;
    if (group_solve(g, where) != 0) { o->status = ACADOS_QP_FAILURE; return o->status; }
    o->status = g->status[0]; o->qp_iter = g->qp_iter[0]; o->sqp_iter = 1; o->solved = true;
    o->time_qp = mpc_batch_last_kernel_ms(g->h) * 1e-3;
    o->time_tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return o->status;
}

void ocp_batch_solve(OcpCapsule** caps, int n, const char* where) {
    if (!caps || n < 1) return;
    if (n == 1) { (void)ocp_solve(caps[0], where); return; }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Ocp*> list((size_t)n);
    for (int i = 0; i < n; ++i) {
        list[i] = ocp_of(caps[i]);
        if (!list[i]) { complain(where, "a capsule of the batch was not created"); return; }
        if (list[i]->rti_phase != 0) { complain(where, "rti_phase 1 / 2 is not supported"); return; }
    }
    Group* g = list[0]->g;
    bool together = g->n == n && g->live == n;
    for (int i = 0; i < n && together; ++i) together = list[i]->g == g && list[i]->idx == i;
    if (!together) {
        const mpc_ocp_spec_t spec = list[0]->g->spec;
        for (int i = 1; i < n; ++i)
            if (memcmp(&list[i]->g->spec, &spec, sizeof(spec)) != 0) { complain(where, "capsules of one batch must share horizon and time step"); return; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (list[i] == list[j]) { complain(where, "a capsule appears twice in the batch"); return; }
        if (!regroup(list.data(), n, spec, where)) return;
        g = list[0]->g;
    }
    for (int i = 0; i < n; ++i)
        if (!take_x0(list[i], where)) return;
    const int rc = group_solve(g, where);
    const double tq = mpc_batch_last_kernel_ms(g->h) * 1e-3;
    const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; ++i) {
        Ocp* o = list[i];
        o->status = rc == 0 ? g->status[i] : (int)ACADOS_QP_FAILURE;
        o->qp_iter = rc == 0 ? g->qp_iter[i] : 0;
        o->sqp_iter = 1; o->solved = rc == 0;
        o->time_qp = tq; o->time_tot = tt;
    }
}

int ocp_reset(OcpCapsule* c, const char* where) {
    Ocp* o = ocp_of(c);
    if (!o) return 1;
    Group* g = o->g;
    const size_t N = (size_t)g->spec.N, j = (size_t)o->idx;
    if (g->mirror_valid) {
        memset(&g->xt[j * (N + 1) * 3], 0, sizeof(double) * (N + 1) * 3);
        memset(&g->ut[j * N * 2], 0, sizeof(double) * N * 2);
        memset(&g->pi[j * N * 3], 0, sizeof(double) * N * 3);
    }
    std::vector<unsigned char> mask((size_t)g->n, 0);
    mask[j] = 1;
    const int rc = mpc_batch_reset(g->h, mask.data());
    if (rc != 0) { complain(where, mpc_batch_last_error(g->h)); return 1; }
    g->u0[2 * j] = g->u0[2 * j + 1] = 0.0;
    return 0;
}

// a new handle for this capsule with another time step, the iterate kept
int ocp_new_time_step(Ocp* o, double Ts, const char* where) {
    mpc_ocp_spec_t spec = o->g->spec;
    spec.Ts = Ts;
    if (!regroup(&o, 1, spec, where)) return 1;
    for (int k = 0; k < spec.N; ++k) o->in->Ts[k] = Ts;
    return 0;
}

bool same(const double* a, const double* b, int n) {
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// ---- integrator capsules ----
struct Sim {
    const char* variant = nullptr;
    mpc_ocp_spec_t spec;
    mpc_batch_t* h = nullptr;          // capacity 1
    mpc_batch_t* hb = nullptr;         // cached by <name>_acados_sim_batch_solve on the first capsule of a batch
    int hb_n = 0;
    double hb_T = 0.0;
    double x[3] = {0, 0, 0}, u[2] = {0, 0}, xn[3] = {0, 0, 0};
    sim_config* config = nullptr;
    sim_dims_* dims = nullptr;
    sim_in* in = nullptr;
    sim_out* out = nullptr;
    sim_opts* opts = nullptr;
    sim_solver* solver = nullptr;
};
Sim* sim_of(SimCapsule* c) { return c ? static_cast<Sim*>(c->mpc_impl) : nullptr; }

int sim_create(SimCapsule* c, const char* variant) {
    if (!c) return 1;
    Sim* s = new Sim();
    s->variant = variant;
    s->spec = variant_spec(variant);
    const int rc = mpc_batch_create(&s->h, &s->spec, 1, nullptr, 0);
    if (rc != 0) {
        char msg[128];
        snprintf(msg, sizeof(msg), "mpc_batch_create returned %d%s", rc, rc == -3 ? " (no GPU)" : "");
        complain(variant, msg);
        delete s;
        return 1;
    }
    s->config = static_cast<sim_config*>(calloc(1, sizeof(sim_config)));
    s->config->ns = 4; s->config->mpc_impl = s;
    s->dims = static_cast<sim_dims_*>(calloc(1, sizeof(sim_dims_)));
    s->dims->nx = MPC_NX; s->dims->nu = MPC_NU; s->dims->nz = 0; s->dims->mpc_impl = s;
    s->in = static_cast<sim_in*>(calloc(1, sizeof(sim_in)));
    s->in->T = s->spec.Ts; s->in->mpc_impl = s;
    s->out = static_cast<sim_out*>(calloc(1, sizeof(sim_out)));
    s->out->mpc_impl = s;
    s->opts = static_cast<sim_opts*>(calloc(1, sizeof(sim_opts)));
    s->opts->num_stages = 4; s->opts->num_steps = 1; s->opts->mpc_impl = s;
    s->solver = static_cast<sim_solver*>(calloc(1, sizeof(sim_solver)));
    s->solver->config = s->config; s->solver->dims = s->dims; s->solver->opts = s->opts; s->solver->mpc_impl = s;
    c->acados_sim_in = s->in; c->acados_sim_out = s->out; c->acados_sim_solver = s->solver; c->acados_sim_opts = s->opts;
    c->acados_sim_config = s->config; c->acados_sim_dims = s->dims;
    c->mpc_impl = s;
    return 0;
}

// the handle's plant step runs at spec.Ts: a new "T" needs a new handle
bool sim_match_T(Sim* s, const char* where) {
    const double T = s->in->T;
    if (T == s->spec.Ts) return true;
    if (!(T > 0.0)) { complain(where, "T must be positive"); return false; }
    mpc_ocp_spec_t spec = s->spec;
    spec.Ts = T;
    mpc_batch_t* h = nullptr;
    if (mpc_batch_create(&h, &spec, 1, nullptr, 0) != 0) { complain(where, "mpc_batch_create failed for the new T"); return false; }
    mpc_batch_free(s->h);
    s->h = h; s->spec = spec;
    return true;
}

int sim_solve(SimCapsule* c, const char* where) {
    Sim* s = sim_of(c);
    if (!s) { complain(where, "capsule was not created"); return ACADOS_NAN_DETECTED; }
    if (!sim_match_T(s, where)) return ACADOS_NAN_DETECTED;
    double xx[3] = {s->x[0], s->x[1], s->x[2]};
    const int rc = mpc_batch_sim_step(s->h, 1, xx, s->u);
    if (rc != 0) { complain(where, mpc_batch_last_error(s->h)); return ACADOS_NAN_DETECTED; }
    s->xn[0] = xx[0]; s->xn[1] = xx[1]; s->xn[2] = xx[2];
    return (xx[0] == xx[0] && xx[1] == xx[1] && xx[2] == xx[2]) ? ACADOS_SUCCESS : ACADOS_NAN_DETECTED;
}

void sim_batch_solve(SimCapsule** caps, int n, const char* where) {
    if (!caps || n < 1) return;
    Sim* s0 = sim_of(caps[0]);
    bool uniform = s0 != nullptr;
    for (int i = 1; i < n && uniform; ++i) {
        Sim* s = sim_of(caps[i]);
        uniform = s && s->in->T == s0->in->T && s->spec.L == s0->spec.L;
    }
    if (!uniform || n == 1) {   // different step lengths: one by one
        for (int i = 0; i < n; ++i) (void)sim_solve(caps[i], where);
        return;
    }
    if (!s0->hb || s0->hb_n != n || s0->hb_T != s0->in->T) {
        if (s0->hb) mpc_batch_free(s0->hb);
        s0->hb = nullptr;
        mpc_ocp_spec_t spec = s0->spec;
        spec.Ts = s0->in->T;
        if (mpc_batch_create(&s0->hb, &spec, n, nullptr, 0) != 0) { complain(where, "mpc_batch_create failed for the batch"); return; }
        s0->hb_n = n; s0->hb_T = spec.Ts;
    }
    std::vector<double> x(3 * (size_t)n), u(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        Sim* s = sim_of(caps[i]);
        memcpy(&x[3 * (size_t)i], s->x, sizeof(s->x));
        memcpy(&u[2 * (size_t)i], s->u, sizeof(s->u));
    }
    if (mpc_batch_sim_step(s0->hb, n, x.data(), u.data()) != 0) { complain(where, mpc_batch_last_error(s0->hb)); return; }
    for (int i = 0; i < n; ++i) memcpy(sim_of(caps[i])->xn, &x[3 * (size_t)i], sizeof(double) * 3);
}

int sim_free(SimCapsule* c) {
    Sim* s = sim_of(c);
    if (!s) return 0;
    if (s->h) mpc_batch_free(s->h);
    if (s->hb) mpc_batch_free(s->hb);
    free(s->config); free(s->dims); free(s->in); free(s->out); free(s->opts); free(s->solver);
    delete s;
    c->acados_sim_in = nullptr; c->acados_sim_out = nullptr; c->acados_sim_solver = nullptr; c->acados_sim_opts = nullptr;
    c->acados_sim_config = nullptr; c->acados_sim_dims = nullptr; c->mpc_impl = nullptr;
    return 0;
}

void print_mat(int m, int n, const double* A, int lda, bool tran, const char* fmt) {
    const int rows = tran ? n : m, cols = tran ? m : n;
    for (int i = 0; i < rows; ++i) {
        for (int j = 0; j < cols; ++j) printf(fmt, tran ? A[j + (size_t)lda * i] : A[i + (size_t)lda * j]);
        printf("\n");
    }
    printf("\n");
}

}  // namespace

extern "C" {

// ---------------- acados_c/ocp_nlp_interface.h ----------------
int ocp_nlp_cost_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in* in, int stage, const char* field, void* value) {
    Ocp* o = in ? static_cast<Ocp*>(in->mpc_impl) : nullptr;
    if (!o || !field || !value) { complain("ocp_nlp_cost_model_set", "null argument"); return MPC_ACADOS_UNSUPPORTED; }
    Group* g = o->g;
    const int N = g->spec.N;
    if (stage < 0 || stage > N) { complain("ocp_nlp_cost_model_set", "stage out of range"); return MPC_ACADOS_UNSUPPORTED; }
    const double* v = static_cast<const double*>(value);
    const size_t j = (size_t)o->idx;
    if (strcmp(field, "yref") == 0 || strcmp(field, "y_ref") == 0) {
        if (stage < N) memcpy(g->yref + (j * N + (size_t)stage) * 5, v, sizeof(double) * 5);
        else memcpy(g->yref_e + j * 3, v, sizeof(double) * 3);
        return ACADOS_SUCCESS;
    }
    if (strcmp(field, "W") == 0) {   // column-major ny x ny; only the generated diagonal
        const int ny = stage < N ? 5 : 3;
        const double* w = stage < N ? g->spec.W : g->spec.W_e;
        bool ok = true;
        for (int r = 0; r < ny; ++r)
            for (int c2 = 0; c2 < ny; ++c2) ok = ok && v[r + ny * c2] == (r == c2 ? w[r] : 0.0);
        if (ok) return ACADOS_SUCCESS;
        complain("ocp_nlp_cost_model_set", "\"W\" differs from the generated weights (they are constants of the handle: create it with another mpc_ocp_spec_t)");
        return MPC_ACADOS_UNSUPPORTED;
    }
    if (strcmp(field, "scaling") == 0) {
        if (v[0] == (stage < N ? g->spec.Ts : 1.0)) return ACADOS_SUCCESS;
        complain("ocp_nlp_cost_model_set", "\"scaling\" differs from the generated one (Ts, 1 at stage N)");
        return MPC_ACADOS_UNSUPPORTED;
    }
    complain("ocp_nlp_cost_model_set", "field not carried by this layer (yref, W, scaling)");
    return MPC_ACADOS_UNSUPPORTED;
}

int ocp_nlp_constraints_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in* in, int stage, const char* field, void* value) {
    Ocp* o = in ? static_cast<Ocp*>(in->mpc_impl) : nullptr;
    if (!o || !field || !value) { complain("ocp_nlp_constraints_model_set", "null argument"); return MPC_ACADOS_UNSUPPORTED; }
    const mpc_ocp_spec_t& sp = o->g->spec;
    const int N = sp.N;
    if (stage < 0 || stage > N) { complain("ocp_nlp_constraints_model_set", "stage out of range"); return MPC_ACADOS_UNSUPPORTED; }
    const double* v = static_cast<const double*>(value);
    const int* iv = static_cast<const int*>(value);
    if (stage == 0 && strcmp(field, "lbx") == 0) { memcpy(o->lbx0, v, sizeof(o->lbx0)); return ACADOS_SUCCESS; }
    if (stage == 0 && strcmp(field, "ubx") == 0) { memcpy(o->ubx0, v, sizeof(o->ubx0)); return ACADOS_SUCCESS; }
    bool ok = false;
    if (stage == 0 && (strcmp(field, "idxbx") == 0 || strcmp(field, "idxbxe") == 0)) ok = iv[0] == 0 && iv[1] == 1 && iv[2] == 2;
    else if (stage > 0 && stage < N && strcmp(field, "lbx") == 0) ok = same(v, sp.lbx, 2);
    else if (stage > 0 && stage < N && strcmp(field, "ubx") == 0) ok = same(v, sp.ubx, 2);
    else if (stage > 0 && stage < N && strcmp(field, "idxbx") == 0) ok = iv[0] == 0 && iv[1] == 1;
    else if (stage < N && strcmp(field, "lbu") == 0) ok = same(v, sp.lbu, 2);
    else if (stage < N && strcmp(field, "ubu") == 0) ok = same(v, sp.ubu, 2);
    else if (stage < N && strcmp(field, "idxbu") == 0) ok = iv[0] == 0 && iv[1] == 1;
    else { complain("ocp_nlp_constraints_model_set", "field not carried by this layer (lbx, ubx, lbu, ubu, idxbx, idxbu)"); return MPC_ACADOS_UNSUPPORTED; }
    if (ok) return ACADOS_SUCCESS;
    complain("ocp_nlp_constraints_model_set", "value differs from the generated bounds (they are constants of the handle: create it with another mpc_ocp_spec_t)");
    return MPC_ACADOS_UNSUPPORTED;
}

int ocp_nlp_dims_get_from_attr(ocp_nlp_config*, ocp_nlp_dims* dims, ocp_nlp_out*, int stage, const char* field) {
    if (!dims || !field || stage < 0 || stage > dims->N) return 0;
    if (strcmp(field, "x") == 0) return MPC_NX;
    if (strcmp(field, "u") == 0) return stage < dims->N ? MPC_NU : 0;
    if (strcmp(field, "pi") == 0) return stage < dims->N ? MPC_NX : 0;
    if (strcmp(field, "kkt_norm_inf") == 0) return 1;
    return 0;
}

void ocp_nlp_out_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_out* out, int stage, const char* field, void* value) {
    Ocp* o = out ? static_cast<Ocp*>(out->mpc_impl) : nullptr;
    if (!o || !field || !value) { complain("ocp_nlp_out_set", "null argument"); return; }
    Group* g = o->g;
    const size_t N = (size_t)g->spec.N, j = (size_t)o->idx, k = (size_t)stage;
    const double* v = static_cast<const double*>(value);
    const bool isx = strcmp(field, "x") == 0, isu = strcmp(field, "u") == 0, isp = strcmp(field, "pi") == 0;
    if (!(isx || isu || isp)) { complain("ocp_nlp_out_set", "field not carried by this layer (x, u, pi)"); return; }
    if (stage < 0 || k > N || (!isx && k == N)) { complain("ocp_nlp_out_set", "stage out of range"); return; }
    if (!ensure_mirror(g, "ocp_nlp_out_set")) return;
    if (isx) memcpy(&g->xt[(j * (N + 1) + k) * 3], v, sizeof(double) * 3);
    else if (isu) memcpy(&g->ut[(j * N + k) * 2], v, sizeof(double) * 2);
    else memcpy(&g->pi[(j * N + k) * 3], v, sizeof(double) * 3);
    if (isu && k == 0) { g->u0[2 * j] = v[0]; g->u0[2 * j + 1] = v[1]; }
    if (!isp) g->dirty = true;   // (pi does not feed back into an SQP_RTI step; it stays in the mirror until the next solve)
}

void ocp_nlp_out_get(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_out* out, int stage, const char* field, void* value) {
    Ocp* o = out ? static_cast<Ocp*>(out->mpc_impl) : nullptr;
    if (!o || !field || !value) { complain("ocp_nlp_out_get", "null argument"); return; }
    Group* g = o->g;
    const size_t N = (size_t)g->spec.N, j = (size_t)o->idx, k = (size_t)stage;
    double* v = static_cast<double*>(value);
    if (strcmp(field, "kkt_norm_inf") == 0) { v[0] = out->inf_norm_res; return; }
    const bool isx = strcmp(field, "x") == 0, isu = strcmp(field, "u") == 0, isp = strcmp(field, "pi") == 0;
    if (!(isx || isu || isp)) { complain("ocp_nlp_out_get", "field not carried by this layer (x, u, pi, kkt_norm_inf)"); return; }
    if (stage < 0 || k > N || (!isx && k == N)) { complain("ocp_nlp_out_get", "stage out of range"); return; }
    // u_0 comes back with every solve (src/optimizer.cpp:296 — the control loop's only read)
    if (isu && k == 0 && !g->mirror_valid && o->solved) { v[0] = g->u0[2 * j]; v[1] = g->u0[2 * j + 1]; return; }
    if (!ensure_mirror(g, "ocp_nlp_out_get")) return;
    if (isx) memcpy(v, &g->xt[(j * (N + 1) + k) * 3], sizeof(double) * 3);
    else if (isu) memcpy(v, &g->ut[(j * N + k) * 2], sizeof(double) * 2);
    else memcpy(v, &g->pi[(j * N + k) * 3], sizeof(double) * 3);
}

void ocp_nlp_solver_opts_set(ocp_nlp_config* config, void*, const char* field, void* value) {
    Ocp* o = config ? static_cast<Ocp*>(config->mpc_impl) : nullptr;
    if (!o || !field || !value) { complain("ocp_nlp_solver_opts_set", "null argument"); return; }
    if (strcmp(field, "rti_phase") == 0) {
        o->rti_phase = *static_cast<int*>(value);
        if (o->rti_phase != 0) complain("ocp_nlp_solver_opts_set", "rti_phase 1 / 2 (split preparation / feedback) is not supported; the next solve will fail");
        return;
    }
    if (strcmp(field, "print_level") == 0) return;
    complain("ocp_nlp_solver_opts_set", "option not carried by this layer (rti_phase, print_level): the solver options are those of the generated code");
}

void ocp_nlp_get(ocp_nlp_config* config, ocp_nlp_solver* solver, const char* field, void* rv) {
    Ocp* o = solver ? static_cast<Ocp*>(solver->mpc_impl) : config ? static_cast<Ocp*>(config->mpc_impl) : nullptr;
    if (!o || !field || !rv) { complain("ocp_nlp_get", "null argument"); return; }
    if (strcmp(field, "time_tot") == 0) *static_cast<double*>(rv) = o->time_tot;
    else if (strcmp(field, "time_qp") == 0 || strcmp(field, "time_qp_sol") == 0 || strcmp(field, "time_qp_solver_call") == 0) *static_cast<double*>(rv) = o->time_qp;
    else if (strcmp(field, "time_lin") == 0 || strcmp(field, "time_reg") == 0 || strcmp(field, "time_sim") == 0 || strcmp(field, "time_glob") == 0 ||
             strcmp(field, "time_sim_la") == 0 || strcmp(field, "time_sim_ad") == 0 || strcmp(field, "time_qp_xcond") == 0)
        *static_cast<double*>(rv) = 0.0;
    else if (strcmp(field, "sqp_iter") == 0 || strcmp(field, "nlp_iter") == 0) *static_cast<int*>(rv) = o->sqp_iter;
    else if (strcmp(field, "qp_iter") == 0) *static_cast<int*>(rv) = o->qp_iter;
    else if (strcmp(field, "status") == 0) *static_cast<int*>(rv) = o->status;
    else if (strcmp(field, "qp_status") == 0 || strcmp(field, "qp_stat") == 0) *static_cast<int*>(rv) = o->status == ACADOS_QP_FAILURE ? 2 : 0;
    else if (strcmp(field, "stat_m") == 0) *static_cast<int*>(rv) = 2;
    else if (strcmp(field, "stat_n") == 0) *static_cast<int*>(rv) = 2;
    else if (strcmp(field, "statistics") == 0) {
        // (stat_n + 1) x nrow, column-major with nrow = min(sqp_iter + 1, stat_m) rows: iter | qp_stat | qp_iter
        double* st = static_cast<double*>(rv);
        const int nrow = o->sqp_iter + 1 < 2 ? o->sqp_iter + 1 : 2;
        for (int i = 0; i < nrow; ++i) {
            st[i] = i;
            st[i + nrow] = i == 0 ? 0.0 : (o->status == ACADOS_QP_FAILURE ? 2.0 : 0.0);
            st[i + 2 * nrow] = i == 0 ? 0.0 : (double)o->qp_iter;
        }
    } else complain("ocp_nlp_get", "field not carried by this layer");
}

void ocp_nlp_out_print(ocp_nlp_dims* dims, ocp_nlp_out* out) {
    Ocp* o = out ? static_cast<Ocp*>(out->mpc_impl) : nullptr;
    if (!o || !dims) return;
    for (int k = 0; k <= dims->N; ++k) {
        double x[3], u[2] = {0, 0}, p[3] = {0, 0, 0};
        ocp_nlp_out_get(o->config, dims, out, k, "x", x);
        if (k < dims->N) { ocp_nlp_out_get(o->config, dims, out, k, "u", u); ocp_nlp_out_get(o->config, dims, out, k, "pi", p); }
        printf("k = %d\n\tx = %e %e %e\n", k, x[0], x[1], x[2]);
        if (k < dims->N) printf("\tu = %e %e\n\tpi = %e %e %e\n", u[0], u[1], p[0], p[1], p[2]);
    }
}

// ---------------- acados_c/sim_interface.h ----------------
int sim_in_set(void*, void*, sim_in* in, const char* field, void* value) {
    Sim* s = in ? static_cast<Sim*>(in->mpc_impl) : nullptr;
    if (!s || !field || !value) { complain("sim_in_set", "null argument"); return MPC_ACADOS_UNSUPPORTED; }
    const double* v = static_cast<const double*>(value);
    if (strcmp(field, "x") == 0) { memcpy(s->x, v, sizeof(s->x)); return ACADOS_SUCCESS; }
    if (strcmp(field, "u") == 0) { memcpy(s->u, v, sizeof(s->u)); return ACADOS_SUCCESS; }
    if (strcmp(field, "T") == 0) { in->T = v[0]; return ACADOS_SUCCESS; }
    complain("sim_in_set", "field not carried by this layer (x, u, T)");
    return MPC_ACADOS_UNSUPPORTED;
}
int sim_out_get(void*, void*, sim_out* out, const char* field, void* value) {
    Sim* s = out ? static_cast<Sim*>(out->mpc_impl) : nullptr;
    if (!s || !field || !value) { complain("sim_out_get", "null argument"); return MPC_ACADOS_UNSUPPORTED; }
    if (strcmp(field, "x") == 0 || strcmp(field, "xn") == 0) { memcpy(value, s->xn, sizeof(s->xn)); return ACADOS_SUCCESS; }
    complain("sim_out_get", "field not carried by this layer (x, xn)");
    return MPC_ACADOS_UNSUPPORTED;
}

// ---------------- blasfeo/include/blasfeo_d_aux_ext_dep.h ----------------
void d_print_mat(int m, int n, double* A, int lda) { print_mat(m, n, A, lda, false, "%9.5f "); }
void d_print_tran_mat(int m, int n, double* A, int lda) { print_mat(m, n, A, lda, true, "%9.5f "); }
void d_print_exp_mat(int m, int n, double* A, int lda) { print_mat(m, n, A, lda, false, "%e\t"); }
void d_print_exp_tran_mat(int m, int n, double* A, int lda) { print_mat(m, n, A, lda, true, "%e\t"); }

// ---------------- acados_solver_<name>.h / acados_sim_solver_<name>.h ----------------
#define MPC_OCP_ENTRY_POINTS(NAME)                                                                                      \
    NAME##_solver_capsule* NAME##_acados_create_capsule(void) {                                                         \
        return static_cast<NAME##_solver_capsule*>(calloc(1, sizeof(NAME##_solver_capsule)));                           \
    }                                                                                                                   \
    int NAME##_acados_free_capsule(NAME##_solver_capsule* c) { free(c); return 0; }                                     \
    int NAME##_acados_create(NAME##_solver_capsule* c) { return ocp_create((OcpCapsule*)c, #NAME, 0, nullptr); }        \
    int NAME##_acados_create_with_discretization(NAME##_solver_capsule* c, int n, double* steps) {                      \
        return ocp_create((OcpCapsule*)c, #NAME, n, steps);                                                             \
    }                                                                                                                   \
    int NAME##_acados_reset(NAME##_solver_capsule* c, int) { return ocp_reset((OcpCapsule*)c, #NAME "_acados_reset"); } \
    int NAME##_acados_update_time_steps(NAME##_solver_capsule* c, int N, double* steps) {                               \
        Ocp* o = ocp_of((OcpCapsule*)c);                                                                                \
        if (!o || !steps) return 1;                                                                                     \
        if (N != o->g->spec.N) { complain(#NAME "_acados_update_time_steps", "N differs from the solver's horizon"); return 1; } \
        for (int k = 1; k < N; ++k)                                                                                     \
            if (fabs(steps[k] - steps[0]) > 1e-12 * fabs(steps[0])) {                                                   \
                complain(#NAME "_acados_update_time_steps", "non-uniform time steps are not supported");                \
                return 1;                                                                                               \
            }                                                                                                           \
        return steps[0] == o->g->spec.Ts ? 0 : ocp_new_time_step(o, steps[0], #NAME "_acados_update_time_steps");       \
    }                                                                                                                   \
    int NAME##_acados_update_qp_solver_cond_N(NAME##_solver_capsule* c, int n) {                                        \
        Ocp* o = ocp_of((OcpCapsule*)c);                                                                                \
        if (o && n == o->g->spec.N) return 0;   /* qp_cond_N = N: no condensing, what every variant is generated with */ \
        complain(#NAME "_acados_update_qp_solver_cond_N", "only qp_cond_N = N (no condensing) is implemented");         \
        return 1;                                                                                                       \
    }                                                                                                                   \
    int NAME##_acados_update_params(NAME##_solver_capsule*, int, double*, int np) {                                     \
        if (np == 0) return 0;                                                                                          \
        complain(#NAME "_acados_update_params", "the model has no parameters (np = 0)");                                \
        return 1;                                                                                                       \
    }                                                                                                                   \
    int NAME##_acados_update_params_sparse(NAME##_solver_capsule*, int, int*, double*, int n_update) {                  \
        if (n_update == 0) return 0;                                                                                    \
        complain(#NAME "_acados_update_params_sparse", "the model has no parameters (np = 0)");                         \
        return 1;                                                                                                       \
    }                                                                                                                   \
    int NAME##_acados_solve(NAME##_solver_capsule* c) { return ocp_solve((OcpCapsule*)c, #NAME "_acados_solve"); }      \
    void NAME##_acados_batch_solve(NAME##_solver_capsule** cs, int n) {                                                 \
        ocp_batch_solve((OcpCapsule**)cs, n, #NAME "_acados_batch_solve");                                              \
    }                                                                                                                   \
    int NAME##_acados_free(NAME##_solver_capsule* c) { return ocp_free((OcpCapsule*)c); }                               \
    void NAME##_acados_print_stats(NAME##_solver_capsule* c) {                                                          \
        Ocp* o = ocp_of((OcpCapsule*)c);                                                                                \
        if (!o) return;                                                                                                 \
        printf("iter\tqp_stat\tqp_iter\n0\t0\t0\n");                                                                    \
        if (o->sqp_iter > 0) printf("1\t%d\t%d\n", o->status == ACADOS_QP_FAILURE ? 2 : 0, o->qp_iter);                 \
    }                                                                                                                   \
    int NAME##_acados_custom_update(NAME##_solver_capsule*, double*, int) {                                             \
        printf("\n" #NAME "_acados_custom_update: no custom update function was generated for this solver.\n");         \
        return 1;                                                                                                       \
    }                                                                                                                   \
    ocp_nlp_in* NAME##_acados_get_nlp_in(NAME##_solver_capsule* c) { return c->nlp_in; }                                \
    ocp_nlp_out* NAME##_acados_get_nlp_out(NAME##_solver_capsule* c) { return c->nlp_out; }                             \
    ocp_nlp_out* NAME##_acados_get_sens_out(NAME##_solver_capsule* c) { return c->sens_out; }                           \
    ocp_nlp_solver* NAME##_acados_get_nlp_solver(NAME##_solver_capsule* c) { return c->nlp_solver; }                    \
    ocp_nlp_config* NAME##_acados_get_nlp_config(NAME##_solver_capsule* c) { return c->nlp_config; }                    \
    void* NAME##_acados_get_nlp_opts(NAME##_solver_capsule* c) { return c->nlp_opts; }                                  \
    ocp_nlp_dims* NAME##_acados_get_nlp_dims(NAME##_solver_capsule* c) { return c->nlp_dims; }                          \
    ocp_nlp_plan_t* NAME##_acados_get_nlp_plan(NAME##_solver_capsule* c) { return c->nlp_solver_plan; }                 \
    /* ---- integrator ---- */                                                                                          \
    NAME##_sim_solver_capsule* NAME##_acados_sim_solver_create_capsule(void) {                                          \
        return static_cast<NAME##_sim_solver_capsule*>(calloc(1, sizeof(NAME##_sim_solver_capsule)));                   \
    }                                                                                                                   \
    int NAME##_acados_sim_solver_free_capsule(NAME##_sim_solver_capsule* c) { free(c); return 0; }                      \
    int NAME##_acados_sim_create(NAME##_sim_solver_capsule* c) { return sim_create((SimCapsule*)c, #NAME); }            \
    int NAME##_acados_sim_solve(NAME##_sim_solver_capsule* c) { return sim_solve((SimCapsule*)c, #NAME "_acados_sim_solve"); } \
    void NAME##_acados_sim_batch_solve(NAME##_sim_solver_capsule** cs, int n) {                                         \
        sim_batch_solve((SimCapsule**)cs, n, #NAME "_acados_sim_batch_solve");                                          \
    }                                                                                                                   \
    int NAME##_acados_sim_free(NAME##_sim_solver_capsule* c) { return sim_free((SimCapsule*)c); }                       \
    int NAME##_acados_sim_update_params(NAME##_sim_solver_capsule*, double*, int np) {                                  \
        if (np == 0) return 0;                                                                                          \
        complain(#NAME "_acados_sim_update_params", "the model has no parameters (np = 0)");                            \
        return 1;                                                                                                       \
    }                                                                                                                   \
    sim_config* NAME##_acados_get_sim_config(NAME##_sim_solver_capsule* c) { return c->acados_sim_config; }             \
    sim_in* NAME##_acados_get_sim_in(NAME##_sim_solver_capsule* c) { return c->acados_sim_in; }                         \
    sim_out* NAME##_acados_get_sim_out(NAME##_sim_solver_capsule* c) { return c->acados_sim_out; }                      \
    void* NAME##_acados_get_sim_dims(NAME##_sim_solver_capsule* c) { return c->acados_sim_dims; }                       \
    sim_opts* NAME##_acados_get_sim_opts(NAME##_sim_solver_capsule* c) { return c->acados_sim_opts; }                   \
    sim_solver* NAME##_acados_get_sim_solver(NAME##_sim_solver_capsule* c) { return c->acados_sim_solver; }

MPC_OCP_ENTRY_POINTS(mobile_robot)
MPC_OCP_ENTRY_POINTS(mobile_robot_18)
MPC_OCP_ENTRY_POINTS(mobile_robot_25)
MPC_OCP_ENTRY_POINTS(mobile_robot_50)
MPC_OCP_ENTRY_POINTS(park)

}  // extern "C"
