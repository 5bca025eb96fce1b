/*
 * acados_compat_main.c — a plain-C caller of the acados-named entry points
 * (include/acados_compat/), written for this repository.  It makes the call sequence of the
 * reference's callers: src/optimizer.cpp:37-63 (create + getters), :117-119 (dims), :251-296 (tick:
 * yref of every stage, lbx/ubx of stage 0, solve, u of stage 0), :323-338 (plant step),
 * include/optimizer.hpp:446-456 (reset), src/c_generated_code/main_mobile_robot.c:114-130
 * (ocp_nlp_out_set / ocp_nlp_out_get of whole trajectories) and acados_solver_mobile_robot.c:951-961
 * (batch_solve).  Compiled once per variant with -DV=<name>.
 *
 * usage: acados_compat_<name> in.bin out.bin
 *   in.bin : int32 T, int32 N, then T records of doubles x0[3], yref[N][5], yref_e[3]
 *   out.bin: doubles, sections in the order they are produced below (tests/test_acados_compat.py)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define STR_(a) #a
#define STR(a) STR_(a)
#define F(suffix) CAT(V, suffix)

#include "acados/utils/math.h"
#include "acados_c/ocp_nlp_interface.h"
#include "acados_c/sim_interface.h"
#include STR(CAT(acados_solver_, V).h)
#include STR(CAT(acados_sim_solver_, V).h)

typedef F(_solver_capsule) capsule_t;
typedef F(_sim_solver_capsule) sim_capsule_t;

static FILE *out;
static void put(const double *v, int n) { fwrite(v, sizeof(double), (size_t)n, out); }
static void put1(double v) { put(&v, 1); }

typedef struct { double x0[3]; double *yref; double yref_e[3]; } tick_t;

/* the tick body of Optimizer::update_and_solve (src/optimizer.cpp:251-296) */
static int tick(capsule_t *c, int N, const tick_t *t, double u[2])
{
    ocp_nlp_config *cfg = F(_acados_get_nlp_config)(c);
    ocp_nlp_dims *dims = F(_acados_get_nlp_dims)(c);
    ocp_nlp_in *in = F(_acados_get_nlp_in)(c);
    ocp_nlp_out *o = F(_acados_get_nlp_out)(c);
    double ye[5] = {t->yref_e[0], t->yref_e[1], t->yref_e[2], 0.0, 0.0};
    ocp_nlp_cost_model_set(cfg, dims, in, N, "yref", ye);
    for (int j = 0; j < N; ++j) ocp_nlp_cost_model_set(cfg, dims, in, j, "yref", t->yref + 5 * j);
    ocp_nlp_constraints_model_set(cfg, dims, in, 0, "lbx", (void *)t->x0);
    ocp_nlp_constraints_model_set(cfg, dims, in, 0, "ubx", (void *)t->x0);
    int status = F(_acados_solve)(c);
    if (status != 0) return status;            /* u_current is left alone (:290-293) */
    ocp_nlp_out_get(cfg, dims, o, 0, "u", u);
    return 0;
}

static void put_traj(capsule_t *c, int N)
{
    ocp_nlp_config *cfg = F(_acados_get_nlp_config)(c);
    ocp_nlp_dims *dims = F(_acados_get_nlp_dims)(c);
    ocp_nlp_out *o = F(_acados_get_nlp_out)(c);
    double v[3];
    for (int k = 0; k <= N; ++k) { ocp_nlp_out_get(cfg, dims, o, k, "x", v); put(v, 3); }
    for (int k = 0; k < N; ++k) { ocp_nlp_out_get(cfg, dims, o, k, "u", v); put(v, 2); }
    for (int k = 0; k < N; ++k) { ocp_nlp_out_get(cfg, dims, o, k, "pi", v); put(v, 3); }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *fi = fopen(argv[1], "rb");
    out = fopen(argv[2], "wb");
    if (!fi || !out) { perror("open"); return 2; }
    int T = 0, Nf = 0;
    if (fread(&T, 4, 1, fi) != 1 || fread(&Nf, 4, 1, fi) != 1) return 2;
    tick_t *tk = (tick_t *)calloc((size_t)T, sizeof(tick_t));
    for (int t = 0; t < T; ++t) {
        tk[t].yref = (double *)malloc(sizeof(double) * 5 * (size_t)Nf);
        if (fread(tk[t].x0, 8, 3, fi) != 3 || fread(tk[t].yref, 8, 5 * (size_t)Nf, fi) != 5 * (size_t)Nf ||
            fread(tk[t].yref_e, 8, 3, fi) != 3) return 2;
    }
    fclose(fi);

    /* ---- section 1: create, dims (src/optimizer.cpp:37-63,117-119) ---- */
    capsule_t *c = F(_acados_create_capsule)();
    int status = F(_acados_create)(c);
    if (status) { printf(STR(V) "_acados_create() returned status %d. Exiting.\n", status); return 1; }
    ocp_nlp_config *cfg = F(_acados_get_nlp_config)(c);
    ocp_nlp_dims *dims = F(_acados_get_nlp_dims)(c);
    ocp_nlp_out *o = F(_acados_get_nlp_out)(c);
    ocp_nlp_solver *solver = F(_acados_get_nlp_solver)(c);
    const int N = dims->N, nx = *dims->nx, nu = *dims->nu;
    if (N != Nf) { fprintf(stderr, "horizon of the input file (%d) != nlp_dims->N (%d)\n", Nf, N); return 2; }
    put1(N); put1(nx); put1(nu);

    /* ---- section 2: T closed ticks on one capsule + plant step; per tick: status, u[2], x_next[3],
     *      qp_iter, sqp_iter ---- */
    sim_capsule_t *sc = F(_acados_sim_solver_create_capsule)();
    status = F(_acados_sim_create)(sc);
    if (status) { printf("acados_create() simulator returned status %d. Exiting.\n", status); return 1; }
    sim_config *scfg = F(_acados_get_sim_config)(sc);
    void *sdims = F(_acados_get_sim_dims)(sc);
    sim_in *sin_ = F(_acados_get_sim_in)(sc);
    sim_out *sout = F(_acados_get_sim_out)(sc);
    double u[2] = {0.0, 0.0};
    double time_max = 0.0;
    for (int t = 0; t < T; ++t) {
        status = tick(c, N, &tk[t], u);
        double xs[3] = {tk[t].x0[0], tk[t].x0[1], tk[t].x0[2]};
        sim_in_set(scfg, sdims, sin_, "x", xs);
        sim_in_set(scfg, sdims, sin_, "u", u);
        int ss = F(_acados_sim_solve)(sc);
        if (ss != ACADOS_SUCCESS) { fprintf(stderr, "acados integrator returned status %d\n", ss); return 1; }
        sim_out_get(scfg, sdims, sout, "x", xs);
        int qp_iter = -1, sqp_iter = -1;
        double tt = 0.0;
        ocp_nlp_get(cfg, solver, "qp_iter", &qp_iter);
        ocp_nlp_get(cfg, solver, "sqp_iter", &sqp_iter);
        ocp_nlp_get(cfg, solver, "time_tot", &tt);
        time_max = MAX(time_max, tt);
        put1(status); put(u, 2); put(xs, 3); put1(qp_iter); put1(sqp_iter);
    }
    /* ---- section 3: the whole iterate after the last tick ---- */
    put_traj(c, N);

    /* ---- section 4: reset, then tick 0 again = the cold tick (include/optimizer.hpp:446-456) ---- */
    status = F(_acados_reset)(c, 1);
    put1(status);
    status = tick(c, N, &tk[0], u);
    put1(status); put(u, 2);

    /* ---- section 5: ocp_nlp_out_set of a whole trajectory, then a solve
     *      (main_mobile_robot.c:114-121): x_k = x0 of tick 1, u_k = (0.1, 0.01) ---- */
    {
        double ui[2] = {0.1, 0.01};
        int rti_phase = 0;
        for (int k = 0; k < N; ++k) {
            ocp_nlp_out_set(cfg, dims, o, k, "x", tk[1].x0);
            ocp_nlp_out_set(cfg, dims, o, k, "u", ui);
        }
        ocp_nlp_out_set(cfg, dims, o, N, "x", tk[1].x0);
        ocp_nlp_solver_opts_set(cfg, F(_acados_get_nlp_opts)(c), "rti_phase", &rti_phase);
        double back[3];
        ocp_nlp_out_get(cfg, dims, o, N / 2, "x", back);    /* reads back what was set */
        put(back, 3);
        status = tick(c, N, &tk[1], u);
        put1(status); put(u, 2);
        put_traj(c, N);
    }

    /* ---- section 6: batch_solve over T capsules (capsule i <- tick i), twice (cold, warm), then one
     *      single solve on capsule 1 and a third batched call (acados_solver_mobile_robot.c:951-961) ---- */
    {
        capsule_t **cs = (capsule_t **)calloc((size_t)T, sizeof(capsule_t *));
        for (int i = 0; i < T; ++i) {
            cs[i] = F(_acados_create_capsule)();
            if (F(_acados_create)(cs[i])) return 1;
        }
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < T; ++i) {
                ocp_nlp_config *ci = F(_acados_get_nlp_config)(cs[i]);
                ocp_nlp_dims *di = F(_acados_get_nlp_dims)(cs[i]);
                ocp_nlp_in *ii = F(_acados_get_nlp_in)(cs[i]);
                double ye[5] = {tk[i].yref_e[0], tk[i].yref_e[1], tk[i].yref_e[2], 0.0, 0.0};
                ocp_nlp_cost_model_set(ci, di, ii, N, "yref", ye);
                for (int j = 0; j < N; ++j) ocp_nlp_cost_model_set(ci, di, ii, j, "yref", tk[i].yref + 5 * j);
                ocp_nlp_constraints_model_set(ci, di, ii, 0, "lbx", tk[i].x0);
                ocp_nlp_constraints_model_set(ci, di, ii, 0, "ubx", tk[i].x0);
            }
            if (rep == 2) {   /* a single solve in between pulls capsule 1 out of the batch handle */
                status = F(_acados_solve)(cs[1]);
                double u1[2];
                ocp_nlp_out_get(F(_acados_get_nlp_config)(cs[1]), F(_acados_get_nlp_dims)(cs[1]), F(_acados_get_nlp_out)(cs[1]), 0, "u", u1);
                put1(status); put(u1, 2);
            }
            F(_acados_batch_solve)(cs, T);
            for (int i = 0; i < T; ++i) {
                int st = -1;
                double ui[2];
                ocp_nlp_get(F(_acados_get_nlp_config)(cs[i]), F(_acados_get_nlp_solver)(cs[i]), "status", &st);
                ocp_nlp_out_get(F(_acados_get_nlp_config)(cs[i]), F(_acados_get_nlp_dims)(cs[i]), F(_acados_get_nlp_out)(cs[i]), 0, "u", ui);
                put1(st); put(ui, 2);
            }
        }
        put_traj(cs[T - 1], N);
        for (int i = 0; i < T; ++i) {
            if (F(_acados_free)(cs[i])) return 1;
            if (F(_acados_free_capsule)(cs[i])) return 1;
        }
        free(cs);
    }

    /* ---- section 7: what this layer refuses (returns MPC_ACADOS_UNSUPPORTED, prints, does not exit) ---- */
    {
        ocp_nlp_in *in = F(_acados_get_nlp_in)(c);
        double ubu_other[2] = {2.0, 0.4};
        double w_bad[25] = {0};
        int r1 = ocp_nlp_constraints_model_set(cfg, dims, in, 3, "ubu", ubu_other);   /* src/optimizer.cpp:273-274 (commented out there) */
        int r2 = ocp_nlp_cost_model_set(cfg, dims, in, 0, "W", w_bad);
        int r3 = ocp_nlp_cost_model_set(cfg, dims, in, 0, "no_such_field", w_bad);
        int idx[3] = {0, 1, 2};
        int r4 = ocp_nlp_constraints_model_set(cfg, dims, in, 0, "idxbx", idx);       /* main_mobile_robot_25.c */
        put1(r1); put1(r2); put1(r3); put1(r4);
        put1(F(_acados_update_params)(c, 0, NULL, 0));
        put1(F(_acados_update_qp_solver_cond_N)(c, N));
    }

    F(_acados_print_stats)(c);
    printf("max time_tot %.3f ms\n", time_max * 1e3);
    status = F(_acados_sim_free)(sc);
    F(_acados_sim_solver_free_capsule)(sc);
    status |= F(_acados_free)(c);
    status |= F(_acados_free_capsule)(c);
    fclose(out);
    for (int t = 0; t < T; ++t) free(tk[t].yref);
    free(tk);
    return status;
}
