/*
 * mpc_batch.h — C-ABI of the MI355X batched SQP-RTI solver (libmpc_amd.so).
 *
 * Plain C, plain pointers and sizes.  This is the drop-in boundary for the
 * reference's hot path.  Every entry point cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Semantics in one paragraph: a handle owns `capacity` independent OCP
 * "capsules" (iterate x/u trajectory, multipliers pi) in device memory.  One
 * `mpc_batch_solve*` call performs, for each of the first B capsules, exactly
 * what one `<name>_acados_solve()` does on one capsule: one SQP-RTI step
 * (ERK4 + forward sensitivities, LINEAR_LS Gauss-Newton cost, BGH box bounds,
 * x0 elimination, HPIPM-style Mehrotra IPM with Riccati recursion, full step),
 * or up to `sqp_max_iter` such steps when `sqp_max_iter > 1`.
 */
#ifndef MPC_BATCH_H
#define MPC_BATCH_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_NX 3 /* x, y, psi          (src/c_generated_code/acados_solver_mobile_robot.h NX) */
#define MPC_NU 2 /* v, delta           (… NU) */
#define MPC_NY 5 /* [x y psi v delta]  (… NY) */
#define MPC_MAX_N 127 /* largest horizon a handle accepts (reference max is 80) */

/* acados return codes reproduced per instance
 * (acados/utils/types.h; used at src/optimizer.cpp:290-293). */
#define MPC_SUCCESS 0
#define MPC_NAN_DETECTED 1
#define MPC_MAXITER 2
#define MPC_MINSTEP 3
#define MPC_QP_FAILURE 4

/* HPIPM modes (hpipm_common.h enum hpipm_mode); the reference uses BALANCE
 * (src/c_generated_code/acados_solver_mobile_robot.c:670). */
#define MPC_HPIPM_SPEED_ABS 0
#define MPC_HPIPM_SPEED 1
#define MPC_HPIPM_BALANCE 2
#define MPC_HPIPM_ROBUST 3

/* What `<name>_acados_create()` bakes into a capsule
 * (src/c_generated_code/acados_solver_mobile_robot.c:358-602, 608-697;
 *  per-variant values: SURVEY.md §2.1). */
typedef struct mpc_ocp_spec {
    int N;            /* shooting intervals (MOBILE_ROBOT_N) */
    double Ts;        /* time step = ERK step = cost scaling of stages 0..N-1 */
    double L;         /* wheelbase; 0 selects the generated literals (0.27 and 3.7037037037037033) */
    double W[5];      /* diag stage weight on [x y psi v delta] */
    double W_e[3];    /* diag terminal weight on [x y psi] */
    double lbu[2], ubu[2]; /* bounds on v, delta, stages 0..N-1 (idxbu = {0,1}) */
    double lbx[2], ubx[2]; /* bounds on x, y, stages 1..N-1   (idxbx = {0,1}) */
    int sqp_max_iter; /* 1 = SQP_RTI (the reference); >1 = full SQP: linearise, NLP residuals, exit if all
                         < nlp_tol (status 0), QP, full step; status 2 (MAXITER) after sqp_max_iter QPs */
    int qp_iter_max;  /* HPIPM iter_max (50) */
    int hpipm_mode;   /* MPC_HPIPM_BALANCE */
    double nlp_tol;   /* used only when sqp_max_iter > 1 (reference JSON: 1e-6) */
} mpc_ocp_spec_t;

typedef struct mpc_batch mpc_batch_t; /* opaque; owns host + device memory */

/* Fills `spec` with one of the reference's generated variants:
 * "mobile_robot" (N=50), "mobile_robot_18", "mobile_robot_25",
 * "mobile_robot_50" (N=80), "park".  Returns 0, or -1 for an unknown name.
 * Replaces: the constants of src/c_generated_code*\/acados_solver_*.c. */
int mpc_spec_variant(const char* name, mpc_ocp_spec_t* spec);

/* Replaces mobile_robot_acados_create_capsule() + mobile_robot_acados_create()
 * (src/c_generated_code/acados_solver_mobile_robot.c:84-111, 755-804) for
 * `capacity` capsules at once.  `device_ids`/`n_devices` select the GPUs the
 * batch is sharded over in contiguous slices (NULL/0 = current device).
 * Initial iterate is all zero (acados_solver_mobile_robot.c:703-729).
 * Returns 0, or a negative code (-1 bad argument, -2 HIP error, -3 no device). */
int mpc_batch_create(mpc_batch_t** out, const mpc_ocp_spec_t* spec, int capacity,
                     const int* device_ids, int n_devices);

/* Replaces the tick body of Optimizer::update_and_solve, src/optimizer.cpp:251-296:
 *   ocp_nlp_cost_model_set(.., j, "yref", ..) for j=0..N        -> yref, yref_e
 *   ocp_nlp_constraints_model_set(.., 0, "lbx"/"ubx", x0)       -> x0
 *   mobile_robot_acados_solve() / mobile_robot_acados_batch_solve()
 *       (acados_solver_mobile_robot.c:942-961)                  -> the solve
 *   ocp_nlp_out_get(.., 0, "u", ..)                             -> u0
 * All pointers are HOST pointers, AoS per instance as the reference stores them:
 *   x0 [B][3], yref [B][N][5], yref_e [B][3], u0 [B][2] (out),
 *   xtraj [B][N+1][3], utraj [B][N][2]: optional, in/out.  If non-NULL they
 *   are loaded as the iterate before the solve (ocp_nlp_out_set "x"/"u") and
 *   receive the new iterate afterwards; NULL keeps the handle's internal
 *   iterate (the capsule's warm start).
 *   status [B] (out, acados codes), qp_iter [B] (out, HPIPM iterations of the
 *   last QP); both optional.
 * An instance whose QP fails (status 4) keeps its previous iterate and u0 is
 * the previous u_0, as the reference keeps u_current (src/optimizer.cpp:290-293).
 * Blocking.  Returns 0 or a negative code.
 * Data movement: batches of up to 64 instances go through pinned staging the kernel reads and
 * writes itself; arrays the caller keeps in pinned memory (hipHostMalloc / hipHostRegister) are
 * read in place by the kernels (single-device handles); pageable arrays are copied to the device,
 * large batches in up to three chunks overlapped with the solve. */
int mpc_batch_solve(mpc_batch_t* h, int B, const double* x0, const double* yref,
                    const double* yref_e, double* u0, double* xtraj, double* utraj,
                    int* status, int* qp_iter);

/* Same call with DEVICE pointers (same AoS layouts) on the handle's first
 * device, enqueued on `stream` (a hipStream_t, NULL = the handle's stream) and
 * NOT synchronised: for callers that keep the batch resident in HBM (bench,
 * Monte-Carlo drivers).  Single-device handles only.  No counterpart in the
 * reference beyond acados_solver_mobile_robot.c:951-961.
 * A handle's calls share its capsules and workspace: with a stream of the caller's, the caller
 * orders this call against the handle's other calls (those run on the handle's own stream and
 * wait for it before they return); two solves of one handle must never overlap. */
int mpc_batch_solve_device(mpc_batch_t* h, int B, const double* d_x0, const double* d_yref,
                           const double* d_yref_e, double* d_u0, double* d_xtraj,
                           double* d_utraj, int* d_status, int* d_qp_iter, void* stream);

/* The one entry point that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot call hipMalloc — code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  It is mpc_batch_solve_device's SQP_RTI step (same kernels,
 * same results) on capsules that live in a fixed array inside the library (one copy per GPU,
 * 36 MiB, held until the process ends) instead of in a handle:
 *   spec            as for mpc_batch_create; sqp_max_iter must be 1 (SQP_RTI)
 *   cold != 0       the first B capsules start from the all-zero iterate (a fresh handle);
 *   cold == 0       they start from the iterate the previous call ON THIS GPU left (a warm handle);
 *                   refused with -1 unless the array holds iterates of the same horizon N for at
 *                   least B capsules (the library records, per GPU, the horizon of the last cold
 *                   call and the largest batch started at that horizon)
 *   d_x0 [B][3], d_yref [B][N][5], d_yref_e [B][3]  device pointers, in
 *   d_u0 [B][2] out (required), d_status [B], d_qp_iter [B] out (may be NULL)
 *   stream          a hipStream_t (NULL = the null stream); the call is NOT synchronised.
 * Limits: B * (N + 1) <= MPC_STATIC_MAX_LINES and B <= MPC_STATIC_MAX_B, i.e. 10280 instances at
 * N = 50; larger batches are refused with -4 and a message on stderr.  Returns 0, -1 (bad
 * argument), -2 (HIP error), -3 (no device), -4 (too large).  It calls no hipMalloc / hipFree or
 * any allocating routine; the first call on a GPU looks up that GPU's copy of the array
 * (hipGetSymbolAddress) and loads the kernels — make one call before starting a graph capture.
 * A call recorded in a HIP graph replays correctly: everything a solve carries from launch to
 * launch (the iterates, the list of instances handed to the sequential kernel and which of its two
 * counters is in use) is device state, nothing is frozen into the kernel arguments at capture time.
 * Every call on a GPU shares that GPU's array and the array keeps the last call's iterates:
 * calls that overlap on one GPU (different streams or threads) race; serialise them. */
#define MPC_STATIC_MAX_LINES 524288 /* capsule lines (64 B each) of the static workspace */
#define MPC_STATIC_MAX_B 262144     /* instances (status words) of the static workspace */
int mpc_batch_solve_static(const mpc_ocp_spec_t* spec, int B, int cold, const double* d_x0,
                           const double* d_yref, const double* d_yref_e, double* d_u0,
                           int* d_status, int* d_qp_iter, void* stream);

/* Replaces mobile_robot_acados_reset(capsule, 1)
 * (acados_solver_mobile_robot.c:830-869; Optimizer::reset_solver,
 * include/optimizer.hpp:446-456): zero x, u, pi of the selected capsules.
 * mask [capacity] (host) selects capsules, NULL = all. */
int mpc_batch_reset(mpc_batch_t* h, const unsigned char* mask);

/* Shifted warm start.  No counterpart in the reference, whose capsules keep
 * the previous iterate unshifted between ticks (src/optimizer.cpp:283-296); it is what an acados
 * user does between RTI calls with ocp_nlp_out_get(stage + 1) / ocp_nlp_out_set(stage) loops:
 * x_k <- x_{k+1}, u_k <- u_{k+1}, pi_k <- pi_{k+1}, lam_k <- lam_{k+1} for k = 0..N-2,
 * x_{N-1} <- x_N; x_N, u_{N-1}, pi_{N-1}, lam_{N-1} are kept (the last interval is repeated).
 * mask [capacity] (host) selects capsules, NULL = all. */
int mpc_batch_shift(mpc_batch_t* h, const unsigned char* mask);

/* Replaces ocp_nlp_out_get(.., stage, "x"|"u"|"pi", ..) for whole trajectories
 * (src/c_generated_code/main_mobile_robot.c:117-120).  Host pointers, any may be
 * NULL: xtraj [B][N+1][3], utraj [B][N][2], pi [B][N][3]. */
int mpc_batch_get(mpc_batch_t* h, int B, double* xtraj, double* utraj, double* pi);

/* Replaces mobile_robot_acados_sim_solve() as used by
 * Optimizer::integrate_next_states (src/optimizer.cpp:321-341;
 * acados_sim_solver_mobile_robot.c:205-214): one ERK4 step of length Ts of the
 * plant, per instance.  Host pointers x [B][3] (in/out), u [B][2]. */
int mpc_batch_sim_step(mpc_batch_t* h, int B, double* x, const double* u);

/* Replaces mobile_robot_acados_free() + free_capsule()
 * (acados_solver_mobile_robot.c:964-993, 98-103). */
void mpc_batch_free(mpc_batch_t* h);

/* ---- Closed loop on the device -----------------------------------------
 * B vehicles following one shared path, every tick entirely on the GPU: waypoint-window
 * selection, reference packing, the solve, the plant step.  Replaces the offline loop
 * Optimizer::run (src/optimizer.cpp:177-206) = per tick Optimizer::update_and_solve
 * (:208-319, with find_next_waypoint :384-431 and find_closest_waypoint :343-382) +
 * Optimizer::integrate_next_states (:321-341), for B independent Optimizer objects at once. */

/* Replaces the path loading of Optimizer::Optimizer (src/optimizer.cpp:138-160):
 * state_refs [M][3] (x, y, psi), input_refs [Mi][2] (v, delta), host pointers, copied. */
int mpc_batch_track_set_path(mpc_batch_t* h, int M, const double* state_refs, int Mi, const double* input_refs);

/* Replaces Optimizer::Optimizer(T, N, v_ref, x_init, y_init, yaw_init) (src/optimizer.cpp:20-176)
 * for B vehicles: x_init [B][3]; waypoint indices, counters and commands start at zero;
 * density = 1/T/v_ref, window limit = floor(3.95 / (min(v_ref, 0.35 if v_ref > 0.3) * T)).
 * Does not touch the capsules' iterates (use mpc_batch_reset for that). */
int mpc_batch_track_init(mpc_batch_t* h, int B, const double* x_init, double T, double v_ref);

/* on != 0: mpc_batch_track_run shifts every capsule's iterate (mpc_batch_shift) at the end of
 * each tick.  Off by default = the reference's behaviour. */
int mpc_batch_track_set_shift(mpc_batch_t* h, int on);

/* Runs n_ticks control ticks for the first B vehicles and returns, after the last tick (any
 * pointer may be NULL): x [B][3] state, u [B][2] command in force (the previous one where the
 * solver failed, src/optimizer.cpp:290-293), target [B] target_waypoint_index, status [B].
 * Blocking. */
int mpc_batch_track_run(mpc_batch_t* h, int B, int n_ticks, double* x, double* u, int* target, int* status);

/* Full SQP only (sqp_max_iter > 1; BASELINE.json config 5, acados nlp_solver_type "SQP", the
 * commented-out option at scripts/mpc_acados.py:234-235; tolerances
 * scripts/mobile_robot_acados_ocp.json "nlp_solver_tol_*"): number of QPs solved per instance
 * and the last NLP residuals {stat, eq, ineq, comp} of the last mpc_batch_solve*.  Replaces
 * ocp_nlp_get(.., "sqp_iter", ..) / ocp_nlp_get(.., "res_stat", ..) ….  Host pointers, may be NULL:
 * sqp_iter [B], res4 [B][4]. */
int mpc_batch_get_sqp(mpc_batch_t* h, int B, int* sqp_iter, double* res4);

/* Per-instance diagnostic bits of the last solve (host pointer diag [B]); 0 on every workload the
 * bench measures.  No counterpart in the reference: they record where the solve took a path of
 * its own next to HPIPM's.
 *   1  REFINED: HPIPM's corrector iterative refinement was PERFORMED in at least one iteration
 *      (itref_corr_max = 2, acados_solver_mobile_robot.c:670 BALANCE mode): the residual of the
 *      linearised KKT system at the corrector's direction was not below (res_g_max, res_b_max,
 *      res_d_max, res_m_max), so the system was solved again for the residual with the same
 *      factorisation, up to twice, and the step lengths recomputed (wave models)
 *   2  LQ: HPIPM's lq_fact = 1 switch was PERFORMED (BALANCE mode): the affine direction of the regular
 *      factorisation left a residual above 1e-5 in the linearised KKT system, so the stage Hessians
 *      were refactorised from their square roots (HPIPM: LQ of [chol(RSQ) | sqrt(Gamma) | [B A]^T L];
 *      here: the same triangularisation by Givens rotations) and the QP stayed with that
 *      factorisation — by the sequential kernel, to which the scan models hand such an instance
 *      (its scan direction is far above the acceptance threshold of bit 4)
 *   4  wave models: a direction computed with the parallel scans left the stationarity or the
 *      dynamics row of the linear system above the acceptance threshold (dynamics rows: 1e-8;
 *      stationarity rows: 2e-6 x the smallest cost-Hessian entry, within [1e-8, 2.5e-7] — 1e-8 with
 *      the reference's weights; mpc_amd/csrc/consts.hpp kScanTolLin); the instance went to the
 *      sequential kernel (sequential Riccati recursion + the refinement of bit 1), launched behind
 *      the scan kernel on the instances it declined.  A scan direction below the threshold is taken as is.
 *      Lane model (the cross-check model), SQP_RTI: its sweeps do not refine; an instance whose QP
 *      reached the refinement of bit 1 was handed to the same sequential kernel
 *   8  lane model in full SQP only: the refinement of bit 1 applied and was NOT performed; the
 *      instance stops with status MPC_QP_FAILURE (4) at its last accepted iterate
 *  16  with bit 4, SQP_RTI handles: the sequential kernel CARRIED ON from the interior-point iterate of
 *      the declined iteration (the iterations before it are the scan kernel's, all below the
 *      threshold) instead of solving the QP again from HPIPM's cold start.  Not set: from the cold
 *      start — full-SQP handles, instances handed over by the lane model, handles created with
 *      MPC_RESUME=0 in the environment, and mpc_batch_solve_static (its array has no room for the
 *      iterates: 192 B per stage).  Either way status and iteration count are HPIPM's */
#define MPC_DIAG_ITREF 1
#define MPC_DIAG_LQ 2
#define MPC_DIAG_SCAN_FALLBACK 4
#define MPC_DIAG_ITREF_SKIPPED 8
#define MPC_DIAG_RESUMED 16
int mpc_batch_get_diag(mpc_batch_t* h, int B, int* diag);

/* Execution model of the solve kernels (no counterpart in the reference; results agree to
 * rounding, ~1e-13, across models).  Leave it on AUTO; the others exist for cross-checks.
 *   AUTO / WAVE  the wave model, kernel chosen by horizon, mode and batch size: lane = stage(s) of
 *            one OCP, IPM workspace in registers and LDS, Riccati recursion and the two
 *            substitutions as parallel scans.  Two stages per lane (two OCPs per wavefront at
 *            N = 50, three at 32 <= N <= 41, four at N <= 31, one at 64 <= N <= 127) for SQP_RTI
 *            batches of more than 16384 instances (6144 where three share a wavefront); one stage
 *            per lane for smaller batches (down to the batch = 1 control loop), N <= 15 and full
 *            SQP with N <= 63;
 *   WAVE1    one stage per lane wherever it applies (N <= 63);   WAVE2: two stages per lane
 *            wherever it applies (N >= 16, SQP_RTI), whatever the batch size;
 *   WAVE_SEQ / WAVE2_SEQ  (one and the same) two stages per lane with the sequential recursions
 *            and HPIPM's iterative refinement — the kernel the scan models hand their declined
 *            instances to (MPC_DIAG_SCAN_FALLBACK), here run on every instance; SQP_RTI only;
 *   LANE     one OCP per lane, IPM workspace in HBM (allocated on the first solve that uses it).
 * The environment variable MPC_MODEL=lane|wave|wave1|wave2|waveseq sets the default of handles
 * left on AUTO. */
#define MPC_MODEL_AUTO 0
#define MPC_MODEL_LANE 1
#define MPC_MODEL_WAVE 2
#define MPC_MODEL_WAVE_SEQ 3
#define MPC_MODEL_WAVE1 4
#define MPC_MODEL_WAVE2 5
#define MPC_MODEL_WAVE2_SEQ 6
int mpc_batch_set_model(mpc_batch_t* h, int model);

/* Introspection used by bench/tests: kernel time of the last solve in ms
 * (hipEvents on the handle's stream around the device work), horizon,
 * capacity, and the last HIP error string of this handle. */
double mpc_batch_last_kernel_ms(mpc_batch_t* h);
double mpc_batch_last_phase0_ms(mpc_batch_t* h); /* device time of the dominant solve kernel (k_wave / first k_phase) */
/* Test hook, no counterpart in the reference (its QP data stay inside libacados): what the
 * preparation phase hands to the QP at the capsules' CURRENT iterate, for yref / yref_e given
 * as host arrays: out [B][N+1][24] (host) = a, b, B00, B10, B20, B01, B11, B21 (the non-trivial
 * entries of A_k = [[1,0,a],[0,1,b],[0,0,1]] and of B_k, row-major names), b_k = phi(x_k,u_k) -
 * x_{k+1} (3), LINEAR_LS gradient in [v, delta, x, y, psi] order (5), BGH residuals lb - z (4)
 * and z - ub (4) on (v, delta, x, y); entries a stage does not have are 0. */
int mpc_batch_debug_linearize(mpc_batch_t* h, int B, const double* yref, const double* yref_e, double* out);
/* Device memory the handle holds right now, in bytes (capsules + what the calls made so far
 * have allocated: host-pointer staging, lane-model workspace, tracker). */
size_t mpc_batch_device_bytes(const mpc_batch_t* h);
int mpc_batch_horizon(const mpc_batch_t* h);
int mpc_batch_capacity(const mpc_batch_t* h);
const char* mpc_batch_last_error(const mpc_batch_t* h);
const char* mpc_batch_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPC_BATCH_H */
