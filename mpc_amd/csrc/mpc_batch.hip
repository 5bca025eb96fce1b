// mpc_batch.hip — libmpc_amd.so: HIP kernels (gfx950) + the C-ABI of include/mpc_batch.h.
//
// Two execution models, both FP64 VALU only — the 2x2 / 3x3 stage blocks
// are not dense contractions, so no MFMA (BASELINE.json north_star):
//   WAVE (default)  k_wave2: lane = two consecutive stages, 64 / GW OCPs per wavefront (two at
//                   N = 50), IPM workspace in registers + LDS, Riccati recursion and substitutions
//                   as parallel scans — rti_wave2.hpp; k_wave: one stage per lane (N <= 15, full
//                   SQP, cross-check model WAVE1) — rti_wave.hpp;
//   LANE            k_phase / k_compact: one OCP per lane, workspace slab in HBM, phased solve
//                   with straggler compaction — rti_core.hpp.
// This file adds the batch I/O kernels, the full-SQP head (k_nlp_res), the device closed loop
// (k_track_*), the handle, sharding over devices and the extern "C" entry points.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "consts.hpp"
#include "rti_wave.hpp"
#include "rti_wave2.hpp"

using namespace mpc;

#define MPC_WAVE 64
#define MPC_MAX_PHASES 8
// Phase schedule: batches up to this size run as one phase (latency path)
#define MPC_SINGLE_PHASE_MAX 2048
// Batches up to this size go through pinned host memory that the kernel reads and writes
// directly (no H2D / D2H copies): the batch = 1 control-loop path
#define MPC_ZEROCOPY_MAX 64
// wavefronts the chip runs at once at one wavefront per SIMD (256 CUs x 4 SIMDs)
#define MPC_SIMD_SLOTS 1024

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------

// what a scan kernel leaves in io.status for an instance it has handed to the list kernel (which
// overwrites it); never seen by a caller
#define MPC_STATUS_PENDING (-1)

// One phase of the SQP-RTI step, one OCP per lane (H4/H15: replaces the serial loop of
// mobile_robot_acados_batch_solve, acados_solver_mobile_robot.c:951-961).
//
// The HPIPM iteration count differs per instance (mean ~8, stragglers 20+), and a wave whose
// lanes have mostly retired still pays the full latency of every sweep.  So the step runs in
// phases: a wave leaves a phase as soon as fewer than `min_active` of its lanes (default 8
// of 64) are still iterating (or all are done); finished lanes write their result to their HOME capsule, the
// others park their scalars, k_compact packs the still-running instances densely into the
// other workspace, and the next phase continues with full waves.  The last phase
// (min_active = 0) runs every instance to completion.  No host synchronisation: the number of
// live instances is read from device memory and surplus blocks exit at once.
__global__ __launch_bounds__(MPC_WAVE) void k_phase(Consts c, Mem cur, Mem home, const int* count_in, int B,
                                                    int first, int min_active, int* flags) {
    const int cnt = first ? B : *count_in;
    const size_t wbase = (size_t)blockIdx.x * MPC_WAVE;
    const unsigned l = threadIdx.x;
    // which of the two list counters this solve uses (Mem::REDO): the first phase takes the scan side's
    // word and publishes it for the later phases and the list kernel
    const int par = home.REDO ? (first ? home.REDO[MPC_REDO_PS] : home.REDO[MPC_REDO_PL]) : 0;
    if (first && home.REDO && blockIdx.x == 0 && l == 0) { home.REDO[par ^ 1] = 0; home.REDO[MPC_REDO_PL] = par; }
    if (wbase >= (size_t)cnt) return;  // wave-uniform
    const bool inrange = wbase + l < (size_t)cnt;
    // full SQP: instances whose residual check passed (or whose QP failed) sit this round out
    const bool valid = inrange && !(c.sqp && first && home.ACTIVE[wbase + l] == 0);
    if (__ballot(valid) == 0ull) {
        if (flags && inrange) flags[wbase + l] = 0;
        return;
    }
    Lane lane(c, cur, wbase, l);
    Lane::State s;
    s.kk = 0; s.diag = 0; s.alpha = 0.0; s.mu = 0.0; s.home = 0.0;
    s.n.g = s.n.b = s.n.d = s.n.m = s.n.musum = 0.0;
    if (valid) {
        if (first) lane.begin(s, wbase + l);
        else lane.load_state(s);
    }
    for (;;) {
        const bool go = valid && lane.cont(s);
        const int nact = __popcll(__ballot(go));
        if (nact == 0 || nact < min_active) break;
        if (go) lane.iterate(s);
    }
    const bool go = valid && lane.cont(s);
    if (valid) {
        if (go) lane.save_state(s);
        else if (home.REDO && (s.diag & MPC_DIAG_ITREF_SKIPPED)) {
            // HPIPM would have refined a direction of this QP and the sweeps of this model do not
            // (rti_core.hpp iterate): nothing is written (the rows of the iterate keep the previous
            // one) and the instance goes to the sequential list kernel, like a declined scan step
            const int h = (int)s.home;
            home.REDO[MPC_REDO_LIST + atomicAdd(&home.REDO[par], 1)] = h;
            home.STATUS[h] = MPC_STATUS_PENDING;
        } else lane.finish(s, home);
    }
    if (flags && inrange) flags[wbase + l] = go ? 1 : 0;
}

// The whole SQP-RTI step with one OCP per GW-lane group (GW = 64: one per wavefront), lane =
// stage, workspace in registers (rti_wave.hpp).  Reads the iterate and the references from the
// instance's workspace column and writes the new iterate, u0, status back to it; nothing else
// touches memory.  Every lane of the wave runs the whole kernel (DPP shifts and reductions need
// the full wave); a group whose instance has converged is masked by `go`.
// cross-lane policy of this thread (rti_wave.hpp): registers/DPP for one wavefront, LDS for the
// two-wavefront workgroup
template <class X> __device__ __forceinline__ X make_policy(int) { return X(); }

// Batch I/O of k_wave in the C-ABI's AoS layouts (device pointers; x0, yref, yref_e and u0 are
// required, status / qp_iter may be null): inputs are read straight from the caller's arrays
// (the lane of stage k reads yref[i][k][0..4], coalesced) and u0 / status / qp_iter are written
// straight back — a wave-model solve is the scan launch plus the (normally empty) list launch,
// no pack / unpack, no workspace in HBM.
struct WaveIO {
    const double* x0;      // [B][3]
    const double* yref;    // [B][N][5]
    const double* yref_e;  // [B][3]
    double* u0;            // [B][2]
    int* status;           // [B]
    int* qp_iter;          // [B]
};

// The arguments of k_wave / k_wave2 as they lie in the kernel-argument segment.  The persistent kernel reads
// them from there again in every pass of its loop over the sets of instances (scalar loads through an
// address the optimiser cannot see through), exactly as a freshly launched workgroup does: kept in
// registers across the loop instead, the ~100 scalar registers of constants and pointers are live
// through every solve — spilled at the kernel's entry and fetched back lane by lane at every use — and
// whatever can be computed from them alone is hoisted out of the loop into vector registers on top.
struct Wave2Args {
    Consts c;
    Mem m;
    int B, sqp_max_iter;
    WaveIO io;
};
typedef const __attribute__((address_space(4))) unsigned* KernArgWords;
template <class T>
__device__ __forceinline__ void load_kernarg(T& dst, KernArgWords src, size_t offset) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    unsigned w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = src[offset / 4 + i];
    __builtin_memcpy(&dst, w, sizeof(T));
}

// PERSIST: see k_wave2.
template <class X, bool SCAN, bool SQP, bool PERSIST = false>
__global__ __launch_bounds__(X::THREADS) void k_wave(Consts c_arg, Mem m_arg, int B_arg, int sqp_max_iter, WaveIO io_arg) {
    static_assert(!PERSIST || (SCAN && !SQP && X::THREADS == 64), "persistent wavefronts: the SQP_RTI scan kernel, one wavefront per workgroup");
    constexpr int G = X::PER_BLOCK;
    constexpr int GW = X::LANES;
  for (;;) {   // (one pass unless PERSIST)
    Consts c = c_arg;
    Mem m = m_arg;
    int B = B_arg;
    WaveIO io = io_arg;
    int l = threadIdx.x;
    int blk = (int)blockIdx.x;
    if (PERSIST) {
        KernArgWords ka = (KernArgWords)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        load_kernarg(c, ka, offsetof(Wave2Args, c));
        load_kernarg(m, ka, offsetof(Wave2Args, m));
        load_kernarg(B, ka, offsetof(Wave2Args, B));
        load_kernarg(io, ka, offsetof(Wave2Args, io));
        asm volatile("" : "+v"(l));
        const int n_sets = (B + G - 1) / G;
        int v = 0;
        if (l == 0) v = atomicAdd(&m.REDO[MPC_REDO_WORK], 1);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= n_sets) {
            if (v == n_sets + (int)gridDim.x - 1 && l == 0) m.REDO[MPC_REDO_WORK] = 0;
            return;
        }
        blk = v;
    }
    const int g = l / GW, k = l % GW;
    const size_t i = (size_t)blk * G + g;
    const int N = c.N;
    const bool inst = i < (size_t)B;
    const size_t ic = inst ? i : (size_t)B - 1;
    const int kcl = k <= N ? k : N;
    double* ck = m.cap + ((size_t)ic * (N + 1) + kcl) * MPC_CAP;   // this lane's capsule line
    X xpol = make_policy<X>(l);
    Wave<X, SCAN> w(c, k, xpol);   // (no LDS: the product cache of the scans lives in registers, rti_wave.hpp Mc)
    if (SCAN && blockIdx.x == 0 && l == 0) {   // this launch's list counter (Mem::REDO; the same values in every pass)
        const int par = m.REDO[MPC_REDO_PS];
        m.REDO[par ^ 1] = 0; m.REDO[MPC_REDO_PL] = par;
    }
    const bool on = inst && k <= N;
    double xi[3], ui[2], x0b[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) xi[q] = on ? ck[C_X + q] : 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) ui[q] = (on && k < N) ? ck[C_U + q] : 0.0;
    {
        const double* yr = k < N ? io.yref + ((size_t)ic * N + kcl) * 5 : io.yref_e + (size_t)ic * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) x0b[q] = (on && k == 0) ? io.x0[3 * ic + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) w.yref[q] = (on && (k < N || q < 3)) ? yr[q] : 0.0;
    }
    const double u_prev0 = ui[0], u_prev1 = ui[1];
    typename Wave<X, SCAN>::State s;
    s.kk = 0; s.diag = 0; s.alpha = 1.0; s.mu = 0.0; s.musum = 0.0; s.open = false;
    int status = 0;
    if (!SQP) {
        // ---- SQP_RTI: one linearisation, one QP, full step ----
        w.prepare(xi, ui, x0b);
        w.begin(s);
        for (;;) {
            const bool go = inst && w.cont(s);
            if (!w.x.any(go)) break;
            if (go) w.iterate(s);
        }
        // status (d_ocp_qp_ipm_solve tail + ocp_qp_hpipm + ocp_nlp_sqp_rti mapping)
        int hs;
        if (s.kk == c.iter_max) hs = 1;
        else if (s.alpha <= c.alpha_min) hs = 2;
        else if (s.mu != s.mu) hs = 3;
        else hs = 0;
        const bool ok = (hs == 0 || hs == 1);
        status = ok ? 0 : 4;
        if (ok) {
            ui[0] = w.z[0]; ui[1] = w.z[1];
#pragma unroll
            for (int q = 0; q < 3; ++q) xi[q] = w.z[2 + q];
        }
        if (!on) { if (PERSIST) continue; return; }
        if (SCAN && s.declined) {   // nothing is written: the sequential kernel takes this QP over (k_wave2<.., LIST>)
            int pos = -1;
            if (k == 0) {
                pos = atomicAdd(&m.REDO[m.REDO[MPC_REDO_PS]], 1);
                m.REDO[MPC_REDO_LIST + pos] = (int)i | (pos < m.ipm_slots ? MPC_REDO_SAVED : 0);
                if (io.status) io.status[i] = MPC_STATUS_PENDING;
            }
            pos = __shfl(pos, g * GW);   // the group's first lane (all lanes of a declined group are here)
            if (pos < m.ipm_slots) w.save_ipm(m.IPM + (size_t)pos * ((size_t)(N + 1) * MPC_IPM_STAGE + MPC_IPM_HEAD), s);
            if (PERSIST) continue;
            return;
        }
        if (k == 0) {
            m.QPITER[i] = s.kk; m.DIAG[i] = s.diag | w.wdiag; m.STATUS[i] = status;
            if (io.status) io.status[i] = status;
            if (io.qp_iter) io.qp_iter[i] = s.kk;
            // on a QP failure the previous u_0 stays in force
            io.u0[2 * i] = ok ? ui[0] : u_prev0; io.u0[2 * i + 1] = ok ? ui[1] : u_prev1;
        }
        if (ok) {
#pragma unroll
            for (int q = 0; q < 3; ++q) ck[C_X + q] = xi[q];
            if (k < N) {
                ck[C_U + 0] = ui[0]; ck[C_U + 1] = ui[1];
#pragma unroll
                for (int q = 0; q < 3; ++q) ck[C_PI + q] = w.pi[q];
            }
        }
        if (PERSIST) continue;
        return;
    }
    // ---- full SQP (acados ocp_nlp_sqp.c order): linearise, residuals, exit test, QP, update;
    //      the whole loop runs inside the kernel, one OCP per group, no host round trip.
    //      As in k_wave2 the iterate (x, u, pi, lam) is not carried in registers across the QP: every
    //      SQP iteration reads it from the capsule and an accepted QP writes its solution back; every
    //      lane linearises and cold-starts whether its OCP still runs or not (a finished group works
    //      on its final iterate and writes nothing), so the solver state is dead across the outer
    //      loop.  A failed or declined QP writes nothing: the capsule keeps the last accepted iterate
    //      (which is where the sequential list kernel carries on after a decline). ----
    double* lk = m.lam + ((size_t)ic * (N + 1) + kcl) * 8;   // this lane's bound multipliers
    double res[4] = {0.0, 0.0, 0.0, 0.0};
    const bool midl = on && k < N;
    bool running = inst;
    bool redo = false;
    int sqp_it = 0, qp_it = 0, diag = 0;
    status = 2;   // MAXITER unless a residual check passes
    for (;;) {
        if (running && sqp_it >= sqp_max_iter) running = false;   // MAXITER: no check after the last QP
        if (!w.x.any(running)) break;
        {
            double pin[3], lamn[8], xn[3], rr[4];
#pragma unroll
            for (int q = 0; q < 3; ++q) xi[q] = on ? ck[C_X + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) ui[q] = midl ? ck[C_U + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) pin[q] = midl ? ck[C_PI + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) lamn[q] = midl ? lk[q] : 0.0;
            w.linearize(xi, ui, xn);
            w.nlp_residuals(xi, ui, x0b, xn, pin, lamn, rr);
            if (running) {
                res[0] = rr[0]; res[1] = rr[1]; res[2] = rr[2]; res[3] = rr[3];
                if (rr[0] < c.nlp_tol && rr[1] < c.nlp_tol && rr[2] < c.nlp_tol && rr[3] < c.nlp_tol) {
                    status = 0;
                    running = false;
                }
            }
        }
        if (!w.x.any(running)) break;
        w.cold_start(xi, ui, x0b);
        w.begin(s);
        for (;;) {
            const bool go = running && w.cont(s);
            if (!w.x.any(go)) break;
            if (go) w.iterate(s);
        }
        if (SCAN && running && s.declined) {
            redo = true;
            running = false;
        }
        if (running) {
            int hs;
            if (s.kk == c.iter_max) hs = 1;
            else if (s.alpha <= c.alpha_min) hs = 2;
            else if (s.mu != s.mu) hs = 3;
            else hs = 0;
            qp_it = s.kk;
            diag |= s.diag;
            ++sqp_it;
            if (hs == 0 || hs == 1) {
                if (on) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) ck[C_X + q] = w.z[2 + q];
                    if (k < N) {
                        ck[C_U + 0] = w.z[0]; ck[C_U + 1] = w.z[1];
#pragma unroll
                        for (int q = 0; q < 3; ++q) ck[C_PI + q] = w.pi[q];
#pragma unroll
                        for (int q = 0; q < 8; ++q) lk[q] = w.has_bound(q & 3) ? w.bn.lam[q] : 0.0;
                    }
                }
            } else {
                status = 4;
                running = false;
            }
        }
    }
    if (!on || k != 0) return;
    if (redo) {
        m.SQPITER[i] = sqp_it;
        m.REDO[MPC_REDO_LIST + atomicAdd(&m.REDO[m.REDO[MPC_REDO_PS]], 1)] = (int)i;
        if (io.status) io.status[i] = MPC_STATUS_PENDING;
        return;
    }
    m.QPITER[i] = qp_it; m.DIAG[i] = diag | w.wdiag; m.STATUS[i] = status; m.SQPITER[i] = sqp_it; m.ACTIVE[i] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) m.NLPRES[4 * i + q] = res[q];
    if (io.status) io.status[i] = status;
    if (io.qp_iter) io.qp_iter[i] = qp_it;
    io.u0[2 * i] = ck[C_U + 0]; io.u0[2 * i + 1] = ck[C_U + 1];   // this lane's own line of the capsule
    return;
  }
}

// The same step with TWO stages per lane (rti_wave2.hpp): lane k of a GW-lane group owns stages 2k
// and 2k+1, 64 / GW OCPs per wavefront (N = 50: two).  Same I/O as k_wave: the lane reads and
// writes two consecutive capsule lines (128 B) and two consecutive yref rows (80 B).
// SCAN: the hot kernel, parallel scans only; an instance whose scan step is declined (rti_wave2.hpp
// iterate) writes nothing and is appended to m.REDO.  LIST (with SCAN = false): the sequential kernel
// over the instances listed in m.REDO — sequential recursions plus HPIPM's iterative refinement —
// launched right behind the hot kernel with the same grid (blocks beyond the list return at once).
// PERSIST (scan kernels of SQP_RTI batches that need more wavefronts than the chip runs at once): the
// grid is one wavefront per SIMD slot and every wavefront draws sets of 64 / GW instances from a
// counter in device memory until none is left, instead of one short-lived workgroup per set — a
// wavefront slot then never waits for a workgroup to be retired and the next one to be dispatched
// (measured on 65536 x N = 50: 32 workgroups per slot, the slots idle 7 % of the kernel's span between
// them).  The counter cleans up after itself: every wavefront makes exactly one fetch that fails, so
// the fetch that returns n_sets + n_wavefronts - 1 is the last of the launch and puts it back to zero
// (no memset between solves; a solve recorded in a HIP graph replays).
template <int GW, bool SCAN, bool SQP, bool LIST, bool PERSIST = false>
__global__ __launch_bounds__(64) void k_wave2(Consts c_arg, Mem m_arg, int B_arg, int sqp_max_iter, WaveIO io_arg) {
    typedef Wave2<GW, SCAN> W;
    static_assert(!(SCAN && LIST), "the list kernel is the sequential one");
    static_assert(!PERSIST || (SCAN && !SQP && !LIST), "persistent wavefronts: the SQP_RTI scan kernel only");
    constexpr int G = W::PER_BLOCK;
    __shared__ mc_d2 mc_lds[W::LDS_SLOTS * 64];   // product cache of the scans + parked per-stage state
  for (;;) {   // (one pass unless PERSIST)
    Consts c = c_arg;
    Mem m = m_arg;
    int B = B_arg;
    WaveIO io = io_arg;
    if (PERSIST) {
        KernArgWords ka = (KernArgWords)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        load_kernarg(c, ka, offsetof(Wave2Args, c));
        load_kernarg(m, ka, offsetof(Wave2Args, m));
        load_kernarg(B, ka, offsetof(Wave2Args, B));
        load_kernarg(io, ka, offsetof(Wave2Args, io));
    }
    const int N = c.N;
    int l = threadIdx.x;
    // (PERSIST: the lane index is made opaque in every pass, or everything derived from it — stage
    //  flags, masks, addresses — would be computed once outside the loop and stay live through every solve)
    if (PERSIST) asm volatile("" : "+v"(l));
    const int g = l / GW, k = l % GW;
    const bool grouped = g < G;   // (GW = 21: lane 63 belongs to no group and carries no instance)
    int blk = (int)blockIdx.x;
    if (PERSIST) {
        const int n_sets = (B + G - 1) / G;
        int v = 0;
        if (l == 0) v = atomicAdd(&m.REDO[MPC_REDO_WORK], 1);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= n_sets) {
            if (v == n_sets + (int)gridDim.x - 1 && l == 0) m.REDO[MPC_REDO_WORK] = 0;
            return;
        }
        blk = v;
    }
#ifdef MPC_PHASE_CLOCK
    const long long tk_entry = clock64(), tw_entry = wall_clock64();
#endif
    size_t i, ic;
    bool inst;
    int list_pos = 0;
    bool resumed = false;
    if (LIST) {
        const int par = m.REDO[MPC_REDO_PL];   // this solve's list counter (Mem::REDO)
        const int cnt = m.REDO[par];
        if (blockIdx.x == 0 && l == 0) m.REDO[MPC_REDO_PS] = par ^ 1;   // the next scan launch counts on the other side
        const int slot0 = (int)blockIdx.x * G;
        if (slot0 >= cnt) return;
        inst = grouped && slot0 + g < cnt;
        list_pos = inst ? slot0 + g : slot0;
        const int entry = m.REDO[MPC_REDO_LIST + list_pos];
        resumed = !SQP && inst && (entry & MPC_REDO_SAVED) != 0;   // its interior-point iterate waits in Mem::IPM
        i = ic = (size_t)(entry & ~MPC_REDO_SAVED);
    } else {
        i = (size_t)blk * G + g;
        inst = grouped && i < (size_t)B;
        ic = inst ? i : (size_t)B - 1;
        if (SCAN && blockIdx.x == 0 && l == 0) {
            const int par = m.REDO[MPC_REDO_PS];
            m.REDO[par ^ 1] = 0; m.REDO[MPC_REDO_PL] = par;
        }
    }
    W w(c, k, mc_lds + l);
    double* ck[2];
    double* lk[2];
    bool on[2];
    double xi[2][3], ui[2][2], x0b[3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = 2 * k + e;
        const int jc = j <= N ? j : N;
        on[e] = inst && j <= N;
        ck[e] = m.cap + ((size_t)ic * (N + 1) + jc) * MPC_CAP;
        lk[e] = SQP ? m.lam + ((size_t)ic * (N + 1) + jc) * 8 : nullptr;
#pragma unroll
        for (int q = 0; q < 3; ++q) xi[e][q] = on[e] ? ck[e][C_X + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) ui[e][q] = (on[e] && j < N) ? ck[e][C_U + q] : 0.0;
        const double* yr = j < N ? io.yref + ((size_t)ic * N + jc) * 5 : io.yref_e + (size_t)ic * 3;
        double y[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) y[q] = (on[e] && (j < N || q < 3)) ? yr[q] : 0.0;
        w.set_yref(e, y);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) x0b[q] = (inst && k == 0) ? io.x0[3 * ic + q] : 0.0;
    const double u_prev0 = ui[0][0], u_prev1 = ui[0][1];
    typename W::State s;
    s.kk = 0; s.diag = 0; s.alpha = 1.0; s.mu = 0.0; s.musum = 0.0; s.open = false;
    int status = 0;
    if (!SQP) {
        // ---- SQP_RTI: one linearisation, one QP, full step ----
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double xn[3];
            w.linearize(e, xi[e], ui[e], xn);
            w.cold_start(e, xi[e], ui[e], x0b);
        }
#ifdef MPC_PHASE_CLOCK
        const long long tk0 = clock64();
#endif
        w.begin(s);
        const size_t ipm_stride = (size_t)(N + 1) * MPC_IPM_STAGE + MPC_IPM_HEAD;
        if (LIST && resumed) w.load_ipm(m.IPM + (size_t)list_pos * ipm_stride, s);   // carry on from the iteration the scan kernel declined
        for (;;) {
            const bool go = inst && w.cont(s);
            if (!w.x.any(go)) break;
            if (go) w.iterate(s);
        }
#ifdef MPC_PHASE_CLOCK
        // development build: cycles per phase of the first 4096 instances, written behind the NLP
        // residuals of a handle created with twice the capacity of the batch (tools/phase_clock.py)
        if (SCAN && inst && k == 0 && i < 4096) {
            double* o = m.NLPRES + 4 * (size_t)B + 8 * i;
            for (int q = 0; q < 5; ++q) o[q] = (double)w.ph[q];
            o[5] = (double)(clock64() - tk0); o[6] = (double)s.kk; o[7] = (double)(tk0 - tk_entry);   // o[7]: load + linearise + cold start
        }
        // every set of instances: when its wavefront took it up and when the interior-point loop ended (100 MHz wall clock)
        if (SCAN && l == 0) {
            double* o = m.NLPRES + 4 * (size_t)B + 8 * 4096 + 2 * (size_t)blk;
            o[0] = (double)tw_entry; o[1] = (double)wall_clock64();
        }
#endif
        int hs;
        if (s.kk == c.iter_max) hs = 1;
        else if (s.alpha <= c.alpha_min) hs = 2;
        else if (s.mu != s.mu) hs = 3;
        else hs = 0;
        const bool ok = (hs == 0 || hs == 1);
        status = ok ? 0 : 4;
        const bool redo = SCAN && s.declined;   // nothing is written; the list kernel takes this QP over
        if (SCAN && w.x.any(redo && inst)) {
            int pos = -1;
            if (redo && inst && k == 0) {
                pos = atomicAdd(&m.REDO[m.REDO[MPC_REDO_PS]], 1);
                m.REDO[MPC_REDO_LIST + pos] = (int)i | (pos < m.ipm_slots ? MPC_REDO_SAVED : 0);
                if (io.status) io.status[i] = MPC_STATUS_PENDING;
            }
            pos = __shfl(pos, g * GW);   // the group's first lane
            if (redo && inst && pos < m.ipm_slots) w.save_ipm(m.IPM + (size_t)pos * ipm_stride, s);
        }
        if (inst && k == 0 && !redo) {
            m.QPITER[i] = s.kk; m.DIAG[i] = s.diag | w.wdiag | (LIST ? MPC_DIAG_SCAN_FALLBACK : 0) | (resumed ? MPC_DIAG_RESUMED : 0); m.STATUS[i] = status;
            if (io.status) io.status[i] = status;
            if (io.qp_iter) io.qp_iter[i] = s.kk;
            // on a QP failure the previous u_0 stays in force
            io.u0[2 * i] = ok ? w.s[0].z[0] : u_prev0; io.u0[2 * i + 1] = ok ? w.s[0].z[1] : u_prev1;
        }
        if (ok && !redo) {
            double pio[2][3];
            w.get_pi(pio);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!on[e]) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) ck[e][C_X + q] = w.s[e].z[2 + q];
                if (2 * k + e < N) {
                    ck[e][C_U + 0] = w.s[e].z[0]; ck[e][C_U + 1] = w.s[e].z[1];
#pragma unroll
                    for (int q = 0; q < 3; ++q) ck[e][C_PI + q] = pio[e][q];
                }
            }
        }
        if (PERSIST) continue;   // the next set of instances
        return;
    }
    // ---- full SQP (acados ocp_nlp_sqp.c order): linearise, residuals, exit test, QP, update.
    //      The iterate (x, u, pi, lam) is NOT carried in registers across the QP: every SQP iteration
    //      reads it from the capsule and an accepted QP writes its solution back (32 doubles per
    //      lane that would otherwise be live through the whole interior-point loop).  A failed QP
    //      writes nothing, so the capsule keeps the last accepted iterate. ----
    //      A declined scan step (SCAN) ends the instance's loop with the capsule at its last accepted
    //      iterate and the count of finished SQP iterations in SQPITER: the list kernel carries on
    //      from exactly there.
    double res[4] = {0.0, 0.0, 0.0, 0.0};
    bool running = inst;
    bool redo = false;
    int sqp_it = (LIST && inst) ? m.SQPITER[i] : 0, qp_it = 0, diag = LIST ? MPC_DIAG_SCAN_FALLBACK : 0;
    status = 2;   // MAXITER unless a residual check passes
    for (;;) {
        if (running && sqp_it >= sqp_max_iter) running = false;   // MAXITER: no check after the last QP
        if (!w.x.any(running)) break;
        // Every lane linearises and cold-starts, whether its OCP still runs or not (a finished
        // group works on its final iterate and writes nothing): the solver state is then
        // overwritten on every path around this loop, i.e. dead during the linearisation, instead
        // of being carried through it in registers on behalf of lanes that will never use it.
        {
            double pin[2][3], lamn[2][8], xn[2][3], rr[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool midl = on[e] && 2 * k + e < N;
#pragma unroll
                for (int q = 0; q < 3; ++q) xi[e][q] = on[e] ? ck[e][C_X + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 2; ++q) ui[e][q] = midl ? ck[e][C_U + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) pin[e][q] = midl ? ck[e][C_PI + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) lamn[e][q] = midl ? lk[e][q] : 0.0;
                w.linearize(e, xi[e], ui[e], xn[e]);
            }
            w.nlp_residuals(xi, ui, x0b, xn, pin, lamn, rr);
            if (running) {
                res[0] = rr[0]; res[1] = rr[1]; res[2] = rr[2]; res[3] = rr[3];
                if (rr[0] < c.nlp_tol && rr[1] < c.nlp_tol && rr[2] < c.nlp_tol && rr[3] < c.nlp_tol) {
                    status = 0;
                    running = false;
                }
            }
        }
        if (!w.x.any(running)) break;
#pragma unroll
        for (int e = 0; e < 2; ++e) w.cold_start(e, xi[e], ui[e], x0b);
        w.begin(s);
        for (;;) {
            const bool go = running && w.cont(s);
            if (!w.x.any(go)) break;
            if (go) w.iterate(s);
        }
        if (SCAN && running && s.declined) {
            redo = true;
            running = false;
        }
        if (running) {
            int hs;
            if (s.kk == c.iter_max) hs = 1;
            else if (s.alpha <= c.alpha_min) hs = 2;
            else if (s.mu != s.mu) hs = 3;
            else hs = 0;
            qp_it = s.kk;
            diag |= s.diag;
            ++sqp_it;
            if (hs == 0 || hs == 1) {
                double pio[2][3];
                w.get_pi(pio);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (!on[e]) continue;
#pragma unroll
                    for (int q = 0; q < 3; ++q) ck[e][C_X + q] = w.s[e].z[2 + q];
                    if (2 * k + e < N) {
                        ck[e][C_U + 0] = w.s[e].z[0]; ck[e][C_U + 1] = w.s[e].z[1];
#pragma unroll
                        for (int q = 0; q < 3; ++q) ck[e][C_PI + q] = pio[e][q];
#pragma unroll
                        for (int q = 0; q < 8; ++q) lk[e][q] = W::has_bound(w.s[e], q & 3) ? w.s[e].bn.lam[q] : 0.0;
                    }
                }
            } else {
                status = 4;
                running = false;
            }
        }
    }
    if (redo && inst && k == 0) {
        m.SQPITER[i] = sqp_it;
        m.REDO[MPC_REDO_LIST + atomicAdd(&m.REDO[m.REDO[MPC_REDO_PS]], 1)] = (int)i;
        if (io.status) io.status[i] = MPC_STATUS_PENDING;
    }
    if (inst && k == 0 && !redo) {
        m.QPITER[i] = qp_it; m.DIAG[i] = diag | w.wdiag; m.STATUS[i] = status; m.SQPITER[i] = sqp_it; m.ACTIVE[i] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) m.NLPRES[4 * i + q] = res[q];
        if (io.status) io.status[i] = status;
        if (io.qp_iter) io.qp_iter[i] = qp_it;
        io.u0[2 * i] = ck[0][C_U + 0]; io.u0[2 * i + 1] = ck[0][C_U + 1];
    }
    return;
  }
}

// Full SQP only (spec.sqp_max_iter > 1; BASELINE config 5): the head of one SQP iteration.
// Linearises every stage of the still-active instances, evaluates the NLP residuals with the
// multipliers of the previous QP and retires the instances that meet nlp_tol (status 0) —
// acados ocp_nlp_sqp.c order: linearise, residuals, exit test, QP, update.  `n_active` counts
// the instances that go on to the QP (one atomic per wave).
__global__ __launch_bounds__(MPC_WAVE) void k_nlp_res(Consts c, Mem m, int B, int sqp_it, int* n_active) {
    const size_t wbase = (size_t)blockIdx.x * MPC_WAVE;
    const unsigned l = threadIdx.x;
    const size_t i = wbase + l;
    const bool inrange = i < (size_t)B;
    if (inrange && sqp_it == 0) { m.ACTIVE[i] = 1; m.SQPITER[i] = 0; m.QPITER[i] = 0; m.DIAG[i] = 0; m.STATUS[i] = 2; }
    const bool valid = inrange && (sqp_it == 0 || m.ACTIVE[i] != 0);
    if (__ballot(valid) == 0ull) return;
    bool go = false;
    if (valid) {
        Lane lane(c, m, wbase, l);
        double r[4];
        lane.nlp_res_and_linearize(r);
        for (int q = 0; q < 4; ++q) m.NLPRES[4 * i + q] = r[q];
        go = !(r[0] < c.nlp_tol && r[1] < c.nlp_tol && r[2] < c.nlp_tol && r[3] < c.nlp_tol);
        if (!go) {
            m.ACTIVE[i] = 0;
            m.STATUS[i] = 0;
            const int N = c.N;
            lane.st(R_U0, N, 0, lane.ld(R_U, 0, 0));
            lane.st(R_U0, N, 1, lane.ld(R_U, 0, 1));
        }
    }
    const int n = __popcll(__ballot(go));
    if (n > 0 && l == 0) atomicAdd(n_active, n);
}

// Packs the instances that are still iterating (flags) densely into the other workspace.
// Placement is by atomic ticket, so the order is not deterministic — results do not depend
// on it (instances are independent and carry their home index).
__global__ __launch_bounds__(MPC_WAVE) void k_compact(Mem src, Mem dst, const int* count_in, int B, int first,
                                                      const int* flags, int* count_out, int N) {
    const int cnt = first ? B : *count_in;
    const size_t wbase = (size_t)blockIdx.x * MPC_WAVE;
    if (wbase >= (size_t)cnt) return;
    const unsigned l = threadIdx.x;
    const bool a = (wbase + l < (size_t)cnt) && flags[wbase + l] != 0;
    const unsigned long long mask = __ballot(a);
    const int n = __popcll(mask);
    if (n == 0) return;
    int base = 0;
    if (l == (unsigned)(__ffsll((long long)mask) - 1)) base = atomicAdd(count_out, n);
    base = __shfl(base, __ffsll((long long)mask) - 1);
    if (!a) return;
    const size_t d = (size_t)base + __popcll(mask & ((1ull << l) - 1ull));
    const double* sc = reinterpret_cast<const double*>(src.slab + (wbase >> 6) * src.wave_bytes) + l;
    double* dc = reinterpret_cast<double*>(dst.slab + (d >> 6) * dst.wave_bytes) + (d & 63);
    for (int k = 0; k <= N; ++k) {
        const size_t o = (size_t)(k * MPC_RPS) << 6;
#pragma unroll 8
        for (int r = MPC_CARRY_A0; r < MPC_CARRY_A1; ++r) dc[o + ((size_t)r << 6)] = sc[o + ((size_t)r << 6)];
#pragma unroll 8
        for (int r = MPC_CARRY_B0; r < MPC_CARRY_B1; ++r) dc[o + ((size_t)r << 6)] = sc[o + ((size_t)r << 6)];
    }
}

// Batch I/O.  The C-ABI hands over the reference's AoS layouts; the workspace is
// [wave][stage][row][lane].  AoS src[B][nst*ncomp] -> rows (stage kfirst+s, row roff+q), through
// a padded LDS tile so that both the global reads (along the AoS row) and the global writes
// (64 lanes of one workspace row) are coalesced.  grid = (ceil(R/32), n_waves), block = (32,8).
__global__ __launch_bounds__(256) void k_pack(const double* __restrict__ src, int nst, int ncomp, int kfirst,
                                              int roff, Mem m, int B) {
    __shared__ double tile[64][33];
    const int R = nst * ncomp;
    const int r0 = blockIdx.x * 32;
    const size_t w = blockIdx.y;
    for (int ii = threadIdx.y; ii < 64; ii += 8) {
        const size_t i = w * 64 + ii;
        const int r = r0 + threadIdx.x;
        if (i < (size_t)B && r < R) tile[ii][threadIdx.x] = src[i * R + r];
    }
    __syncthreads();
    double* blk = reinterpret_cast<double*>(m.slab + w * m.wave_bytes);
    for (int rr = threadIdx.y; rr < 32; rr += 8) {
        const int r = r0 + rr;
        if (r >= R) continue;
        const int st = r / ncomp, q = r - st * ncomp;
        double* row = blk + ((size_t)((kfirst + st) * MPC_RPS + roff + q) << 6);
        for (int h = 0; h < 2; ++h) {
            const int ln = threadIdx.x + 32 * h;
            if (w * 64 + ln < (size_t)B) row[ln] = tile[ln][rr];
        }
    }
}

// The reverse: workspace rows -> AoS dst[B][nst*ncomp].
__global__ __launch_bounds__(256) void k_unpack(double* __restrict__ dst, int nst, int ncomp, int kfirst, int roff,
                                                Mem m, int B) {
    __shared__ double tile[64][33];
    const int R = nst * ncomp;
    const int r0 = blockIdx.x * 32;
    const size_t w = blockIdx.y;
    const double* blk = reinterpret_cast<const double*>(m.slab + w * m.wave_bytes);
    for (int rr = threadIdx.y; rr < 32; rr += 8) {
        const int r = r0 + rr;
        if (r >= R) continue;
        const int st = r / ncomp, q = r - st * ncomp;
        const double* row = blk + ((size_t)((kfirst + st) * MPC_RPS + roff + q) << 6);
        for (int h = 0; h < 2; ++h) {
            const int ln = threadIdx.x + 32 * h;
            if (w * 64 + ln < (size_t)B) tile[ln][rr] = row[ln];
        }
    }
    __syncthreads();
    for (int ii = threadIdx.y; ii < 64; ii += 8) {
        const size_t i = w * 64 + ii;
        const int r = r0 + threadIdx.x;
        if (i < (size_t)B && r < R) dst[i * R + r] = tile[ii][threadIdx.x];
    }
}

// AoS capsule <-> C-ABI AoS arrays: src/dst [B][nst*ncomp] <-> cap[i][kfirst+s][slot+q].
__global__ void k_cap_pack(const double* __restrict__ src, int nst, int ncomp, int kfirst, int slot, Mem m, int N, int B) {
    const size_t R = (size_t)nst * ncomp;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * R) return;
    const size_t i = idx / R, r = idx - i * R;
    const int st = (int)(r / ncomp), q = (int)(r - (size_t)st * ncomp);
    m.cap[(i * (N + 1) + kfirst + st) * MPC_CAP + slot + q] = src[idx];
}
__global__ void k_cap_unpack(double* __restrict__ dst, int nst, int ncomp, int kfirst, int slot, Mem m, int N, int B) {
    const size_t R = (size_t)nst * ncomp;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * R) return;
    const size_t i = idx / R, r = idx - i * R;
    const int st = (int)(r / ncomp), q = (int)(r - (size_t)st * ncomp);
    dst[idx] = m.cap[(i * (N + 1) + kfirst + st) * MPC_CAP + slot + q];
}
// Capsule (and, on full-SQP handles, the bound multipliers) <-> the workspace rows R_X, R_U, R_PI,
// R_LAM of the lane-per-OCP kernels: only when a handle is driven through MPC_MODEL_LANE.
// One thread per (instance, stage).
__global__ void k_cap_slab(Mem m, int N, int B, int to_slab) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * (N + 1)) return;
    const size_t i = idx / (N + 1);
    const int k = (int)(idx - i * (N + 1));
    double* cl = m.cap + idx * MPC_CAP;
    double* ll = m.lam ? m.lam + idx * 8 : nullptr;
    double* blk = reinterpret_cast<double*>(m.slab + (i >> 6) * m.wave_bytes) + (i & 63) + ((size_t)(k * MPC_RPS) << 6);
    if (to_slab) {
        for (int r = 0; r < 8; ++r) blk[(size_t)r << 6] = cl[r];
        for (int r = 0; r < 8; ++r) blk[(size_t)(R_LAM + r) << 6] = ll ? ll[r] : 0.0;
    } else {
        for (int r = 0; r < 8; ++r) cl[r] = blk[(size_t)r << 6];
        if (ll) for (int r = 0; r < 8; ++r) ll[r] = blk[(size_t)(R_LAM + r) << 6];
    }
}

// H14: zero x, u, pi (and the bound multipliers) of the selected capsules
// (mobile_robot_acados_reset, :830-869).  One thread per (instance, stage).
__global__ void k_reset(Mem m, int N, const unsigned char* mask, int cap) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)cap * (N + 1)) return;
    const size_t i = idx / (N + 1);
    if (mask && !mask[i]) return;
    double* cl = m.cap + idx * MPC_CAP;
#pragma unroll
    for (int r = 0; r < MPC_CAP; ++r) cl[r] = 0.0;
    if (m.lam) {
#pragma unroll
        for (int r = 0; r < 8; ++r) m.lam[idx * 8 + r] = 0.0;
    }
}

// Shifted warm start on the capsule lines cap[instance][stage][MPC_CAP] (and the multiplier
// lines): one workgroup per instance; every thread reads its elements of line k+1 before anyone
// writes line k.  `lines` = cap (width MPC_CAP, x carried into line N-1) or lam (everything of
// line N-1 kept).
#define MPC_SHIFT_THREADS 256
#define MPC_SHIFT_PER_THREAD (((MPC_MAX_N + 1) * 8 + MPC_SHIFT_THREADS - 1) / MPC_SHIFT_THREADS)
static_assert(MPC_CAP == 8, "k_shift assumes 8 doubles per capsule line");
__global__ __launch_bounds__(MPC_SHIFT_THREADS) void k_shift(double* lines, int N, const unsigned char* mask, int n, int x_cols) {
    const size_t i = blockIdx.x;
    if (i >= (size_t)n) return;
    if (mask && !mask[i]) return;   // uniform per workgroup
    double* line = lines + i * (size_t)(N + 1) * 8;
    const int tot = N * 8;    // destination elements: lines 0..N-1
    double v[MPC_SHIFT_PER_THREAD];
#pragma unroll
    for (int q = 0; q < MPC_SHIFT_PER_THREAD; ++q) {
        const int e = q * MPC_SHIFT_THREADS + threadIdx.x;
        // line N only carries x_N: the controls and multipliers of the last interval stay
        const bool take = e < tot && (e < (N - 1) * 8 || (e % 8) < x_cols);
        v[q] = take ? line[e + 8] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MPC_SHIFT_PER_THREAD; ++q) {
        const int e = q * MPC_SHIFT_THREADS + threadIdx.x;
        const bool take = e < tot && (e < (N - 1) * 8 || (e % 8) < x_cols);
        if (take) line[e] = v[q];
    }
}

// H16: plant step (mobile_robot_acados_sim_solve: ERK4, one step of Ts).
__device__ __forceinline__ void plant_step(const Consts& c, double& x0, double& x1, double& psi, double v, double dl) {
    const double h = c.Ts;
    const double om = (v / c.L) * tan(dl);
    const double ah = 0.5 * h, a1 = 1.0 * h;
    const double bw0 = h * (1.0 / 6.0), bw1 = h * (2.0 / 6.0);
    double sA, cA, sB, cB, sD, cD;
    sincos(psi, &sA, &cA);
    sincos(psi + ah * om, &sB, &cB);
    sincos(psi + a1 * om, &sD, &cD);
    const double n0 = (((x0 + bw0 * (v * cA)) + bw1 * (v * cB)) + bw1 * (v * cB)) + bw0 * (v * cD);
    const double n1 = (((x1 + bw0 * (v * sA)) + bw1 * (v * sB)) + bw1 * (v * sB)) + bw0 * (v * sD);
    const double n2 = (((psi + bw0 * om) + bw1 * om) + bw1 * om) + bw0 * om;
    x0 = n0; x1 = n1; psi = n2;
}

// x [B][3] in/out, u [B][2].
__global__ void k_sim_step(Consts c, double* x, const double* u, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double x0 = x[3 * i], x1 = x[3 * i + 1], psi = x[3 * i + 2];
    plant_step(c, x0, x1, psi, u[2 * i], u[2 * i + 1]);
    x[3 * i] = x0; x[3 * i + 1] = x1; x[3 * i + 2] = psi;
}

// Development / test hook (mpc_batch_debug_linearize): what the preparation phase hands to the QP
// at the capsules' current iterate, one thread per (instance, stage), written AoS as
// out[i][k][24] = a, b, B00, B10, B20, B01, B11, B21 (the non-trivial entries of A_k, B_k; H5/H7),
// b_k = phi(x_k, u_k) - x_{k+1} (3), the LINEAR_LS gradient in [v, delta, x, y, psi] order (5; H8),
// the BGH bound residuals lb - z (4) and z - ub (4) on (v, delta, x, y) (H9).  Entries a stage
// does not have (stage N: dynamics, controls, bounds; stage 0: state gradient and state bounds,
// x0 being eliminated) are 0.  Uses the same Algebra::linearize_math as the solve kernels.
__global__ void k_debug_lin(Consts c, Mem m, const double* __restrict__ yref, const double* __restrict__ yref_e, int B,
                            double* __restrict__ out) {
    const int N = c.N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * (N + 1)) return;
    const size_t i = idx / (N + 1);
    const int k = (int)(idx - i * (N + 1));
    const double* cl = m.cap + idx * MPC_CAP;
    double* o = out + idx * 24;
    for (int q = 0; q < 24; ++q) o[q] = 0.0;
    const double x0 = cl[C_X], x1 = cl[C_X + 1], psi = cl[C_X + 2];
    if (k < N) {
        const double v = cl[C_U], dl = cl[C_U + 1];
        Algebra alg(c);
        Algebra::AB8 ab;
        double cv[3], xn[3];
        alg.linearize_math(x0, x1, psi, v, dl, ab, cv, xn);
        o[0] = ab.a; o[1] = ab.b; o[2] = ab.B00; o[3] = ab.B10; o[4] = ab.B20; o[5] = ab.B01; o[6] = ab.B11; o[7] = ab.B21;
        const double* cn = cl + MPC_CAP;
        for (int q = 0; q < 3; ++q) o[8 + q] = xn[q] - cn[C_X + q];
        const double* yr = yref + (i * N + k) * 5;
        const double z[5] = {v, dl, x0, x1, psi};
        o[11] = c.wg[0] * (z[0] - yr[3]);
        o[12] = c.wg[1] * (z[1] - yr[4]);
        for (int q = 0; q < 3; ++q) o[13 + q] = k > 0 ? c.wg[2 + q] * (z[2 + q] - yr[q]) : 0.0;
        const int nb = k == 0 ? 2 : 4;
        for (int j = 0; j < nb; ++j) { o[16 + j] = c.lb[j] - z[j]; o[20 + j] = z[j] - c.ub[j]; }
    } else {
        const double z[3] = {x0, x1, psi};
        for (int q = 0; q < 3; ++q) o[13 + q] = c.we[q] * (z[q] - yref_e[3 * i + q]);
    }
}

// ---- closed loop on the device: H2 + the packing half of H1 + H16 --------
// Per-vehicle tracker state; one shared path table (state_refs [M][3], input_refs [Mi][2]).
struct Track {
    const double* sref;   // [M][3]
    const double* iref;   // [Mi][2]
    int M, Mi;
    double* x;            // [S][3] current state (Optimizer::x_current) = the x0 of the next solve
    double* u;            // [S][2] command in force (Optimizer::u_current)
    int* idx;             // [S][4] target_waypoint_index, last_waypoint_index, closest_waypoint_index, count
    double* yref;         // [S][N][5] references of the tick, the C-ABI's AoS layout
    double* yref_e;       // [S][3]
    double density;       // 1 / T / v_ref                     (src/optimizer.cpp:128)
    int limit;            // floor(3.95 / (v_ref * T))         (src/optimizer.cpp:387-391)
    int lookahead;        // 1
};

// Optimizer::find_closest_waypoint (src/optimizer.cpp:343-382): downward scan, strict '<'.
__device__ __forceinline__ int closest_waypoint(const Track& tr, double cx, double cy, int min_index, int max_index,
                                                int last, int target) {
    if (min_index < 0) min_index = last < tr.M - 1 ? last : tr.M - 1;
    if (max_index < 0) max_index = target + tr.limit < tr.M - 1 ? target + tr.limit : tr.M - 1;
    const double cn = cx * cx + cy * cy;
    double best = 1.7976931348623157e308;
    int closest = -1;
    for (int i = max_index; i >= min_index; --i) {
        const double px = tr.sref[3 * i], py = tr.sref[3 * i + 1];
        const double d = (px * px + py * py) - 2 * (px * cx + py * cy) + cn;
        if (d < best) { best = d; closest = i; }
    }
    return closest;
}

// One thread per vehicle: Optimizer::find_next_waypoint (src/optimizer.cpp:384-431).
__global__ __launch_bounds__(MPC_WAVE) void k_track_select(Track tr, int B) {
    const size_t i = (size_t)blockIdx.x * MPC_WAVE + threadIdx.x;
    if (i >= (size_t)B) return;
    const double cx = tr.x[3 * i], cy = tr.x[3 * i + 1];
    int target = tr.idx[4 * i], last = tr.idx[4 * i + 1], count = tr.idx[4 * i + 3];
    int closest = closest_waypoint(tr, cx, cy, -1, -1, last, target);
    if (closest < 0) closest = last < tr.M - 1 ? last : tr.M - 1;   // empty window (the reference would index -1)
    const double dx = tr.sref[3 * closest] - cx, dy = tr.sref[3 * closest + 1] - cy;
    const double dist = sqrt(dx * dx + dy * dy);
    if (dist > 1.2) {
        const double lo = (double)closest - dist * tr.density * 1.2;
        const int min_index = (int)(lo > 0.0 ? lo : 0.0);
        const int c2 = closest_waypoint(tr, cx, cy, min_index, -1, last, target);
        if (c2 >= 0) closest = c2;
    }
    if (count >= 8) { target = closest + tr.lookahead; count = 0; }
    else { target = target + 1; count = count + 1; }
    if (target > tr.M - 1) target = tr.M - 1;
    tr.idx[4 * i] = target; tr.idx[4 * i + 1] = target; tr.idx[4 * i + 2] = closest; tr.idx[4 * i + 3] = count;
}
// The reference packing of Optimizer::update_and_solve (:240-281), one thread per (vehicle,
// stage), written in the C-ABI's AoS layout that the solve reads (coalesced across the stages of
// a vehicle); stage N writes the terminal reference.
__global__ void k_track_pack(Track tr, int N, int B) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * (N + 1)) return;
    const size_t i = idx / (N + 1);
    const int j = (int)(idx - i * (N + 1));
    const int target = tr.idx[4 * i];
    if (j < N) {
        const bool past = target + j >= tr.M;
        const int r = past ? tr.M - 1 : target + j;
        double* y = tr.yref + (i * N + j) * 5;
        y[0] = tr.sref[3 * r]; y[1] = tr.sref[3 * r + 1]; y[2] = tr.sref[3 * r + 2];
        y[3] = (past || r >= tr.Mi) ? 0.0 : tr.iref[2 * r];
        y[4] = 0.0;   // "set steer rate to 0" (:271)
    } else {
        double* y = tr.yref_e + 3 * i;   // terminal reference = the FIRST waypoint of the window (:244-251)
        y[0] = tr.sref[3 * target]; y[1] = tr.sref[3 * target + 1]; y[2] = tr.sref[3 * target + 2];
    }
}

// After the solve: adopt u0 unless the solver failed (src/optimizer.cpp:290-296), then
// Optimizer::integrate_next_states (:321-341).
__global__ __launch_bounds__(MPC_WAVE) void k_track_advance(Consts c, Mem m, Track tr, const double* u0, int B) {
    const size_t i = (size_t)blockIdx.x * MPC_WAVE + threadIdx.x;
    if (i >= (size_t)B) return;
    double v = tr.u[2 * i], dl = tr.u[2 * i + 1];
    if (m.STATUS[i] == 0) {
        v = u0[2 * i]; dl = u0[2 * i + 1];
        tr.u[2 * i] = v; tr.u[2 * i + 1] = dl;
    }
    double x0 = tr.x[3 * i], x1 = tr.x[3 * i + 1], psi = tr.x[3 * i + 2];
    plant_step(c, x0, x1, psi, v, dl);
    tr.x[3 * i] = x0; tr.x[3 * i + 1] = x1; tr.x[3 * i + 2] = psi;
}

// ------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------
#define MPC_MAX_CHUNKS 3
#define MPC_CHUNK_MIN 16384   // instances: below two chunks of this size a solve is not split
struct Shard {
    int device = -1;
    int first = 0;     // first global instance of this shard
    int cap = 0;       // instances this shard can hold
    size_t S = 0;      // padded stride
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the device work of the last solve ...
    bool outer_events = false;                 // ... if that solve recorded them (the batch <= 64 path does not)
    hipEvent_t ev_ph[2] = {};   // around the dominant solve kernel of the last solve
    // host-pointer solves of large batches are pipelined in up to three chunks (65536 x N = 50 from
    // pinned arrays: 1 chunk 8.9 ms, 2: 8.0, 3: 7.9, 4: 8.1, 8: 9.1 — every chunk waits for its slowest wavefront): the inputs of chunk c + 1 travel
    // on copy_stream while the kernels of chunk c run on `stream` (created on first use)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_chunk[MPC_MAX_CHUNKS] = {};
    int n_ph = 0;
    int sqp_rounds = 0;  // QP rounds of the last full-SQP solve (lane model)
    // allocated by mpc_batch_create: the capsules, the small per-instance arrays, u0
    double* capbuf = nullptr;  // Mem::cap
    double* ipmbuf = nullptr;  // Mem::IPM
    double* lambuf = nullptr;  // Mem::lam (full-SQP handles only)
    int* islab = nullptr;
    double* d_u0 = nullptr;    // [S][2] u_0 of the last solve when the caller passes none
    // lane model only, allocated on its first use (ensure_lane)
    char* slab = nullptr;    // workspace: inputs, iterate rows, IPM workspace
    char* slab2 = nullptr;   // secondary workspace for compacted stragglers (first large lane solve)
    int* phase_ints = nullptr;  // [MPC_MAX_PHASES] live counts, then [S] flags
    Mem mem, mem2;
    // AoS staging for the host-pointer entry points, allocated on first use (ensure_staging):
    // device buffers + pinned host mirrors (so that the copies of several shards overlap)
    double *s_x0 = nullptr, *s_yref = nullptr, *s_yrefe = nullptr, *s_u0 = nullptr;
    double *s_xt = nullptr, *s_ut = nullptr, *s_pi = nullptr;
    unsigned char* s_mask = nullptr;
    bool staged = false, staged_traj = false;
    // pinned, device-visible host staging for small batches (zero-copy latency path)
    double* z_in = nullptr;   // x0 [Z][3] | yref [Z][N][5] | yref_e [Z][3]
    double* z_out = nullptr;  // u0 [Z][2] | status [Z] (int) | qp_iter [Z] (int)
    // closed-loop tracker (mpc_batch_track_*): path table and per-vehicle state
    double *t_sref = nullptr, *t_iref = nullptr, *t_x = nullptr, *t_u = nullptr, *t_yref = nullptr, *t_yrefe = nullptr;
    int* t_idx = nullptr;
    int t_M = 0, t_Mi = 0;
    int t_n = 0;               // vehicles initialised by the last mpc_batch_track_init
    size_t dev_bytes = 0;      // device memory this shard holds
};

struct mpc_batch {
    mpc_ocp_spec_t spec;
    Consts consts;
    int capacity = 0;
    std::vector<Shard> shards;
    double last_ms = 0.0;
    int model = MPC_MODEL_AUTO;
    int track_shift = 0;   // mpc_batch_track_set_shift
    double t_density = 0.0;   // tracker parameters (mpc_batch_track_init)
    int t_limit = 0;
    std::string err;
};

// Restores the caller's current device on every exit path of an entry point (HIPCHK returns early).
struct DeviceGuard {
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIPCHK(h, call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

// hipMalloc + accounting of the shard's device footprint
#define DEVALLOC(h, s, ptr, nbytes)                                                      \
    do {                                                                                 \
        HIPCHK(h, hipMalloc((void**)&(ptr), (nbytes)));                                  \
        (s).dev_bytes += (size_t)(nbytes);                                               \
    } while (0)

static void free_shard(Shard& s) {
    if (s.device < 0) return;
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    (void)hipFree(s.slab); (void)hipFree(s.slab2); (void)hipFree(s.islab); (void)hipFree(s.phase_ints);
    (void)hipFree(s.capbuf); (void)hipFree(s.lambuf); (void)hipFree(s.d_u0); (void)hipFree(s.ipmbuf);
    (void)hipFree(s.s_x0); (void)hipFree(s.s_yref); (void)hipFree(s.s_yrefe); (void)hipFree(s.s_u0);
    (void)hipFree(s.s_xt); (void)hipFree(s.s_ut); (void)hipFree(s.s_pi); (void)hipFree(s.s_mask);
    if (s.z_in) (void)hipHostFree(s.z_in);
    if (s.z_out) (void)hipHostFree(s.z_out);
    (void)hipFree(s.t_sref); (void)hipFree(s.t_iref); (void)hipFree(s.t_x); (void)hipFree(s.t_u); (void)hipFree(s.t_idx);
    (void)hipFree(s.t_yref); (void)hipFree(s.t_yrefe);
    if (s.ev0) (void)hipEventDestroy(s.ev0);
    if (s.ev1) (void)hipEventDestroy(s.ev1);
    for (hipEvent_t e : s.ev_ph) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s.ev_chunk) if (e) (void)hipEventDestroy(e);
    if (s.copy_stream) { (void)hipStreamSynchronize(s.copy_stream); (void)hipStreamDestroy(s.copy_stream); }
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Shard();
}

// The device-visible address of a host array that is pinned (hipHostMalloc / hipHostRegister), or
// null for pageable memory: kernels then read and write the caller's arrays over the host link
// themselves instead of going through staging buffers and copies.
template <class T>
static T* pinned_device_ptr(T* host, size_t count) {
    if (!host || count == 0) return nullptr;
    hipPointerAttribute_t a, b;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) {
        (void)hipGetLastError();   // pageable memory is "invalid value" to the runtime
        return nullptr;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    // the whole span must be pinned, and in one piece: an array of which only a prefix is registered
    // (a view that runs past a hipHostRegister'ed region, or spans two allocations) goes through
    // the staging copies instead of faulting in a kernel
    const char* last = reinterpret_cast<const char*>(host) + count * sizeof(T) - 1;
    if (hipPointerGetAttributes(&b, last) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (b.type != hipMemoryTypeHost || !b.devicePointer) return nullptr;
    if (static_cast<const char*>(b.devicePointer) - static_cast<const char*>(a.devicePointer) != last - reinterpret_cast<const char*>(host))
        return nullptr;
    return static_cast<T*>(a.devicePointer);
}

// Waits for the stream of a latency-path solve (batches of up to 64 instances, ~0.1 ms of device
// work).  hipStreamSynchronize may put the thread to sleep and the wake-up then lands in the tail of
// the tick-time distribution (bench.py: p99 = 2.3 x p50 on the driver's box in round 2); polling
// the stream keeps the thread on its core for the few hundred microseconds a tick can take and falls
// back to the blocking wait after that.  MPC_SYNC=block restores the plain blocking wait.
static hipError_t wait_short(hipStream_t st) {
    static const bool block = [] { const char* e = getenv("MPC_SYNC"); return e && strcmp(e, "block") == 0; }();
    if (!block) {
        for (int spin = 0; spin < 20000; ++spin) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
        }
        (void)hipGetLastError();   // hipErrorNotReady is sticky in the last-error slot
    }
    return hipStreamSynchronize(st);
}

// Every shard, even on an error: the work already enqueued on the other shards is waited for
// before an entry point returns, so the caller's buffers are never written after the return.
static void sync_all(mpc_batch* h) {
    for (Shard& s : h->shards) {
        if (s.device < 0 || !s.stream) continue;
        (void)hipSetDevice(s.device);
        if (s.copy_stream) (void)hipStreamSynchronize(s.copy_stream);
        (void)hipStreamSynchronize(s.stream);
    }
}

// Lane model: its workspace slab is allocated on the first solve that uses it, so a handle that
// only ever runs the wave model holds the capsules and little else.
// (buffer by buffer: a call that fails half-way — out of memory — leaves what it has got to the
//  next call and to free_shard, and nothing is allocated twice)
static int ensure_lane(mpc_batch* h, Shard& s, hipStream_t st) {
    const int N = h->spec.N;
    const size_t capa = (size_t)(s.cap > 0 ? s.cap : 1);
    if (!s.slab) {
        const size_t nbytes = slab_bytes_for(N, capa);
        DEVALLOC(h, s, s.slab, nbytes);
        HIPCHK(h, hipMemsetAsync(s.slab, 0, nbytes, st));
        s.mem.slab = s.slab;
    }
    if (!s.phase_ints) {
        DEVALLOC(h, s, s.phase_ints, (MPC_MAX_PHASES + s.S) * sizeof(int));
        HIPCHK(h, hipMemsetAsync(s.phase_ints, 0, (MPC_MAX_PHASES + s.S) * sizeof(int), st));
    }
    return 0;
}
// Device staging of the host-pointer entry points, allocated on their first use.
static int ensure_staging(mpc_batch* h, Shard& s, bool traj) {
    const int N = h->spec.N;
    const size_t capa = (size_t)(s.cap > 0 ? s.cap : 1);
    if (!s.staged) {
        if (!s.s_x0) DEVALLOC(h, s, s.s_x0, sizeof(double) * 3 * capa);
        if (!s.s_yref) DEVALLOC(h, s, s.s_yref, sizeof(double) * 5 * N * capa);
        if (!s.s_yrefe) DEVALLOC(h, s, s.s_yrefe, sizeof(double) * 3 * capa);
        if (!s.s_u0) DEVALLOC(h, s, s.s_u0, sizeof(double) * 2 * capa);
        if (!s.s_mask) DEVALLOC(h, s, s.s_mask, capa);
        s.staged = true;
    }
    if (traj && !s.staged_traj) {
        if (!s.s_xt) DEVALLOC(h, s, s.s_xt, sizeof(double) * 3 * (N + 1) * capa);
        if (!s.s_ut) DEVALLOC(h, s, s.s_ut, sizeof(double) * 2 * N * capa);
        if (!s.s_pi) DEVALLOC(h, s, s.s_pi, sizeof(double) * 3 * N * capa);
        s.staged_traj = true;
    }
    return 0;
}

static dim3 tgrid(int R, int B) { return dim3((R + 31) / 32, (B + 63) / 64); }
static void pack_iter(const Mem& m, int N, hipStream_t st, const double* src, int nst, int ncomp, int slot, int n) {
    const size_t tot = (size_t)n * nst * ncomp;
    hipLaunchKernelGGL(k_cap_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, nst, ncomp, 0, slot, m, N, n);
}
static void unpack_iter(const Mem& m, int N, hipStream_t st, double* dst, int nst, int ncomp, int slot, int n) {
    const size_t tot = (size_t)n * nst * ncomp;
    hipLaunchKernelGGL(k_cap_unpack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dst, nst, ncomp, 0, slot, m, N, n);
}

// MPC_MODEL_LANE or a wave model, after AUTO / environment / applicability are resolved
static int pick_model(const mpc_batch* h) {
    static const char* model_env = getenv("MPC_MODEL");
    int model = h->model;
    if (model == MPC_MODEL_AUTO && model_env) {
        if (strcmp(model_env, "lane") == 0) model = MPC_MODEL_LANE;
        else if (strcmp(model_env, "waveseq") == 0 || strcmp(model_env, "wave2seq") == 0) model = MPC_MODEL_WAVE_SEQ;
        else if (strcmp(model_env, "wave") == 0) model = MPC_MODEL_WAVE;
        else if (strcmp(model_env, "wave1") == 0) model = MPC_MODEL_WAVE1;
        else if (strcmp(model_env, "wave2") == 0) model = MPC_MODEL_WAVE2;
    }
    if (model == MPC_MODEL_AUTO) model = MPC_MODEL_WAVE;
    return model;
}

// The lane model's sweeps do not refine a direction (HPIPM's itref_corr_max): an SQP_RTI instance
// whose QP would have been refined is handed to the sequential list kernel of the wave model, which
// solves it again with the refinement.  Full SQP (host-driven rounds in this model) keeps the lane
// model's own result and MPC_DIAG_ITREF_SKIPPED.
static bool lane_hands_over(const mpc_batch* h) { return h->spec.sqp_max_iter <= 1; }

// The lane-model solve on inputs already packed into the workspace rows (R_X0, R_YREF, R_YREFE):
// capsule -> rows, one RTI step (phased, with straggler compaction) or the host-driven SQP loop,
// rows -> capsule.
static int enqueue_lane(mpc_batch* h, Shard& s, int n, hipStream_t st) {
    const int N = h->spec.N;
    Mem& m = s.mem;
    const bool sqp = h->spec.sqp_max_iter > 1;
    const int iters = sqp ? h->spec.sqp_max_iter : 1;
    const dim3 wgrid((n + MPC_WAVE - 1) / MPC_WAVE), wblk(MPC_WAVE);
    s.sqp_rounds = 0;
    if (n > MPC_SINGLE_PHASE_MAX && !s.slab2) {
        // second workspace for the compacted stragglers of the phased solve
        const size_t nbytes = slab_bytes_for(N, (size_t)(s.cap > 0 ? s.cap : 1));
        DEVALLOC(h, s, s.slab2, nbytes);
        HIPCHK(h, hipMemsetAsync(s.slab2, 0, nbytes, st));
        s.mem2 = s.mem;
        s.mem2.slab = s.slab2;
    }
    const unsigned cgrid = (unsigned)(((size_t)n * (N + 1) + 255) / 256);
    // `home` as the phases see it: with the list of instances handed to the sequential list kernel
    // (lane_hands_over), or without one
    Mem mh = m;
    if (!lane_hands_over(h)) mh.REDO = nullptr;
    hipLaunchKernelGGL(k_cap_slab, dim3(cgrid), dim3(256), 0, st, m, N, n, 1);   // capsule -> workspace rows
    for (int it = 0; it < iters; ++it) {
        if (sqp) {
            // head of an SQP iteration; the host reads the number of instances still iterating
            // (one 4-byte copy + sync per iteration: full SQP is the Monte-Carlo mode, not the
            // control-loop one) and stops as soon as every instance has retired.
            int* n_active = s.phase_ints + MPC_MAX_PHASES - 1;   // last counter slot is never a phase count (P <= 7)
            HIPCHK(h, hipMemsetAsync(n_active, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_nlp_res, wgrid, wblk, 0, st, h->consts, m, n, it, n_active);
            int na = 0;
            HIPCHK(h, hipMemcpyAsync(&na, n_active, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            if (na == 0) break;
            ++s.sqp_rounds;
        }
        if (n <= MPC_SINGLE_PHASE_MAX || !s.slab2) {
            HIPCHK(h, hipEventRecord(s.ev_ph[0], st));
            hipLaunchKernelGGL(k_phase, wgrid, wblk, 0, st, h->consts, m, mh, (const int*)nullptr, n, 1, 0, (int*)nullptr);
            HIPCHK(h, hipEventRecord(s.ev_ph[1], st));
            s.n_ph = 1;
        } else {
            // phases 0..P-1; a wave hands over when fewer than MINACT of its lanes still iterate.
            // Measured on MI355X (65536 x N=50): P=2, MINACT=8 is the optimum (20.4 ms/tick vs 25.0
            // single-phase; more phases or an earlier hand-over cost more in compaction traffic
            // than they save).
            const int P = 2;
            const int MINACT = MPC_WAVE / 8;
            int* counts = s.phase_ints;
            int* flags = s.phase_ints + MPC_MAX_PHASES;
            HIPCHK(h, hipMemsetAsync(counts, 0, MPC_MAX_PHASES * sizeof(int), st));
            for (int p = 0; p < P; ++p) {
                const Mem& cur = (p % 2 == 0) ? m : s.mem2;
                const Mem& oth = (p % 2 == 0) ? s.mem2 : m;
                const bool last = (p == P - 1);
                if (p == 0) HIPCHK(h, hipEventRecord(s.ev_ph[0], st));
                hipLaunchKernelGGL(k_phase, wgrid, wblk, 0, st, h->consts, cur, mh, counts + p, n, p == 0 ? 1 : 0,
                                   last ? 0 : MINACT, flags);
                if (p == 0) HIPCHK(h, hipEventRecord(s.ev_ph[1], st));
                if (!last)
                    hipLaunchKernelGGL(k_compact, wgrid, wblk, 0, st, cur, oth, counts + p, n, p == 0 ? 1 : 0, flags,
                                       counts + p + 1, N);
            }
            s.n_ph = 1;
        }
    }
    hipLaunchKernelGGL(k_cap_slab, dim3(cgrid), dim3(256), 0, st, m, N, n, 0);   // workspace rows -> capsule
    HIPCHK(h, hipGetLastError());
    return 0;
}

// The wave-model launch: reads the C-ABI's arrays and the capsules and writes them back.
// wio.x0 / yref / yref_e / u0 must be set.  seq: sequential recursions instead of the scans
// (MPC_MODEL_WAVE_SEQ, development / cross-check; RTI only).
// part: MPC_LAUNCH_BOTH, or the scan kernel / the list kernel alone (the batch = 1 latency path
// launches the list kernel only when the scan kernel has declined something: io.status == PENDING).
enum { MPC_LAUNCH_BOTH = 0, MPC_LAUNCH_SCAN = 1, MPC_LAUNCH_LIST_ONLY = 2 };
// Lanes per OCP of the two-stages-per-lane kernels: N + 1 stages need ceil((N + 1) / 2) lanes.
// 16 / 32 / 64 lanes align with the DPP rows; 21 lanes put three OCPs of 33..42 stages (the
// reference's N = 40 variants) into a wavefront instead of two.
// Wavefront slots of the current device for the persistent scan kernels (k_wave2<.., PERSIST>): four
// per CU — one wavefront per SIMD is what 512 registers and 40 KB of LDS allow — or 0 = one workgroup
// per set of instances (MPC_PERSIST=0, or the device could not be queried).
static int persistent_slots() {
    static const bool off = [] { const char* e = getenv("MPC_PERSIST"); return e && strcmp(e, "0") == 0; }();
    if (off) return 0;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = -1; }
        cus[dev] = n;
    }
    return cus[dev] > 0 ? 4 * cus[dev] : 0;
}
// MPC_RESUME=0 when a handle is created: no Mem::IPM pool, the list kernel solves declined instances
// again from the cold start — what the no-allocation entry point does; A/B runs and tests
static bool ipm_pool_enabled() {
    const char* e = getenv("MPC_RESUME");
    return !(e && strcmp(e, "0") == 0);
}
static int wave2_group_width(int N) { return N + 1 <= 32 ? 16 : N + 1 <= 42 ? 21 : N + 1 <= 64 ? 32 : 64; }
static void launch_wave(const Consts& c, const Mem& m, int n, int sqp_max_iter, const WaveIO& wio, bool seq, hipStream_t st,
                        bool one_stage_per_lane = false, bool two_stages_per_lane = false, int part = MPC_LAUNCH_BOTH) {
    const int N = c.N;
    const bool sqp = sqp_max_iter > 1;
    seq = seq && !sqp;
    // the sequential kernel over the instances the scan kernel declined (two stages per lane, group
    // width by horizon); same grid as a full launch, blocks beyond the list return at once
#define MPC_LAUNCH_LIST()                                                                                                 \
    do {                                                                                                                  \
        if (part == MPC_LAUNCH_SCAN) break;                                                                               \
        const int gw = wave2_group_width(N);                                                                              \
        const int per = 64 / gw;                                                                                          \
        const dim3 grd((n + per - 1) / per), blk(64);                                                                     \
        if (sqp) {                                                                                                        \
            if (gw == 16) hipLaunchKernelGGL((k_wave2<16, false, true, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);      \
            else if (gw == 21) hipLaunchKernelGGL((k_wave2<21, false, true, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio); \
            else if (gw == 32) hipLaunchKernelGGL((k_wave2<32, false, true, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio); \
            else hipLaunchKernelGGL((k_wave2<64, false, true, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);               \
        } else {                                                                                                          \
            if (gw == 16) hipLaunchKernelGGL((k_wave2<16, false, false, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);      \
            else if (gw == 21) hipLaunchKernelGGL((k_wave2<21, false, false, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio); \
            else if (gw == 32) hipLaunchKernelGGL((k_wave2<32, false, false, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio); \
            else hipLaunchKernelGGL((k_wave2<64, false, false, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);               \
        }                                                                                                                 \
    } while (0)
#define MPC_LAUNCH_WAVE(X_)                                                                                               \
    do {                                                                                                                  \
        const dim3 grd((n + X_::PER_BLOCK - 1) / X_::PER_BLOCK), blk(X_::THREADS);                                        \
        if (part == MPC_LAUNCH_LIST_ONLY) {                                                                               \
        } else if (sqp) hipLaunchKernelGGL((k_wave<X_, true, true>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);        \
        else if (persistent_slots() > 0 && (int)grd.x > persistent_slots())                          \
            hipLaunchKernelGGL((k_wave<X_, true, false, true>), dim3(persistent_slots()), blk, 0, st, \
                               c, m, n, sqp_max_iter, wio);                                                               \
        else hipLaunchKernelGGL((k_wave<X_, true, false>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);                  \
        MPC_LAUNCH_LIST();                                                                                                \
    } while (0)
#define MPC_LAUNCH_WAVE2(GW_)                                                                                             \
    do {                                                                                                                  \
        const int per = 64 / GW_;                                                                                         \
        const dim3 grd((n + per - 1) / per), blk(64);                                                                     \
        if (seq) {                                                                                                        \
            hipLaunchKernelGGL((k_wave2<GW_, false, false, false>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);         \
        } else {                                                                                                          \
            const int slots = persistent_slots();                                                                         \
            if (part == MPC_LAUNCH_LIST_ONLY) {                                                                           \
            } else if (slots > 0 && (int)grd.x > slots && GW_ != 64)                                                      \
                hipLaunchKernelGGL((k_wave2<GW_ == 64 ? 32 : GW_, true, false, false, true>), dim3(slots), blk, 0, st, c, m, n, sqp_max_iter, wio); \
            else                                                                                                          \
                hipLaunchKernelGGL((k_wave2<GW_, true, false, false>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);      \
            MPC_LAUNCH_LIST();                                                                                            \
        }                                                                                                                 \
    } while (0)
    const int gw2 = wave2_group_width(N);
    if (seq) {
        // the sequential model (cross-check; RTI only): sequential recursions + refinement, two stages per lane
        if (gw2 == 16) MPC_LAUNCH_WAVE2(16);
        else if (gw2 == 21) MPC_LAUNCH_WAVE2(21);
        else if (gw2 == 32) MPC_LAUNCH_WAVE2(32);
        else MPC_LAUNCH_WAVE2(64);
    } else if (N + 1 > 64) {
        // 64 <= N <= 127: two stages per lane, one OCP per wavefront
        const dim3 grd(n), blk(64);
        if (sqp) {
            if (part != MPC_LAUNCH_LIST_ONLY)
                hipLaunchKernelGGL((k_wave2<64, true, true, false>), grd, blk, 0, st, c, m, n, sqp_max_iter, wio);
            MPC_LAUNCH_LIST();
        } else MPC_LAUNCH_WAVE2(64);
    } else if (one_stage_per_lane || sqp || N + 1 <= 16 ||
               (!two_stages_per_lane && n <= (gw2 == 21 ? 6 : 16) * MPC_SIMD_SLOTS)) {
        // One stage per lane: N + 1 <= 16 fills a 16-lane group as it is (four OCPs per wavefront
        // either way); full SQP keeps one OCP per wavefront where it can — OCPs that share a
        // wavefront wait for each other, and SQP iteration counts spread from 1 to 100; and batches
        // up to 16384 instances (from the batch = 1 control loop to a few rounds of wavefronts over
        // the chip's 1024 SIMDs): a wavefront that carries half as many stages per lane finishes an
        // IPM iteration in 60 % of the time, and nothing waits for a partner OCP.  Measured kernel
        // time one stage / two stages per lane at N = 50: 1024 instances 0.26 / 0.45 ms, 4096:
        // 0.59 / 0.72, 8192: 1.00 / 1.08, 16384: 1.80 / 1.81, 65536: 5.95 / 5.60 (N = 25 alike).  With three
        // OCPs per wavefront (N = 33..41) the two-stage kernel wins earlier: N = 40, 4096: 0.51 / 0.56,
        // 8192: 0.86 / 0.77, 16384: 1.56 / 1.24.
        if (N + 1 <= 16) MPC_LAUNCH_WAVE(XWave<16>);
        else if (N + 1 <= 32) MPC_LAUNCH_WAVE(XWave<32>);
        else MPC_LAUNCH_WAVE(XWave<64>);
    } else {
        // SQP_RTI: two stages per lane, four (N + 1 <= 32), three (<= 42) or two (<= 64) OCPs per wavefront
        if (gw2 == 16) MPC_LAUNCH_WAVE2(16);
        else if (gw2 == 21) MPC_LAUNCH_WAVE2(21);
        else MPC_LAUNCH_WAVE2(32);
    }
#undef MPC_LAUNCH_LIST
#undef MPC_LAUNCH_WAVE
#undef MPC_LAUNCH_WAVE2
}

// The shard's per-instance arrays as seen by a launch that starts at instance i0 (chunked solves;
// wave models only: the lane model's slab is blocked by wavefront).  REDO is shared: the list holds
// indices relative to the slice, and the hot kernel and its list kernel get the same slice.
static Mem mem_slice(const Mem& m, int i0, int N) {
    Mem r = m;
    const size_t o = (size_t)i0;
    r.STATUS += o; r.QPITER += o; r.DIAG += o; r.SQPITER += o; r.ACTIVE += o;
    r.NLPRES += 4 * o;
    r.cap += o * (size_t)(N + 1) * MPC_CAP;
    if (r.lam) r.lam += o * (size_t)(N + 1) * 8;
    if (r.u0) r.u0 += 2 * o;
    return r;
}

// wio: the arrays of the n instances of this launch (already offset); i0: where they sit in the shard
static int enqueue_wave(mpc_batch* h, Shard& s, int n, hipStream_t st, const WaveIO& wio, int model, int i0 = 0, bool first = true,
                        int part = MPC_LAUNCH_BOTH) {
    if (first) HIPCHK(h, hipEventRecord(s.ev_ph[0], st));
    launch_wave(h->consts, i0 ? mem_slice(s.mem, i0, h->spec.N) : s.mem, n, h->spec.sqp_max_iter, wio,
                model == MPC_MODEL_WAVE_SEQ || model == MPC_MODEL_WAVE2_SEQ, st,
                model == MPC_MODEL_WAVE1, model == MPC_MODEL_WAVE2, part);
    HIPCHK(h, hipEventRecord(s.ev_ph[1], st));
    s.n_ph = 1;
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Enqueue the device work of one solve on shard s for its first `n` instances; all pointers are
// device (or device-visible) pointers in the C-ABI's AoS layouts; d_u0 / d_xt / d_ut / d_status /
// d_iter may be null.
static int enqueue_solve(mpc_batch* h, Shard& s, int n, const double* d_x0, const double* d_yref,
                         const double* d_yrefe, double* d_u0, double* d_xt, double* d_ut, int* d_status,
                         int* d_iter, hipStream_t st, int part = MPC_LAUNCH_BOTH) {
    const int N = h->spec.N;
    const dim3 tb(32, 8);
    Mem& m = s.mem;
    const int model = pick_model(h);
    if (!d_u0) d_u0 = s.d_u0;
    if (d_xt) pack_iter(m, N, st, d_xt, N + 1, 3, (int)C_X, n);
    if (d_ut) pack_iter(m, N, st, d_ut, N, 2, (int)C_U, n);
    if (model != MPC_MODEL_LANE) {
        const WaveIO io = {d_x0, d_yref, d_yrefe, d_u0, d_status, d_iter};
        int rc = enqueue_wave(h, s, n, st, io, model, 0, part != MPC_LAUNCH_LIST_ONLY, part);
        if (rc != 0) return rc;
    } else {
        int rc = ensure_lane(h, s, st);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(k_pack, tgrid(3, n), tb, 0, st, d_x0, 1, 3, N, (int)R_X0, m, n);
        hipLaunchKernelGGL(k_pack, tgrid(5 * N, n), tb, 0, st, d_yref, N, 5, 0, (int)R_YREF, m, n);
        hipLaunchKernelGGL(k_pack, tgrid(3, n), tb, 0, st, d_yrefe, 1, 3, N, (int)R_YREFE, m, n);
        rc = enqueue_lane(h, s, n, st);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(k_unpack, tgrid(2, n), tb, 0, st, d_u0, 1, 2, N, (int)R_U0, m, n);
        if (d_status) HIPCHK(h, hipMemcpyAsync(d_status, m.STATUS, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
        if (d_iter) HIPCHK(h, hipMemcpyAsync(d_iter, m.QPITER, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
        if (lane_hands_over(h)) {
            // the list kernel reads the caller's arrays and the capsules and overwrites u0 / status /
            // qp_iter of the instances the phases have handed over (normally none)
            const WaveIO io = {d_x0, d_yref, d_yrefe, d_u0, d_status, d_iter};
            launch_wave(h->consts, m, n, 1, io, false, st, false, false, MPC_LAUNCH_LIST_ONLY);
        }
    }
    if (d_xt) unpack_iter(m, N, st, d_xt, N + 1, 3, (int)C_X, n);
    if (d_ut) unpack_iter(m, N, st, d_ut, N, 2, (int)C_U, n);
    HIPCHK(h, hipGetLastError());
    return 0;
}

extern "C" {

const char* mpc_batch_version(void) { return "mpc_amd 0.3 (gfx950 FP64: wave-per-OCP parallel-in-time + lane-per-OCP)"; }

int mpc_spec_variant(const char* name, mpc_ocp_spec_t* s) {
    if (!name || !s) return -1;
    struct Row { const char* n; int N; double Ts, W[5], We[3], lbu[2], ubu[2], lbx[2], ubx[2]; };
    // SURVEY.md §2.1 (generated acados_solver_<name>.c of every variant); pinned against the
    // reference's own files by tests/golden/ocp_constants.json (tests/test_variants_golden.py)
    static const Row rows[] = {
        {"mobile_robot", 50, 0.05, {2, 2, 0.5, 1, 0.1}, {2, 2, 0.5}, {-0.25, -0.42}, {0.375, 0.42}, {-2, -2}, {22, 16}},
        {"mobile_robot_18", 40, 0.125, {2, 2, 0.5, 1, 0.1}, {2, 2, 0.5}, {-0.4, -0.42}, {0.4, 0.42}, {-1, -1}, {22, 16}},
        {"mobile_robot_25", 40, 0.1, {2, 2, 0.5, 1, 0.25}, {2, 2, 0.5}, {-1, -0.4}, {1, 0.4}, {-1, -1}, {22, 16}},
        {"mobile_robot_50", 80, 0.05, {2, 2, 0.5, 1, 0.25}, {2, 2, 0.5}, {-1.5, -0.4}, {1.5, 0.4}, {-1, -1}, {22, 16}},
        {"park", 49, 0.05, {1, 7, 1, 0.5, 0.05}, {1, 7, 1}, {-0.5, -0.4}, {0.5, 0.4}, {-4, -4}, {4, 4}},
    };
    for (const Row& r : rows) {
        if (strcmp(name, r.n) == 0) {
            memset(s, 0, sizeof(*s));
            s->N = r.N; s->Ts = r.Ts; s->L = 0.0;
            memcpy(s->W, r.W, sizeof(r.W)); memcpy(s->W_e, r.We, sizeof(r.We));
            memcpy(s->lbu, r.lbu, sizeof(r.lbu)); memcpy(s->ubu, r.ubu, sizeof(r.ubu));
            memcpy(s->lbx, r.lbx, sizeof(r.lbx)); memcpy(s->ubx, r.ubx, sizeof(r.ubx));
            s->sqp_max_iter = 1; s->qp_iter_max = 50; s->hpipm_mode = MPC_HPIPM_BALANCE; s->nlp_tol = 1e-6;
            return 0;
        }
    }
    return -1;
}

int mpc_batch_create(mpc_batch_t** out, const mpc_ocp_spec_t* spec, int capacity, const int* device_ids,
                     int n_devices) {
    if (!out) return -1;
    *out = nullptr;
    if (const char* why = spec_problem(spec)) {
        fprintf(stderr, "mpc_batch_create: %s\n", why);
        return -1;
    }
    if (capacity < 1) return -1;
    int ndev_avail = 0;
    if (hipGetDeviceCount(&ndev_avail) != hipSuccess || ndev_avail < 1) return -3;
    std::vector<int> devs;
    if (device_ids && n_devices > 0) {
        for (int d = 0; d < n_devices; ++d) {
            if (device_ids[d] < 0 || device_ids[d] >= ndev_avail) return -1;
            devs.push_back(device_ids[d]);
        }
    } else {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return -3;
        devs.push_back(cur);
    }
    mpc_batch* h = new mpc_batch();
    h->spec = *spec;
    h->consts = make_consts(*spec);
    h->capacity = capacity;
    const int N = spec->N;
    const int nd = (int)devs.size();
    const int per = (capacity + nd - 1) / nd;
    DeviceGuard device_guard;
    h->shards.resize(nd);
    int rc = 0;
    for (int d = 0; d < nd && rc == 0; ++d) {
        Shard& s = h->shards[d];
        s.device = devs[d];
        s.first = d * per;
        s.cap = capacity - s.first < per ? capacity - s.first : per;
        if (s.cap < 0) s.cap = 0;
        const int capa = s.cap > 0 ? s.cap : 1;
        s.S = ((size_t)capa + MPC_WAVE - 1) / MPC_WAVE * MPC_WAVE;
        auto body = [&]() -> int {
            HIPCHK(h, hipSetDevice(s.device));
            HIPCHK(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreate(&s.ev0));
            HIPCHK(h, hipEventCreate(&s.ev1));
            for (hipEvent_t& e : s.ev_ph) HIPCHK(h, hipEventCreate(&e));
            DEVALLOC(h, s, s.islab, islab_ints_for((size_t)capa) * sizeof(int));
            HIPCHK(h, hipMemsetAsync(s.islab, 0, islab_ints_for((size_t)capa) * sizeof(int), s.stream));
            s.mem = make_mem(s.islab, (size_t)capa, N);
            const size_t cb = cap_doubles_for(N, (size_t)capa) * sizeof(double);
            DEVALLOC(h, s, s.capbuf, cb);
            HIPCHK(h, hipMemsetAsync(s.capbuf, 0, cb, s.stream));
            s.mem.cap = s.capbuf;
            if (spec->sqp_max_iter > 1) {   // the bound multipliers only feed back in full SQP
                DEVALLOC(h, s, s.lambuf, cb);
                HIPCHK(h, hipMemsetAsync(s.lambuf, 0, cb, s.stream));
                s.mem.lam = s.lambuf;
            }
            DEVALLOC(h, s, s.d_u0, sizeof(double) * 2 * s.S);
            HIPCHK(h, hipMemsetAsync(s.d_u0, 0, sizeof(double) * 2 * s.S, s.stream));
            if (spec->sqp_max_iter <= 1 && ipm_pool_enabled()) {
                // Mem::IPM: a record for every capsule — with fewer, which of the declined instances get one
                // would depend on the order of their atomics and results would not repeat from run to run
                const int slots = capa;
                const size_t pb = (size_t)slots * ((size_t)(N + 1) * MPC_IPM_STAGE + MPC_IPM_HEAD) * sizeof(double);
                DEVALLOC(h, s, s.ipmbuf, pb);
                s.mem.IPM = s.ipmbuf;
                s.mem.ipm_slots = slots;
            }
            s.mem2 = s.mem;
            {
                const size_t Z = capa < MPC_ZEROCOPY_MAX ? capa : MPC_ZEROCOPY_MAX;
                HIPCHK(h, hipHostMalloc((void**)&s.z_in, sizeof(double) * Z * (6 + 5 * (size_t)N), hipHostMallocMapped));
                HIPCHK(h, hipHostMalloc((void**)&s.z_out, sizeof(double) * Z * 3, hipHostMallocMapped));
            }
            HIPCHK(h, hipStreamSynchronize(s.stream));
            return 0;
        };
        rc = body();
    }
    if (rc != 0) {
        fprintf(stderr, "mpc_batch_create: %s\n", h->err.c_str());
        for (Shard& s : h->shards) free_shard(s);
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void mpc_batch_free(mpc_batch_t* h) {
    if (!h) return;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) free_shard(s);
    delete h;
}

int mpc_batch_solve_device(mpc_batch_t* h, int B, const double* d_x0, const double* d_yref,
                           const double* d_yref_e, double* d_u0, double* d_xtraj, double* d_utraj,
                           int* d_status, int* d_qp_iter, void* stream) {
    if (!h || B < 1 || B > h->capacity || !d_x0 || !d_yref || !d_yref_e) return -1;
    if (h->shards.size() != 1) { h->err = "mpc_batch_solve_device needs a single-device handle"; return -1; }
    if ((d_xtraj == nullptr) != (d_utraj == nullptr)) { h->err = "xtraj and utraj must be given together"; return -1; }
    Shard& s = h->shards[0];
    hipStream_t st = stream ? (hipStream_t)stream : s.stream;
    DeviceGuard device_guard;
    HIPCHK(h, hipSetDevice(s.device));
    HIPCHK(h, hipEventRecord(s.ev0, st));
    int rc = enqueue_solve(h, s, B, d_x0, d_yref, d_yref_e, d_u0, d_xtraj, d_utraj, d_status, d_qp_iter, st);
    if (rc == 0) { HIPCHK(h, hipEventRecord(s.ev1, st)); s.outer_events = true; }
    return rc;
}

int mpc_batch_solve(mpc_batch_t* h, int B, const double* x0, const double* yref, const double* yref_e,
                    double* u0, double* xtraj, double* utraj, int* status, int* qp_iter) {
    if (!h || B < 1 || B > h->capacity || !x0 || !yref || !yref_e) return -1;
    if ((xtraj == nullptr) != (utraj == nullptr)) { h->err = "xtraj and utraj must be given together"; return -1; }
    const int N = h->spec.N;
    DeviceGuard device_guard;
    if (h->shards.size() == 1 && B <= MPC_ZEROCOPY_MAX && !xtraj && pick_model(h) != MPC_MODEL_LANE) {
        // latency path (the reference's batch = 1 control loop): inputs are staged in pinned
        // host memory that k_wave reads directly, outputs are written straight into pinned
        // host memory — one launch and one synchronisation, no copies
        Shard& s = h->shards[0];
        HIPCHK(h, hipSetDevice(s.device));
        double* zx0 = s.z_in;
        double* zyr = zx0 + 3 * (size_t)B;
        double* zye = zyr + 5 * (size_t)N * B;
        memcpy(zx0, x0, sizeof(double) * 3 * B);
        memcpy(zyr, yref, sizeof(double) * 5 * N * B);
        memcpy(zye, yref_e, sizeof(double) * 3 * B);
        double* zu0 = s.z_out;
        int* zst = reinterpret_cast<int*>(zu0 + 2 * (size_t)B);
        int* zit = zst + B;
        // the scan kernel alone; the sequential list kernel follows only if something was declined.
        // (The only events of this path are the pair enqueue_wave records around the kernels: two
        //  more around the call cost 10 us of the 110 us a batch = 1 tick took end to end.)
        int rc = enqueue_solve(h, s, B, zx0, zyr, zye, zu0, nullptr, nullptr, zst, zit, s.stream, MPC_LAUNCH_SCAN);
        if (rc != 0) { sync_all(h); return rc; }
        HIPCHK(h, wait_short(s.stream));
        bool pending = false;
        for (int i = 0; i < B; ++i) pending = pending || zst[i] == MPC_STATUS_PENDING;
        if (pending) {
            rc = enqueue_solve(h, s, B, zx0, zyr, zye, zu0, nullptr, nullptr, zst, zit, s.stream, MPC_LAUNCH_LIST_ONLY);
            if (rc != 0) { sync_all(h); return rc; }
            HIPCHK(h, wait_short(s.stream));
        }
        if (u0) memcpy(u0, zu0, sizeof(double) * 2 * B);
        if (status) memcpy(status, zst, sizeof(int) * B);
        if (qp_iter) memcpy(qp_iter, zit, sizeof(int) * B);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev_ph[0], s.ev_ph[1]) == hipSuccess) h->last_ms = ms;
        else (void)hipGetLastError();
        s.outer_events = false;
        return 0;
    }
    // enqueue on every shard, then wait for all.  (Copies from pageable host memory are staged by
    // the runtime and serialise with the host; callers that want the shards' transfers to overlap
    // pass pinned arrays — hipHostMalloc / hipHostRegister: read in place, below — or use one
    // handle per device.)
    auto body = [&]() -> int {
        for (Shard& s : h->shards) {
            const int n = B - s.first < s.cap ? B - s.first : s.cap;
            if (n <= 0) continue;
            const size_t o = (size_t)s.first;
            HIPCHK(h, hipSetDevice(s.device));
            int rc = 0;
            const int model = pick_model(h);
            // Pinned caller arrays: no staging at all — the wave kernels read x0 / yref / yref_e from
            // the caller's memory over the host link while they compute (2 KB per instance at its
            // start, 23 GB/s at the headline rate) and write u0 straight back.  Handles with several
            // devices: every shard asks for the arrays' address under its own device (pinned host
            // memory is mapped into every device's address space on ROCm) and reads its own slice, so
            // the shards' transfers overlap instead of queueing behind staged copies.
            const double* px0 = nullptr; const double* pyref = nullptr; const double* pyrefe = nullptr;
            if (model != MPC_MODEL_LANE && !xtraj) {
                px0 = pinned_device_ptr(x0, 3 * (size_t)B); pyref = pinned_device_ptr(yref, 5 * (size_t)N * B); pyrefe = pinned_device_ptr(yref_e, 3 * (size_t)B);
            }
            if (px0 && pyref && pyrefe) {
                double* pu0 = pinned_device_ptr(u0, 2 * (size_t)B);
                HIPCHK(h, hipEventRecord(s.ev0, s.stream));
                rc = enqueue_solve(h, s, n, px0 + 3 * o, pyref + 5 * N * o, pyrefe + 3 * o, pu0 ? pu0 + 2 * o : s.d_u0,
                                   nullptr, nullptr, nullptr, nullptr, s.stream);
                if (rc != 0) return rc;
                HIPCHK(h, hipEventRecord(s.ev1, s.stream)); s.outer_events = true;
                if (u0 && !pu0) HIPCHK(h, hipMemcpyAsync(u0 + 2 * o, s.d_u0, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s.stream));
                if (status) HIPCHK(h, hipMemcpyAsync(status + o, s.mem.STATUS, sizeof(int) * n, hipMemcpyDeviceToHost, s.stream));
                if (qp_iter) HIPCHK(h, hipMemcpyAsync(qp_iter + o, s.mem.QPITER, sizeof(int) * n, hipMemcpyDeviceToHost, s.stream));
                continue;
            }
            rc = ensure_staging(h, s, xtraj != nullptr);
            if (rc != 0) return rc;
            // (not for full SQP: its kernel runs 50x longer than the copy and every chunk would
            //  wait for its own slowest instance — measured 351 -> 410 ms on 131072 instances)
            const int nchunks = (model == MPC_MODEL_LANE || xtraj || h->spec.sqp_max_iter > 1) ? 1
                                : n / MPC_CHUNK_MIN > MPC_MAX_CHUNKS ? MPC_MAX_CHUNKS
                                                                         : n / MPC_CHUNK_MIN > 1 ? n / MPC_CHUNK_MIN : 1;
            if (nchunks > 1) {
                // Pipelined: chunk c + 1's inputs are copied (copy_stream) while chunk c is solved
                // (stream).  With pageable host arrays every hipMemcpyAsync returns when its data has
                // left the host buffer, so the loop below still runs one chunk ahead of the kernels.
                if (!s.copy_stream) {
                    HIPCHK(h, hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking));
                    for (hipEvent_t& e : s.ev_chunk) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                }
                HIPCHK(h, hipEventRecord(s.ev0, s.stream));
                // the staging buffers may still be read by the previous solve's kernels
                HIPCHK(h, hipStreamWaitEvent(s.copy_stream, s.ev0, 0));
                const int per = ((n + nchunks - 1) / nchunks + 63) / 64 * 64;
                for (int c = 0, i0 = 0; i0 < n; ++c, i0 += per) {
                    const int nc = n - i0 < per ? n - i0 : per;
                    const size_t oc = o + (size_t)i0;
                    HIPCHK(h, hipMemcpyAsync(s.s_x0 + 3 * (size_t)i0, x0 + 3 * oc, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, s.copy_stream));
                    HIPCHK(h, hipMemcpyAsync(s.s_yref + 5 * (size_t)N * i0, yref + 5 * N * oc, sizeof(double) * 5 * N * nc, hipMemcpyHostToDevice, s.copy_stream));
                    HIPCHK(h, hipMemcpyAsync(s.s_yrefe + 3 * (size_t)i0, yref_e + 3 * oc, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, s.copy_stream));
                    HIPCHK(h, hipEventRecord(s.ev_chunk[c], s.copy_stream));
                    HIPCHK(h, hipStreamWaitEvent(s.stream, s.ev_chunk[c], 0));
                    const WaveIO io = {s.s_x0 + 3 * (size_t)i0, s.s_yref + 5 * (size_t)N * i0, s.s_yrefe + 3 * (size_t)i0,
                                       s.s_u0 + 2 * (size_t)i0, nullptr, nullptr};
                    rc = enqueue_wave(h, s, nc, s.stream, io, model, i0, c == 0);
                    if (rc != 0) return rc;
                }
                HIPCHK(h, hipEventRecord(s.ev1, s.stream)); s.outer_events = true;
                if (u0) HIPCHK(h, hipMemcpyAsync(u0 + 2 * o, s.s_u0, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s.stream));
            } else {
            HIPCHK(h, hipMemcpyAsync(s.s_x0, x0 + 3 * o, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s.stream));
            HIPCHK(h, hipMemcpyAsync(s.s_yref, yref + 5 * N * o, sizeof(double) * 5 * N * n, hipMemcpyHostToDevice, s.stream));
            HIPCHK(h, hipMemcpyAsync(s.s_yrefe, yref_e + 3 * o, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s.stream));
            if (xtraj) {
                HIPCHK(h, hipMemcpyAsync(s.s_xt, xtraj + 3 * (N + 1) * o, sizeof(double) * 3 * (N + 1) * n, hipMemcpyHostToDevice, s.stream));
                HIPCHK(h, hipMemcpyAsync(s.s_ut, utraj + 2 * N * o, sizeof(double) * 2 * N * n, hipMemcpyHostToDevice, s.stream));
            }
            HIPCHK(h, hipEventRecord(s.ev0, s.stream));
            rc = enqueue_solve(h, s, n, s.s_x0, s.s_yref, s.s_yrefe, s.s_u0, xtraj ? s.s_xt : nullptr,
                               xtraj ? s.s_ut : nullptr, nullptr, nullptr, s.stream);
            if (rc != 0) return rc;
            HIPCHK(h, hipEventRecord(s.ev1, s.stream)); s.outer_events = true;
            if (u0) HIPCHK(h, hipMemcpyAsync(u0 + 2 * o, s.s_u0, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s.stream));
            if (xtraj) {
                HIPCHK(h, hipMemcpyAsync(xtraj + 3 * (N + 1) * o, s.s_xt, sizeof(double) * 3 * (N + 1) * n, hipMemcpyDeviceToHost, s.stream));
                HIPCHK(h, hipMemcpyAsync(utraj + 2 * N * o, s.s_ut, sizeof(double) * 2 * N * n, hipMemcpyDeviceToHost, s.stream));
            }
            }
            if (status) HIPCHK(h, hipMemcpyAsync(status + o, s.mem.STATUS, sizeof(int) * n, hipMemcpyDeviceToHost, s.stream));
            if (qp_iter) HIPCHK(h, hipMemcpyAsync(qp_iter + o, s.mem.QPITER, sizeof(int) * n, hipMemcpyDeviceToHost, s.stream));
        }
        return 0;
    };
    const int rc = body();
    if (rc != 0) { sync_all(h); return rc; }
    double worst = 0.0;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        HIPCHK(h, hipSetDevice(s.device));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess && ms > worst) worst = ms;
    }
    h->last_ms = worst;
    return 0;
}

// mask [capacity] (host) -> the shard's device copy; returns the device pointer or null for "all"
static int stage_mask(mpc_batch* h, Shard& s, const unsigned char* mask, const unsigned char** d_mask) {
    *d_mask = nullptr;
    if (!mask) return 0;
    int rc = ensure_staging(h, s, false);
    if (rc != 0) return rc;
    HIPCHK(h, hipMemcpyAsync(s.s_mask, mask + s.first, s.cap, hipMemcpyHostToDevice, s.stream));
    *d_mask = s.s_mask;
    return 0;
}

int mpc_batch_reset(mpc_batch_t* h, const unsigned char* mask) {
    if (!h) return -1;
    const int N = h->spec.N;
    DeviceGuard device_guard;
    int rc = 0;
    for (Shard& s : h->shards) {
        if (s.cap <= 0 || rc != 0) continue;
        auto body = [&]() -> int {
            HIPCHK(h, hipSetDevice(s.device));
            const unsigned char* dm = nullptr;
            int r = stage_mask(h, s, mask, &dm);
            if (r != 0) return r;
            const size_t tot = (size_t)s.cap * (N + 1);
            hipLaunchKernelGGL(k_reset, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s.stream, s.mem, N, dm, s.cap);
            HIPCHK(h, hipGetLastError());
            return 0;
        };
        rc = body();
    }
    sync_all(h);
    return rc;
}

int mpc_batch_shift(mpc_batch_t* h, const unsigned char* mask) {
    if (!h) return -1;
    const int N = h->spec.N;
    DeviceGuard device_guard;
    int rc = 0;
    for (Shard& s : h->shards) {
        if (s.cap <= 0 || rc != 0) continue;
        auto body = [&]() -> int {
            HIPCHK(h, hipSetDevice(s.device));
            const unsigned char* dm = nullptr;
            int r = stage_mask(h, s, mask, &dm);
            if (r != 0) return r;
            hipLaunchKernelGGL(k_shift, dim3(s.cap), dim3(MPC_SHIFT_THREADS), 0, s.stream, s.mem.cap, N, dm, s.cap, (int)C_U);
            if (s.mem.lam)
                hipLaunchKernelGGL(k_shift, dim3(s.cap), dim3(MPC_SHIFT_THREADS), 0, s.stream, s.mem.lam, N, dm, s.cap, 0);
            HIPCHK(h, hipGetLastError());
            return 0;
        };
        rc = body();
    }
    sync_all(h);
    return rc;
}

int mpc_batch_track_set_shift(mpc_batch_t* h, int on) {
    if (!h) return -1;
    h->track_shift = on ? 1 : 0;
    return 0;
}

int mpc_batch_get(mpc_batch_t* h, int B, double* xtraj, double* utraj, double* pi) {
    if (!h || B < 1 || B > h->capacity) return -1;
    const int N = h->spec.N;
    DeviceGuard device_guard;
    int rc = 0;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0 || rc != 0) continue;
        const size_t o = (size_t)s.first;
        auto body = [&]() -> int {
            HIPCHK(h, hipSetDevice(s.device));
            int r = ensure_staging(h, s, true);
            if (r != 0) return r;
            if (xtraj) {
                unpack_iter(s.mem, N, s.stream, s.s_xt, N + 1, 3, (int)C_X, n);
                HIPCHK(h, hipMemcpyAsync(xtraj + 3 * (N + 1) * o, s.s_xt, sizeof(double) * 3 * (N + 1) * n, hipMemcpyDeviceToHost, s.stream));
            }
            if (utraj) {
                unpack_iter(s.mem, N, s.stream, s.s_ut, N, 2, (int)C_U, n);
                HIPCHK(h, hipMemcpyAsync(utraj + 2 * N * o, s.s_ut, sizeof(double) * 2 * N * n, hipMemcpyDeviceToHost, s.stream));
            }
            if (pi) {
                unpack_iter(s.mem, N, s.stream, s.s_pi, N, 3, (int)C_PI, n);
                HIPCHK(h, hipMemcpyAsync(pi + 3 * N * o, s.s_pi, sizeof(double) * 3 * N * n, hipMemcpyDeviceToHost, s.stream));
            }
            HIPCHK(h, hipGetLastError());
            return 0;
        };
        rc = body();
    }
    sync_all(h);
    return rc;
}

int mpc_batch_sim_step(mpc_batch_t* h, int B, double* x, const double* u) {
    if (!h || B < 1 || B > h->capacity || !x || !u) return -1;
    DeviceGuard device_guard;
    int rc = 0;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0 || rc != 0) continue;
        const size_t o = (size_t)s.first;
        auto body = [&]() -> int {
            HIPCHK(h, hipSetDevice(s.device));
            int r = ensure_staging(h, s, false);
            if (r != 0) return r;
            HIPCHK(h, hipMemcpyAsync(s.s_x0, x + 3 * o, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s.stream));
            HIPCHK(h, hipMemcpyAsync(s.s_u0, u + 2 * o, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s.stream));
            hipLaunchKernelGGL(k_sim_step, dim3((n + 255) / 256), dim3(256), 0, s.stream, h->consts, s.s_x0, s.s_u0, n);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(x + 3 * o, s.s_x0, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s.stream));
            return 0;
        };
        rc = body();
    }
    sync_all(h);
    return rc;
}

double mpc_batch_last_kernel_ms(mpc_batch_t* h) {
    if (!h) return -1.0;
    if (h->shards.size() == 1 && h->shards[0].outer_events) {  // device entry point: events may still be in flight
        Shard& s = h->shards[0];
        float ms = 0.f;
        if (hipEventSynchronize(s.ev1) == hipSuccess && hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess)
            h->last_ms = ms;
        else (void)hipGetLastError();   // (events never recorded: not an error of the next launch)
    }
    return h->last_ms;
}
// ---- closed loop on the device ------------------------------------------------------------
static Track make_track(const mpc_batch* h, const Shard& s) {
    Track tr;
    tr.sref = s.t_sref; tr.iref = s.t_iref; tr.M = s.t_M; tr.Mi = s.t_Mi;
    tr.x = s.t_x; tr.u = s.t_u; tr.idx = s.t_idx; tr.yref = s.t_yref; tr.yref_e = s.t_yrefe;
    tr.density = h->t_density; tr.limit = h->t_limit; tr.lookahead = 1;
    return tr;
}

int mpc_batch_track_set_path(mpc_batch_t* h, int M, const double* state_refs, int Mi, const double* input_refs) {
    if (!h || M < 1 || Mi < 1 || !state_refs || !input_refs) return -1;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {
        if (s.cap <= 0) continue;
        HIPCHK(h, hipSetDevice(s.device));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        if (s.t_sref) s.dev_bytes -= sizeof(double) * (3 * (size_t)s.t_M + 2 * (size_t)s.t_Mi);
        if (s.t_iref && !s.t_sref) s.dev_bytes -= sizeof(double) * 2 * (size_t)s.t_Mi;   // left by a call that failed half-way
        (void)hipFree(s.t_sref); (void)hipFree(s.t_iref);
        s.t_sref = s.t_iref = nullptr;
        s.t_M = 0;
        // (t_iref first: mpc_batch_track_init tests t_sref, which is only set once both exist)
        s.t_Mi = Mi;
        DEVALLOC(h, s, s.t_iref, sizeof(double) * 2 * Mi);
        DEVALLOC(h, s, s.t_sref, sizeof(double) * 3 * M);
        HIPCHK(h, hipMemcpy(s.t_sref, state_refs, sizeof(double) * 3 * M, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(s.t_iref, input_refs, sizeof(double) * 2 * Mi, hipMemcpyHostToDevice));
        s.t_M = M; s.t_Mi = Mi;
    }
    return 0;
}

int mpc_batch_track_init(mpc_batch_t* h, int B, const double* x_init, double T, double v_ref) {
    if (!h || B < 1 || B > h->capacity || !x_init || !(T > 0.0) || !(v_ref > 0.0)) return -1;
    // T only sizes the waypoint windows (density, limit), exactly as in the reference, where the
    // constructor's T is ignored by the solver and the plant step: those run on the generated
    // variant's own time step (src/optimizer.cpp:117-119; the reference's own launch files pass
    // T = 0.1 to the Ts = 0.05 variant).  So T != spec.Ts is legal here too.
    // Optimizer::Optimizer (src/optimizer.cpp:128, 174) and find_next_waypoint's statics (:387-391)
    h->t_density = 1.0 / T / v_ref;
    const double v_lim = v_ref > 0.3 ? 0.35 : v_ref;
    h->t_limit = (int)floor(3.95 / (v_lim * T));
    const int N = h->spec.N;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        s.t_n = 0;
        if (n <= 0) continue;
        if (!s.t_sref) { h->err = "mpc_batch_track_init: call mpc_batch_track_set_path first"; return -1; }
        HIPCHK(h, hipSetDevice(s.device));
        // (buffer by buffer, t_x last: mpc_batch_track_run tests t_x, and a call that ran out of
        //  memory half-way is completed by the next one)
        if (!s.t_u) DEVALLOC(h, s, s.t_u, sizeof(double) * 2 * s.S);
        if (!s.t_idx) DEVALLOC(h, s, s.t_idx, sizeof(int) * 4 * s.S);
        if (!s.t_yref) DEVALLOC(h, s, s.t_yref, sizeof(double) * 5 * N * s.S);
        if (!s.t_yrefe) DEVALLOC(h, s, s.t_yrefe, sizeof(double) * 3 * s.S);
        if (!s.t_x) DEVALLOC(h, s, s.t_x, sizeof(double) * 3 * s.S);
        HIPCHK(h, hipMemsetAsync(s.t_x, 0, sizeof(double) * 3 * s.S, s.stream));
        HIPCHK(h, hipMemcpyAsync(s.t_x, x_init + 3 * (size_t)s.first, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s.stream));
        HIPCHK(h, hipMemsetAsync(s.t_u, 0, sizeof(double) * 2 * s.S, s.stream));
        HIPCHK(h, hipMemsetAsync(s.t_idx, 0, sizeof(int) * 4 * s.S, s.stream));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        s.t_n = n;
    }
    return 0;
}

int mpc_batch_track_run(mpc_batch_t* h, int B, int n_ticks, double* x, double* u, int* target, int* status) {
    if (!h || B < 1 || B > h->capacity || n_ticks < 0) return -1;
    const int N = h->spec.N;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {   // validate before anything is enqueued
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        if (!s.t_x) { h->err = "mpc_batch_track_run: call mpc_batch_track_init first"; return -1; }
        if (n > s.t_n) { h->err = "mpc_batch_track_run: B exceeds the vehicles given to mpc_batch_track_init"; return -1; }
    }
    auto body = [&]() -> int {
        for (Shard& s : h->shards) {
            const int n = B - s.first < s.cap ? B - s.first : s.cap;
            if (n <= 0) continue;
            HIPCHK(h, hipSetDevice(s.device));
            const Track tr = make_track(h, s);
            const dim3 wgrid((n + MPC_WAVE - 1) / MPC_WAVE), wblk(MPC_WAVE);
            const unsigned pgrid = (unsigned)(((size_t)n * (N + 1) + 255) / 256);
            HIPCHK(h, hipEventRecord(s.ev0, s.stream));
            for (int t = 0; t < n_ticks; ++t) {
                hipLaunchKernelGGL(k_track_select, wgrid, wblk, 0, s.stream, tr, n);
                hipLaunchKernelGGL(k_track_pack, dim3(pgrid), dim3(256), 0, s.stream, tr, N, n);
                int rc = enqueue_solve(h, s, n, s.t_x, s.t_yref, s.t_yrefe, s.d_u0, nullptr, nullptr, nullptr, nullptr, s.stream);
                if (rc != 0) return rc;
                hipLaunchKernelGGL(k_track_advance, wgrid, wblk, 0, s.stream, h->consts, s.mem, tr, (const double*)s.d_u0, n);
                if (h->track_shift) {
                    hipLaunchKernelGGL(k_shift, dim3(n), dim3(MPC_SHIFT_THREADS), 0, s.stream, s.mem.cap, N,
                                       (const unsigned char*)nullptr, n, (int)C_U);
                    if (s.mem.lam)
                        hipLaunchKernelGGL(k_shift, dim3(n), dim3(MPC_SHIFT_THREADS), 0, s.stream, s.mem.lam, N,
                                           (const unsigned char*)nullptr, n, 0);
                }
            }
            HIPCHK(h, hipEventRecord(s.ev1, s.stream)); s.outer_events = true;
            HIPCHK(h, hipGetLastError());
        }
        return 0;
    };
    const int rc = body();
    if (rc != 0) { sync_all(h); return rc; }
    double worst = 0.0;
    std::vector<int> idx4;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        const size_t o = (size_t)s.first;
        HIPCHK(h, hipSetDevice(s.device));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess && ms > worst) worst = ms;
        if (x) HIPCHK(h, hipMemcpy(x + 3 * o, s.t_x, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
        if (u) HIPCHK(h, hipMemcpy(u + 2 * o, s.t_u, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
        if (status) HIPCHK(h, hipMemcpy(status + o, s.mem.STATUS, sizeof(int) * n, hipMemcpyDeviceToHost));
        if (target) {
            idx4.resize(4 * (size_t)n);
            HIPCHK(h, hipMemcpy(idx4.data(), s.t_idx, sizeof(int) * 4 * n, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) target[o + i] = idx4[4 * (size_t)i];
        }
    }
    h->last_ms = worst;
    return 0;
}

int mpc_batch_get_diag(mpc_batch_t* h, int B, int* diag) {
    if (!h || B < 1 || B > h->capacity || !diag) return -1;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        HIPCHK(h, hipSetDevice(s.device));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        HIPCHK(h, hipMemcpy(diag + s.first, s.mem.DIAG, sizeof(int) * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

int mpc_batch_set_model(mpc_batch_t* h, int model) {
    if (!h || model < MPC_MODEL_AUTO || model > MPC_MODEL_WAVE2_SEQ) return -1;
    h->model = model;
    return 0;
}

// Full SQP (sqp_max_iter > 1): per-instance number of QPs solved and the last NLP residuals
// (stat, eq, ineq, comp) of the last solve.  Host pointers, either may be NULL.
int mpc_batch_get_sqp(mpc_batch_t* h, int B, int* sqp_iter, double* res4) {
    if (!h || B < 1 || B > h->capacity) return -1;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        HIPCHK(h, hipSetDevice(s.device));
        HIPCHK(h, hipStreamSynchronize(s.stream));
        if (sqp_iter) HIPCHK(h, hipMemcpy(sqp_iter + s.first, s.mem.SQPITER, sizeof(int) * n, hipMemcpyDeviceToHost));
        if (res4) HIPCHK(h, hipMemcpy(res4 + 4 * (size_t)s.first, s.mem.NLPRES, sizeof(double) * 4 * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

int mpc_batch_debug_linearize(mpc_batch_t* h, int B, const double* yref, const double* yref_e, double* out) {
    if (!h || B < 1 || B > h->capacity || !yref || !yref_e || !out) return -1;
    const int N = h->spec.N;
    DeviceGuard device_guard;
    for (Shard& s : h->shards) {
        const int n = B - s.first < s.cap ? B - s.first : s.cap;
        if (n <= 0) continue;
        const size_t o = (size_t)s.first;
        HIPCHK(h, hipSetDevice(s.device));
        int rc = ensure_staging(h, s, false);
        if (rc != 0) return rc;
        double* d_out = nullptr;
        const size_t ob = sizeof(double) * 24 * (size_t)n * (N + 1);
        HIPCHK(h, hipMalloc((void**)&d_out, ob));
        hipError_t e = hipMemcpyAsync(s.s_yref, yref + 5 * N * o, sizeof(double) * 5 * N * n, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.s_yrefe, yref_e + 3 * o, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) {
            const size_t tot = (size_t)n * (N + 1);
            hipLaunchKernelGGL(k_debug_lin, dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, s.stream, h->consts, s.mem,
                               (const double*)s.s_yref, (const double*)s.s_yrefe, n, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out + 24 * o * (N + 1), d_out, ob, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
        else (void)hipStreamSynchronize(s.stream);
        (void)hipFree(d_out);
        if (e != hipSuccess) { h->err = std::string("mpc_batch_debug_linearize: ") + hipGetErrorString(e); return -2; }
    }
    return 0;
}

// Device time of the dominant solve kernel of the last solve on the first shard (k_wave, or the
// dense first k_phase launch of the lane model), from HIP events on the launch stream;
// synchronises.  Used by bench.py for the roofline.
double mpc_batch_last_phase0_ms(mpc_batch_t* h) {
    if (!h || h->shards.empty()) return -1.0;
    Shard& s = h->shards[0];
    float ms = 0.f;
    if (s.n_ph < 1 || hipEventSynchronize(s.ev_ph[1]) != hipSuccess ||
        hipEventElapsedTime(&ms, s.ev_ph[0], s.ev_ph[1]) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
    }
    return ms;
}
size_t mpc_batch_device_bytes(const mpc_batch_t* h) {
    size_t tot = 0;
    if (h) for (const Shard& s : h->shards) tot += s.dev_bytes;
    return tot;
}
// ---- the entry point that allocates nothing (include/mpc_batch.h: mpc_batch_solve_static) ----
// Its capsules and per-instance status words live in a file-scope device array instead of a
// handle.  `scratch` is a host-side stand-in for the symbol, not a device address: the address
// each GPU holds its copy at is looked up once per device (static_scratch) and only that pointer
// is ever handed to kernels, hipMemsetAsync or copies.
struct StaticScratch {
    double cap[(size_t)MPC_STATIC_MAX_LINES * MPC_CAP];   // capsule lines [instance][stage 0..N][8]
    int status[MPC_STATIC_MAX_B], qp_iter[MPC_STATIC_MAX_B], diag[MPC_STATIC_MAX_B];
    int redo[MPC_REDO_LIST + MPC_STATIC_MAX_B];   // Mem::REDO (zero when the module is loaded)
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticScratch)];
#define MPC_STATIC_MAX_DEVICES 64
static StaticScratch* g_scratch_ptr[MPC_STATIC_MAX_DEVICES];
static int g_scratch_N[MPC_STATIC_MAX_DEVICES];   // horizon of the iterates the array holds (0: none yet) ...
static int g_scratch_B[MPC_STATIC_MAX_DEVICES];   // ... and how many capsules of it have been initialised
static std::mutex g_scratch_mutex;
// The current device's copy of `scratch` (set-up on the first call on that device), or null.
static StaticScratch* static_scratch(int dev) {
    if (dev < 0 || dev >= MPC_STATIC_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            fprintf(stderr, "mpc_batch_solve_static: hipGetSymbolAddress(scratch): %s\n", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = static_cast<StaticScratch*>(p);
    }
    return g_scratch_ptr[dev];
}

int mpc_batch_solve_static(const mpc_ocp_spec_t* spec, int B, int cold, const double* d_x0, const double* d_yref,
                           const double* d_yref_e, double* d_u0, int* d_status, int* d_qp_iter, void* stream) {
    if (const char* why = spec_problem(spec)) {
        fprintf(stderr, "mpc_batch_solve_static: %s\n", why);
        return -1;
    }
    if (spec->sqp_max_iter != 1) {
        fprintf(stderr, "mpc_batch_solve_static: SQP_RTI only (sqp_max_iter must be 1)\n");
        return -1;
    }
    if (B < 1 || !d_x0 || !d_yref || !d_yref_e || !d_u0) {
        fprintf(stderr, "mpc_batch_solve_static: B must be >= 1 and x0, yref, yref_e, u0 non-null\n");
        return -1;
    }
    const int N = spec->N;
    if (B > MPC_STATIC_MAX_B || (long long)B * (N + 1) > (long long)MPC_STATIC_MAX_LINES) {
        fprintf(stderr, "mpc_batch_solve_static: B = %d at N = %d needs %lld capsule lines; the static workspace holds %d "
                        "(and at most %d instances): use a handle (mpc_batch_create) for larger batches\n",
                B, N, (long long)B * (N + 1), MPC_STATIC_MAX_LINES, MPC_STATIC_MAX_B);
        return -4;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -3;
    StaticScratch* sc = static_scratch(dev);
    if (!sc) return -2;
    {
        // capsule lines are indexed [instance][0..N]: a warm call continues iterates of the same
        // horizon only, and only for capsules an earlier call has initialised
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!cold && (g_scratch_N[dev] != N || B > g_scratch_B[dev])) {
            fprintf(stderr, "mpc_batch_solve_static: warm call (cold = 0) with N = %d, B = %d, but this GPU's array holds %d "
                            "capsules of horizon %d: start them with cold = 1\n", N, B, g_scratch_B[dev], g_scratch_N[dev]);
            return -1;
        }
        if (g_scratch_N[dev] != N) g_scratch_B[dev] = 0;
        g_scratch_N[dev] = N;
        if (B > g_scratch_B[dev]) g_scratch_B[dev] = B;
    }
    hipStream_t st = (hipStream_t)stream;
    Mem m;
    memset(&m, 0, sizeof(m));
    m.cap = sc->cap;
    m.STATUS = sc->status; m.QPITER = sc->qp_iter; m.DIAG = sc->diag;
    m.REDO = sc->redo;
    if (cold) {   // the all-zero iterate of a fresh capsule (acados_solver_mobile_robot.c:703-729)
        const hipError_t e = hipMemsetAsync(sc->cap, 0, sizeof(double) * MPC_CAP * (size_t)B * (N + 1), st);
        if (e != hipSuccess) {
            fprintf(stderr, "mpc_batch_solve_static: hipMemsetAsync: %s\n", hipGetErrorString(e));
            return -2;
        }
    }
    const WaveIO io = {d_x0, d_yref, d_yref_e, d_u0, d_status, d_qp_iter};
    launch_wave(make_consts(*spec), m, B, 1, io, false, st);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "mpc_batch_solve_static: launch: %s\n", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

int mpc_batch_horizon(const mpc_batch_t* h) { return h ? h->spec.N : -1; }
int mpc_batch_capacity(const mpc_batch_t* h) { return h ? h->capacity : -1; }
const char* mpc_batch_last_error(const mpc_batch_t* h) { return h ? h->err.c_str() : "null handle"; }

}  // extern "C"
