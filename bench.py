#!/usr/bin/env python3
"""bench.py — headline benchmark: batched SQP-RTI solves per second (BASELINE.json metric
"OCP solves/sec, mobile_robot N=50 batch 64K @1-8 GPUs; batch=1 p50 latency").

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one control tick of the whole batch: one mpc_batch_solve_device call (one SQP-RTI
step per instance; u0, status and iteration counts out) on inputs that are already resident in
HBM.  Every rank owns one GPU and 65536 instances (weak scaling, the default: the instances are
independent, no collective on the data path; mpc_amd/sharding.py carries the barrier and the
max-over-ranks time); value = instances solved by all ranks / max-over-ranks wall time.  The same
run also times ONE 65536 batch sharded over the ranks (strong scaling, BASELINE.json configs[2]) and
reports it as "strong_scaling"; --scaling strong makes that the headline value instead.  Tick t tracks a window
shifted by t waypoints from the previous iterate (the reference's warm start: no shifting of the
trajectory, src/optimizer.cpp:208-296).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 FP64 FMA lanes x 2 x 2.4 GHz (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# Counter-derived figures of the dominant kernel (HBM bytes moved, FP64 flops executed per launch)
# come from the rocprofv3 passes committed under profiles/; the summary names the sources it was
# taken on, and the bench line says `stale: true` when the kernel sources have changed since.
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "r03", "summary.json")
KERNEL_SOURCES = ("mpc_amd/csrc/mpc_batch.hip", "mpc_amd/csrc/rti_wave2.hpp", "mpc_amd/csrc/rti_wave.hpp",
                  "mpc_amd/csrc/rti_core.hpp", "mpc_amd/csrc/consts.hpp", "include/mpc_hpipm_args.h")


def kernel_source_sha():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile_summary(B, N):
    """Per-launch counter figures of the profiled configuration, or None."""
    try:
        s = json.load(open(PROFILE_SUMMARY))
    except (OSError, ValueError):
        return None
    if s.get("batch") != B or s.get("horizon") != N:
        return None
    s["stale"] = s.get("kernel_source_sha") != kernel_source_sha()
    return s


def algorithmic_bytes_per_solve(N):
    """batch I/O of one tick with the iterate kept in the capsule:
    in x0 (24) + yref (N*40) + yref_e (24); out u0 (16) + status/qp_iter (8)."""
    return 24 + 40 * N + 24 + 16 + 8


# Algorithmic FLOPs of one SQP-RTI solve = F0 + F1 * (HPIPM iterations): add + mul + div + sqrt of
# the dense SEQUENTIAL restatement counted on an operation-counting scalar, fitted per horizon on
# the bench workload (tests/test_flops_model.py re-derives the numbers with the oracle and fails if
# they drift by more than 3 %).  The 16 N transcendental calls of the RK4 linearisation are reported
# separately.  Horizons without an entry scale the N = 50 row linearly in N.
FLOPS_MODEL = {15: (1950.0, 21360.0), 25: (4768.0, 35755.0), 40: (26850.0, 54739.0),
               50: (39354.0, 67903.0), 80: (50700.0, 110198.0)}


def flops_model(N):
    if N in FLOPS_MODEL:
        f0, f1 = FLOPS_MODEL[N]
    else:
        f0, f1 = (v * N / 50.0 for v in FLOPS_MODEL[50])
    return f0, f1, 16 * N


# The STRUCTURED count of the same run (oracle/op_count.hpp): operations of the dense restatement
# whose operands are not exact structural zeros / ones — what an implementation that exploits this
# OCP's sparsity (diagonal Hessian blocks, A = I + (a, b), selections Vx / Vu, x0 eliminated) has
# to execute.  Fitted on a WARM tick of the bench workload (on the all-zero cold iterate most of the
# data is accidentally zero); tests/test_flops_model.py re-derives it (5 %).  Trig calls separate.
FLOPS_MODEL_STRUCTURED = {15: (3643.0, 13540.0), 25: (6580.0, 22818.0), 40: (11583.0, 36873.0),
                          50: (18551.0, 45813.0), 80: (333.0, 76739.0)}


def flops_model_structured(N):
    if N in FLOPS_MODEL_STRUCTURED:
        return FLOPS_MODEL_STRUCTURED[N]
    return tuple(v * N / 50.0 for v in FLOPS_MODEL_STRUCTURED[50])


def flops_survey_8d(N, qp_iter):
    """SURVEY.md section 8(d)'s closed form: N * 390 (RK4 on the 18-vector) + qp_iter * N * 620
    (structured Riccati factorisation 162 + two solves 256 + barrier / step algebra ~200)."""
    return N * 390.0 + qp_iter * N * 620.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU (weak scaling) / in all (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU solves --batch instances; strong: ONE batch of --batch instances is sharded over the GPUs "
                         "(BASELINE.json configs[2]).  The other mode is measured too and reported as <mode>_scaling")
    ap.add_argument("--variant", default="mobile_robot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-stress", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a rehearsal of the multi-rank flow)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    from mpc_amd import sharding
    rank, local_rank, world = sharding.env_rank()
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    from mpc_amd import capi, workload

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = sharding.connect(args.backend, rank, world, dev, log=lambda m: print(f"[bench] {m}", file=sys.stderr))

    K, W = args.steps, args.warmup

    def measure(scaling):
        """K timed ticks of this rank's slice of the global batch (sharding.rank_slice); inputs of
        every tick resident in HBM before the timed region.  Returns the max-over-ranks wall time
        and this rank's per-step device times."""
        lo, hi, total = sharding.rank_slice(scaling, args.batch, world, rank)
        n = hi - lo
        sp_, wl_ = workload.make(args.variant, n, first=lo)
        solver_ = capi.BatchSolver(sp_, n, devices=[local_rank])
        ticks = []
        for t in range(W + K):
            x0, yref, yref_e = wl_.tick(t)
            ticks.append((torch.from_numpy(x0).to(dev), torch.from_numpy(yref).to(dev), torch.from_numpy(yref_e).to(dev)))
        d_u0 = torch.zeros(n, 2, dtype=torch.float64, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        d_it = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def step(t):
            a, b, c = ticks[t]
            solver_.solve_device(n, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr(), 0, 0, d_st.data_ptr(), d_it.data_ptr())

        def sync():
            solver_.last_kernel_ms()  # synchronises the handle's stream
            torch.cuda.synchronize()

        kernel_ms, launch_ms = [], []

        def record(_t):
            # per-step device time from HIP events on the launch stream (this waits for the step;
            # steps are dependent through the warm start anyway)
            kernel_ms.append(solver_.last_kernel_ms())
            launch_ms.append(solver_.last_phase0_ms())

        elapsed = sharding.timed_steps(step, sync, W, K, ranks, after_step=record)
        return dict(elapsed=elapsed, total=total, n=n, sp=sp_, wl=wl_, solver=solver_, kernel_ms=kernel_ms, launch_ms=launch_ms,
                    qp_iter=d_it.cpu().numpy(), status=d_st.cpu().numpy())

    m = measure(args.scaling)
    # the other reading of "batch 64K @ 1-8 GPUs", measured in the same run (identical at one GPU)
    other = "strong" if args.scaling == "weak" else "weak"
    if world > 1:
        mo = measure(other)
        mo["solver"].close()
    else:
        mo = m
    B, sp, wl, solver = m["n"], m["sp"], m["wl"], m["solver"]
    N = sp.N
    elapsed, kernel_ms, launch_ms, qp_iter_last, status_last = m["elapsed"], m["kernel_ms"], m["launch_ms"], m["qp_iter"], m["status"]

    if rank == 0:
        value = m["total"] * K / elapsed
        per = "per GPU" if args.scaling == "weak" else "in all, sharded over the GPUs"
        out = {
            "metric": f"OCP solves/sec, mobile_robot N=50 SQP-RTI, {args.batch} instances {per}",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.variant} N={N} Ts={sp.Ts} SQP-RTI (HPIPM BALANCE, qp_iter_max {sp.qp_iter_max}), "
                                   f"{args.batch} arc-tracking instances {per}, warm iterate, seed {workload.SEED}",
                       "batch_per_gpu": B, "global_batch": m["total"], "horizon": N, "parallelism": f"dp{world} (independent shards, no collective)",
                       "barrier_backend": ranks.backend},
            "mean_qp_iter": float(qp_iter_last.mean()), "max_qp_iter": int(qp_iter_last.max()),
            "failed_instances": int((status_last != 0).sum()),
            "kernel_ms_per_step": float(np.mean(kernel_ms)),
            "handle_device_bytes": solver.device_bytes(),
            # BASELINE.json configs[2] can be read either way; the other reading, same run, same protocol
            other + "_scaling": {"scaling": other, "global_batch": mo["total"], "batch_per_gpu": mo["n"],
                                 "value": mo["total"] * K / mo["elapsed"], "ms_per_step": 1e3 * mo["elapsed"] / K,
                                 "kernel_ms_per_step": float(np.mean(mo["kernel_ms"]))},
        }
        # ---- roofline of the dominant kernel: the wave-model solve launch.  The binding roof is the
        #      FP64 vector unit (~60 flop per byte of batch I/O, no MFMA); the HBM
        #      figures of the same launch ride along. ----
        f0, f1, trig = flops_model(N)
        fs0, fs1 = flops_model_structured(N)
        flops_launch = float((f0 + f1 * qp_iter_last).sum())
        bytes_launch = float(algorithmic_bytes_per_solve(N) * B)
        l_ms = float(np.mean(launch_ms))
        ach_tf = flops_launch / (l_ms * 1e-3) / 1e12
        ach_gbs = bytes_launch / (l_ms * 1e-3) / 1e9
        prof = load_profile_summary(B, N)
        roof = {"bound": "fp64_vector", "kernel": "k_wave2<32,scan,rti,persistent>" if 42 < N + 1 <= 64 and B > 16384 else "wave-model solve kernel",
                "launch_ms": l_ms, "achieved": ach_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach_tf / FP64_VECTOR_PEAK_TFLOPS,
                # the same duration priced with the two smaller counts: operations on non-structural
                # operands only (FLOPS_MODEL_STRUCTURED), and SURVEY.md 8(d)'s closed form
                "frac_structured": float((fs0 + fs1 * qp_iter_last).sum()) / (l_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "frac_survey_8d": float(flops_survey_8d(N, qp_iter_last).sum()) / (l_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "flops_per_solve": flops_launch / B, "flops_per_solve_structured": float((fs0 + fs1 * qp_iter_last).mean()),
                "flops_per_solve_survey_8d": float(flops_survey_8d(N, qp_iter_last).mean()), "trig_calls_per_solve": trig,
                "model": f"frac: algorithmic flops = {f0:.0f} + {f1:.0f}*qp_iter per solve (op count of the dense SEQUENTIAL restatement, "
                         f"multiplies by structural zeros included); frac_structured: {fs0:.0f} + {fs1:.0f}*qp_iter (the same run, operations "
                         "on structural 0 / 1 operands left out); frac_survey_8d: N*390 + qp_iter*N*620 (SURVEY.md 8d).  The parallel scans "
                         "execute more than any of them (executed_*).  Summed over the launch / live HIP-event duration",
                "traffic": None,
                "hbm": {"algorithmic_bytes_per_solve": algorithmic_bytes_per_solve(N), "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS}}
        if prof:
            roof["traffic"] = prof.get("hbm_traffic_bytes_per_launch")
            if roof["traffic"]:
                roof["hbm"]["traffic_rate"] = roof["traffic"] / (l_ms * 1e-3) / 1e9          # GB/s actually moved
                roof["hbm"]["traffic_over_algorithmic"] = roof["traffic"] / bytes_launch
            ex = prof.get("executed_fp64_flop_per_launch")
            if ex:
                roof["executed_achieved"] = ex / (l_ms * 1e-3) / 1e12
                roof["executed_frac"] = roof["executed_achieved"] / FP64_VECTOR_PEAK_TFLOPS
                roof["executed_over_algorithmic"] = ex / flops_launch   # (the dense count of `frac`)
            roof["counters_from"] = {"file": os.path.relpath(PROFILE_SUMMARY, ROOT), "commit": prof.get("commit"),
                                     "kernel_source_sha": prof.get("kernel_source_sha"), "stale": prof["stale"]}
        out["roofline"] = roof
        # ---- batch=1 latency (second half of the metric; BASELINE.md §3: >= 1000 ticks) ----
        # (the three secondary measurements below run at N = 1 only: the other ranks of a multi-GPU
        #  run would sit in the final barrier meanwhile)
        if not args.no_latency and world == 1:
            sp1, wl1 = workload.make(args.variant, 1)
            s1 = capi.BatchSolver(sp1, 1, devices=[local_rank])
            lat, klat, qit = [], [], []
            for t in range(1010):
                x0, yref, yref_e = wl1.tick(t % 64)
                ta = time.perf_counter()
                r1 = s1.solve(x0, yref, yref_e)
                lat.append(1e3 * (time.perf_counter() - ta))
                klat.append(s1.last_kernel_ms())
                qit.append(int(r1["qp_iter"][0]))
            s1.close()
            lat, klat, qit = np.array(lat[10:]), np.array(klat[10:]), np.array(qit[10:])
            slow = lat > 1.5 * np.percentile(lat, 50)
            out["latency_b1_ms"] = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                                    "kernel_p50": float(np.percentile(klat, 50)), "kernel_p99": float(np.percentile(klat, 99)),
                                    # where the tail comes from: the host side of a tick (wall - device time) and the
                                    # HPIPM iteration count of the ticks that take more than 1.5 x the median
                                    "host_overhead_p50": float(np.percentile(lat - klat, 50)), "host_overhead_p99": float(np.percentile(lat - klat, 99)),
                                    "qp_iter_p50": float(np.percentile(qit, 50)), "qp_iter_max": int(qit.max()),
                                    "slow_ticks": int(slow.sum()), "slow_ticks_mean_qp_iter": float(qit[slow].mean()) if slow.any() else None,
                                    "ticks": 1000,
                                    "note": "mpc_batch_solve with host pointers in/out, end to end (inputs staged in pinned host memory the kernel reads "
                                            "directly; the library polls the stream instead of sleeping on it).  The ticks above 1.5 x p50 are the "
                                            "workload's hard QPs (tick 28 of every 64: about 20 HPIPM iterations instead of 7), not host jitter"}
        # ---- the same tick through the host-pointer entry point (PCIe-inclusive; never `value`) ----
        if not args.no_e2e and world == 1:
            e2e = {}
            hs = capi.BatchSolver(sp, B, devices=[local_rank])
            for kind in ("pageable", "pinned"):
                ms = []
                for t in range(4):
                    arrs = [np.ascontiguousarray(a) for a in wl.tick(t)]
                    if kind == "pinned":
                        keep = [torch.from_numpy(a).pin_memory() for a in arrs]
                        arrs = [k.numpy() for k in keep]
                    ta = time.perf_counter()
                    hs.solve(*arrs)
                    ms.append(1e3 * (time.perf_counter() - ta))
                e2e[kind] = float(np.median(ms[1:]))
                hs.reset()
            hs.close()
            out["e2e_host_ms_per_step"] = dict(e2e, note="mpc_batch_solve on host numpy arrays: H2D of x0/yref/yref_e, the solve, D2H of u0/status/qp_iter; "
                                               "median of 3 warm ticks (u0 returned into pageable memory in both cases)")
        # ---- the hand-over path under load: a bounds-active, heavy-weight workload (cost weights x 1e4,
        #      +-0.3 m state box around the start) on which the scan kernel declines instances and the
        #      sequential list kernel solves them again; never `value` ----
        if not args.no_stress and world == 1:
            Bs = min(B, 65536)
            sps, wls = workload.make(args.variant, Bs)
            sps = sps.copy(lbx=(-0.3, -0.3), ubx=(0.3, 0.3), W=tuple(1e4 * w for w in sps.W), W_e=tuple(1e4 * w for w in sps.W_e))
            hs = capi.BatchSolver(sps, Bs, devices=[local_rank])
            d_us = torch.zeros(Bs, 2, dtype=torch.float64, device=dev)
            d_ss = torch.zeros(Bs, dtype=torch.int32, device=dev)
            d_is = torch.zeros(Bs, dtype=torch.int32, device=dev)
            ms, dec, fails, its = [], [], 0, []
            for t in range(5):
                x0, yref, yref_e = (a.copy() for a in wls.tick(t))
                shift = x0[:, :2].copy()
                x0[:, :2] -= shift; yref[:, :, :2] -= shift[:, None, :]; yref_e[:, :2] -= shift
                dd = [torch.from_numpy(a).to(dev) for a in (x0, yref, yref_e)]
                torch.cuda.synchronize()
                hs.solve_device(Bs, dd[0].data_ptr(), dd[1].data_ptr(), dd[2].data_ptr(), d_us.data_ptr(), 0, 0, d_ss.data_ptr(), d_is.data_ptr())
                k_ms = hs.last_kernel_ms()
                torch.cuda.synchronize()
                if t > 0:
                    ms.append(k_ms)
                    dec.append(float(((hs.get_diag(Bs) & 4) != 0).mean()))
                    fails += int((d_ss.cpu().numpy() != 0).sum())
                    its.append(float(d_is.cpu().numpy().mean()))
            hs.close()
            out["stress_workload"] = {"workload": f"{args.variant} N={N}, cost weights x 1e4, state box +-0.3 m around the start, {Bs} instances, warm ticks 1-4",
                                      "ms_per_step": float(np.mean(ms)), "solves_per_s": Bs / (1e-3 * float(np.mean(ms))),
                                      "declined_fraction": float(np.mean(dec)), "mean_qp_iter": float(np.mean(its)), "failed_instances": fails,
                                      "note": "device time of scan launch + list launch (instances whose scan direction missed the acceptance "
                                              "threshold carry on from that iteration in the sequential kernel, which also performs HPIPM's iterative "
                                              "refinement and LQ refactorisation)"}
        # ---- CPU baseline: the oracle (a port, not acados) on a bounded sample, on rank 0 only ----
        if not args.no_cpu_baseline and world == 1:
            from tests import oracle_lib   # the only use of the oracle in this file: the thing timed as the CPU baseline
            nthr = oracle_lib.lib().oracle_max_threads()   # the box's CPU share (OMP_NUM_THREADS), not the whole machine
            ns = min(B, 32768)

            def timed(threads, n, nt):
                xo = uo = None
                tcpu = 0.0
                for t in range(nt + 1):
                    x0, yref, yref_e = wl.tick(t)
                    ta = time.perf_counter()
                    r = oracle_lib.rti_batch(sp, x0[:n], yref[:n], yref_e[:n], xo, uo, nthreads=threads)
                    if t > 0:
                        tcpu += time.perf_counter() - ta
                    xo, uo = r["xtraj"], r["utraj"]
                return nt * n / tcpu

            v_all = timed(nthr, ns, 3)
            v_one = timed(1, min(ns, 2048), 3)
            out["cpu_baseline"] = {"value": v_all, "unit": "solves/s", "cores": nthr, "kind": "port",
                                   "host_cores": os.cpu_count(), "single_thread_value": v_one,
                                   "sample": f"first {ns} instances of the same workload, ticks 1-3 (warm), dense C++ oracle -O2 + OpenMP on the "
                                             f"{nthr} OpenMP threads this job may use (os.cpu_count() reports {os.cpu_count()}); single thread: "
                                             f"first {min(ns, 2048)} instances; acados itself is unavailable offline"}
        print(json.dumps(out), flush=True)
    solver.close()
    ranks.close()


if __name__ == "__main__":
    main()
