#!/usr/bin/env python3
"""bench.py — headline benchmark: batched SQP-RTI solves per second (BASELINE.json metric
"OCP solves/sec, mobile_robot N=50 batch 64K @1-8 GPUs; batch=1 p50 latency").

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one control tick of the whole batch: one mpc_batch_solve_device call (AoS->SoA
packing, one SQP-RTI step per instance, u0/status/iterate-independent outputs) on inputs that
are already resident in HBM.  Every rank owns one GPU and 65536 instances (weak scaling: the
instances are independent, no collective on the data path); value = instances solved by all
ranks / max-over-ranks wall time.  Tick t tracks a window shifted by t waypoints from the
previous iterate (the reference's warm start: no shifting of the trajectory,
src/optimizer.cpp:208-296).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 FP64 FMA lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# HBM bytes of one k_wave launch (65536 x N=50) from the rocprofv3 PMC passes in profiles/r01 (v12):
# FETCH_SIZE 2.909e5 KiB x 2 (gfx950 correction of the guide) + WRITE_SIZE 3.69e5 KiB
MEASURED_TRAFFIC_BYTES_PER_LAUNCH = 0.97e9
# FP64 flops one k_wave launch (65536 x N=50, warm ticks) EXECUTES, from the SQ counters in
# profiles/r01/v12_pmc_sq.csv: (2 x 8.58e8 FMA + 5.31e8 MUL + 2.61e8 ADD wave-instructions) x 64 lanes
MEASURED_FP64_FLOP_PER_LAUNCH = 1.60e11


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--variant", default="mobile_robot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a rehearsal of the multi-rank flow)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import torch.distributed as dist
    from mpc_amd import capi, workload

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The collective backend only carries the barrier and the max-over-ranks time (no
        # collective on the data path); if RCCL cannot start on this node the measurement falls
        # back to gloo for those two calls and says so in the JSON line.
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: nccl/RCCL unavailable ({e}); using gloo for barrier/all_reduce", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                args.backend = "gloo"
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    B, K, W = args.batch, args.steps, args.warmup
    sp, wl = workload.make(args.variant, B, first=rank * B)
    N = sp.N
    solver = capi.BatchSolver(sp, B, devices=[local_rank])

    # inputs of every tick resident in HBM before the timed region
    ticks = []
    for t in range(W + K):
        x0, yref, yref_e = wl.tick(t)
        ticks.append((torch.from_numpy(x0).to(dev), torch.from_numpy(yref).to(dev), torch.from_numpy(yref_e).to(dev)))
    d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(t):
        a, b, c = ticks[t]
        solver.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr(), 0, 0, d_st.data_ptr(), d_it.data_ptr())

    def sync():
        solver.last_kernel_ms()  # synchronises the handle's stream
        torch.cuda.synchronize()

    for t in range(W):
        step(t)
    sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, phase0_ms = [], []
    qp_hist = []
    t0 = time.perf_counter()
    for t in range(W, W + K):
        step(t)
        # per-step kernel time from HIP events on the launch stream (this waits for the step;
        # steps are dependent through the warm start anyway)
        kernel_ms.append(solver.last_kernel_ms())
        phase0_ms.append(solver.last_phase0_ms())
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    qp_iter_last = d_it.cpu().numpy()
    status_last = d_st.cpu().numpy()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        total = world * B * K
        value = total / elapsed
        mean_kernel_ms = float(np.mean(kernel_ms))
        mean_it = float(qp_iter_last.mean())
        out = {
            "metric": "OCP solves/sec, mobile_robot N=50 SQP-RTI, 65536 instances per GPU",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.variant} N={N} Ts={sp.Ts} SQP-RTI (HPIPM BALANCE, qp_iter_max {sp.qp_iter_max}), "
                                   f"{B} arc-tracking instances per GPU, warm iterate, seed {workload.SEED}",
                       "batch_per_gpu": B, "horizon": N, "parallelism": f"dp{world} (independent shards, no collective)",
                       "barrier_backend": (("rccl" if args.backend == "nccl" else args.backend) if world > 1 else None)},
            "mean_qp_iter": mean_it, "max_qp_iter": int(qp_iter_last.max()),
            "failed_instances": int((status_last != 0).sum()),
            "kernel_ms_per_step": mean_kernel_ms,
        }
        # ---- roofline of the dominant kernel: k_wave (one OCP per wavefront, workspace in registers) ----
        f0, f1, trig = flops_model(N)
        flops_step = float((f0 + f1 * qp_iter_last).sum())
        bytes_step = algorithmic_bytes_per_solve(N) * B
        p0_ms = float(np.mean(phase0_ms))
        ach_gbs = bytes_step / (p0_ms * 1e-3) / 1e9
        traffic = MEASURED_TRAFFIC_BYTES_PER_LAUNCH if (B == 65536 and N == 50) else None
        out["roofline"] = {"bound": "hbm", "kernel": "k_wave<64,scan>", "launch_ms": p0_ms,
                           "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                           "traffic": traffic,
                           "note": "achieved = ALGORITHMIC bytes (batch I/O: x0, yref, yref_e in; u0, status, qp_iter out = "
                                   f"{algorithmic_bytes_per_solve(N)} B/solve) / live HIP-event duration of the launch; the kernel "
                                   "keeps the IPM workspace in registers and is bound by FP64 VALU issue, not by HBM: see roofline_fp64"}
        if traffic:
            out["roofline"]["hbm_traffic_rate"] = traffic / (p0_ms * 1e-3) / 1e9  # GB/s actually moved
        ach_tf = flops_step / (p0_ms * 1e-3) / 1e12
        out["roofline_fp64"] = {"bound": "fp64_vector", "kernel": "k_wave<64,scan>", "achieved": ach_tf, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": ach_tf / FP64_VECTOR_PEAK_TFLOPS,
                                "flops_per_solve": flops_step / B, "trig_calls_per_solve": trig,
                                "model": f"{f0:.0f} + {f1:.0f}*qp_iter (op count of the dense SEQUENTIAL restatement = algorithmic "
                                         "flops; the parallel scans execute more, see executed_*) over the k_wave launch"}
        if traffic:   # same configuration as the PMC passes
            ex_tf = MEASURED_FP64_FLOP_PER_LAUNCH / (p0_ms * 1e-3) / 1e12
            out["roofline_fp64"]["executed_achieved"] = ex_tf
            out["roofline_fp64"]["executed_frac"] = ex_tf / FP64_VECTOR_PEAK_TFLOPS
            out["roofline_fp64"]["executed_note"] = ("FP64 flops the launch executes by the SQ instruction counters (rocprofv3 --pmc, "
                                                     "profiles/r01/v12_pmc_sq.csv; all 64 lanes counted, idle stages and scan overhead included)")
        # ---- batch=1 latency (second half of the metric) ----
        if not args.no_latency:
            sp1, wl1 = workload.make(args.variant, 1)
            s1 = capi.BatchSolver(sp1, 1, devices=[local_rank])
            lat, klat = [], []
            for t in range(60):
                x0, yref, yref_e = wl1.tick(t)
                ta = time.perf_counter()
                s1.solve(x0, yref, yref_e)
                lat.append(1e3 * (time.perf_counter() - ta))
                klat.append(s1.last_kernel_ms())
            s1.close()
            out["latency_b1_ms"] = {"p50": float(np.percentile(lat[10:], 50)), "p99": float(np.percentile(lat[10:], 99)),
                                    "kernel_p50": float(np.percentile(klat[10:], 50)),
                                    "note": "mpc_batch_solve with host pointers in/out, end to end (inputs staged in pinned host memory the kernel reads directly)"}
        # ---- CPU baseline: the oracle (a port, not acados) on a bounded sample ----
        if not args.no_cpu_baseline:
            from tests import oracle_lib   # the only use of the oracle in this file: the thing timed as the CPU baseline
            nthr = min(oracle_lib.lib().oracle_max_threads(), 16)
            ns = min(B, 8192)
            xo = uo = None
            tcpu = 0.0
            for t in range(4):
                x0, yref, yref_e = wl.tick(t)
                ta = time.perf_counter()
                r = oracle_lib.rti_batch(sp, x0[:ns], yref[:ns], yref_e[:ns], xo, uo, nthreads=nthr)
                if t > 0:
                    tcpu += time.perf_counter() - ta
                xo, uo = r["xtraj"], r["utraj"]
            out["cpu_baseline"] = {"value": 3 * ns / tcpu, "unit": "solves/s", "cores": nthr, "kind": "port",
                                   "sample": f"first {ns} instances of the same workload, ticks 1-3 (warm), "
                                             f"dense C++ oracle -O2 + OpenMP; acados itself is unavailable offline"}
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
