#!/usr/bin/env python3
"""Soak: (1) 20 consecutive warm ticks, 8192 instances, GPU (capsule warm start) vs oracle fed
its own iterate; (2) 65536 vehicles x 300 closed-loop ticks on the device; (3) full SQP, 131072
instances (BASELINE config 5, one GPU's share)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload, variants
from mpc_amd.optimizer import load_txt
from tests import oracle_lib

B = 8192
sp, wl = workload.make("mobile_robot", B)
s = capi.BatchSolver(sp, B)
xo = uo = None
worst, same_min, diag_any = 0.0, 1.0, 0
for t in range(20):
    x0, yref, yref_e = wl.tick(t)
    ro = oracle_lib.rti_batch(sp, x0, yref, yref_e, xo, uo); xo, uo = ro["xtraj"], ro["utraj"]
    rg = s.solve(x0, yref, yref_e)
    _, ug, _ = s.get(B, want_pi=False)
    worst = max(worst, np.abs(ug - uo).max()); same_min = min(same_min, (ro["qp_iter"] == rg["qp_iter"]).mean())
    diag_any |= int(s.get_diag(B).max()); assert np.array_equal(ro["status"], rg["status"])
print(f"(1) 20 ticks x {B}: max|du| {worst:.2e}, min fraction equal qp_iter {same_min:.5f}, diag bits {diag_any}", flush=True)
s.close()

g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
sr = load_txt(os.path.join(g, "state_refs_path150_head.txt")); ur = load_txt(os.path.join(g, "input_refs_path150_head.txt"))
Bv = 65536
rng = np.random.Generator(np.random.PCG64(2))
start = sr[rng.integers(0, 40, Bv)] + rng.normal(0, 1, (Bv, 3)) * np.array([0.05, 0.05, 0.1])
h = capi.BatchSolver(variants.spec("mobile_robot"), Bv)
h.track_set_path(sr, ur); h.track_init(start, 0.1, 0.5)
t0 = time.perf_counter(); nfail = 0
for blk in range(6):
    r = h.track_run(50); nfail += int((r["status"] != 0).sum())
dt = time.perf_counter() - t0
d = np.min(np.hypot(sr[None, :, 0] - r["x"][:4096, None, 0], sr[None, :, 1] - r["x"][:4096, None, 1]), axis=1)
print(f"(2) {Bv} vehicles x 300 ticks in {dt:.2f} s ({dt/300*1e3:.2f} ms/tick): failed at block ends {nfail}, diag {int(h.get_diag(Bv).max())}, "
      f"distance to path p50 {np.median(d):.3f} max {d.max():.3f} m, targets {r['target'].min()}..{r['target'].max()}", flush=True)
h.close()

Bs = 131072
sp5, wl5 = workload.make("mobile_robot", Bs, sqp_max_iter=100, nlp_tol=1e-6)
x0, yref, yref_e = wl5.tick(0)
s5 = capi.BatchSolver(sp5, Bs)
t0 = time.perf_counter(); r = s5.solve(x0, yref, yref_e); dt = time.perf_counter() - t0
sit, res = s5.get_sqp(Bs)
print(f"(3) full SQP {Bs} instances: {dt:.3f} s wall, kernel {s5.last_kernel_ms():.0f} ms, status counts {np.bincount(r['status'], minlength=5).tolist()}, "
      f"QPs {int(sit.sum())}, converged residual max {res[r['status']==0].max():.2e}", flush=True)
s5.close()
