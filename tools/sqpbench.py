#!/usr/bin/env python3
"""BASELINE config 5 (one GPU's share): full SQP to nlp_tol on perturbed arc-tracking instances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_amd import capi, workload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for model in sys.argv[2:] or ["wave", "lane"]:
    sp, wl = workload.make("mobile_robot", B, sqp_max_iter=100, nlp_tol=1e-6)
    x0, yref, yref_e = wl.tick(0)
    s = capi.BatchSolver(sp, B); s.set_model(model)
    for rep in range(2):
        s.reset()
        t0 = time.perf_counter(); r = s.solve(x0, yref, yref_e); dt = time.perf_counter() - t0
    sit, res = s.get_sqp(B)
    print(f"{model}: B={B} full SQP (<=100 it, tol 1e-6) wall {dt*1e3:.1f} ms, kernel {s.last_kernel_ms():.1f} ms, "
          f"status counts {np.bincount(r['status'], minlength=5).tolist()}, sqp_iter p50/p90/max {np.percentile(sit,50):.0f}/{np.percentile(sit,90):.0f}/{sit.max()}, "
          f"QPs solved {sit.sum()}, {sit.sum()/ (s.last_kernel_ms()*1e-3):.3g} QP/s", flush=True)
    s.close()
