"""Config 5: where does the time of the full-SQP solve go (scan kernel vs list kernel), how many
instances are declined?  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload
B = 131072
sp, wl = workload.make("mobile_robot", B, sqp_max_iter=100, nlp_tol=1e-6)
x0, yref, yref_e = wl.tick(0)
s = capi.BatchSolver(sp, B)
for rep in range(2):
    s.reset()
    t0 = time.perf_counter(); r = s.solve(x0, yref, yref_e); dt = time.perf_counter() - t0
dg = s.get_diag(B)
sit, res = s.get_sqp(B)
print(f"wall {dt*1e3:.1f} ms kernel {s.last_kernel_ms():.1f} ms; declined (bit 4) {int(((dg & 4) != 0).sum())}, refined {int(((dg & 1) != 0).sum())}, "
      f"sqp_iter of declined: {sit[(dg & 4) != 0][:20]}")
