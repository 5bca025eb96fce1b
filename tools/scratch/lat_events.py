#!/usr/bin/env python3
"""batch = 1 end-to-end latency of mpc_batch_solve (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload
sp, wl = workload.make("mobile_robot", 1)
s = capi.BatchSolver(sp, 1)
ticks = [wl.tick(t) for t in range(64)]
lat = []
for t in range(3010):
    x0, yref, yref_e = ticks[t % 64]
    ta = time.perf_counter()
    s.solve(x0, yref, yref_e)
    lat.append(1e3 * (time.perf_counter() - ta))
print(os.environ.get("MPC_LAT_EVENTS", "4"), "p50 %.4f p10 %.4f p99 %.4f ms" % tuple(np.percentile(lat[10:], q) for q in (50, 10, 99)))
