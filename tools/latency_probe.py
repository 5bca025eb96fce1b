"""batch = 1 tick-time distribution through mpc_batch_solve (host pointers): wall time against the
device time of the same tick, percentiles and the slowest ticks.  MPC_SYNC=block|spin selects how
the library waits for the stream (mpc_batch.hip wait_short).
usage: python tools/latency_probe.py [ticks]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd import capi, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sp, wl = workload.make("mobile_robot", 1)
ticks = [wl.tick(t) for t in range(64)]
s = capi.BatchSolver(sp, 1)
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    wall, dev = [], []
    for t in range(n + 20):
        x0, yref, yref_e = ticks[t % 64]
        a = time.perf_counter()
        s.solve(x0, yref, yref_e)
        wall.append(1e3 * (time.perf_counter() - a))
        dev.append(s.last_kernel_ms())
    wall, dev = np.array(wall[20:]), np.array(dev[20:])
    host = wall - dev
    pc = lambda a: " ".join(f"{np.percentile(a, q):.4f}" for q in (50, 90, 99, 99.9, 100))
    print(f"[{os.environ.get('MPC_SYNC', 'spin')}, {mode}] wall p50/90/99/99.9/max {pc(wall)} | device {pc(dev)} | wall-device {pc(host)}")
    slow = np.argsort(wall)[-5:]
    print("   slowest ticks (index, wall, device):", [(int(i), round(float(wall[i]), 4), round(float(dev[i]), 4)) for i in slow])
s.close()
