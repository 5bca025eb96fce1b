"""Kernel time of the one-stage and the two-stage wave kernels over batch sizes (dispatch threshold)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
for name, N in (("mobile_robot", 50), ("mobile_robot_25", 25), ("mobile_robot_18", 40)):
    for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        row = []
        for model in ("wave1", "wave2"):
            sp, wl = workload.make(name, B, N=N)
            s = capi.BatchSolver(sp, B); s.set_model(model)
            ms = []
            for t in range(6):
                x0, yref, yref_e = wl.tick(t)
                s.solve(x0, yref, yref_e); ms.append(s.last_kernel_ms())
            s.close()
            row.append(float(np.median(ms[2:])))
        print(f"{name} N={N} B={B}: wave1 {row[0]:.3f} ms  wave2 {row[1]:.3f} ms  ratio {row[0] / row[1]:.3f}", flush=True)
