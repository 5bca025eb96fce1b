#!/usr/bin/env python3
"""Development probe: device time of one cold tick as a function of qp_iter_max (all lanes run
exactly that many IPM iterations), which separates prepare cost from per-iteration cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import capi, workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
for itmax in (0, 1, 2, 4, 8):
    sp, wl = workload.make("mobile_robot", B, qp_iter_max=itmax)
    x0, yref, yref_e = wl.tick(0)
    a, b, c = (torch.from_numpy(v).to(dev) for v in (x0, yref, yref_e))
    d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    solver = capi.BatchSolver(sp, B, devices=[0])
    ms = []
    for rep in range(3):
        solver.reset()
        torch.cuda.synchronize()
        solver.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr())
        ms.append(solver.last_kernel_ms())
    print(f"B={B} qp_iter_max={itmax}: {min(ms):.3f} ms", flush=True)
    solver.close()
