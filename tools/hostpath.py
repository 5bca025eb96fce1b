#!/usr/bin/env python3
"""Wall time of mpc_batch_solve (host pointers in/out, the C-ABI a reference-side binding calls)
at large batch sizes, next to the kernel time of the same call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_amd import capi, workload

for B in [int(a) for a in sys.argv[1:]] or [65536, 8192, 1024]:
    sp, wl = workload.make("mobile_robot", B)
    s = capi.BatchSolver(sp, B)
    ticks = [wl.tick(t) for t in range(8)]
    if os.environ.get("PINNED"):
        import torch
        def pin(a):
            t = torch.empty(a.shape, dtype=torch.float64, pin_memory=True); v = t.numpy(); v[...] = a; keep.append(t); return v
        keep = []
        ticks = [tuple(pin(a) for a in tk) for tk in ticks]
    wall, kern = [], []
    for t in range(8):
        x0, yref, yref_e = ticks[t]
        ta = time.perf_counter()
        r = s.solve(x0, yref, yref_e)
        wall.append((time.perf_counter() - ta) * 1e3); kern.append(s.last_kernel_ms())
    print(f"B={B}: wall ms " + " ".join(f"{w:.2f}" for w in wall) + " | device ms " + " ".join(f"{k:.2f}" for k in kern)
          + f" | {B / (np.median(wall[3:]) * 1e-3):.3g} solves/s end to end", flush=True)
    s.close()
