#!/usr/bin/env python3
"""Per-tick device time of every variant / horizon class at 65536 instances (warm ticks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import capi, workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
for name, over in [("mobile_robot", {}), ("mobile_robot_18", {}), ("mobile_robot_25", {}), ("mobile_robot_25", {"N": 25}),
                   ("mobile_robot", {"N": 15}), ("park", {}), ("mobile_robot_50", {})]:
    sp, wl = workload.make(name, B, **over)
    s = capi.BatchSolver(sp, B, devices=[0])
    d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev); d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    ms = []
    for t in range(4):
        a, b, c = (torch.from_numpy(v).to(dev) for v in wl.tick(t))
        torch.cuda.synchronize()
        s.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr(), 0, 0, 0, d_it.data_ptr())
        ms.append(s.last_kernel_ms())
    it = d_it.cpu().numpy()
    print(f"{name:16s} N={sp.N:3d} B={B}: ms/tick {' '.join(f'{m:.2f}' for m in ms)} | {B/ms[-1]*1e3:.3g} solves/s | qp_iter mean {it.mean():.1f} max {it.max()}", flush=True)
    s.close()
