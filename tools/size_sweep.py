#!/usr/bin/env python3
"""Device time per warm tick against the batch size, one stage per lane (wave1) against two stages
per lane (wave2) and what AUTO picks: the data behind launch_wave's batch-size threshold
(mpc_batch.hip) and behind the strong-scaling rows (65536 instances cut over 2 / 4 / 8 GPUs).
usage (GPU box): python tools/size_sweep.py [variant] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpc_amd import capi, workload

name = sys.argv[1] if len(sys.argv) > 1 else "mobile_robot"
over = {"N": int(sys.argv[2])} if len(sys.argv) > 2 else {}
dev = torch.device("cuda:0")
print(f"{name} {over}: median device ms of warm ticks 3..9")
for B in (1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 65536):
    sp, wl = workload.make(name, B, **over)
    ticks = [[torch.from_numpy(v).to(dev) for v in wl.tick(t)] for t in range(10)]
    row = []
    for model in ("wave1", "wave2", "auto"):
        s = capi.BatchSolver(sp, B, devices=[0])
        s.set_model(model)
        d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        ms = []
        for t in range(10):
            a, b, c = ticks[t]
            torch.cuda.synchronize()
            s.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr())
            ms.append(s.last_kernel_ms())
        s.close()
        row.append(float(np.median(ms[3:])))
    print(f"  B={B:6d}: wave1 {row[0]:.3f}  wave2 {row[1]:.3f}  auto {row[2]:.3f}   ({B / row[2] * 1e3:.3g} solves/s)", flush=True)
