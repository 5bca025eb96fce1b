#!/usr/bin/env python3
"""Quick kernel timing for development: per-tick device time of mpc_batch_solve_device at a few
batch sizes (cold tick 0, then warm ticks), plus a parity spot check against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpc_amd import capi, workload

def run(name, B, ticks=5, check=True):
    sp, wl = workload.make(name, B)
    dev = torch.device("cuda:0")
    solver = capi.BatchSolver(sp, B, devices=[0])
    d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev); d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    ms, its = [], []
    for t in range(ticks):
        x0, yref, yref_e = wl.tick(t)
        a, b, c = (torch.from_numpy(v).to(dev) for v in (x0, yref, yref_e))
        torch.cuda.synchronize()
        solver.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr(), 0, 0, d_st.data_ptr(), d_it.data_ptr())
        ms.append(solver.last_kernel_ms()); it = d_it.cpu().numpy(); its.append((it.mean(), it.max())); pc = None; dg = np.bincount(solver.get_diag(B), minlength=8)
    print(f"{name} B={B}: ms/tick " + " ".join(f"{m:.2f}" for m in ms) + " | iters(mean,max) " + " ".join(f"{a:.1f}/{b}" for a, b in its) + f" | last phase counts {pc.tolist()} diag {dg.tolist()}", flush=True)
    if check:
        from tests import oracle_lib
        n = min(B, 128)
        xo = uo = None
        for t in range(ticks):
            x0, yref, yref_e = wl.tick(t)
            ro = oracle_lib.rti_batch(sp, x0[:n], yref[:n], yref_e[:n], xo, uo); xo, uo = ro["xtraj"], ro["utraj"]
        xg, ug, _ = solver.get(n)
        print(f"   parity vs oracle after {ticks} ticks (first {n}): max|du| {np.abs(ug-uo).max():.2e} max|dx| {np.abs(xg-xo).max():.2e} status_fail {(d_st.cpu().numpy()!=0).sum()}", flush=True)
    solver.close()

if __name__ == "__main__":
    sizes = [int(s) for s in sys.argv[1:]] or [65536, 256, 1]
    for B in sizes:
        run("mobile_robot", B)
