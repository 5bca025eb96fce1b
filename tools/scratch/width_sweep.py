#!/usr/bin/env python3
"""Every horizon 16..66 on the two-stage kernels (group widths 16 / 21 / 32 / 64) with random weights and
bounds around the reference's, against the oracle (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload
from tests import oracle_lib
rng = np.random.default_rng(7)
B = 301
worst = 0.0
bad = 0
for N in range(16, 67):
    name = ("mobile_robot", "mobile_robot_18", "mobile_robot_25", "park")[N % 4]
    sp, wl = workload.make(name, B, N=N, seed=1000 + N)
    W = tuple(float(w * np.exp(rng.uniform(-1, 1))) for w in sp.W)
    We = tuple(float(w * np.exp(rng.uniform(-1, 1))) for w in sp.W_e)
    sp = sp.copy(W=W, W_e=We, lbu=tuple(float(v * rng.uniform(0.7, 1.3)) for v in sp.lbu), ubu=tuple(float(v * rng.uniform(0.7, 1.3)) for v in sp.ubu))
    s = capi.BatchSolver(sp, B); s.set_model("wave2")
    xo = uo = None
    for t in range(3):
        x0, yref, yref_e = wl.tick(t)
        ro = oracle_lib.rti_batch(sp, x0, yref, yref_e, xo, uo); xo, uo = ro["xtraj"], ro["utraj"]
        rg = s.solve(x0, yref, yref_e)
        _, ug, _ = s.get(B, want_pi=False)
        same = ro["qp_iter"] == rg["qp_iter"]
        du = np.abs(ug - uo)[same].max()
        worst = max(worst, du)
        ok = np.array_equal(ro["status"], rg["status"]) and same.mean() > 0.99 and du < 1e-9
        bad += 0 if ok else 1
        if not ok:
            print(f"N={N} {name} tick {t}: du {du:.2e} same {same.mean():.4f} status {np.array_equal(ro['status'], rg['status'])}", flush=True)
    s.close()
print(f"horizons 16..66: worst |du| {worst:.2e}, {bad} failing (horizon, tick) pairs")
