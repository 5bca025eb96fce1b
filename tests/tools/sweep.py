#!/usr/bin/env python3
"""Randomised parity sweep (development aid): several seeds x variants x warm ticks, GPU (default
model) against the oracle; prints the worst deviation and any status / iteration-count mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload
from tests import oracle_lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
worst = 0.0
for name, over in [("mobile_robot", {}), ("mobile_robot_18", {}), ("mobile_robot_25", {"N": 25}), ("mobile_robot", {"N": 15}),
                   ("park", {}), ("mobile_robot_50", {})]:
    for seed in range(100, 100 + seeds):
        sp, wl = workload.make(name, B, seed=seed, **over)
        s = capi.BatchSolver(sp, B)
        xo = uo = None
        line = []
        for t in range(3):
            x0, yref, yref_e = wl.tick(t)
            ro = oracle_lib.rti_batch(sp, x0, yref, yref_e, xo, uo); xo, uo = ro["xtraj"], ro["utraj"]
            rg = s.solve(x0, yref, yref_e)
            _, ug, _ = s.get(B, want_pi=False)
            du = np.abs(ug - uo).max(); worst = max(worst, du)
            same = (ro["qp_iter"] == rg["qp_iter"]).mean(); st = np.array_equal(ro["status"], rg["status"])
            dg = int(np.bitwise_or.reduce(s.get_diag(B)))
            line.append(f"{du:.1e}/{same:.4f}/{'ok' if st else 'STATUS'}/d{dg}")
        print(f"{name:16s} N={sp.N:3d} seed {seed}: " + "  ".join(line), flush=True)
        s.close()
print(f"worst |du| {worst:.2e}")
