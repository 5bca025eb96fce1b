#!/usr/bin/env python3
"""Stress of the scan numerics: state bounds tightened around the start positions so that the
barrier terms of the x/y bounds (which enter J of the Riccati scan) become active."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_amd import capi, workload
from tests import oracle_lib

B = 2048
for name, box in [("mobile_robot", 0.3), ("mobile_robot", 0.05), ("park", 0.2), ("mobile_robot_25", 0.1), ("mobile_robot_25", 0.03)]:
    sp, wl = workload.make(name, B)
    x0, yref, yref_e = wl.tick(0)
    # one common box for the whole batch does not fit arbitrary starts: shift every instance to the origin
    shift = x0[:, :2].copy()
    x0 = x0.copy(); yref = yref.copy(); yref_e = yref_e.copy()
    x0[:, :2] -= shift; yref[:, :, :2] -= shift[:, None, :]; yref_e[:, :2] -= shift
    sp = sp.copy(lbx=(-box, -box), ubx=(box, box))
    for model in ["wave", "waveseq", "lane"]:
        s = capi.BatchSolver(sp, B); s.set_model(model)
        xo = uo = None
        out = []
        for t in range(4):
            ro = oracle_lib.rti_batch(sp, x0, yref, yref_e, xo, uo); xo, uo = ro["xtraj"], ro["utraj"]
            rg = s.solve(x0, yref, yref_e)
            xg, ug, _ = s.get(B, want_pi=False)
            dg = s.get_diag(B)
            same = ro["qp_iter"] == rg["qp_iter"]
            du = np.abs(ug - uo).reshape(B, -1).max(1)
            out.append(f"t{t}: same_it {same.mean():.4f} status_eq {(ro['status']==rg['status']).mean():.4f} fail {int((ro['status']!=0).sum())} "
                       f"max|du| {du.max():.2e} (same: {du[same].max():.2e}) it {ro['qp_iter'].mean():.1f}/{ro['qp_iter'].max()} diag {np.bincount(dg, minlength=8).tolist()}")
        print(name, "box", box, model, " | ".join(out), flush=True)
        s.close()
