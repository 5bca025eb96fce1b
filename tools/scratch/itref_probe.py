"""Exploration: oracle vs device on the refinement workload (W x 1e4, +-0.3 m box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
from tests import oracle_lib as oracle
B = 2048
sp, wl = workload.make("mobile_robot", B)
x0, yref, yref_e = (a.copy() for a in wl.tick(0))
shift = x0[:, :2].copy()
x0[:, :2] -= shift; yref[:, :, :2] -= shift[:, None, :]; yref_e[:, :2] -= shift
sp2 = sp.copy(lbx=(-0.3, -0.3), ubx=(0.3, 0.3), W=tuple(1e4 * w for w in sp.W), W_e=tuple(1e4 * w for w in sp.W_e))
models = sys.argv[1:] or ["wave2", "wave2seq"]
solvers = {m: capi.BatchSolver(sp2, B) for m in models}
for m, s in solvers.items():
    s.set_model(m)
xo, uo = np.zeros((B, sp.N + 1, 3)), np.zeros((B, sp.N, 2))
for t in range(4):
    ro = oracle.rti_batch(sp2, x0, yref, yref_e, xo, uo, want_stats=True)
    ref = ro["stats"][:, 3] > 0
    for m, s in solvers.items():
        rg = s.solve(x0, yref, yref_e, xtraj=xo, utraj=uo)
        dg = s.get_diag(B)
        same = ro["qp_iter"] == rg["qp_iter"]
        dev = np.abs(rg["utraj"] - ro["utraj"]).reshape(B, -1).max(1)
        dref = (dg & 1) != 0
        print(t, m, "status eq", np.array_equal(ro["status"], rg["status"]), "same iters", same.mean(), "dev(same)", dev[same].max(),
              "dev(refined by oracle & same)", dev[ref & same].max() if (ref & same).any() else None,
              "oracle refined", ref.sum(), "device refined", dref.sum(), "both", (ref & dref).sum(), "fallback", ((dg & 4) != 0).sum(),
              "lq", ((dg & 2) != 0).sum(), "iters differ on refined", (~same & ref).sum(), "ms", s.last_kernel_ms() if hasattr(s, "last_kernel_ms") else None)
    xo, uo = ro["xtraj"], ro["utraj"]
