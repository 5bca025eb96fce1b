"""How much of the pair divergence (two OCPs per wavefront wait for each other) could pairing by the
previous tick's iteration count remove?  Prints sum over pairs of max(K1, K2) for several pairings."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
B = 65536
sp, wl = workload.make("mobile_robot", B)
s = capi.BatchSolver(sp, B)
prev = None
for t in range(8):
    x0, yref, yref_e = wl.tick(t)
    r = s.solve(x0, yref, yref_e)
    K = r["qp_iter"].astype(np.int64)
    def cost(order):
        k = K[order]
        return np.maximum(k[0::2], k[1::2]).sum() * 2
    ident = cost(np.arange(B))
    ideal = cost(np.argsort(K, kind="stable"))
    line = f"t={t} mean K {K.mean():.3f}  identity {ident / B:.3f}  ideal {ideal / B:.3f}"
    if prev is not None:
        pred = cost(np.argsort(prev, kind="stable"))
        line += f"  by previous tick {pred / B:.3f}  corr {np.corrcoef(prev, K)[0, 1]:.3f}"
    print(line, "hist", np.bincount(K)[:14])
    prev = K
