"""Reads the per-phase cycle counters of an INSTRUMENTED build (s_memtime around phases, written
behind the NLP residuals; the instrumentation is applied as a temporary patch, see the session
notes in profiles/r02/README.md) — garbage with the committed library."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
B = 65536
names = sys.argv[1:] or ["p0", "p1", "p2", "p3", "p4", "p5", "p6", "iters"]
sp, wl = workload.make("mobile_robot", B)
s = capi.BatchSolver(sp, 2 * B)
for t in range(4):
    x0, yref, yref_e = wl.tick(t)
    r = s.solve(x0, yref, yref_e)
_, res = s.get_sqp(2 * B)
d = np.asarray(res).reshape(-1)[4 * B:4 * B + 8 * 4096].reshape(4096, 8)
for n, v in zip(names, d.mean(0)):
    print(f"{n:28s} {v:12.1f}   per iteration {v / d[:, 7].mean():10.1f}")
print("kernel ms", s.last_kernel_ms())
