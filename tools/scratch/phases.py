import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
B = 65536
sp, wl = workload.make("mobile_robot", B)
s = capi.BatchSolver(sp, 2 * B)
for t in range(4):
    x0, yref, yref_e = wl.tick(t)
    r = s.solve(x0, yref, yref_e)
_, res = s.get_sqp(2 * B)
d = np.asarray(res).reshape(-1)[4 * B:4 * B + 8 * 4096].reshape(4096, 8)
names = ["factor(+rhs first)", "fw affine", "bw(+rhs)", "fw final", "bu", "begin", "loop total", "iters"]
tot = d[:, 6].mean()
for n, v in zip(names, d.mean(0)):
    print(f"{n:22s} {v:12.1f}  {v / tot:6.3f}")
print("per-iteration cycles", (d[:, 6] / d[:, 7]).mean(), "kernel ms", s.last_kernel_ms())
