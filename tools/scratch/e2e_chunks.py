"""End-to-end time of a host-pointer solve (pinned / pageable inputs) — experiment script."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_amd import workload, capi
B = 65536
sp, wl = workload.make("mobile_robot", B)
hs = capi.BatchSolver(sp, B)
res = {}
for kind in ("pinned", "pageable"):
    ms = []
    for t in range(6):
        arrs = [np.ascontiguousarray(a) for a in wl.tick(t)]
        if kind == "pinned":
            keep = [torch.from_numpy(a).pin_memory() for a in arrs]
            arrs = [k.numpy() for k in keep]
        ta = time.perf_counter(); hs.solve(*arrs); ms.append(1e3 * (time.perf_counter() - ta))
    res[kind] = float(np.median(ms[2:]))
    hs.reset()
print(os.environ.get("MPC_CHUNKS_EXPERIMENT"), res, "kernel span", hs.last_kernel_ms())
