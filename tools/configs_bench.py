#!/usr/bin/env python3
"""The five BASELINE.json configurations at their own sizes on one MI355X (device-resident inputs,
warm ticks; config 5 = one GPU's share of the 1 M instances)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd import capi, workload

dev = torch.device("cuda:0")


def ticks_ms(name, B, n=8, **over):
    sp, wl = workload.make(name, B, **over)
    s = capi.BatchSolver(sp, B, devices=[0])
    d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev); d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    ms = []
    for t in range(n):
        a, b, c = (torch.from_numpy(v).to(dev) for v in wl.tick(t))
        torch.cuda.synchronize()
        s.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr(), 0, 0, d_st.data_ptr(), d_it.data_ptr())
        ms.append(s.last_kernel_ms())
    it = d_it.cpu().numpy(); fail = int((d_st.cpu().numpy() != 0).sum())
    s.close()
    return sp, np.array(ms), it, fail


# 1: mobile_robot_18 (N=40, Ts=0.125), batch = 1, closed-loop latency through the host-pointer entry point
sp, wl = workload.make("mobile_robot_18", 1)
s = capi.BatchSolver(sp, 1)
lat = []
for t in range(200):
    x0, yref, yref_e = wl.tick(t)
    ta = time.perf_counter(); s.solve(x0, yref, yref_e); lat.append(1e3 * (time.perf_counter() - ta))
s.close()
lat = np.array(lat[20:])
print(f"config 1  mobile_robot_18 N={sp.N} B=1: end-to-end p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms", flush=True)
# 2: N=25, batch 4096
for over in ({"N": 25}, {}):
    sp, ms, it, fail = ticks_ms("mobile_robot_25", 4096, **over)
    print(f"config 2  mobile_robot_25 N={sp.N} B=4096: {np.median(ms[3:]):.3f} ms/tick = {4096 / np.median(ms[3:]) * 1e3:.3g} solves/s, qp_iter mean {it.mean():.1f}, failed {fail}", flush=True)
# 3: headline
sp, ms, it, fail = ticks_ms("mobile_robot", 65536)
print(f"config 3  mobile_robot N={sp.N} B=65536: {np.median(ms[3:]):.2f} ms/tick = {65536 / np.median(ms[3:]) * 1e3:.3g} solves/s, qp_iter mean {it.mean():.1f}, failed {fail}", flush=True)
# 4: park, batch 16384
sp, ms, it, fail = ticks_ms("park", 16384)
print(f"config 4  park N={sp.N} B=16384: {np.median(ms[3:]):.2f} ms/tick = {16384 / np.median(ms[3:]) * 1e3:.3g} solves/s, qp_iter mean {it.mean():.1f} max {it.max()}, failed {fail}", flush=True)
# 5: full SQP, 131072 instances (1/8 of 1 M)
B = 131072
sp, wl = workload.make("mobile_robot", B, sqp_max_iter=100, nlp_tol=1e-6)
x0, yref, yref_e = wl.tick(0)
s = capi.BatchSolver(sp, B)
for rep in range(2):
    s.reset(); t0 = time.perf_counter(); r = s.solve(x0, yref, yref_e); dt = time.perf_counter() - t0
sit, res = s.get_sqp(B)
print(f"config 5  full SQP N={sp.N} B={B}: wall {dt * 1e3:.0f} ms (kernel {s.last_kernel_ms():.0f} ms), converged {(r['status'] == 0).mean() * 100:.1f} %, "
      f"QPs {int(sit.sum())} = {sit.sum() / dt:.3g} QP/s", flush=True)
# 5 at its full size on ONE GPU: 1 M instances as eight consecutive solves of 131072 through the same
# handle (0.5 GB of capsules + multipliers), each with its own seeded draw of instances
conv, qps, t_solve = 0, 0, 0.0
for part in range(8):
    _, wl8 = workload.make("mobile_robot", B, first=part * B, sqp_max_iter=100, nlp_tol=1e-6)   # instances part*B .. (part+1)*B of the seeded stream
    x0, yref, yref_e = wl8.tick(0)
    s.reset()
    ta = time.perf_counter(); r = s.solve(x0, yref, yref_e); t_solve += time.perf_counter() - ta
    sit, _ = s.get_sqp(B)
    conv += int((r["status"] == 0).sum()); qps += int(sit.sum())
print(f"config 5  full SQP N={sp.N}, 1048576 instances on ONE GPU (8 x {B} through one handle, host arrays in): "
      f"{t_solve:.2f} s in the solves, converged {conv / (8 * B) * 100:.1f} %, QPs {qps} = {qps / t_solve:.3g} QP/s", flush=True)
s.close()
