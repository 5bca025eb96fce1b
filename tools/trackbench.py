#!/usr/bin/env python3
"""Device closed loop at scale: B vehicles on the recorded path head, n ticks per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_amd import capi, variants
from mpc_amd.optimizer import load_txt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
s = load_txt(os.path.join(g, "state_refs_path150_head.txt")); u = load_txt(os.path.join(g, "input_refs_path150_head.txt"))
rng = np.random.Generator(np.random.PCG64(1))
start = s[rng.integers(0, 60, B)] + rng.normal(0, 1, (B, 3)) * np.array([0.05, 0.05, 0.1])
sp = variants.spec("mobile_robot")
h = capi.BatchSolver(sp, B)
h.track_set_path(s, u); h.track_init(start, 0.1, 0.5)
h.track_run(2)
t0 = time.perf_counter(); r = h.track_run(n); dt = time.perf_counter() - t0
d = np.min(np.hypot(s[None, :, 0] - r["x"][:2048, None, 0], s[None, :, 1] - r["x"][:2048, None, 1]), axis=1)
print(f"B={B}: {n} closed-loop ticks in {dt*1e3:.1f} ms = {dt/n*1e3:.2f} ms/tick = {B*n/dt:.3g} vehicle-ticks/s; "
      f"failed {int((r['status']!=0).sum())}, max distance to path (first 2048) {d.max():.3f} m, target idx {r['target'].min()}..{r['target'].max()}")
