#!/usr/bin/env python3
"""Where the time of an interior-point iteration of the headline kernel goes, measured with
s_memtime around its phases.  Builds a development variant of the library with -DMPC_PHASE_CLOCK
(cycle counters in k_wave2<.., scan, rti>, written behind the NLP residuals of a handle created with
twice the batch's capacity), runs four ticks of the bench workload and prints cycles per phase.
The counters cost registers (the instrumented kernel is ~5-10 % slower), so read the shares, not
the absolute numbers.     usage (on the GPU box, through gpurun): python tools/phase_clock.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "ab", "phase_clock.so")
if not os.path.exists(lib) or "--build" in sys.argv:
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DMPC_PHASE_CLOCK", "-mllvm", "-disable-machine-licm", "-shared",
                    "-o", lib, "mpc_batch.hip"], cwd=os.path.join(ROOT, "mpc_amd", "csrc"), check=True)
if "--build" in sys.argv:
    sys.exit(0)
os.environ["MPC_AMD_LIB"] = lib
import numpy as np
from mpc_amd import capi, workload

B = 65536
sp, wl = workload.make("mobile_robot", B)
s = capi.BatchSolver(sp, 2 * B)
s.set_model("wave2")
import torch
dev = torch.device("cuda", 0)
d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
for t in range(4):   # one launch per tick (device pointers: a host-pointer solve of this size is chunked)
    a, b, c = (torch.from_numpy(v).to(dev) for v in wl.tick(t))
    torch.cuda.synchronize()
    s.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d_u0.data_ptr())
    s.last_kernel_ms()
    torch.cuda.synchronize()
_, res = s.get_sqp(2 * B)
d = np.asarray(res).reshape(-1)[4 * B:4 * B + 8 * 4096].reshape(4096, 8)
names = ["factorisation scan + first right-hand side", "predictor forward pass", "corrector right-hand side",
         "corrector forward pass, acceptance, update", "residuals and exit test"]
tot = d[:, 5].mean()
print(f"cycles per solve (begin + IPM loop) {tot:.0f}, iterations {d[:, 6].mean():.2f}")
for n, v in zip(names, d[:, :5].mean(0)):
    print(f"  {n:44s} {v:10.0f}  {100 * v / tot:5.1f} %")
print(f"  {'idle behind the partner OCP, loop overhead':44s} {tot - d[:, :5].sum(1).mean():10.0f}  {100 * (1 - d[:, :5].sum(1).mean() / tot):5.1f} %")
print(f"  load + linearise + cold start (before the loop) {d[:, 7].mean():10.0f}  = {100 * d[:, 7].mean() / (tot + d[:, 7].mean()):.1f} % of a wavefront's life")
# how well the 32768 wavefronts pack onto the 1024 SIMD slots: start / end stamps of every wavefront
w = np.asarray(res).reshape(-1)[4 * B + 8 * 4096:4 * B + 8 * 4096 + 2 * (B // 2)].reshape(-1, 2)
t0, t1 = w[:, 0].min(), w[:, 1].max()
span = t1 - t0
busy = (w[:, 1] - w[:, 0]).sum() / (1024 * span)   # (1024 = the SIMD slots of 256 CUs)
last_start = w[:, 0].max()
order = np.sort(w[:, 1])
print(f"kernel span {span / 100:.1f} us (100 MHz clock); wavefront-busy share of 1024 slots {100 * busy:.1f} %")
print(f"  last wavefront starts at {100 * (last_start - t0) / span:.1f} % of the span; ends of the last 1024 wavefronts spread over "
      f"{(order[-1] - order[-1024]) / 100:.1f} us; mean wavefront life {(w[:, 1] - w[:, 0]).mean() / 100:.1f} us, max {(w[:, 1] - w[:, 0]).max() / 100:.1f} us")
