#!/usr/bin/env python3
"""Development aid: compile ONE instantiation of a kernel template from mpc_amd/csrc/mpc_batch.hip
on its own (seconds instead of a minute) and print its register / scratch / LDS usage.
usage: python tools/k2probe.py [k_wave2<32,true,false>]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "mpc_amd", "csrc")
inst = sys.argv[1] if len(sys.argv) > 1 else "k_wave2<32, true, false, false>"
name = inst.split("<")[0]
src = open(os.path.join(CS, "mpc_batch.hip")).read()
io = src[src.index("struct WaveIO {"):]
io = io[:io.index("};") + 2]
pol = src[src.index("template <class X> __device__ __forceinline__ X make_policy(int)"):]
pol = pol[:pol.index("// Batch I/O of k_wave")]
m = re.search(r"template <[^>]*>\s*__global__ __launch_bounds__\([^)]*\) void %s\(" % name, src)
body = src[m.start():]
# the kernel ends at the first line that is exactly "}"
end = body.index("\n}\n") + 3
kern = body[:end]
out = os.path.join("/tmp", "k2probe.hip")
with open(out, "w") as f:
    f.write('#include <hip/hip_runtime.h>\n#include "consts.hpp"\n#include "rti_wave.hpp"\n#include "rti_wave2.hpp"\nusing namespace mpc;\n')
    f.write("#define MPC_STATUS_PENDING (-1)\n" + pol + "\n" + io + "\n" + kern + "\n")
    f.write("template __global__ void %s(Consts, Mem, int, int, WaveIO);\n" % inst)
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", CS, "-Rpass-analysis=kernel-resource-usage",
       "--cuda-device-only", "-c", "-o", "/tmp/k2probe.o", out] + extra
r = subprocess.run(cmd, capture_output=True, text=True)
for line in r.stderr.splitlines():
    if "error" in line or any(k in line for k in ("VGPRs", "AGPRs", "Scratch", "Spill", "LDS Size", "Occupancy")):
        print(line.split("remark:")[-1].strip() if "remark" in line else line)
