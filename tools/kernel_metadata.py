#!/usr/bin/env python3
"""Register, spill, scratch and LDS figures of every solve kernel in the built library, read from the
gfx950 code object's metadata (llvm-readelf --notes) — the compiler's own numbers, independent of what
a profiler prints per dispatch (rocprofv3 7.2 reports VGPR_Count in units of 4 from a granule count of
8 and leaves Accum_VGPR_Count at 0 on gfx950: profiles/r03/README.md).
usage: python tools/kernel_metadata.py [library] > profiles/rNN/kernel_metadata.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mpc_amd", "libmpc_amd.so")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
print(f"# {os.path.relpath(lib, ROOT)}: code-object metadata of the solve kernels (llvm-readelf --notes)")
print(f"# {'kernel':58s} regs  agpr  sgpr  sgpr_spill  vgpr_spill  scratch_B  lds_B   (regs = arch + accumulation registers per lane)")
for blk in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if "k_wave" not in name and "k_phase" not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)  # noqa: E731
    print(f"{dem:60s} {g('vgpr_count'):>4s}  {blk.split()[0]:>4s}  {g('sgpr_count'):>4s}  {g('sgpr_spill_count'):>10s}  "
          f"{g('vgpr_spill_count'):>10s}  {g('private_segment_fixed_size'):>9s}  {g('group_segment_fixed_size'):>5s}")
