"""Per-basic-block instruction categories of one kernel in the -save-temps assembly.
usage: python tools/asmblocks.py <mangled-name-substring> [min_block_size]"""
import re
import sys

import os
path = os.environ.get("ASM", "mpc_amd/csrc/_asm/mpc_batch-hip-amdgcn-amd-amdhsa-gfx950.s")
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sys.argv[1] in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def cat(op):
    if op.startswith("ds_"): return "ds"
    if "accvgpr" in op: return "acc"
    if op.startswith("v_mov"): return "mov"
    if op.startswith("v_cndmask"): return "sel"
    if op.endswith("_f64") or "_f64_" in op: return "f64"
    if op.startswith("s_"): return "s"
    if op.startswith("scratch"): return "scr"
    return "oth"


cur, n, cats, br, out = "entry", 0, {}, [], []
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        out.append((cur, n, cats, br)); cur = m.group(1); n = 0; cats = {}; br = []
    elif l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";"):
        op = l.strip().split()[0]; n += 1; c = cat(op); cats[c] = cats.get(c, 0) + 1
        if "branch" in op: br.append(l.strip().split()[-1])
out.append((cur, n, cats, br))
tot = {}
for o in out:
    for k, v in o[2].items(): tot[k] = tot.get(k, 0) + v
    if o[1] >= minsz: print(o[0], o[1], o[2], o[3])
print("total", tot)
