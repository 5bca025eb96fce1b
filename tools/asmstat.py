"""Instruction histogram of one kernel in the -save-temps assembly (make -C mpc_amd/csrc asm).
usage: python tools/asmstat.py <mangled-name-substring> [file.s]"""
import collections
import re
import sys

path = sys.argv[2] if len(sys.argv) > 2 else "mpc_amd/csrc/_asm/mpc_batch-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sys.argv[1] in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
body = lines[start:end]


def is_ins(l):
    t = l.strip()
    return l.startswith("\t") and t and not t.startswith(".") and not t.startswith(";")


ins = [l.strip().split()[0] for l in body if is_ins(l)]
print(lines[start].split(":")[0], len(ins), "instructions")
for k, v in collections.Counter(ins).most_common(40):
    print(f"  {k:28s}{v}")
bb, cur, n = [], "entry", 0
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l):
        bb.append((cur, n)); cur = l.split(":")[0]; n = 0
    elif is_ins(l):
        n += 1
bb.append((cur, n))
print("blocks > 150:", [b for b in bb if b[1] > 150])
