"""Attribute the instructions of a kernel (assembly built with -gline-tables-only) to the member
functions of rti_wave2.hpp by the last .loc that pointed into that file.  Rough (inlined helpers
from other headers inherit the last rti_wave2.hpp line seen), good enough to see which phase
pays for register shuffling.   usage: ASM=/tmp/k2hot_g.s python tools/asmattr.py [file-substring]"""
import os, re, sys, bisect
asm = os.environ.get("ASM", "/tmp/k2hot_g.s")
hdr = sys.argv[1] if len(sys.argv) > 1 else "rti_wave2.hpp"
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpc_amd", "csrc", hdr)).read().split("\n")
starts, names = [], []
for i, l in enumerate(src, 1):
    m = re.match(r"\s*(?:template <[^>]*>\s*)?__device__ __forceinline__ .*?(\w+)\(", l)
    if m:
        starts.append(i); names.append(m.group(1))
fileno = None
cur = "?"
tab = {}
def cat(op):
    if op.startswith("ds_"): return "ds"
    if "accvgpr" in op: return "acc"
    if op.startswith("v_mov"): return "mov"
    if op.startswith("v_cndmask"): return "sel"
    if "lane" in op: return "lane"
    if op.endswith("_f64") or "_f64_" in op: return "f64"
    if op.startswith("s_"): return "s"
    return "oth"
for l in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]*)"', l)
    if m and m.group(2).endswith(hdr):
        fileno = m.group(1)
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        if m.group(1) == fileno:
            j = bisect.bisect_right(starts, int(m.group(2))) - 1
            cur = names[j] if j >= 0 else "?"
        continue
    if l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";"):
        op = l.strip().split()[0]
        d = tab.setdefault(cur, {})
        d[cat(op)] = d.get(cat(op), 0) + 1
cats = ["f64", "acc", "mov", "sel", "lane", "ds", "s", "oth"]
print("%-16s" % "function", " ".join("%6s" % c for c in cats), "  total")
for k, d in sorted(tab.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-16s" % k, " ".join("%6d" % d.get(c, 0) for c in cats), "%7d" % sum(d.values()))
