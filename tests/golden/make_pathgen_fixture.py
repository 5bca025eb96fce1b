#!/usr/bin/env python3
"""Copies DATA files of the reference (not source) that pin the path-generation port
mpc_amd/pathgen.py: the interpolated runs scripts/run1..5.txt (np.savetxt of `runs[i]`,
scripts/path2.py:283-284) and the references the reference generated from them,
scripts/paths/{state_refs,input_refs,kappa}.txt (scripts/mpc_acados.py:248-254)."""
import os, shutil
SRC = "/root/reference/scripts"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pathgen")
for i in range(1, 6):
    shutil.copy(f"{SRC}/run{i}.txt", f"{DST}/run{i}.txt")
for n in ("state_refs", "input_refs", "kappa"):
    shutil.copy(f"{SRC}/paths/{n}.txt", f"{DST}/{n}.txt")
