#!/usr/bin/env python3
"""Cuts the first 700 waypoints of the reference's path DATA files for the 18 cm/s variant,
scripts/paths/{state_refs,input_refs,wp_attributes,wp_normals}_path118.txt (what
Optimizer::Optimizer loads for v_ref = 0.18, src/optimizer.cpp:138-160), into
tests/golden/path118/ under their own names, so that the constructor's file-loading path can be
run on them (BASELINE.json configs[0]).  Data, not source; rows are copied as text, unchanged.
Run here (where /root/reference is mounted):  python tests/golden/make_path118_fixture.py"""
import os

SRC = "/root/reference/scripts/paths"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "path118")
N = 700
os.makedirs(OUT, exist_ok=True)
for kind in ("state_refs", "input_refs", "wp_attributes", "wp_normals"):
    name = f"{kind}_path118.txt"
    with open(os.path.join(SRC, name)) as f:
        rows = [next(f) for _ in range(N)]
    with open(os.path.join(OUT, name), "w") as f:
        f.writelines(rows)
    print(name, len(rows))
