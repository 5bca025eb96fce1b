#!/usr/bin/env python3
"""Generate tests/golden/model_vectors.json (G1 of SURVEY.md §8c).

Runs THE REFERENCE'S OWN CasADi-generated model functions — compiled by oracle/Makefile
from the sources where they lie under /root/reference into oracle/_ref/libref_model.so —
on seeded inputs and stores inputs and outputs as C99 hex floats (bit-exact).  Only data
is written; no reference source text is stored.  Run in the build container:

    make -C oracle && python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as O  # noqa: E402

VARIANTS = ["mobile_robot", "mobile_robot_18", "mobile_robot_25", "mobile_robot_50", "park"]


def hexlist(a):
    return [float(v).hex() for v in np.asarray(a, float).ravel()]


def main():
    rng = np.random.Generator(np.random.PCG64(20241110))
    cases = []
    n = 64
    xs = np.column_stack([rng.uniform(-2, 22, n), rng.uniform(-2, 16, n), rng.uniform(-7, 7, n)])
    us = np.column_stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.42, 0.42, n)])
    Sxs = rng.normal(size=(n, 9))
    Sus = rng.normal(size=(n, 6))
    # edge cases: identity seed (what ERK stage 1 sees), zeros, bounds, large yaw
    xs[0], us[0], Sxs[0], Sus[0] = (0.3, -0.2, 0.7), (0.35, 0.2), np.eye(3).ravel(), np.zeros(6)
    xs[1], us[1], Sxs[1], Sus[1] = (0, 0, 0), (0, 0), np.eye(3).ravel(), np.zeros(6)
    xs[2], us[2] = (22, 16, np.pi), (0.375, 0.42)
    xs[3], us[3] = (-2, -2, -np.pi), (-0.25, -0.42)
    xs[4], us[4] = (5, 5, 100.0), (1.5, 0.4)
    for i in range(n):
        per_variant = {}
        for v in VARIANTS:
            f = O.ref_call(v + "_expl_vde_forw", [xs[i], Sxs[i], Sus[i], us[i], np.zeros(0)], [3, 9, 6])
            o = O.ref_call(v + "_expl_ode_fun", [xs[i], us[i], np.zeros(0)], [3])
            per_variant[v] = {"vde_forw": hexlist(np.concatenate(f)), "ode_fun": hexlist(o[0])}
        # all variants share one model: store once, assert equality here
        first = per_variant[VARIANTS[0]]
        for v in VARIANTS[1:]:
            assert per_variant[v] == first, f"variant {v} differs from {VARIANTS[0]}"
        cases.append({"x": hexlist(xs[i]), "Sx": hexlist(Sxs[i]), "Su": hexlist(Sus[i]), "u": hexlist(us[i]),
                      "vde_forw": first["vde_forw"], "ode_fun": first["ode_fun"]})
    out = {
        "generator": "oracle/gen_golden.py",
        "source": "reference CasADi C compiled with gcc -O2 -std=c99 (oracle/Makefile target ref)",
        "functions": "<variant>_expl_vde_forw (x, Sx[3x3 col-major], Su[3x2 col-major], u) -> 18; <variant>_expl_ode_fun (x,u) -> 3",
        "variants_identical": VARIANTS,
        "cases": cases,
    }
    path = os.path.join(ROOT, "tests", "golden", "model_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(cases), "cases")


if __name__ == "__main__":
    main()
