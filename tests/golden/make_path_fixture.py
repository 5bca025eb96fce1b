#!/usr/bin/env python3
"""Cuts the first 600 waypoints of the reference's path data files
scripts/paths/{state_refs,input_refs}_path150.txt (DATA, not source) into a small fixture.
Run here (where /root/reference is mounted):  python tests/golden/make_path_fixture.py"""
import os
import numpy as np

SRC = "/root/reference/scripts/paths"
HERE = os.path.dirname(os.path.abspath(__file__))
n = 600
s = np.loadtxt(os.path.join(SRC, "state_refs_path150.txt"))[:n]
u = np.loadtxt(os.path.join(SRC, "input_refs_path150.txt"))[:n]
np.savetxt(os.path.join(HERE, "state_refs_path150_head.txt"), s, fmt="%.8f")
np.savetxt(os.path.join(HERE, "input_refs_path150_head.txt"), u, fmt="%.8f")
print(s.shape, u.shape)
