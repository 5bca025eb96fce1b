"""bench.py's algorithmic-FLOP model (roofline_fp64) is a committed table; this test re-derives it
with the oracle's operation-counting scalar so the table cannot drift from the algorithm."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpc_amd import workload


@pytest.mark.parametrize("N", sorted(bench.FLOPS_MODEL))
def test_flops_table_matches_operation_counts(oracle, N):
    sp, wl = workload.make("mobile_robot", 24, N=N)
    x0, yref, yref_e = wl.tick(0)
    f0, f1, trig = bench.flops_model(N)
    for i in range(24):
        c = oracle.count_ops(sp, x0[i], yref[i], yref_e[i], np.zeros((N + 1, 3)), np.zeros((N, 2)))
        flops = c["add"] + c["mul"] + c["div"] + c["sqrt"]
        assert abs(f0 + f1 * c["qp_iter"] - flops) < 0.03 * flops
        assert c["trig"] == trig


def test_other_horizons_scale_the_n50_row():
    f0, f1, trig = bench.flops_model(49)
    assert abs(f1 - 67903.0 * 49 / 50) < 1e-6 and trig == 16 * 49
