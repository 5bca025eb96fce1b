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


@pytest.mark.parametrize("N", sorted(bench.FLOPS_MODEL_STRUCTURED))
def test_structured_flops_table_matches_operation_counts(oracle, N):
    """The structured count (operations on non-structural operands, oracle/op_count.hpp) on a WARM
    tick of the bench workload, against bench.py's second table; it must sit well below the dense
    count and above SURVEY 8(d)'s closed form's order of magnitude."""
    sp, wl = workload.make("mobile_robot", 16, N=N)
    x0, yref, yref_e = wl.tick(0)
    r = oracle.rti_batch(sp, x0, yref, yref_e)
    x1, y1, e1 = wl.tick(1)
    f0, f1 = bench.flops_model_structured(N)
    tot_model = tot = tot_dense = 0.0
    for i in range(16):
        c = oracle.count_ops(sp, x1[i], y1[i], e1[i], r["xtraj"][i], r["utraj"][i])
        flops = c["s_add"] + c["s_mul"] + c["s_div"] + c["sqrt"]
        assert abs(f0 + f1 * c["qp_iter"] - flops) < 0.05 * flops
        tot_model += f0 + f1 * c["qp_iter"]; tot += flops
        tot_dense += c["add"] + c["mul"] + c["div"] + c["sqrt"]
    assert abs(tot_model - tot) < 0.02 * tot
    assert 0.5 * tot_dense < tot < 0.8 * tot_dense
