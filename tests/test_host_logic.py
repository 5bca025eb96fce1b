"""CPU checks of the product's host-side logic and of the kernel algorithm compiled for the
host (tests/hostbuild): structure-exploiting Riccati/IPM vs the dense oracle, workload
generator, sharding arithmetic."""
import numpy as np
import pytest

from mpc_amd import sharding, variants, workload
from tests import hostcore_lib


@pytest.mark.parametrize("name", list(variants.VARIANTS))
def test_kernel_algorithm_on_host_matches_dense_oracle(oracle, name):
    B = 96
    sp, wl = workload.make(name, B)
    xo = uo = None
    for t in range(3):
        x0, yref, yref_e = wl.tick(t)
        ro = oracle.rti_batch(sp, x0, yref, yref_e, xo, uo)
        rh = hostcore_lib.rti_batch(sp, x0, yref, yref_e, xo, uo)
        assert np.array_equal(ro["status"], rh["status"]) and (rh["status"] == 0).all()
        assert np.array_equal(ro["qp_iter"], rh["qp_iter"])
        assert not rh["diag"].any()     # HPIPM's refinement / LQ tests never fire
        assert np.abs(ro["utraj"] - rh["utraj"]).max() < 1e-10
        assert np.abs(ro["xtraj"] - rh["xtraj"]).max() < 1e-10
        assert np.abs(ro["pi"] - rh["pi"]).max() < 1e-9
        xo, uo = ro["xtraj"], ro["utraj"]


def test_kernel_algorithm_edge_horizons(oracle):
    for N in (1, 2, 3, 18, 25):
        sp, wl = workload.make("mobile_robot_25", 16, N=N)
        x0, yref, yref_e = wl.tick(0)
        ro = oracle.rti_batch(sp, x0, yref, yref_e)
        rh = hostcore_lib.rti_batch(sp, x0, yref, yref_e)
        assert np.array_equal(ro["qp_iter"], rh["qp_iter"])
        assert np.abs(ro["utraj"] - rh["utraj"]).max() < 1e-10


def test_kernel_algorithm_active_bounds_and_nan(oracle):
    # park: both controls saturate; plus a NaN instance that must fail without touching its iterate
    sp, wl = workload.make("park", 32)
    x0, yref, yref_e = wl.tick(0)
    bad = yref.copy(); bad[7, 2, 1] = np.nan
    xt = np.full((32, sp.N + 1, 3), 0.1); ut = np.full((32, sp.N, 2), 0.05)
    ro = oracle.rti_batch(sp, x0, bad, yref_e, xt, ut)
    rh = hostcore_lib.rti_batch(sp, x0, bad, yref_e, xt, ut)
    assert np.array_equal(ro["status"], rh["status"]) and rh["status"][7] == variants.QP_FAILURE
    assert np.array_equal(rh["xtraj"][7], xt[7]) and np.array_equal(rh["u0"][7], ut[7, 0])
    ok = rh["status"] == 0
    assert np.abs(ro["utraj"][ok] - rh["utraj"][ok]).max() < 1e-10
    sat = np.isclose(np.abs(rh["utraj"][ok][:, :, 1]), 0.4, atol=1e-6).any()
    assert sat, "expected saturated steering in the parking workload"


def test_workload_is_deterministic_and_shardable():
    sp, full = workload.make("mobile_robot", 64)
    a = full.tick(2)
    _, part = workload.make("mobile_robot", 24, first=40)
    b = part.tick(2)
    for fa, fb in zip(a, b):
        assert np.array_equal(fa[40:], fb)
    # reference quirks reproduced by the generator (src/optimizer.cpp:244-251, 271)
    x0, yref, yref_e = a
    assert np.array_equal(yref_e, yref[:, 0, :3]) and not yref[:, :, 4].any()
    ds = np.hypot(np.diff(yref[:, :, 0], axis=1), np.diff(yref[:, :, 1], axis=1))
    np.testing.assert_allclose(ds, variants.cruise_speed("mobile_robot") * sp.Ts, rtol=2e-2)


def test_shard_ranges_partition_the_batch():
    for total in (1, 7, 64, 65536, 100003):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = sharding.shard_range(total, world, r)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == total
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)
